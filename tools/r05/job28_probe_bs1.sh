#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python tools/probe_sparse.py 1 9 > gpurun_out/r05_probe_bs1.txt 2>&1
