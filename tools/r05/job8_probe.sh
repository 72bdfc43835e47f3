#!/bin/bash
O=gpurun_out; mkdir -p $O
python tools/probe_sparse.py 1 32 > $O/r05_sparse_probe.txt 2>&1
