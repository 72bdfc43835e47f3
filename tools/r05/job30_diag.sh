#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 300 python tools/mt_batch_digits_time.py 16 32 > gpurun_out/r05_diag.txt 2>&1
