#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
GF2BV_STREAM_PROBE=1 timeout 300 python tools/mt_batch_digits_time.py 16 32 2>&1 | cut -c1-220 > gpurun_out/r05_diag_probe.txt
