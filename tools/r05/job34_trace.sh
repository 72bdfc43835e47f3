#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
GF2BV_TRACE=1 timeout 300 python tools/mt_batch_digits_time.py 16 32 2>&1 | cut -c1-200 > gpurun_out/r05_diag_trace2.txt
