#!/bin/bash
O=gpurun_out; mkdir -p $O
{ python tools/mt_many_time.py 8 32; python tools/mt_many_time.py 32 32; python tools/mt_many_time.py 8 17; } > $O/r05_mt_many.txt 2>&1
