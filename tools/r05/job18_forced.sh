#!/bin/bash
# the whole parity file with the round-5 plans forced onto every test
O=gpurun_out; mkdir -p $O
GF2BV_TWO_LEVEL=2 GF2BV_THREE_LEVEL=2 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x > $O/r05_forced_3l_a.log 2>&1; echo "K2 P2 rc=$?" > $O/r05_forced.summary
GF2BV_TWO_LEVEL=3 GF2BV_THREE_LEVEL=2 GF2BV_STRASSEN=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x > $O/r05_forced_3l_b.log 2>&1; echo "K3 P2 L1 rc=$?" >> $O/r05_forced.summary
GF2BV_SPARSE_FAST=0 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x > $O/r05_forced_nosparse.log 2>&1; echo "no sparse rc=$?" >> $O/r05_forced.summary
