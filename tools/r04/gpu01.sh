#!/bin/bash
# round 4, first lease: full-rank seed at 262144; the batch leg (configs[3]) under rocprofv3: kernel stats, trace, PMC traffic of one gang
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python tools/find_full_rank_seed.py 262144 > $O/r04_seed262144.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_batch_stats -- python $R/tools/profile_batch.py 32768 48 2 > $O/r04_batch_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_update16<" --kernel-trace --output-format csv -d $O/r04_batch_fetch -- python $R/tools/profile_batch.py 32768 24 1 > $O/r04_batch_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_update16<" --kernel-trace --output-format csv -d $O/r04_batch_write -- python $R/tools/profile_batch.py 32768 24 1 > $O/r04_batch_write.log 2>&1
cd $R
{ python tools/pmc_summary.py $O/r04_batch_fetch k_update16; python tools/pmc_summary.py $O/r04_batch_write k_update16; } > $O/r04_batch_pmc.txt 2>&1
find $O/r04_batch_fetch $O/r04_batch_write -name "*.csv" -delete
ls -la $O/r04_batch_stats/*/ > $O/r04_batch_stats.ls 2>&1
# keep the trace only if it is small enough to come back
find $O/r04_batch_stats -name "*kernel_trace.csv" -size +40M -delete
