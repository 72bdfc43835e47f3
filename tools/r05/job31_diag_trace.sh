#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
KEEP_TRACE=1 bash tools/jobs/kernel_stats.sh r05_diag python tools/mt_batch_digits_time.py 16 32
tail -n 12 gpurun_out/r05_diag_trace.log > gpurun_out/r05_diag_run.txt
python - <<'PY' > gpurun_out/r05_diag_kernels.txt 2>&1
import csv, glob, collections
f = glob.glob('/root/repo/gpurun_out/r05_diag_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
# per 50 ms window: count and mean duration of k_block_sparse by template, and of k_update16
win = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name']
    key = None
    if 'k_block_sparse' in n: key = n[n.index('k_block_sparse'):][:30]
    elif 'k_update16<' in n: key = 'k_update16'
    elif 'k_prio_window' in n: key = 'k_prio_window'
    elif 'k_gate' in n: key = 'k_gate'
    elif 'k_narrow_all' in n: key = 'k_narrow_all'
    elif 'k_block_trsm' in n: key = 'k_block_trsm'
    if key is None: continue
    w = (int(r['Start_Timestamp']) - t0) // 100_000_000
    win[w][key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for w in sorted(win):
    print(f"t={w * 0.1:.1f}s", {k: (len(v), round(sum(v) / len(v), 1)) for k, v in sorted(win[w].items())})
PY
find gpurun_out/r05_diag_trace -name "*kernel_trace.csv" -delete
