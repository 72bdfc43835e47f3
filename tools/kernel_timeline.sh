#!/bin/bash
# kernel timeline of a few blocks: rocprofv3 kernel trace -> per-kernel start/end
R=$1; N=$2; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$N -- python $R/tools/profile_one.py $N 2 > $R/gpurun_out/tl_$N.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/tl_$N/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# second solve: find the last k_to_tiled... just take kernels after the middle
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_block_fast" in n]
half = idx[len(idx) // 2:]            # second solve
for pick in (half[len(half) // 8], half[len(half) // 2], half[len(half) * 7 // 8]):
    t0 = int(rows[pick]["Start_Timestamp"])
    print("---- block at trace index", pick)
    for r in rows[pick:pick + 24]:
        print(f'{(int(r["Start_Timestamp"]) - t0) / 1000:9.1f} {(int(r["End_Timestamp"]) - t0) / 1000:9.1f} us  q{r.get("Queue_Id", "?")}  {r["Kernel_Name"][:40]}  grid {r.get("Grid_Size_X", r.get("Grid_Size", "?"))}')
PY
