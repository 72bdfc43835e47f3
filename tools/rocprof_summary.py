"""Turn a rocprofv3 (rocpd sqlite) result into the per-kernel stats table we commit under profiles/."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
print(f"# {'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
    print(f"  {calls:>7} {total:>12.1f} {avg:>10.2f} {pct:>6.2f}  {name.split('(')[0]}")
