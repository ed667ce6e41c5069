"""Turns a rocprofv3 rocpd sqlite database (--kernel-trace --stats) into the text summary kept under profiles/."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print(f"{'kernel':60s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
for name, calls, tot, avg, pct in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{short[:60]:60s} {calls:7d} {tot:12.1f} {avg:10.2f} {pct:7.2f}")
try:
    q = ("select name, count(*), avg(duration)/1000.0, min(duration)/1000.0, max(duration)/1000.0, max(vgpr_count), max(sgpr_count), "
         "max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc")
    print()
    print(f"{'kernel':44s} {'n':>5s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'grid':>9s} {'wg':>5s}")
    for r in cur.execute(q):
        short = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print(f"{short[:44]:44s} {r[1]:5d} {r[2]:9.2f} {r[3]:9.2f} {r[4]:9.2f} {r[5]:5d} {r[6]:5d} {r[7]:7d} {r[8]:9d} {r[9]:5d}")
except Exception as e:  # noqa
    print("(no per-dispatch table)", e)
