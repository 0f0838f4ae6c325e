"""Per-dispatch timeline of one update cycle from a rocprofv3 rocpd SQLite database: start offset, duration and the gap to
the previous dispatch, for the cycle(s) in the middle of the run.  Usage: timeline.py trace_results.db [cycles]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ncycles = int(sys.argv[2]) if len(sys.argv) > 2 else 2
views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
rows = None
for q in ("select name, start, end from kernels order by start",
          "select kernel_name, start, end from kernels order by start"):
    try:
        rows = db.execute(q).fetchall()
        break
    except sqlite3.Error:
        continue
if rows is None:
    print("no usable kernel view; objects:", views)
    for v in views:
        if "kernel" in v.lower() or "dispatch" in v.lower():
            try:
                cols = [c[1] for c in db.execute(f"pragma table_info('{v}')")]
                print(v, cols)
            except sqlite3.Error:
                pass
    sys.exit(1)

def short(n):
    return n.replace("mcl::(anonymous namespace)::", "").split("(")[0].replace("void ", "")

rows = [(short(n), s, e) for n, s, e in rows]
starts = [i for i, r in enumerate(rows) if r[0].startswith("k_propagate")]  # first kernel of a cycle
if len(starts) < ncycles + 2:
    print("too few cycles in the trace")
    sys.exit(1)
mid = len(starts) // 2
for c in range(ncycles):
    a, b = starts[mid + c], starts[mid + c + 1]
    t0 = rows[a][1]
    prev_end = rows[a - 1][2] if a > 0 else t0
    print(f"--- cycle {mid + c}: {b - a} dispatches, first start -> next cycle's first start = {(rows[b][1] - t0) / 1e3:.1f} us; "
          f"idle before this cycle's first kernel = {(t0 - prev_end) / 1e3:.1f} us")
    busy = 0.0
    for i in range(a, b):
        n, s, e = rows[i]
        gap = (s - prev_end) / 1e3
        busy += (e - s) / 1e3
        print(f"  {n:44s} start {(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}")
        prev_end = e
    print(f"  kernel time {busy:.1f} us, last end at {(prev_end - t0) / 1e3:.1f} us")

# every launch of the dominant kernel in order: shows how its duration moves as the set converges
lf = [(e - s) / 1e3 for n, s, e in rows if n.startswith("k_reweight_lf") or n.startswith("k_reweight_beam")]
if lf:
    print("sensor kernel, every launch in order (us):", " ".join(f"{v:.0f}" for v in lf))
