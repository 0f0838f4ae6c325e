"""Prints the per-kernel summary (calls, total/avg duration, %) of a rocprofv3 rocpd SQLite database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
# median and maximum per kernel from the dispatches themselves (a single launch that the profiler's own activity stretched to
# milliseconds - seen now and then on a stage-level call - shows here instead of hiding in the average)
per = {}
for q in ("select name, start, end from kernels", "select kernel_name, start, end from kernels"):
    try:
        for name, start, end in db.execute(q):
            per.setdefault(name, []).append((end - start) / 1e3)
        break
    except sqlite3.Error:
        continue
print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>7s} {'median_us':>10s} {'max_us':>10s}")
for name, calls, total, avg, pct in rows:
    short = name.replace("mcl::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    d = sorted(per.get(name, []))
    extra = f" {d[len(d) // 2]:10.2f} {d[-1]:10.2f}" if d else ""
    print(f"{short:90s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:7.2f}{extra}")
if len(sys.argv) > 2:
    for (kernel, counter, value, n) in db.execute(
            "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        short = kernel.replace("mcl::(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace(", ", ",")  # (one field: no blanks)
        print(f"PMC {short:60s} {counter:36s} {value:18.1f} (n={n})")
