"""Prints the per-kernel summary (calls, total/avg duration, %) of a rocprofv3 rocpd SQLite database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>7s}")
for name, calls, total, avg, pct in rows:
    short = name.replace("mcl::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print(f"{short:90s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:7.2f}")
if len(sys.argv) > 2:
    for (kernel, counter, value, n) in db.execute(
            "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        short = kernel.replace("mcl::(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace(", ", ",")  # (one field: no blanks)
        print(f"PMC {short:60s} {counter:36s} {value:18.1f} (n={n})")
