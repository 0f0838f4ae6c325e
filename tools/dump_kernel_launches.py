"""Every launch of the kernels whose name contains <pattern>, in order, with the kernels right before it: dump_kernel_launches.py trace_results.db <pattern>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = None
for q in ("select name, start, end from kernels order by start", "select kernel_name, start, end from kernels order by start"):
    try:
        rows = db.execute(q).fetchall()
        break
    except sqlite3.Error:
        continue


def short(n):
    return n.replace("mcl::(anonymous namespace)::", "").split("(")[0].replace("void ", "")


rows = [(short(n), s, e) for n, s, e in rows]
for i, (n, s, e) in enumerate(rows):
    if sys.argv[2] in n:
        before = " <- ".join(f"{rows[j][0]} {(rows[j][2] - rows[j][1]) / 1e3:.1f}" for j in range(i - 1, max(i - 4, -1), -1))
        print(f"{i:5d} {n:32s} {(e - s) / 1e3:10.1f} us   after: {before}")
