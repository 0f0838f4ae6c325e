"""Per-dispatch PMC values of one kernel from a rocprofv3 rocpd database, in dispatch order (the LF kernel cycle by cycle)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_reweight_lf_patch"
cols = [r[1] for r in db.execute("PRAGMA table_info(counters_collection)")]
if "--cols" in sys.argv:
    print(cols)
order = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else cols[0])
rows = db.execute(f"select {order}, counter_name, sum(value) from counters_collection where kernel_name like ? group by {order}, counter_name order by {order}",
                  (f"%{kernel}%",)).fetchall()
names = sorted({r[1] for r in rows})
by = {}
for d, name, v in rows:
    by.setdefault(d, {})[name] = v
print("dispatch " + " ".join(f"{n:>24s}" for n in names))
for k, d in enumerate(sorted(by)):
    print(f"{k:8d} " + " ".join(f"{by[d].get(n, float('nan')):24.0f}" for n in names))
