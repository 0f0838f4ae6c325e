"""Per-launch values of the counters of a rocprofv3 --pmc database for the kernels whose name contains argv[2] (default k_reweight_lf), in launch order."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "k_reweight_lf"
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
key = "dispatch_id" if "dispatch_id" in cols else "id"
rows = db.execute(f"select {key}, kernel_name, counter_name, sum(value) from counters_collection group by {key}, kernel_name, counter_name order by {key}").fetchall()
by = collections.OrderedDict()
for d, k, c, v in rows:
    if pat in k:
        by.setdefault(d, {})[c] = v
names = sorted({c for v in by.values() for c in v})
print("launch " + " ".join(f"{n:>22s}" for n in names))
for i, (d, v) in enumerate(by.items()):
    print(f"{i:6d} " + " ".join(f"{v.get(n, float('nan')):22.0f}" for n in names))
