"""Writes the HBM-traffic record bench.py reads (profiles/lf_kernel_traffic.json) from two rocprofv3 PMC databases
(FETCH_SIZE and WRITE_SIZE passes of the default bench).  Usage: make_traffic_record.py fetch.db write.db out.json [particles]"""
import hashlib, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter(db_path, name):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (name,)).fetchall()
    rows = [r for r in rows if "k_reweight_lf_patch" in r[0]] or [r for r in rows if "k_reweight_lf_palette" in r[0]]
    assert rows, f"no LF kernel in {db_path}"
    return rows[0][0], rows[0][1], rows[0][2]


kernel, fetch, n1 = counter(sys.argv[1], "FETCH_SIZE")
_, write, n2 = counter(sys.argv[2], "WRITE_SIZE")
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the key of the record: SHA-256 of the LF kernels' source)

sha = bench.lf_kernel_source_sha(os.path.join(ROOT, "beluga_amd", "csrc", "kernels.hip"))
rec = {"kernel": kernel.replace("void ", "").replace("mcl::(anonymous namespace)::", "").split("(")[0], "particles": int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000,
       "fetch_size_kb": fetch, "write_size_kb": write, "launches_averaged": [n1, n2], "kernels_hip_sha256": sha,
       "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python bench.py`; HBM bytes per launch = "
               "2 x FETCH_SIZE + WRITE_SIZE (gfx950 reports half of every read: profiles/r01_pmc_traffic_calibration.txt); "
               "kernels_hip_sha256 = SHA-256 of the LF kernels' source, the part of kernels.hip between its [lf-kernels-begin] and "
               "[lf-kernels-end] markers (bench.lf_kernel_source_sha)"}
with open(sys.argv[3], "w") as fh:
    json.dump(rec, fh, indent=1)
print(rec)
