"""Per launch of the LF kernel: the share of the launch's duration its waves are alive (SQ_WAVE_CYCLES, quad-cycles, x 4 / (SQ_WAVES x duration x clock)) -
what the end of a launch (workgroups finishing their last blocks at different times) and its ramp cost.  From a rocprofv3 --kernel-trace --pmc SQ_WAVES
SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE database: lf_alive.py pmc_results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [c[1] for c in db.execute("pragma table_info('counters_collection')")]
print("# columns:", cols)
idc = "dispatch_id" if "dispatch_id" in cols else cols[0]
rows = db.execute(f"select {idc}, kernel_name, counter_name, value, start, end from counters_collection where kernel_name like '%k_reweight_lf_patch%' order by start").fetchall()
by = {}
for d, k, c, v, s, e in rows:
    r = by.setdefault(d, {"start": s, "end": e})
    r[c] = r.get(c, 0.0) + v
print("# launch  dur_us  waves  wave_cycles(quad)  gui_active  clock_GHz  alive_share")
for i, (d, r) in enumerate(sorted(by.items(), key=lambda kv: kv[1]["start"])):
    dur_ns = r["end"] - r["start"]
    gui = r.get("GRBM_GUI_ACTIVE", 0.0)
    clock = gui / 8.0 / dur_ns if gui else 2.2  # (summed over the eight XCDs)
    waves = r.get("SQ_WAVES", 0.0)
    wc = r.get("SQ_WAVE_CYCLES", 0.0)
    alive = wc * 4.0 / (waves * dur_ns * clock) if waves else 0.0
    print(f"{i:3d} {dur_ns / 1e3:8.1f} {waves:7.0f} {wc:14.0f} {gui:12.0f} {clock:6.3f} {alive:6.3f}")
