"""HBM traffic summary for bench.py from two rocprofv3 PMC passes (separate runs, one counter each):
python tools/pmc_traffic_json.py OUT.json KEY FETCH_DB WRITE_DB LAUNCHES "command" [kernel-name filter]
  LAUNCHES = launches of the workload in each trace (train steps, or dispatches of the filtered kernel)
Adds / replaces entry KEY of OUT.json: raw FETCH_SIZE / WRITE_SIZE bytes per launch (bench.py applies the gfx950 x2 read correction)."""
import json
import os
import sqlite3
import sys

out, key, fdb, wdb, launches, cmd = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5]), sys.argv[6]
flt = sys.argv[7] if len(sys.argv) > 7 else ""


def total(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name = ? and kernel_name like ? "
                     "group by kernel_name order by 2 desc", (counter, f"%{flt}%")).fetchall()
    return sum(r[1] for r in rows), sum(r[2] for r in rows), [(r[0][:90], r[1], r[2]) for r in rows[:12]]


f, nf, topf = total(fdb, "FETCH_SIZE")
w, nw, topw = total(wdb, "WRITE_SIZE")
if flt:   # per dispatch of the filtered kernel
    launches_f, launches_w = nf, nw
else:
    launches_f = launches_w = launches
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
entry = {"fetch_size_bytes": f * 1024 / launches_f, "write_size_bytes": w * 1024 / launches_w, "launches_in_trace": launches_f, "command": cmd,
         "note": "raw rocprofv3 counters (KB x 1024) per launch; readers apply the gfx950 correction FETCH_SIZE x 2 for wide coalesced reads",
         "top_fetch_kernels_KB_per_trace": topf, "top_write_kernels_KB_per_trace": topw}
data = {}
if os.path.exists(out):
    with open(out) as fh:
        data = json.load(fh)
data[key] = entry
with open(out, "w") as fh:
    json.dump(data, fh, indent=1)
print(key, "fetch GB/launch (x2 corrected)", 2 * entry["fetch_size_bytes"] / 1e9, "write GB/launch", entry["write_size_bytes"] / 1e9)
