"""per-kernel statistics (by name and launch geometry) of a rocprofv3 rocpd kernel trace: python tools/prof_kernels.py db [name filter]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
q = ("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(end-start), min(end-start), max(end-start) from kernels "
     "where name like ? group by name, grid_x, grid_y, grid_z order by min(start)")
print(f"{'kernel':90s} {'wgs':>7s} {'calls':>6s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s}")
for name, gx, gy, gz, wx, n, avg, mn, mx in c.execute(q, (f"%{flt}%",)):
    name = re.sub(r"\(anonymous namespace\)::|void ", "", name)
    print(f"{name[:90]:90s} {gx * gy * gz // max(wx, 1):7d} {n:6d} {avg / 1e3:8.1f} {mn / 1e3:8.1f} {mx / 1e3:8.1f}")
