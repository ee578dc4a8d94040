"""kernel sequence of ONE steady-state step from a rocprofv3 rocpd kernel trace: start offset, duration, gap to the previous kernel on the
same queue, queue, name.   python tools/prof_timeline.py db marker_substring [nth_from_last]
The step is delimited by consecutive occurrences of a kernel whose name contains `marker_substring` (e.g. adamw)."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
marker = sys.argv[2]
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = c.execute(f"select start, end, name, {qcol}, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
# steps end with the last marker kernel of a run of markers
ends = [i for k, i in enumerate(marks) if k + 1 == len(marks) or marks[k + 1] != i + 1]
a, b = ends[-nth - 1] + 1, ends[-nth] + 1
seg = rows[a:b]
t0 = seg[0][0]
lastq = {}
print(f"# {len(seg)} kernels, span {(seg[-1][1] - t0) / 1e3:.1f} us")
for s, e, name, q, gx, gy, gz, wx in seg:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    gap = (s - lastq[q]) / 1e3 if q in lastq else 0.0
    lastq[q] = e
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {gap:6.1f} q{q} wgs {gx * gy * gz // max(wx, 1):6d}  {name[:110]}")
