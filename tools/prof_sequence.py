"""the launch SEQUENCE of the last window of a rocprofv3 kernel trace (dispatches after the last idle gap > gap_ms), with the idle time in front of each:
    python tools/prof_sequence.py trace.db out.txt [gap_ms=100] [max_rows=0]"""
import re
import sqlite3
import sys


def short(s):
    s = re.sub(r"\(anonymous namespace\)::", "", s)
    return re.sub(r"\(.*", "", s)[:70]


db, out = sys.argv[1], sys.argv[2]
gap_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
max_rows = int(sys.argv[4]) if len(sys.argv) > 4 else 0
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall()
cut, last_end = 0, rows[0][2]
for i, (_, s0, e0, _) in enumerate(rows[1:], 1):
    if s0 - last_end > gap_ms * 1e6:
        cut = i
    last_end = max(last_end, e0)
rows = rows[cut:]
with open(out, "w") as f:
    f.write(f"# {len(rows)} dispatches, wall {(max(r[2] for r in rows) - rows[0][1]) / 1e6:.3f} ms, kernel time {sum(r[2] - r[1] for r in rows) / 1e6:.3f} ms\n")
    f.write(f"{'#':>5s} {'t_us':>9s} {'idle_us':>8s} {'dur_us':>8s} {'grid_x':>9s}  kernel\n")
    t0, prev = rows[0][1], rows[0][1]
    for i, (n, s0, e0, gx) in enumerate(rows):
        if max_rows and i >= max_rows:
            break
        f.write(f"{i:5d} {(s0 - t0) / 1e3:9.1f} {(s0 - prev) / 1e3:8.1f} {(e0 - s0) / 1e3:8.1f} {gx:9d}  {short(n)}\n")
        prev = max(prev, e0)
