"""per-queue view of the last window of a rocprofv3 kernel trace: busy time, idle time between consecutive dispatches, the largest idle gaps with the kernels on
either side (which launches the main chain waits for):  python tools/prof_queues.py trace.db out.txt [gap_ms=300] [steps=5]"""
import re
import sqlite3
import sys


def short(s):
    s = re.sub(r"\(anonymous namespace\)::", "", s)
    s = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", s)
    return re.sub(r"\(.*", "", s)[:60]


db, out = sys.argv[1], sys.argv[2]
gap_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute(f"select name, start, end, {qcol or 0} from kernels order by start").fetchall()
cut, last_end = 0, rows[0][2]
for i, (_, s0, e0, _) in enumerate(rows[1:], 1):
    if s0 - last_end > gap_ms * 1e6:
        cut = i
    last_end = max(last_end, e0)
rows = rows[cut:]
t0, t1 = rows[0][1], max(r[2] for r in rows)
with open(out, "w") as f:
    f.write(f"# columns of `kernels`: {cols}\n# window: {len(rows)} dispatches, {(t1 - t0) / 1e6:.3f} ms = {(t1 - t0) / 1e6 / steps:.3f} ms per step; queue column: {qcol}\n")
    qs = {}
    for r in rows:
        qs.setdefault(r[3], []).append(r)
    for q, rs in sorted(qs.items(), key=lambda kv: -len(kv[1])):
        busy = sum(r[2] - r[1] for r in rs)
        gaps = []
        prev = rs[0]
        for r in rs[1:]:
            gaps.append((r[1] - prev[2], prev, r))
            prev = r if r[2] > prev[2] else prev
        idle = sum(max(0, g[0]) for g in gaps)
        f.write(f"\nqueue {q}: {len(rs)} dispatches, busy {busy / 1e6 / steps:.3f} ms per step, idle between its dispatches {idle / 1e6 / steps:.3f} ms per step "
                f"(gaps < 3 us: {sum(max(0, g[0]) for g in gaps if g[0] < 3000) / 1e6 / steps:.3f} ms in {sum(1 for g in gaps if 0 <= g[0] < 3000) // steps} per step)\n")
        for g, a, b in sorted(gaps, key=lambda x: -x[0])[:25]:
            f.write(f"   idle {g / 1e3:8.1f} us between {short(a[0])} and {short(b[0])}\n")
