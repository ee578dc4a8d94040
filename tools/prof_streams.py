"""per-queue busy time and the tail where only the side queue runs: python tools/prof_streams.py db"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute(f"select start, end, name, {qcol} from kernels order by start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + (t1 - t0) * 3 // 4
rows = [r for r in rows if r[0] >= cut]
busy = collections.Counter(); n = collections.Counter()
for s, e, _, q in rows:
    busy[q] += e - s; n[q] += 1
span = rows[-1][1] - rows[0][0]
print(f"window {span/1e6:.2f} ms")
for q in busy: print(f"queue {q}: {n[q]} kernels, busy {busy[q]/1e6:.2f} ms")
# intervals where exactly one queue is active
main = max(busy, key=lambda q: n[q])
ev = []
for s, e, _, q in rows:
    ev.append((s, 1, q)); ev.append((e, -1, q))
ev.sort()
act = collections.Counter(); last = ev[0][0]; only = collections.Counter(); both = 0; idle = 0
for t, d, q in ev:
    live = [k for k, v in act.items() if v > 0]
    if len(live) == 1: only[live[0]] += t - last
    elif len(live) >= 2: both += t - last
    else: idle += t - last
    act[q] += d; last = t
print("only:", {k: round(v/1e6, 2) for k, v in only.items()}, "both %.2f ms" % (both/1e6), "idle %.2f ms" % (idle/1e6), "main queue =", main)
