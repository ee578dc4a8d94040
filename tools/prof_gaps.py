"""busy time (union of kernel intervals), idle gaps and concurrency from a rocprofv3 rocpd kernel trace: python tools/prof_gaps.py db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select start, end, name from kernels order by start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
# restrict to the last third of the trace (steady-state graph replays)
cut = t0 + (t1 - t0) * 2 // 3
rows = [r for r in rows if r[0] >= cut]
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]; gaps = []
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = cur_e - rows[0][0]
tot = sum(e - s for s, e, _ in rows)
print(f"window {span/1e6:.2f} ms: busy(union) {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms, sum of kernel times {tot/1e6:.2f} ms, {len(rows)} kernels, {len(gaps)} gaps")
gaps.sort(reverse=True)
print("largest gaps (us):", [round(g[0] / 1e3, 1) for g in gaps[:15]])
import collections
h = collections.Counter(min(int(g[0] / 1e3), 20) for g in gaps)
print("gap histogram (us -> count):", sorted(h.items()))
print("total of gaps < 20us: %.2f ms" % (sum(g[0] for g in gaps if g[0] < 20e3) / 1e6))
