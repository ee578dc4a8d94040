"""which kernels surround a given kernel in a rocprofv3 rocpd trace: python tools/prof_neighbors.py db pattern"""
import sqlite3, sys, collections, re
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
rows = c.execute("select start, end, name, grid_x from kernels order by start").fetchall()
def short(s): return re.sub(r"\(anonymous namespace\)::|void ", "", s)[:60]
cnt = collections.Counter(); dur = collections.Counter()
for i, r in enumerate(rows):
    if pat in r[2] and 0 < i < len(rows) - 1:
        k = (short(rows[i - 1][2]), short(rows[i + 1][2]), r[3])
        cnt[k] += 1; dur[k] += r[1] - r[0]
for k, n in cnt.most_common(25):
    print(f"{n:5d} {dur[k] / n / 1e3:7.1f}us grid_x={k[2]:<9d} prev={k[0]:60s} next={k[1]}")
