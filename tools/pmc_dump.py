"""print per-kernel PMC counter averages from a rocprofv3 rocpd database: python tools/pmc_dump.py db [name-filter]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t == "counters_collection"]
rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (f"%{flt}%",)).fetchall()
agg = collections.defaultdict(dict)
for k, n, v, d in rows:
    agg[k][n] = v / max(d, 1)
for k, d in agg.items():
    print(k[:110])
    for n, v in sorted(d.items()):
        print(f"    {n:36s} {v:16.0f}")
