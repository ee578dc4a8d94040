"""sum a PMC counter over all dispatches of a rocprofv3 rocpd database, grouped by kernel: python tools/pmc_total.py db COUNTER nsteps"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1]); name = sys.argv[2]; steps = float(sys.argv[3])
rows = c.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name = ? group by kernel_name order by 2 desc", (name,)).fetchall()
tot = sum(r[1] for r in rows)
print(f"# {name}: total {tot:.4g} over {sum(r[2] for r in rows)} dispatches = {tot / steps:.4g} per step ({steps:g} steps)")
for k, v, n in rows[:25]:
    k = re.sub(r"\(anonymous namespace\)::|void ", "", k)[:80]
    print(f"{k:80s} {n:6d} {v / steps:14.1f}")
