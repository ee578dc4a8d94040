"""per-dispatch counter values in dispatch order: python tools/pmc_list.py db COUNTER [filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); flt = sys.argv[3] if len(sys.argv) > 3 else ""
for k, d, v in c.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? and kernel_name like ? group by dispatch_id order by dispatch_id", (sys.argv[2], f"%{flt}%")):
    print(d, k[-60:], v)
