"""kernels around the big host-to-device copies of the LAST sample() call in a rocprofv3 --kernel-trace --memory-copy-trace database:
python3 tools/prof_copies.py DB OUT"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
mc = [t for t in tabs if t.startswith("rocpd_memory_copy")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({mc})")]
kern = sorted((s, e, n) for n, s, e in db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id"))
cop = sorted((s, e, b) for s, e, b in db.execute(f"select start, end, size from {mc}") if b > 50e6)
big = cop[-2:] if len(cop) >= 2 and cop[-1][0] - cop[-2][0] < 10e6 else cop[-1:]
t0 = big[0][0]
lo, hi = t0 - 2e6, big[-1][1] + 4e6
with open(sys.argv[2], "w") as f:
    f.write(f"# the last call's image copies ({len(big)}) and the kernels from 2 ms before the first to 4 ms after the last; t = 0 at the first copy's start\n")
    ev = [(s, e, f"COPY {b / 1e6:.1f} MB") for s, e, b in big] + [(s, e, "K " + n[:70]) for s, e, n in kern if lo <= s <= hi]
    ev.sort()
    for s, e, n in ev:
        f.write(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  {n}\n")
    last_before = max((e for s, e, n in kern if e <= t0), default=None)
    if last_before:
        f.write(f"# last kernel before the first copy ended at {(last_before - t0) / 1e3:.1f} us\n")
