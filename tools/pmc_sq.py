"""per-kernel shader-instruction mix from ONE rocprofv3 pass, e.g.
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- prog
      python tools/pmc_sq.py db out.txt "header"
Columns are per DISPATCH averages: instruction counts are wave-instructions summed over the chip; cyc = GRBM_GUI_ACTIVE / 8 (one XCD's clock);
valu/simd = SQ_INSTS_VALU / 1024 SIMDs (x ~4 cycles each = the VALU-issue floor of the kernel in cycles)."""
import re
import sqlite3
import sys

db, out, header = sys.argv[1], sys.argv[2], sys.argv[3]
c = sqlite3.connect(db)
per = {}
for k, name, v, n in c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
    d = per.setdefault(k, {"n": 0})
    d[name] = v
    d["n"] = max(d["n"], n)
names = sorted({n for d in per.values() for n in d if n != "n"})
with open(out, "w") as f:
    f.write(header + "\n")
    f.write(f"{'kernel':70s} {'calls':>6s} {'cyc':>9s} " + " ".join(f"{n.replace('SQ_', '')[:13]:>13s}" for n in names if n != 'GRBM_GUI_ACTIVE') + f" {'valu/simd':>10s}\n")
    for k, d in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
        nm = re.sub(r"\(anonymous namespace\)::|void ", "", k)[:70]
        n = d["n"]
        f.write(f"{nm:70s} {n:6d} {d.get('GRBM_GUI_ACTIVE', 0) / 8 / n:9.0f} " + " ".join(f"{d.get(x, 0) / n:13.0f}" for x in names if x != 'GRBM_GUI_ACTIVE')
                + f" {d.get('SQ_INSTS_VALU', 0) / n / 1024:10.0f}\n")
