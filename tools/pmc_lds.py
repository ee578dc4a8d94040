"""LDS and issue counters per kernel from ONE rocprofv3 pass `--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
--kernel-trace` (rocpd database):  python tools/pmc_lds.py db out.txt "header" [name filter]
lds_active = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 x 256): the fraction of CU-cycles the LDS index path was busy (summed over 256 CUs);
conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: the share of those cycles spent on bank conflicts."""
import re
import sqlite3
import sys


def main(db, out, header, flt):
    c = sqlite3.connect(db)
    per = {}
    for k, name, v, n in c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
        d = per.setdefault(k, {"n": 0})
        d[name] = v
        d["n"] = max(d["n"], n)
    with open(out, "w") as f:
        f.write(header + "\n")
        f.write(f"{'kernel':70s} {'calls':>6s} {'lds_active':>10s} {'conflict':>9s} {'lds_insts/call':>15s} {'valu_insts/call':>16s} {'gui_active/call':>16s}\n")
        for k, d in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
            if flt and not re.search(flt, k):
                continue
            g, act, bc = d.get("GRBM_GUI_ACTIVE", 0.0), d.get("SQ_LDS_IDX_ACTIVE", 0.0), d.get("SQ_LDS_BANK_CONFLICT", 0.0)
            name = re.sub(r"\(anonymous namespace\)::|void ", "", k)[:70]
            n = max(d["n"], 1)
            f.write(f"{name:70s} {d['n']:6d} {100 * act / max(g / 8 * 256, 1):9.1f}% {100 * bc / max(act, 1):8.1f}% {d.get('SQ_INSTS_LDS', 0.0) / n:15.0f} {d.get('SQ_INSTS_VALU', 0.0) / n:16.0f} {g / n / 8:16.0f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
