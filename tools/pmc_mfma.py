"""MFMA utilisation per kernel from ONE rocprofv3 pass `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace`
(rocpd database):  python tools/pmc_mfma.py db out.txt "header" nsteps
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024): MFMA_BUSY is summed over the chip's 1024 SIMDs, GRBM_GUI_ACTIVE
over the 8 XCDs (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_32x32x16_bf16)."""
import re
import sqlite3
import sys


def main(db, out, header, steps):
    c = sqlite3.connect(db)
    per = {}
    for k, name, v, n in c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
        d = per.setdefault(k, {"n": 0})
        d[name] = v
        d["n"] = max(d["n"], n)
    tot_m = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in per.values())
    tot_g = sum(d.get("GRBM_GUI_ACTIVE", 0.0) for d in per.values())
    tot_b = sum(d.get("SQ_BUSY_CYCLES", 0.0) for d in per.values())
    with open(out, "w") as f:
        f.write(header + "\n")
        f.write(f"# whole trace ({steps} steps): SQ_VALU_MFMA_BUSY_CYCLES {tot_m:.4g}, GRBM_GUI_ACTIVE {tot_g:.4g}, SQ_BUSY_CYCLES {tot_b:.4g} -> mfma_busy = "
                f"{tot_m / max(tot_g / 8 * 1024, 1):.4f} of the SIMD-cycles while a kernel was running (eager, kernels back to back on one stream)\n")
        f.write(f"# per step: {tot_m / steps:.4g} MFMA-busy cycles = {tot_m / steps / 32:.4g} v_mfma_f32_32x32x16_bf16 equivalents = {tot_m / steps / 32 * 32768 / 1e12:.3f} TFLOP issued on the matrix cores\n")
        f.write(f"{'kernel':86s} {'calls':>6s} {'mfma_busy_cyc':>15s} {'gui_active_cyc':>15s} {'mfma_util':>9s} {'time_share':>10s}\n")
        for k, d in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
            m, g = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
            name = re.sub(r"\(anonymous namespace\)::|void ", "", k)[:86]
            f.write(f"{name:86s} {d['n']:6d} {m:15.0f} {g:15.0f} {100 * m / max(g / 8 * 1024, 1):8.2f}% {100 * g / max(tot_g, 1):9.2f}%\n")
    return tot_m / max(tot_g / 8 * 1024, 1), tot_m / steps


if __name__ == "__main__":
    frac, cyc = main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]))
    print("mfma_busy", frac)
    if len(sys.argv) > 6:      # json file, key: the figure bench.py quotes in its roofline note
        import json
        import os
        path, key = sys.argv[5], sys.argv[6]
        d = json.load(open(path)) if os.path.exists(path) else {}
        d[key] = {"mfma_busy": frac, "mfma_busy_cycles_per_step": cyc, "source": os.path.basename(sys.argv[2]), "command": sys.argv[3].lstrip("# ")}
        json.dump(d, open(path, "w"), indent=1)
