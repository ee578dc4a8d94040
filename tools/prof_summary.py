"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table (committed under profiles/)."""
import re
import sqlite3
import sys


def short(s):
    s = re.sub(r"\(anonymous namespace\)::", "", s)
    s = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", s)
    return s[:100]


def main(db, out, header, steps, gap_ms=None):
    """gap_ms: keep only the dispatches after the LAST idle gap longer than this (set-up | timed replays, tools/encdec_once.py)"""
    c = sqlite3.connect(db)
    where, wall = "", None
    if gap_ms:
        ts = c.execute("select start, end from kernels order by start").fetchall()
        cut, last_end = ts[0][0], ts[0][1]
        for s0, e0 in ts[1:]:
            if s0 - last_end > gap_ms * 1e6:
                cut = s0
            last_end = max(last_end, e0)
        where = f" where start >= {cut}"
        wall = (last_end - cut) / 1e6
        header += f"\n# window: dispatches after the last idle gap > {gap_ms:g} ms only (graph replays; set-up, eager warm-up and capture excluded): {wall:.2f} ms of wall time = {wall / steps:.3f} ms per step"
    rows = c.execute(f"select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels{where} group by name order by 3 desc").fetchall()
    tot, n = sum(r[2] for r in rows), sum(r[1] for r in rows)
    with open(out, "w") as f:
        f.write(header + "\n")
        f.write(f"# total kernel time {tot / 1e6:.1f} ms over {n} dispatches (= {tot / 1e6 / steps:.2f} ms and {n // steps} dispatches per step, {steps} steps in the trace)\n")
        f.write(f"{'kernel':100s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'pct':>6s}\n")
        for r in rows:
            f.write(f"{short(r[0]):100s} {r[1]:7d} {r[2] / 1e6:9.2f} {r[3] / 1e3:9.1f} {r[4] / 1e3:8.1f} {r[5] / 1e3:9.1f} {r[2] / tot * 100:6.2f}\n")
        f.write("\n# GEMM and transformer-layer dispatches by launch geometry (threads_x, splits, batch)\n")
        gw = (where + " and " if where else " where ") + "(name like '%gemm%kernel%' or name like '%tlayer%' or name like '%attn_bwd_fused%')"
        for r in c.execute(f"select name, grid_x, grid_y, grid_z, count(*), sum(end-start), avg(end-start) from kernels{gw} group by name, grid_x, grid_y, grid_z order by 6 desc limit 60"):
            f.write(f"{short(r[0])[:70]:70s} grid=({r[1]},{r[2]},{r[3]}) calls {r[4]:5d} total_ms {r[5] / 1e6:8.2f} avg_us {r[6] / 1e3:8.1f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), float(sys.argv[5]) if len(sys.argv) > 5 else None)
