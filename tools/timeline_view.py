"""condensed view of a tools/prof_timeline.py listing: one line per kernel, indented by hardware queue
    python tools/timeline_view.py gpurun_out/TAG/encdec_timeline.txt [t0 t1]"""
import re
import sys

rows = [l.rstrip() for l in open(sys.argv[1])][1:]
t0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
ev = []
for l in rows:
    m = re.match(r"\s*([\d.]+)\s+([\d.]+) gap\s+(-?[\d.]+) q(\d+) wgs\s+(\d+)\s+(.*)", l)
    s, d, g, q, w, n = m.groups()
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = re.sub(r"void ", "", n)
    ev.append((float(s), float(d), int(q), int(w), n[:44]))
qs = sorted(set(e[2] for e in ev))
print("queue busy us:", {q: round(sum(e[1] for e in ev if e[2] == q)) for q in qs}, "span", round(ev[-1][0] + ev[-1][1]))
for e in ev:
    if t0 <= e[0] <= t1:
        print(f"{e[0]:8.1f} {e[1]:6.1f} {'          ' * qs.index(e[2])}q{e[2]} {e[3]:5d} {e[4]}")
