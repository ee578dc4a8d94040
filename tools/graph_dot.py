"""structure of the captured forward + backward hipGraph (RALF_GRAPH_DOT=path, engine.TrainStep._capture): for every kernel node its direct
predecessors and fan-out -- to tell true dependencies from executor artefacts in a rocprofv3 timeline.
    RALF_GRAPH_DOT=/tmp/g.dot python tools/encdec_once.py 1 ; python tools/graph_dot.py /tmp/g.dot [out.txt]"""
import re
import sys
from collections import defaultdict

txt = open(sys.argv[1]).read()
labels = {}
for m in re.finditer(r'"(graph_\d+_node_\d+)"\[[^\]]*?label="\{\s*(\w+)\s*\|\s*\{ID \| (\d+) \| ([^\\<}]*)', txt):
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", m.group(4)).strip()
    grid = re.search(re.escape(m.group(4)) + r"\\<\\<\\<\((\d+),(\d+),(\d+)\)", txt[m.start():m.start() + 600])
    wgs = int(grid.group(1)) * int(grid.group(2)) * int(grid.group(3)) if grid else 0
    labels[m.group(1)] = (int(m.group(3)), m.group(2), name[:48], wgs)
edges = re.findall(r'"(graph_\d+_node_\d+)"\s*->\s*"(graph_\d+_node_\d+)"', txt)
pred, succ = defaultdict(list), defaultdict(list)
for a, b in edges:
    pred[b].append(a)
    succ[a].append(b)
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print(f"# {len(labels)} nodes, {len(edges)} edges; roots: {[labels[n][0] for n in labels if not pred[n]]}", file=out)
for n, (i, kind, name, wgs) in sorted(labels.items(), key=lambda kv: kv[1][0]):
    ps = sorted(labels[p][0] for p in pred[n] if p in labels)
    print(f"{i:4d} {kind[:6]:6s} {wgs:6d} {name:48s} <- {ps}  -> {sorted(labels[c][0] for c in succ[n] if c in labels)}", file=out)
