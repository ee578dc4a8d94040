"""GPU: ENDURANCE of the captured train step.  The fast path is a stack of hipGraphs (forward + backward, the parameter-gradient
side graph behind hand-added event nodes or branches of the main graph, clip + AdamW) kept per batch-shape signature; the equality tests elsewhere
compare the forms over a handful of steps.  Here: 2 000 replays that alternate three shape signatures (full batch, the ragged last batch of an
epoch -- train/train.py:166 --, a shorter constraint sequence -- helpers/task_preprocessor.py: kmax = n_valid.max()), the Python collector enabled
and forced between steps, TrainStep objects dropped and re-created (re-captured) mid-run, in BOTH forms of the side work; losses equal to a twin
model stepped WITHOUT graphs at fixed checkpoints, finite throughout, and no growth of device memory once every shape has been captured."""
import gc
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _slice(tree, n, lc=None):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out[k] = _slice(v, n)
        elif torch.is_tensor(v) and v.dim() > 0:
            v = v[:n]
            if lc is not None and k in ("seq_layout_const", "seq_layout_const_pad_mask"):
                v = v[:, :lc]
            out[k] = v.contiguous()
        else:
            out[k] = v
    return out


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("form", ["auto", "1", "0"])   # RALF_SIDE_GRAPH: both forms captured and timed, the faster kept (the default) / the parameter gradients as a second graph on the side stream / as branches of the main graph
def test_two_thousand_replays_over_three_shapes(form, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    from ralf_amd.engine import TrainStep
    from ralf_amd.synthetic import make_batch, to_device

    monkeypatch.setenv("RALF_SIDE_GRAPH", form)
    dev = torch.device("cuda", 0)
    B, N, HW, STEPS, TWIN = 4, 10, 128, 2000, 300
    models = []
    for _ in range(2):
        torch.manual_seed(5)
        m = bench.build_model(dev, N, "bfloat16", task="c")
        m.rt.drop_p = lambda p: 0.0               # train mode (BatchNorm on batch statistics) without dropout: the twin sees the same arithmetic
        models.append(m)
    graphed_m, eager_m = models
    i, t = graphed_m.preprocess(make_batch(B, N, H=HW, W=HW, seed=3))
    i, t = to_device(i, dev), to_device(t, dev)
    i["retrieved"] = {k: v for k, v in i["retrieved"].items() if k != "image"}
    Lc = i["seq_layout_const"].shape[1]
    assert Lc >= 3
    shapes = [(i, t), (_slice(i, B - 1), _slice(t, B - 1)), (_slice(i, B, Lc - 1), t)]
    order = [0, 0, 1, 0, 2, 2, 0, 1]              # the batch shapes of consecutive steps (every pair of neighbours occurs)

    def new_step():
        s = TrainStep(graphed_m, lr=1e-4, weight_decay=1e-4, max_norm=0.1, use_graph=True)
        s.max_graph_shapes = 3
        return s
    step, twin = new_step(), TrainStep(eager_m, lr=1e-4, weight_decay=1e-4, max_norm=0.1, use_graph=False)
    assert gc.isenabled()
    losses, mem = [], {}
    for k in range(STEPS):
        b = shapes[order[k % len(order)]]
        loss = step(*b)
        if k < TWIN:
            ref = twin(*b)
            if k in (0, 1, 7, 50, 150, TWIN - 1):
                a, r = float(loss), float(ref)
                assert abs(a - r) <= 3e-2 * max(1.0, abs(r)), (form, k, a, r)
        if k % 100 == 0 or k == STEPS - 1:
            losses.append(float(loss))
            assert losses[-1] == losses[-1], (form, k)                       # not NaN
        if k % 37 == 0:
            gc.collect()                                                      # a collector pass between steps (never inside a capture: engine._capture)
        if k in (500, 1100):                                                  # the step object dropped mid-run: its graphs go, a new one captures again
            assert step.eager_fallbacks == 0 and len(step._by_shape) + 1 >= 3
            del step
            gc.collect()
            step = new_step()
        if k in (700, 1300, STEPS - 1):
            torch.cuda.synchronize()
            mem[k] = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
    torch.cuda.synchronize()
    print(f"RALF_SIDE_GRAPH={form}: losses {[round(x, 3) for x in losses]}; side graph chosen: {getattr(step, 'side_graph', None)}; memory (allocated, reserved) {mem}")
    assert step.eager_fallbacks == 0 and step.captures == 3
    assert losses[-1] < losses[0] - 0.5, losses                               # it trained all the way (the same four samples: the loss falls)
    # no growth between equivalent points of the run (all three shapes captured in each of them): allocated bytes to 2 %, the reserved pool may not grow
    # ("auto" decides per capture by timing: the two forms hold 1.9 / 2.3 GB of graph memory here, so a flipped choice moves the level, not a leak)
    tol = 1.30 if form == "auto" else 1.02
    assert mem[STEPS - 1][0] <= tol * mem[700][0] and mem[1300][0] <= tol * mem[700][0], mem
    assert mem[STEPS - 1][1] <= (1.30 if form == "auto" else 1.05) * mem[1300][1], mem
