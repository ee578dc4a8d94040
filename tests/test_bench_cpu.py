"""CPU: the parts of bench.py that do not need a GPU -- the self-launcher's clean failure on a node with too few GPUs, and the
reader of the committed PMC traffic summaries (bench.py never carries a hard-coded traffic figure)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_gpus_n_without_enough_gpus_fails_with_a_message_not_a_traceback():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs 2 GPUs" in r.stderr and "Traceback" not in r.stderr
    assert r.stdout.strip() == ""            # stdout is reserved for the one JSON line


def test_traffic_comes_from_the_newest_committed_profile(tmp_path, monkeypatch):
    import bench

    tb, entry = bench.measured_traffic("train_step_B64_N10_bf16")
    assert tb and entry["file"].startswith("profiles/") and entry["file"].endswith("_hbm_traffic.json")
    assert tb == 2.0 * entry["fetch_size_bytes"] + entry["write_size_bytes"]      # gfx950: FETCH_SIZE counts half of a wide read
    assert 20e9 < tb < 80e9
    kb, ke = bench.measured_traffic("knn_scores_nq16_61548x1792")
    assert 0.43e9 < kb < 0.50e9 and ke["file"] == entry["file"]                   # 441.2 MB algorithmic, read once
    assert bench.measured_traffic("no_such_workload") == (None, None)
    # the newest file by name wins
    prof = tmp_path / "profiles"
    prof.mkdir()
    for name, fetch in (("r03a_hbm_traffic.json", 1.0e9), ("r02z_hbm_traffic.json", 9.0e9)):
        (prof / name).write_text(json.dumps({"train_step_B64_N10_bf16": {"fetch_size_bytes": fetch, "write_size_bytes": 5.0e8}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    tb2, e2 = bench.measured_traffic("train_step_B64_N10_bf16")
    assert tb2 == 2.5e9 and e2["file"].endswith("r03a_hbm_traffic.json")


def test_committed_rocprof_average_reader():
    """roofline_knn quotes the HIP-event time AND the rocprofv3 average of the same kernel from the newest committed summary"""
    import bench

    avg, mn, f = bench.committed_kernel_avg_us(r"knn_scores_kernel<16, *1, *2")
    assert f and f.startswith("profiles/") and f.endswith("_knn_kernel_stats.txt")
    assert 60.0 < mn <= avg < 120.0          # 441 MB at 3.7 ... 7.3 TB/s
    assert bench.committed_kernel_avg_us(r"no_such_kernel") == (None, None, None)


def test_the_package_pins_the_hardware_queue_count_before_the_device_is_touched():
    """GPU_MAX_HW_QUEUES (ralf_amd/__init__.py): set to 4 at import when it is UNSET; a value the user exported is respected (warning) unless
    RALF_FORCE_HW_QUEUES=1 (what bench.py, a launcher of its own, sets for itself); HW_QUEUES reports the effective value"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, warnings; warnings.simplefilter('ignore'); import ralf_amd; "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'), ralf_amd.HW_QUEUES['set_by_package'], ralf_amd.HW_QUEUES['effective'])")
    for extra, want in (({}, "4 True 4"), ({"GPU_MAX_HW_QUEUES": "8"}, "8 False 8"), ({"GPU_MAX_HW_QUEUES": "8", "RALF_FORCE_HW_QUEUES": "1"}, "4 True 4")):
        env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "RALF_FORCE_HW_QUEUES")}
        env.update(extra, PYTHONPATH=root)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and out.stdout.strip() == want, (extra, out.stdout, out.stderr[-500:])


def test_bench_compiles_without_warnings_and_its_cpu_baseline_leg_runs():
    """bench.py is only exercised end to end on the GPU box: catch on the CPU what can be caught -- the file compiles with warnings as errors
    (an implicit string concatenation next to a parenthesis once turned a JSON field into a call), and the cpu_baseline leg (oracle train
    steps, the one part of the bench that needs no GPU) runs at a tiny size"""
    import warnings

    with open(os.path.join(ROOT, "bench.py")) as f:
        src = f.read()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        compile(src, "bench.py", "exec")
    sys.path.insert(0, ROOT)
    import bench

    r = bench.cpu_baseline_train(N=10, B=2, steps=1, warm=0, hw=64, B_autoreg=2, steps_autoreg=1, warm_autoreg=0)
    assert r["value"] > 0 and r["kind"] == "port" and "BOUND" in r["sample"] and r["autoreg_baseline"]["value"] > 0
