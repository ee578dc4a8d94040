import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """npz fixture with '/'-separated nested keys -> nested access returning torch tensors."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name))
        self.keys = list(self._z.keys())

    def __getitem__(self, key):
        import torch

        return torch.from_numpy(self._z[key])

    def sub(self, prefix):
        import torch

        prefix = prefix.rstrip("/") + "/"
        out = {}
        for k in self.keys:
            if k.startswith(prefix):
                rest = k[len(prefix):]
                node = out
                parts = rest.split("/")
                for p in parts[:-1]:
                    node = node.setdefault(p, {})
                node[parts[-1]] = torch.from_numpy(self._z[k])
        return out


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


@pytest.fixture(scope="session", autouse=True)
def _dist_cleanup():
    """a test may create a 1-rank RCCL group (tests/test_engine_gpu.py): shut it down once, at the end of the session"""
    yield
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    except Exception:
        pass


def free_port() -> int:
    """a TCP port nobody listens on right now (rendezvous of the multi-process tests): asked from the kernel, not derived from the pid --
    a fixed scheme met a lingering listener once in a few hundred runs (EADDRINUSE)"""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
