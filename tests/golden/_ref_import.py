"""Container-only helper: import the reference (read-only at /root/reference) with the
missing third-party import-time names stubbed.  Used ONLY by
make_golden.py to generate fixtures; never imported by tests, smoke() or bench.py and
never present on the GPU box (the reference does not travel)."""
import sys
import types
from unittest.mock import MagicMock

REF = "/root/reference"


def install():
    import datasets  # noqa: F401  must come before torchvision is stubbed
    import torch  # noqa: F401

    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name in [
        "hydra", "hydra.utils", "hydra.conf", "hydra.core", "hydra.core.config_store",
        "timm", "torchvision", "torchvision.models", "torchvision.models.feature_extraction",
        "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
        "cv2", "seaborn", "faiss", "dreamsim", "prdc", "pytorch_fid", "pytorch_fid.fid_score",
    ]:
        if name not in sys.modules:
            sys.modules[name] = MagicMock()

    class DictConfig(dict):
        __getattr__ = dict.get

        def __setattr__(self, k, v):
            self[k] = v

    om = types.ModuleType("omegaconf")
    om.DictConfig = DictConfig
    om.OmegaConf = MagicMock()
    om.open_dict = MagicMock()
    sys.modules["omegaconf"] = om
    return DictConfig
