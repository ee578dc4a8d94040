"""Constructor-time weight files, resolved and loaded the way the reference constructors do it.

  * ResNet-50 body: `resnet50_a1_0-14fe96d1.pth` in the working directory, else under `./cache/PRECOMPUTED_WEIGHT_DIR`
    (image2layout/train/models/common/image.py:38-48, global_variables.py:21); a missing file is an AssertionError as in the reference.
  * frozen layout encoder: `tmp/fidnet/<dataset>/model_best.pth.tar`, else `./cache/PRECOMPUTED_WEIGHT_DIR/fidnet/<dataset>/model_best.pth.tar`,
    `pku` -> `pku10` (image2layout/train/fid/model.py:131-175); the file holds {"state_dict": FIDNetV3 weights}; the reference loads them
    strictly into the full FIDNetV3 and then deletes the decoder-side modules, so exactly those keys may be dropped here.

Both files are pickles written by the authors.  They are read with torch.load(weights_only=True); a checkpoint that carries other Python
objects needs RALF_TRUST_CHECKPOINTS=1 (the reference's plain torch.load, i.e. arbitrary unpickling).
"""
from __future__ import annotations

import logging
import os

import torch

logger = logging.getLogger(__name__)

PRECOMPUTED_WEIGHT_DIR = "./cache/PRECOMPUTED_WEIGHT_DIR"   # global_variables.py:21 (relative to the working directory, like the reference)
RESNET50_FILE = "resnet50_a1_0-14fe96d1.pth"
FIDNET_CKPT_DIR = "tmp/fidnet"
FIDNET_FILE = "model_best.pth.tar"
# FIDNetV3 modules that load_fidnet_feature_extractor deletes after the strict load (fid/model.py:168-173)
FIDNET_DROPPED = ("pos_token", "dec_transformer.", "fc_out_disc.", "fc_out_cls.", "fc_out_bbox.")


def _torch_load(path: str):
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:   # noqa: BLE001 -- the unpickler's error types vary between torch versions
        if os.environ.get("RALF_TRUST_CHECKPOINTS") == "1":
            return torch.load(path, map_location="cpu", weights_only=False)
        raise RuntimeError(f"{path}: not loadable with torch.load(weights_only=True) ({e}); set RALF_TRUST_CHECKPOINTS=1 to unpickle it "
                           "without restrictions, as the reference's torch.load does") from e


def resnet50_weight_path() -> str:
    """common/image.py:40-45"""
    path = RESNET50_FILE
    if not os.path.exists(path):
        path = os.path.join(PRECOMPUTED_WEIGHT_DIR, RESNET50_FILE)
        assert os.path.exists(path), f"{path} does not exist"
    return path


def load_resnet50_state() -> dict:
    path = resnet50_weight_path()
    sd = _torch_load(path)
    logger.info("Load resnet50: %s", path)
    return sd


def fidnet_weight_path(dataset_name: str, ckpt_dir: str = FIDNET_CKPT_DIR) -> str:
    """fid/model.py:131-140,164-166"""
    if dataset_name == "pku":
        dataset_name = "pku10"
    path = os.path.join(ckpt_dir, dataset_name, FIDNET_FILE)
    if not os.path.exists(path):
        path = os.path.join(PRECOMPUTED_WEIGHT_DIR, "fidnet", dataset_name, FIDNET_FILE)
    return path


def load_fidnet_state(dataset_name: str, ckpt_dir: str = FIDNET_CKPT_DIR) -> dict:
    path = fidnet_weight_path(dataset_name, ckpt_dir)
    if not os.path.exists(path):
        raise FileNotFoundError(f"FIDNetV3 checkpoint {path} does not exist (looked under {ckpt_dir} and {PRECOMPUTED_WEIGHT_DIR}/fidnet)")
    logger.info("Loading FIDNetV3 (weight_path=%s) ...", path)
    return _torch_load(path)["state_dict"]


def fidnet_encoder_state(state_dict: dict, own_keys) -> dict:
    """the part of a full FIDNetV3 state dict that survives load_fidnet_feature_extractor, checked as strictly as the reference's
    load_state_dict: every surviving key present, nothing unknown"""
    own = set(own_keys)
    keep = {k: v for k, v in state_dict.items() if k in own}
    missing = sorted(own - set(keep))
    unknown = sorted(k for k in state_dict if k not in own and not k.startswith(FIDNET_DROPPED))
    if missing or unknown:
        raise RuntimeError(f"FIDNetV3 checkpoint does not match the layout encoder: missing {missing[:5]}, unexpected {unknown[:5]}")
    return keep
