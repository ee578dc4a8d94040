"""`RetrievalAugmentation` -- the retrieval block shared by the reference's *_ra baselines (CGL-GAN-RA, DS-GAN-RA,
LayoutDM-RA).  Mirror of image2layout/train/models/common/retrieval_augment.py:18-101: same
constructor keywords, `preprocess_retrieved_samples`, `forward(image_backbone, img_feature, retrieved_layouts)` and
state_dict layout (`layout_encoder.*`, `pos_emb_1d.pe`, `layout_adapter.net.*`, `attn.*`, `head.net.*`), so a checkpoint
of a reference *_ra generator's `retrieval_augment` sub-module loads with strict=True.

It is the same computation as rows a5-a7 of the RALF generator, on the same HIP kernels: frozen layout encoder over all
B*K exemplars in one batch -> adapter FFN -> *sqrt(d) + PE[0:K] -> cross-attention from the image tokens -> head FFN
over [image tokens | attended | exemplars].  The baselines' own bodies (GAN / diffusion heads) are out of scope.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as RF
from .. import nn as RN
from ..functional import Runtime
from .ralf import _DTYPES

LAYOUT_KEYS = ("label", "mask", "center_x", "center_y", "width", "height")


class RetrievalAugmentation(nn.Module):
    def __init__(self, *, d_model: int, dataset_name: str, top_k: int, num_classes: int, max_seq_length: int, use_reference_image: bool,
                 compute_dtype="float32", pretrained: bool = True):
        """pretrained (extension; default = the reference): the frozen layout encoder is the trained FIDNetV3 of `dataset_name`, loaded from the
        reference's locations (fid/model.py:131-175); pretrained=False keeps random weights (tests)"""
        super().__init__()
        if use_reference_image:
            raise NotImplementedError("use_reference_image=True (exemplar IMAGES through the backbone) is not on the accelerated path")
        self.top_k, self.use_reference_image, self.d_model = top_k, use_reference_image, d_model
        self.dataset_name, self.max_seq_length = dataset_name, max_seq_length
        self.rt = Runtime(_DTYPES[compute_dtype] if isinstance(compute_dtype, str) else compute_dtype)
        self.layout_encoder = RN.LayoutEncoder(num_classes)       # load_fidnet_feature_extractor: frozen, decoder side deleted
        if pretrained:
            self.layout_encoder.load_fidnet(dataset_name)
        for p in self.layout_encoder.parameters():
            p.requires_grad = False
        self.pos_emb_1d = RN.PosEnc1d(d_model, 5000)
        self.layout_adapter = RN.FeedForward(256, 4 * d_model, d_model)
        self.attn = RN.FuseAttention(d_model, d_model, heads=8, dim_head=64)
        self.head = RN.FeedForward(d_model, 4 * d_model)

    def train(self, mode: bool = True):
        super().train(mode)
        self.rt.training = mode
        return self

    def preprocess_retrieved_samples(self, retrieved):
        """retrieval_augment.py:54-66: unwrap the one-element list of the non-training collate, RGB + saliency -> 4 channels"""
        if isinstance(retrieved, list):
            assert len(retrieved) == 1
            retrieved = retrieved[0]
        retrieved["image"] = torch.cat([retrieved["image"], retrieved["saliency"]], dim=2)
        assert retrieved["image"].size(2) == 4, f"{retrieved['image'].shape=}"
        return retrieved

    def forward(self, image_backbone, img_feature: torch.Tensor, retrieved_layouts: dict) -> torch.Tensor:
        """img_feature [B, hw, d] (compute dtype, on the GPU) -> memory [B, 2*hw + K, d]"""
        rt = self.rt.to(img_feature.device)
        K = self.top_k
        B = retrieved_layouts["label"].shape[0]
        flat = {k: retrieved_layouts[k][:, :K].reshape(B * K, -1).to(img_feature.device) for k in LAYOUT_KEYS}
        f = self.layout_encoder.extract_features(flat, rt)                   # [B*K, 256], no grad
        f = self.layout_adapter(f, rt).view(B, K, -1)
        ref = RF.ScalePEDropFn.apply(f, self.pos_emb_1d.pe[0, :K].contiguous(), self.d_model ** 0.5, rt.drop_p(self.pos_emb_1d.p), rt)
        ca = self.attn(img_feature, ref, rt)
        return self.head(RF.concat_rows([img_feature.contiguous(), ca, ref], rt), rt)
