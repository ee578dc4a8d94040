"""Constraint serialisation for the user-constraint encoder (host integer path).

Follows image2layout/train/models/layoutformerpp/task_preprocessor.py:54-622: the constraint
vocabulary extends the layout tokenizer's by 7 task tokens, 3 separators, N element markers
(RelElement A..), 6 location and 4 size relations; sequences are
  [bos, TASK, end_of_task, v1 | v2 | ... , eos, pad...]   with `sep` between elements.
Element shuffling (tasks c / partial) consumes torch.randperm exactly like the reference, so a
seeded run reproduces the reference's sequences.
"""
from __future__ import annotations

import copy

import torch

from .task import VARS

TASK_TOKENS = ["end_of_task", "label", "label_size", "relationship", "refinement", "completion", "uncondition"]
SEP_TOKENS = ["sep", "relation_sep", "canvas"]
N_REL_LOC, N_REL_SIZE, MAX_REL_ELEMENTS = 6, 4, 11  # RelLoc, RelSize, RelElement (helpers/relationships.py:10-52)


class BasePreprocessor:
    TASK = ""
    VAR: list = []
    SHUFFLE = False

    def __init__(self, tokenizer, global_task_embedding: bool = False):
        if tokenizer.max_seq_length > MAX_REL_ELEMENTS:
            raise ValueError(f"max_seq_length={tokenizer.max_seq_length} > {MAX_REL_ELEMENTS}: the reference's constraint vocabulary "
                             "has only 11 element markers (task_preprocessor.py:65-68)")
        self.tokenizer = tokenizer
        self.global_task_embedding = global_task_embedding
        n_extra = len(TASK_TOKENS) + len(SEP_TOKENS) + tokenizer.max_seq_length + N_REL_LOC + N_REL_SIZE
        self.N_total = tokenizer.N_total + n_extra
        self._ids = dict(tokenizer._special_token_name_to_id)
        for i, name in enumerate(TASK_TOKENS + SEP_TOKENS):
            self._ids[name] = tokenizer.N_total + i
        for i, name in enumerate(tokenizer._label_feature.names):
            self._ids[name] = i
        self.device = torch.device("cpu")

    def name_to_id(self, name):
        return self._ids[name]

    def tok(self, name, batch):
        return torch.full((batch, 1), self._ids[name], dtype=torch.long, device=self.device)

    def _head(self, B):
        parts = [self.tok("bos", B)]
        if not self.global_task_embedding:
            parts += [self.tok(self.TASK, B), self.tok("end_of_task", B)]
        return parts

    def __call__(self, cond):
        return self.serialise(cond, self.SHUFFLE)

    def split_vars(self, seq: torch.Tensor, shuffle: bool) -> dict:
        """[B, 5N+1] (with bos) -> per-variable [B, N]; eos -> pad IN PLACE like the reference."""
        pad, eos = self._ids["pad"], self._ids["eos"]
        seq[seq == eos] = pad
        t = seq[:, 1:].reshape(seq.size(0), -1, 5).permute(0, 2, 1)  # [B, 5, N]
        if shuffle:
            counts = (t != pad).sum(dim=2)[..., 0]
            perms = [torch.randperm(int(n)) for n in counts]
            out = t.clone()
            for i, perm in enumerate(perms):
                out[i, :, : len(perm)] = t[i][:, perm]
            t = out
        return {name: t[:, i] for i, name in enumerate(self.tokenizer.var_order)}

    def serialise(self, cond, shuffle: bool) -> dict:
        pad, eos = self._ids["pad"], self._ids["eos"]
        v = self.split_vars(cond.seq, shuffle)
        self.device = v["label"].device
        B = v["label"].size(0)
        n_valid = ((v["label"] != pad) & (v["label"] != eos)).sum(dim=1)
        nv, kmax = len(self.VAR), int(n_valid.max())
        n_tok = nv * n_valid
        head = 1 if self.global_task_embedding else 3
        total = head + 1 + n_tok + torch.div(n_tok - 1, nv, rounding_mode="floor")  # + eos + separators
        body_len = (nv + 1) * kmax - 1
        body = torch.stack([*[v[k][:, :kmax] for k in self.VAR], self.tok("sep", B).repeat(1, kmax)], dim=2).view(B, -1)[:, :-1]
        assert body.size(1) == body_len
        valid = torch.arange(body_len, device=self.device).unsqueeze(0) < (total.unsqueeze(-1) - (head + 1))
        body = torch.where(valid, body, torch.full_like(body, pad))
        seq = torch.cat(self._head(B) + [body, self.tok("pad", B)], dim=1)
        seq.scatter_(1, total.unsqueeze(-1) - 1, eos)
        return {"seq": seq, "pad_mask": seq == pad}


class UnconditionalPreprocessor(BasePreprocessor):
    TASK = "uncondition"

    def __call__(self, cond):
        B = cond.image.size(0)
        self.device = cond.image.device
        seq = torch.cat(self._head(B) + [self.tok("eos", B)], dim=-1)
        return {"seq": seq, "pad_mask": seq == self._ids["pad"]}


class LabelPreprocessor(BasePreprocessor):
    TASK, VAR, SHUFFLE = "label", VARS["c"], True


class LabelSizePreprocessor(BasePreprocessor):
    TASK, VAR = "label_size", VARS["cwh"]

    def __call__(self, cond):
        assert cond.task == "cwh"
        return self.serialise(cond, False)


class RefinementPreprocessor(BasePreprocessor):
    TASK, VAR = "refinement", VARS["refinement"]

    def __call__(self, cond):
        assert cond.task == "refinement"
        return self.serialise(cond, False)


class PartialPreprocessor(BasePreprocessor):
    TASK, VAR = "completion", VARS["partial"]

    def __call__(self, cond):
        assert cond.task == "partial"
        cond = copy.deepcopy(cond)
        cond.seq[~cond.mask] = self._ids["pad"]
        return self.serialise(cond, True)


class RelationshipPreprocessor(BasePreprocessor):
    TASK, VAR = "relationship", VARS["relation"]

    def __init__(self, *a, **k):
        raise NotImplementedError("relation task needs the authors' relationship table (cache/pku_cgl_relationships_dic_...pt)")


PREPROCESSOR = {
    None: UnconditionalPreprocessor, "none": UnconditionalPreprocessor, "uncond": UnconditionalPreprocessor,
    "c": LabelPreprocessor, "cwh": LabelSizePreprocessor, "partial": PartialPreprocessor,
    "refinement": RefinementPreprocessor, "relation": RelationshipPreprocessor,
}
