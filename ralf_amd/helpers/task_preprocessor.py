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
import random

import torch

from .relationships import RelElement, RelLoc, RelSize
from .task import VARS

TASK_TOKENS = ["end_of_task", "label", "label_size", "relationship", "refinement", "completion", "uncondition"]
SEP_TOKENS = ["sep", "relation_sep", "canvas"]
N_REL_LOC, N_REL_SIZE, MAX_REL_ELEMENTS = 6, 4, 11  # RelLoc, RelSize, RelElement (helpers/relationships.py:10-52)


class BasePreprocessor:
    TASK = ""
    VAR: list = []
    SHUFFLE = False

    def __init__(self, tokenizer, global_task_embedding: bool = False):
        # the reference's constraint vocabulary has one element marker per slot taken from the 11-member RelElement enum
        # (task_preprocessor.py:65-68), so it cannot be built for max_seq_length > 11.  Longer layouts (the <= 32-element
        # measurement config) get anonymous marker ids beyond K: every task except `relation` is unaffected.
        self.tokenizer = tokenizer
        self.global_task_embedding = global_task_embedding
        n_extra = len(TASK_TOKENS) + len(SEP_TOKENS) + tokenizer.max_seq_length + N_REL_LOC + N_REL_SIZE
        self.N_total = tokenizer.N_total + n_extra
        self._ids = dict(tokenizer._special_token_name_to_id)
        for i, name in enumerate(TASK_TOKENS + SEP_TOKENS):
            self._ids[name] = tokenizer.N_total + i
        # relationship vocabulary: element markers A.. (one per element slot), then RelLoc, then RelSize members; the enum
        # MEMBERS are the keys, like the reference's table entries (task_preprocessor.py:65-87,119-124)
        n_el = tokenizer.max_seq_length
        markers = list(RelElement)[:n_el] + [f"element_{i}" for i in range(MAX_REL_ELEMENTS, n_el)]
        rel_tokens = markers + list(RelLoc) + list(RelSize)
        for i, member in enumerate(rel_tokens):
            self._ids[member] = tokenizer.N_total + len(TASK_TOKENS) + len(SEP_TOKENS) + i
        for i, name in enumerate(tokenizer._label_feature.names):
            self._ids[name] = i
        self._names = {v: k for k, v in self._ids.items()}
        self.device = torch.device("cpu")

    def name_to_id(self, name):
        return self._ids[name]

    def id_to_name(self, idx: int):
        return self._names[idx]

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


class _RefEnumUnpickler:
    """pickle module shim for torch.load: the authors' table pickles the reference's enum classes
    (image2layout.train.helpers.relationships.*); map them onto ralf_amd.helpers.relationships."""
    import pickle as _pk

    class Unpickler(_pk.Unpickler):
        def find_class(self, module, name):
            if module.endswith("helpers.relationships"):
                from . import relationships
                return getattr(relationships, name)
            return super().find_class(module, name)

    load = staticmethod(lambda f, **kw: _RefEnumUnpickler.Unpickler(f, **kw).load())
    __name__ = "pickle"


def _load_relationship_table(path: str) -> dict:
    return torch.load(path, pickle_module=_RefEnumUnpickler, weights_only=False)


class RelationshipPreprocessor(BasePreprocessor):
    """Gen-R (task_preprocessor.py:488-606): [bos, relationship, end_of_task, l1 | l2 | ..., relation_sep,
    (label_i, elem_i, relation, label_j, elem_j) sep ... eos] with RELATION_SIZE % of the sample's relations drawn by
    `random.sample`.

    `table`: dict str(data id) -> list of [label name, RelElement, RelLoc | RelSize, label name | "canvas", RelElement | "pad"]
    (helpers.relationships.relationship_table builds one from layouts), or a path to the authors' torch-saved table.  Like
    the reference, every list is shuffled once at construction with Python's global `random`."""
    TASK, VAR = "relationship", VARS["relation"]
    TABLE_PATH = "cache/pku_cgl_relationships_dic_using_canvas_sort_label_lexico.pt"

    def __init__(self, tokenizer, global_task_embedding: bool = False, RELATION_SIZE: int = 10, table=None):
        super().__init__(tokenizer, global_task_embedding)
        if tokenizer.max_seq_length > MAX_REL_ELEMENTS:
            raise ValueError(f"relation task: max_seq_length={tokenizer.max_seq_length} > {MAX_REL_ELEMENTS} element markers (RelElement)")
        self.RELATION_SIZE = RELATION_SIZE
        if table is None or isinstance(table, str):
            table = _load_relationship_table(table or self.TABLE_PATH)
        self.table = {k: random.sample(v, len(v)) for k, v in table.items()}
        self.label_preprocessor = LabelPreprocessor(tokenizer, global_task_embedding)

    def set_relation_size(self, RELATION_SIZE: int) -> None:
        self.RELATION_SIZE = RELATION_SIZE

    def __call__(self, cond):
        ids = cond.id.cpu().tolist() if torch.is_tensor(cond.id) else cond.id
        relations = [self.table[str(i)] for i in ids]
        self.split_vars(cond.seq, True)          # consumes the element-shuffle randomness the reference spends here (result unused)
        lab = self.label_preprocessor(cond)      # ... and shuffles again for the sequence that is kept
        seq_label, pad_mask = lab["seq"], lab["pad_mask"]
        self.device = seq_label.device
        B = seq_label.size(0)
        if not self.global_task_embedding:
            seq_label[:, 1] = self._ids[self.TASK]
        seq_label[seq_label == self._ids["eos"]] = self._ids["relation_sep"]
        rows = []
        for b in range(B):
            head = seq_label[b][~pad_mask[b]]
            if len(relations[b]) == 0:
                rows.append(torch.cat([head, self.tok("eos", 1)[0]]))
                continue
            k = max(len(relations[b]) * self.RELATION_SIZE // 100, 1)
            picked = random.sample(relations[b], k)
            rel = torch.tensor([[self._ids[e] for e in r] for r in picked], device=self.device)
            rel = torch.cat([rel, self.tok("sep", rel.size(0))], dim=1).view(-1)
            rel[-1] = self._ids["eos"]
            rows.append(torch.cat([head, rel]))
        # the output width is the longest row AMONG SAMPLES THAT HAVE RELATIONS (task_preprocessor.py:559-585)
        longest = max((r.size(0) for r, rl in zip(rows, relations) if len(rl) > 0), default=-1)
        if longest < 0:
            raise ValueError("relation task: no sample of the batch has relations (the reference fails here as well)")
        seq = torch.full((B, longest), self._ids["pad"], dtype=torch.long, device=self.device)
        for b, r in enumerate(rows):
            seq[b, : r.size(0)] = r
        return {"seq": seq, "pad_mask": seq == self._ids["pad"]}


PREPROCESSOR = {
    None: UnconditionalPreprocessor, "none": UnconditionalPreprocessor, "uncond": UnconditionalPreprocessor,
    "c": LabelPreprocessor, "cwh": LabelSizePreprocessor, "partial": PartialPreprocessor,
    "refinement": RefinementPreprocessor, "relation": RelationshipPreprocessor,
}
