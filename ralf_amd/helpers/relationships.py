"""Pairwise size / location relationships between layout elements (host integer path, `relation` task).

Behavioural mirror of image2layout/train/helpers/relationships.py:10-166 (enums, detectors,
`compute_relation`) and of the table producer image2layout/preprocess/precompute_relationship.py:32-126
(`describe_relationships`): the reference ships neither tests nor the pre-computed table
(cache/pku_cgl_relationships_dic_using_canvas_sort_label_lexico.pt), so `relationship_table` below
re-creates a table of the same shape from a batch of layouts with the same rules.

The enum VALUES are observable (bit positions of `edge_attributes`, vocabulary order of the constraint
encoder) and therefore identical to the reference's.
"""
from __future__ import annotations

import random
from enum import IntEnum
from itertools import combinations

import torch

GEO_KEYS = ["center_x", "center_y", "width", "height"]


class RelSize(IntEnum):
    UNKNOWN = 0
    SMALLER = 1
    EQUAL = 2
    LARGER = 3


class RelLoc(IntEnum):
    UNKNOWN = 4
    LEFT = 5
    TOP = 6
    RIGHT = 7
    BOTTOM = 8
    CENTER = 9


class RelElement(IntEnum):  # "the i-th element carrying this label": A = first, B = second, ...
    A = 10
    B = 11
    C = 12
    D = 13
    E = 14
    F = 15
    G = 16
    H = 17
    I = 18  # noqa: E741
    J = 19
    K = 20


# relation seen from the other element
RELATIVE_RELATION = {
    RelLoc.LEFT: RelLoc.RIGHT, RelLoc.RIGHT: RelLoc.LEFT, RelLoc.TOP: RelLoc.BOTTOM, RelLoc.BOTTOM: RelLoc.TOP,
    RelLoc.CENTER: RelLoc.CENTER, RelLoc.UNKNOWN: RelLoc.UNKNOWN,
    RelSize.SMALLER: RelSize.LARGER, RelSize.LARGER: RelSize.SMALLER, RelSize.EQUAL: RelSize.EQUAL, RelSize.UNKNOWN: RelSize.UNKNOWN,
}

REL_SIZE_ALPHA = 0.1


def _ltrb(box):
    xc, yc, w, h = box
    return xc - w / 2, yc - h / 2, xc + w / 2, yc + h / 2


def detect_size_relation(b1, b2) -> RelSize:
    """size of box 2 relative to box 1 (xywh): EQUAL inside a +-10 % area band."""
    a1, a2 = b1[2] * b1[3], b2[2] * b2[3]
    if (1 - REL_SIZE_ALPHA) * a1 < a2 < (1 + REL_SIZE_ALPHA) * a1:
        return RelSize.EQUAL
    return RelSize.LARGER if a1 < a2 else RelSize.SMALLER


def detect_loc_relation_between_elements(b1, b2) -> RelLoc:
    """where box 2 lies relative to box 1; vertical separation is tested before horizontal, overlap = CENTER."""
    l1, t1, r1, bt1 = _ltrb(b1)
    l2, t2, r2, bt2 = _ltrb(b2)
    if bt2 <= t1:
        return RelLoc.TOP
    if bt1 <= t2:
        return RelLoc.BOTTOM
    if r2 <= l1:
        return RelLoc.LEFT
    if r1 <= l2:
        return RelLoc.RIGHT
    return RelLoc.CENTER


def detect_loc_relation_between_element_and_canvas(box) -> RelLoc:
    yc = box[1]
    if yc < 1.0 / 3:
        return RelLoc.TOP
    if yc < 2.0 / 3:
        return RelLoc.CENTER
    return RelLoc.BOTTOM


def compute_relation(batch: dict, edge_ratio: float = 0.1) -> dict:
    """random subset (probability `edge_ratio` per pair, Python's global `random`, pairs in combinations order) of the
    relations between the canvas (node 0) and the elements (nodes 1..S): edge list + bit-coded attributes."""
    B, S = batch["label"].shape
    geo = {k: torch.cat([torch.full((B, 1), 0.5 if k.startswith("center") else 1.0), batch[k]], dim=1) for k in GEO_KEYS}
    n_nodes = batch["mask"].sum(dim=1) + 1   # + canvas
    E = (S + 1) * (S + 2) // 2
    unknown = (1 << RelSize.UNKNOWN) | (1 << RelLoc.UNKNOWN)
    edge_indexes = torch.full((B, E, 2), -1, dtype=torch.long)
    edge_attributes = torch.full((B, E), unknown, dtype=torch.long)
    for b in range(B):
        n, cnt = int(n_nodes[b]), 0
        for i, j in combinations(range(S + 1), 2):
            if n <= i or n <= j:
                continue
            if random.random() > edge_ratio:
                continue
            bi, bj = [geo[k][b][i] for k in GEO_KEYS], [geo[k][b][j] for k in GEO_KEYS]
            loc = detect_loc_relation_between_element_and_canvas(bj) if i == 0 else detect_loc_relation_between_elements(bi, bj)
            edge_indexes[b, cnt, 0], edge_indexes[b, cnt, 1] = i, j
            edge_attributes[b, cnt] = (1 << detect_size_relation(bi, bj)) | (1 << loc)
            cnt += 1
    return {"edge_indexes": edge_indexes, "edge_attributes": edge_attributes}


def relationship_table(batch: dict, label_names) -> dict:
    """data id -> list of [label_i, RelElement_i, relation, label_j | "canvas", RelElement_j | "pad"]: every element
    pair (location, then size) and every element-canvas location, elements visited last to first -- the shape and
    order of the authors' pre-computed table (precompute_relationship.py:57-126), which `RelationshipPreprocessor`
    samples from."""
    out = {}
    B = batch["label"].size(0)
    for b in range(B):
        labels, masks = batch["label"][b].tolist(), batch["mask"][b].tolist()
        seen: dict = {}
        unique = []
        for lab, m in zip(labels, masks):
            if not m:
                unique.append(None)
                continue
            seen[lab] = seen.get(lab, 0) + 1
            unique.append([label_names[lab], list(RelElement)[seen[lab] - 1]])
        valid = [i for i, m in enumerate(masks) if m][::-1]
        box = lambda i: [batch[k][b, i].item() for k in GEO_KEYS]  # noqa: E731
        loc_pairs, size_pairs, canvas = [], [], []
        for pos, i in enumerate(valid):
            for j in valid[pos + 1:]:
                loc_pairs.append([*unique[i], detect_loc_relation_between_elements(box(i), box(j)), *unique[j]])
                size_pairs.append([*unique[i], detect_size_relation(box(i), box(j)), *unique[j]])
            canvas.append([*unique[i], detect_loc_relation_between_element_and_canvas(box(i)), "canvas", "pad"])
        out[batch["id"][b]] = loc_pairs + size_pairs + canvas
    return out
