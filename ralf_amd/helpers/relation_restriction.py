"""Decoding-space restriction for the `relation` task (host integer / set logic).

Behavioural mirror of `TransformerSortByDictRelationConstraint`
(image2layout/train/models/layoutformerpp/relation_restriction.py:354-825; originally LayoutFormer++'s
constrained decoding): for one sample decoded token by token it returns, per step, the vocabulary mask
(True = forbidden) implied by the label order and by the size / location relations to already generated
elements (or to the canvas), plus the position to back-track to when the step turns out infeasible.

Written from the reference's observable behaviour (its interval arithmetic, rounding directions, the quirks
noted inline) as a small table-driven interpreter: every (attribute slot, relation) pair maps to an integer
interval [lo, hi) of admissible bins; intervals of all constraints on the current element are intersected.
"""
from __future__ import annotations

from math import ceil, floor
from typing import List, Optional, Tuple

import torch

from .relationships import REL_SIZE_ALPHA, RELATIVE_RELATION, RelElement, RelLoc, RelSize

TYPE, WIDTH, HEIGHT, CX, CY = range(5)   # attribute slot of a sequence position (position % 5), var_order of the tokenizer
CANVAS = "canvas"


class _State:
    """progress after a given number of decoded tokens"""

    def __init__(self, num_elements: int):
        self.num_elements = num_elements
        self.curr_element = 0
        self.num_bbox = 0
        self.pred_labels: list = []
        self.pred_bbox: list = []   # per element: [w, h, cx, cy] bins decoded so far

    @property
    def finished(self) -> bool:
        return self.curr_element == self.num_elements and self.num_bbox >= 4

    def copy(self) -> "_State":
        """what copy.deepcopy gives for this record (ints and lists of ints), without its generic walk: 20 us per decode step and sample otherwise"""
        c = _State.__new__(_State)
        c.num_elements, c.curr_element, c.num_bbox = self.num_elements, self.curr_element, self.num_bbox
        c.pred_labels = list(self.pred_labels)
        c.pred_bbox = [list(b) for b in self.pred_bbox]
        return c


def _nth_occurrence(types: torch.Tensor, label: torch.Tensor, ordinal: int) -> int:
    return int(torch.nonzero(types == label)[ordinal])


class RelationConstraint:
    def __init__(self, preprocessor):
        self.pre = preprocessor
        tok = preprocessor.tokenizer
        self.nbin = int(tok._num_bin)
        self.canvas = self.nbin - 1            # bins span [0, nbin-1]
        self._token_mask = tok.token_mask
        self.V = self._token_mask.size(-1)
        # first bin token of every geometric slot; the slot's admissible set ends with the two special tokens
        self.start = {}
        for slot in (WIDTH, HEIGHT, CX, CY):
            nz = self._token_mask[slot].nonzero()
            lo, hi = int(nz[0]), int(nz[-3]) + 1
            assert hi - lo == self.nbin
            self.start[slot] = lo
        self.ordinal = {e: i for i, e in enumerate(RelElement)}
        self._eos = int(tok.name_to_id("eos"))
        self._all_bins = (0, self.nbin)   # admissible bins are always ONE run [lo, hi): the full range met with intervals
        self.history: List[_State] = []
        self.types: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------------------------------
    def prepare(self, seq: torch.Tensor) -> list:
        """constraint-encoder sequence of ONE sample -> per element the list of its constraints:
        ("canvas", RelLoc) or (relation, index of the EARLIER element it refers to).  Resets the decode history."""
        pre = self.pre
        eos = int(torch.argmax((seq == pre.name_to_id("eos")).float()))
        rsep = int(torch.argmax((seq == pre.name_to_id("relation_sep")).float()))
        seq = seq[:eos]
        types = seq[3:rsep][::2]               # labels, `sep` between them; [bos, task, end_of_task] in front
        self.types = types
        rel = seq[rsep + 1:]
        rel = rel[rel != pre.name_to_id("sep")].reshape(-1, 5)
        n = types.size(0)
        self.history = [_State(n)]
        out: list = [[] for _ in range(n)]
        for label_i, elem_i, rel_id, label_j, elem_j in rel:
            kind = pre.id_to_name(int(rel_id))
            pi = _nth_occurrence(types, label_i, self.ordinal[pre.id_to_name(int(elem_i))])
            if pre.id_to_name(int(label_j)) == CANVAS:
                out[pi].append((CANVAS, kind))
                continue
            pj = _nth_occurrence(types, label_j, self.ordinal[pre.id_to_name(int(elem_j))])
            if pj > pi:                        # always constrain the LATER element by the earlier one
                pi, pj, kind = pj, pi, RELATIVE_RELATION[kind]
            assert pi > pj, f"{pi=} {pj=} {kind=}"
            out[pi].append((kind, pj))
        self.label_tokens = [pre.name_to_id(name) for name in pre.tokenizer._label_feature.names]
        return out

    # ---- admissible bin interval [lo, hi) of one constraint on one attribute slot (None = unconstrained) --------------
    def _canvas_cy(self, kind, h) -> Tuple[int, int]:
        cs, half = self.canvas, h / 2
        if kind == RelLoc.TOP:
            return ceil(half), floor(cs / 3 - half)
        if kind == RelLoc.CENTER:
            return ceil(cs / 3 + half), floor(2 * cs / 3 - half)
        if kind == RelLoc.BOTTOM:
            return ceil(2 * cs / 3 + half), floor(cs - half)
        raise ValueError(f"Unknown rel_type: {kind}")

    def _interval(self, slot, kind, tgt, cur) -> Optional[Tuple[int, int]]:
        cs, nb = self.canvas, self.nbin
        if slot == CX:
            w = cur[0]
            tw, tcx = tgt[0], tgt[2]
            if kind == RelLoc.LEFT:
                return floor(tcx + tw / 2 + w / 2), ceil(cs - w / 2)
            if kind == RelLoc.RIGHT:
                return floor(w / 2), ceil(tcx - tw / 2 - w / 2)
            if kind == RelLoc.CENTER:
                return ceil(tcx - tw / 2 + w / 2), floor(tcx + tw / 2 - w / 2)
            return floor(w / 2), ceil(cs - w / 2)
        if slot == CY:
            h = cur[1]
            th, tcy = tgt[1], tgt[3]
            if kind == RelLoc.TOP:
                return floor(tcy + th / 2 + h / 2), ceil(cs - h / 2)
            if kind == RelLoc.BOTTOM:
                return floor(h / 2), ceil(tcy - th / 2 - h / 2)
            if kind == RelLoc.CENTER:   # (sic) the reference widens the band by the current height here
                return ceil(tcy - th / 2 - h / 2), floor(tcy + th / 2 + h / 2)
            return floor(h / 2), ceil(cs - h / 2)
        if slot == WIDTH:
            tw, th, tcx = tgt[0], tgt[1], tgt[2]
            area = tw * th
            if kind == RelLoc.LEFT:
                return 0, ceil(cs - tcx - tw / 2)
            if kind == RelLoc.RIGHT:
                return 0, ceil(tcx - tw / 2)
            if kind == RelLoc.CENTER:
                return 0, (floor(cs - tcx + tw / 2) if tcx < nb // 2 else floor(tcx + tw / 2))
            if kind == RelSize.SMALLER:
                area /= 1 - REL_SIZE_ALPHA
                return min(ceil(area / cs), cs), ceil(area)
            if kind == RelSize.LARGER:
                area /= 1 + REL_SIZE_ALPHA
                return 0, floor(area / cs)
            if kind == RelSize.EQUAL:
                return floor(area / (1 + REL_SIZE_ALPHA) / cs), ceil(area / (1 - REL_SIZE_ALPHA) / cs)
            return None
        # HEIGHT
        w = cur[0]
        th, tcy = tgt[1], tgt[3]
        area = tgt[0] * th
        if kind == RelLoc.TOP:
            return 0, ceil(tcy - th / 2)
        if kind == RelLoc.BOTTOM:
            return 0, floor(tcy - th / 2)
        if kind == RelLoc.CENTER:
            return 0, (floor(cs - tcy + th / 2) if tcy < nb // 2 else floor(tcy + th / 2))
        if kind == RelSize.SMALLER:
            area /= 1 - REL_SIZE_ALPHA
            return (cs if w == 0 else min(ceil(area / w), cs)), nb
        if kind == RelSize.LARGER:
            area /= 1 + REL_SIZE_ALPHA
            return 0, (nb if w == 0 else min(floor(area / w), nb))
        if kind == RelSize.EQUAL:
            w = 1 if w == 0 else w
            return floor(area / (1 + REL_SIZE_ALPHA) / w), ceil(area / (1 - REL_SIZE_ALPHA) / w)
        return None

    @staticmethod
    def _meet(allowed, interval):
        """intersection of the admissible run [a, b) with [lo, hi); an EMPTY interval leaves it unchanged (the reference's `_intersect`).
        (The reference filters a set of bins; every set it can reach is a run, the empty one included.)"""
        if interval is None:
            return allowed
        lo, hi = interval
        if hi <= lo:
            return allowed
        a, b = allowed
        a, b = (a if a > lo else lo), (b if b < hi else hi)
        return (a, b) if b > a else (0, 0)

    def _mask_from(self, allowed, slot: int) -> torch.Tensor:
        mask = torch.ones(self.V, dtype=torch.bool)
        a, b = allowed
        if b > a:
            mask[self.start[slot] + int(a):self.start[slot] + int(b)] = False
        return mask

    # ------------------------------------------------------------------------------------------
    def __call__(self, token_ids: torch.Tensor, rel_constraints: list):
        """token_ids [1, L] (bos + L-1 decoded tokens) -> (mask [V] True = forbidden, back-track position or None)"""
        what, back = self.step(token_ids.size(1) - 1, int(token_ids[0, -1]), rel_constraints)
        if what[0] == "only":
            mask = torch.ones(self.V, dtype=torch.bool)
            mask[what[1]] = False
        elif what[0] == "bins":
            mask = self._mask_from(what[2], what[1])
        else:
            mask = ~self._token_mask[what[1]].clone()
        return mask, back

    def step(self, n_decoded: int, last: int, rel_constraints: list):
        """the same step without tensors (the batched decode loop builds the masks of all samples at once): n_decoded tokens are decoded, `last`
        is the latest of them -> (what is admissible, back-track position or None):
          ("only", token)        that one token
          ("bins", slot, (a, b)) the bin tokens start[slot] + a .. start[slot] + b - 1 (an EMPTY run admits nothing)
          ("slot", n_decoded)    whatever the tokenizer admits at this position"""
        self.history = self.history[: n_decoded + 1]
        st = self.history[-1].copy()
        slot = n_decoded % 5
        if n_decoded > 0:
            if last in self.label_tokens:
                st.pred_labels.append(last)
                st.pred_bbox.append([])
            else:   # the token of the previous slot, as a bin index (KeyError for a non-label at a type slot, like the reference)
                st.pred_bbox[-1].append(int(last - self.start[{HEIGHT: WIDTH, CX: HEIGHT, CY: CX, TYPE: CY}[slot]]))
        back = None
        if st.finished:
            return ("only", self._eos), back            # (a finished state is not recorded)
        if slot == TYPE:
            st.curr_element += 1
            st.num_bbox = 0
            what = ("only", int(self.types[n_decoded // 5]))
        else:
            cons = rel_constraints[st.curr_element - 1]
            cur = st.pred_bbox[-1]
            if st.curr_element == 1:     # first element: only canvas constraints, and only on cy
                allowed = self._all_bins
                for kind, arg in cons:
                    if slot == CY and kind == CANVAS:
                        allowed = self._meet(allowed, self._canvas_cy(arg, cur[1]))
                what = ("bins", slot, allowed)
            elif len(cons) == 0:
                what = ("slot", n_decoded)
            else:
                allowed = self._all_bins
                for kind, arg in cons:
                    if kind == CANVAS:
                        back = None      # (the reference overwrites the back-track target with every constraint it visits)
                        if slot != CY:
                            continue
                        allowed = self._meet(allowed, self._canvas_cy(arg, cur[1]))
                        continue
                    back = arg * 5 + st.num_bbox + 1
                    allowed = self._meet(allowed, self._interval(slot, kind, st.pred_bbox[arg], cur))
                what = ("bins", slot, allowed)
            st.num_bbox += 1
        self.history.append(st)
        return what, back
