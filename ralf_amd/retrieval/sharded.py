"""Multi-GPU forms of the exact top-k search: one process per GPU, `torch.distributed` (RCCL on the GPUs,
gloo in the CPU tests).

  * query-sharded replicas (default: the 441 MB index fits every GPU): rank r searches a contiguous block of the queries
    against its own full copy; `all_gather` only hands every rank the whole table -- no collective on the data path;
  * index-sharded: rank r holds rows [offset_r, offset_r + n_r) of the index, every rank scans ALL queries against its rows,
    the local top-k lists (score, GLOBAL index; nq * k * 12 bytes per rank) are all-gathered and merged with the scan's own
    order (score descending, index ascending) -- only worthwhile for indexes far beyond 60 k rows.

`local_search(queries, k) -> (scores [nq, k], idx [nq, k])` is `FlatIPIndex.search` on the GPU; the tests inject a NumPy scan.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


def query_block(nq: int, rank: int, world: int) -> slice:
    """contiguous block of the queries owned by `rank` (sizes differ by at most one)"""
    base, extra = divmod(nq, world)
    lo = rank * base + min(rank, extra)
    return slice(lo, lo + base + (1 if rank < extra else 0))


def merge_topk(scores: torch.Tensor, idx: torch.Tensor, k: int):
    """[nq, m] candidate lists (entries with idx < 0 are padding) -> best k per row by (score desc, index asc)"""
    s = torch.where(idx < 0, torch.full_like(scores, float("-inf")), scores)
    # stable two-key ordering: sort by index first, then stably by descending score
    o1 = torch.argsort(torch.where(idx < 0, torch.full_like(idx, torch.iinfo(torch.int64).max), idx), dim=1, stable=True)
    s1, i1 = torch.gather(s, 1, o1), torch.gather(idx, 1, o1)
    o2 = torch.argsort(s1, dim=1, descending=True, stable=True)
    return torch.gather(s1, 1, o2)[:, :k].contiguous(), torch.gather(i1, 1, o2)[:, :k].contiguous()


def search_query_sharded(local_search: Callable, queries: torch.Tensor, k: int, group=None):
    """every rank returns the full (scores [nq, k], idx [nq, k]) table; each searched only its block of the queries"""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nq = queries.shape[0]
    blk = query_block(nq, rank, world)
    s, i = local_search(queries[blk], k) if blk.stop > blk.start else (queries.new_empty(0, k), torch.empty(0, k, dtype=torch.int64, device=queries.device))
    width = -(-nq // world)
    ps = torch.full((width, k), float("-inf"), dtype=torch.float32, device=s.device)
    pi = torch.full((width, k), -1, dtype=torch.int64, device=s.device)
    ps[: s.shape[0]], pi[: i.shape[0]] = s, i
    gs = [torch.empty_like(ps) for _ in range(world)]
    gi = [torch.empty_like(pi) for _ in range(world)]
    dist.all_gather(gs, ps, group=group)
    dist.all_gather(gi, pi, group=group)
    out_s = torch.cat([gs[r][: query_block(nq, r, world).stop - query_block(nq, r, world).start] for r in range(world)])
    out_i = torch.cat([gi[r][: query_block(nq, r, world).stop - query_block(nq, r, world).start] for r in range(world)])
    return out_s, out_i


def search_index_sharded(local_search: Callable, row_offset: int, queries: torch.Tensor, k: int, group=None):
    """`local_search` scans this rank's rows (local indices); returns the GLOBAL top-k on every rank"""
    world = dist.get_world_size(group)
    s, i = local_search(queries, k)
    i = torch.where(i >= 0, i + row_offset, i)          # a shard smaller than k pads with (-inf, -1)
    gs = [torch.empty_like(s) for _ in range(world)]
    gi = [torch.empty_like(i) for _ in range(world)]
    dist.all_gather(gs, s.contiguous(), group=group)
    dist.all_gather(gi, i.contiguous(), group=group)
    return merge_topk(torch.cat(gs, dim=1), torch.cat(gi, dim=1), k)
