"""Retrieval with the reference's signature (src/openmatch/retriever/dense_retriever.py:37-97):
`distributed_parallel_retrieve(args, topk) -> {qid: {docid: score}}` over the pickle shards in
`args.output_dir`, this rank's query shard(s) against ALL corpus shards.

The per-shard `torch.matmul` + `torch.topk` (dense_retriever.py:13-34) is replaced by the
HBM-resident HipIndex: corpus shards are appended to one device index and searched with the
fused bf16-MFMA similarity + bitonic top-k kernel, fp32 re-scored.  Like the reference, the
default result is the UNION of the per-shard top-k lists (up to k * n_shards docs per query,
dense_retriever.py:79-92 — `save_as_trec` writes all of them); `global_topk=True` keeps only the
global top-k (one search over the concatenated index: the fast path when only depth k is
evaluated).

`sharded_search` is the MI355X multi-GPU path of BASELINE.json: corpus rows sharded across
ranks, local top-k per rank, ONE RCCL all-gather of the packed [nq, k] (score, global id) words,
on-device merge."""
from __future__ import annotations

import logging
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from .engine import HipIndex, topk_merge
from .utils import list_shards, read_shard

logger = logging.getLogger(__name__)


def _device_index(args) -> int:
    """args.device (reference: dense_retriever.py:23, `.to(args.device)`) -> cuda index; a bare
    'cuda' / missing device means this process's own GPU (LOCAL_RANK, else the current device)."""
    from .modeling import _device_index as _idx, default_device
    dev = getattr(args, "device", None)
    if dev is None:
        lr = getattr(args, "local_rank", None)
        return int(lr) if lr is not None and int(lr) >= 0 else default_device()
    i = _idx(dev)
    return default_device() if i is None else i


def _load_queries(args) -> Tuple[np.ndarray, List[str]]:
    parts = list_shards(args.output_dir, "query", args.process_index)
    logger.info("query_all_partitions = %s", parts)
    reps, ids = [], []
    for p in parts:
        r, i = read_shard(p)
        if len(i) == 0:
            continue
        reps.append(r)
        ids.extend(i)
    if not reps:
        raise ValueError("No pre-computed query embeddings found")
    return np.concatenate(reps), ids


def distributed_parallel_retrieve(args, topk: int, global_topk: bool = False) -> Dict[str, Dict[str, float]]:
    queries, qids = _load_queries(args)
    corpus_parts = list_shards(args.output_dir, "corpus")
    if len(corpus_parts) == 0:
        raise ValueError("No pre-computed document embeddings found")
    logger.info("corpus_all_partitions = %s", corpus_parts)
    dev = _device_index(args)
    result: Dict[str, Dict[str, float]] = {q: {} for q in qids}
    dim = queries.shape[1]
    if not global_topk:
        for p in corpus_parts:
            reps, ids = read_shard(p)
            if len(ids) == 0:
                continue
            ix = HipIndex(dim, len(ids), dev)
            ix.add(reps)
            sc, idx = ix.search(queries, topk)
            ix.close()
            for qi, q in enumerate(qids):
                for s, j in zip(sc[qi], idx[qi]):
                    if j >= 0:
                        result[q][ids[int(j)]] = float(s)
        return result
    shards = [read_shard(p) for p in corpus_parts]
    total = sum(len(i) for _, i in shards)
    ix = HipIndex(dim, max(total, 1), dev)
    all_ids: List[str] = []
    for reps, ids in shards:
        if len(ids):
            ix.add(reps)
            all_ids.extend(ids)
    sc, idx = ix.search(queries, topk)
    ix.close()
    for qi, q in enumerate(qids):
        for s, j in zip(sc[qi], idx[qi]):
            if j >= 0:
                result[q][all_ids[int(j)]] = float(s)
    return result


def pack_topk(scores: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """[nq, k] (fp32 score, global id < 2^31) -> ONE int64 per entry: score bits << 32 | id.  An empty
    slot (id -1) keeps id bits 0xFFFFFFFF.  8 bytes per entry: 80 KB per rank at nq = 1000, k = 10."""
    bits = scores.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    return (bits << 32) | (ids.to(torch.int64) & 0xFFFFFFFF)


def unpack_topk(packed: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    sc = (packed >> 32).to(torch.int32).view(torch.float32)          # arithmetic shift, then truncation: exact bits
    ids = (packed & 0xFFFFFFFF)
    ids = torch.where(ids == 0xFFFFFFFF, torch.full_like(ids, -1), ids)
    return sc, ids


def sharded_search(index, queries: torch.Tensor, k: int, id_offset: int = 0, group=None,
                   local_search: Optional[Callable] = None, merge: Optional[Callable] = None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Every rank holds `index` = its corpus shard (row j has global id id_offset + j) and the
    same `queries` [nq, dim].  Returns the global (scores, ids) [nq, k] on every rank.

    One exchange step: ONE all_gather_into_tensor of the packed [nq, k] (score, id) words (RCCL
    over xGMI on GPUs), then the on-device merge.  `local_search(queries, k)` / `merge(all_sc,
    all_ids)` default to the HipIndex search and vr_topk_merge; the CPU (gloo) test injects the
    oracle's search and `merge_topk_host` so that THIS function is what runs under world_size 2."""
    sc, ids = (local_search or index.search)(queries, k)
    if not isinstance(sc, torch.Tensor):
        sc, ids = torch.from_numpy(np.ascontiguousarray(sc)), torch.from_numpy(np.ascontiguousarray(ids))
    ids = torch.where(ids >= 0, ids + id_offset, ids)
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sc, ids
    if int(ids.max()) >= 2 ** 31 - 1:
        raise ValueError("global row ids must stay below 2^31 - 1 for the packed exchange")
    world = dist.get_world_size(group)
    mine = pack_topk(sc, ids)
    nq = mine.shape[0]
    dev = mine.device
    if mine.is_cuda and dist.get_backend(group) == "gloo":
        mine = mine.cpu()        # gloo rendezvous (no RCCL: e.g. two ranks sharing one GPU in the tests): exchange on the host
    gathered = torch.empty((world * nq,) + tuple(mine.shape[1:]), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine, group=group)             # the one collective of the path
    gathered = gathered.to(dev)
    all_sc, all_ids = unpack_topk(gathered.view((world, nq) + tuple(mine.shape[1:])))
    return (merge or topk_merge)(all_sc.contiguous(), all_ids.contiguous())


def merge_topk_host(all_sc: np.ndarray, all_ids: np.ndarray, k: int):
    """Host-side statement of the merge rule (score desc, id asc) used by the gloo CPU tests."""
    P, nq, kk = all_sc.shape
    sc = np.transpose(all_sc, (1, 0, 2)).reshape(nq, P * kk)
    ids = np.transpose(all_ids, (1, 0, 2)).reshape(nq, P * kk)
    sc = np.where(ids >= 0, sc, -np.inf)
    order = np.lexsort((ids, -sc), axis=1)[:, :k]
    return np.take_along_axis(sc, order, 1), np.take_along_axis(ids, order, 1)
