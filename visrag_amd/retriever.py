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
ranks, local top-k per rank emitted as packed 64-bit (score, global id) keys by the search's own
merge kernel, ONE RCCL all-gather of those [nq, k] words, on-device merge of the gathered buffer."""
from __future__ import annotations

import logging
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from .engine import HipIndex, topk_merge_keys
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


def sharded_search(index, queries: torch.Tensor, k: int, id_offset: int = 0, group=None,
                   local_search_keys: Optional[Callable] = None, merge_keys: Optional[Callable] = None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Every rank holds `index` = its corpus shard (row j has global id id_offset + j) and the
    same `queries` [nq, dim].  Returns the global (scores, ids) [nq, k] on every rank.

    The data path is three library calls and one collective, no tensor arithmetic in between:
      1. `index.search_keys(queries, k, id_offset)` — the local fused search, whose merge kernel writes
         each result as ONE 64-bit word, orderable(score) << 32 | ~global_id (vr_index_search_keys);
      2. ONE `all_gather_into_tensor` of those [nq, k] words (80 KB per rank at nq = 1000, k = 10;
         RCCL over xGMI when the group's backend is nccl) — the only exchange step of the path;
      3. `vr_topk_merge_keys` over the gathered buffer as it is.
    `local_search_keys(queries, k, id_offset)` / `merge_keys(keys[world, nq, k])` default to those calls;
    the CPU (gloo) test injects the host statements below so that THIS function runs under world_size 2."""
    n_local = len(index) if index is not None else 0
    if id_offset < 0 or id_offset + n_local >= 2 ** 32 - 1:
        raise ValueError("global row ids must stay below 2^32 - 1 for the packed exchange")
    keys = (local_search_keys or index.search_keys)(queries, k, id_offset)
    if not isinstance(keys, torch.Tensor):
        keys = torch.from_numpy(np.ascontiguousarray(keys))
    merge = merge_keys or topk_merge_keys
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return merge(keys.view((1,) + tuple(keys.shape)))
    world = dist.get_world_size(group)
    nq, dev = keys.shape[0], keys.device
    mine = keys.contiguous()
    if mine.is_cuda and dist.get_backend(group) == "gloo":
        mine = mine.cpu()        # gloo rendezvous (no RCCL: e.g. two ranks sharing one GPU in the tests): exchange on the host
    gathered = torch.empty((world * nq, k), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine, group=group)             # the one collective of the path
    return merge(gathered.to(dev).view(world, nq, k))


# ---- host statements of the exchange format (tests; the product path above never calls them) ----
def pack_keys_host(scores: np.ndarray, ids: np.ndarray, id_offset: int = 0) -> np.ndarray:
    """(fp32 score, local row id or -1) -> the packed key of include/visrag_hip.h as int64 bit patterns."""
    u = np.ascontiguousarray(scores, dtype=np.float32).view(np.uint32).astype(np.uint64)
    ordr = np.where(u >> np.uint64(31), u ^ np.uint64(0xFFFFFFFF), u ^ np.uint64(0x80000000))
    gid = (np.asarray(ids, dtype=np.int64) + id_offset).astype(np.uint64) & np.uint64(0xFFFFFFFF)
    keys = (ordr << np.uint64(32)) | (gid ^ np.uint64(0xFFFFFFFF))
    return np.where(np.asarray(ids) >= 0, keys, np.uint64(0)).view(np.int64)


def unpack_keys_host(keys: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    k = np.ascontiguousarray(keys).view(np.uint64)
    o = (k >> np.uint64(32)).astype(np.uint32)
    u = np.where(o >> np.uint32(31), o ^ np.uint32(0x80000000), o ^ np.uint32(0xFFFFFFFF)).astype(np.uint32)
    sc = u.view(np.float32)
    ids = ((k & np.uint64(0xFFFFFFFF)) ^ np.uint64(0xFFFFFFFF)).astype(np.int64)
    none = k == 0
    return np.where(none, -np.inf, sc).astype(np.float32), np.where(none, -1, ids)


def merge_keys_host(keys: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """[world, nq, kk] packed keys -> (scores, global ids) [nq, k]: larger key first — the merge rule
    (score desc, id asc) is the integer order of the keys."""
    P, nq, kk = keys.shape
    u = np.transpose(np.ascontiguousarray(keys).view(np.uint64), (1, 0, 2)).reshape(nq, P * kk)
    order = np.argsort(~u, axis=1, kind="stable")[:, :k]
    return unpack_keys_host(np.take_along_axis(u, order, 1).view(np.int64))


def merge_topk_host(all_sc: np.ndarray, all_ids: np.ndarray, k: int):
    """Host-side statement of the merge rule (score desc, id asc) used by the gloo CPU tests."""
    P, nq, kk = all_sc.shape
    sc = np.transpose(all_sc, (1, 0, 2)).reshape(nq, P * kk)
    ids = np.transpose(all_ids, (1, 0, 2)).reshape(nq, P * kk)
    sc = np.where(ids >= 0, sc, -np.inf)
    order = np.lexsort((ids, -sc), axis=1)[:, :k]
    return np.take_along_axis(sc, order, 1), np.take_along_axis(ids, order, 1)
