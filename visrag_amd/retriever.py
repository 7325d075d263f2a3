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

Two ways to run it under torchrun (one process per GPU), same result, byte for byte in the TREC file:
  * replicated (the reference's: dense_retriever.py:48-69): every rank loads ALL corpus shards into its own GPU
    and searches its own queries — N copies of the index, no exchange;
  * corpus-sharded (`sharded=True`, `args.sharded_corpus`, or VISRAG_SHARDED_RETRIEVE=1; BASELINE.json north_star):
    rank r loads only the corpus shards rank r wrote (`embeddings.corpus.rank.{r}*`: the rows it embedded),
    every rank searches ALL queries against its shard, ONE all-gather of the packed per-shard top-k keys,
    and each rank returns the entries of ITS queries — so the caller (driver/eval.py:210-232: save_as_trec per
    rank, rank-0 merge) is unchanged.

`sharded_search` is the data path of the second form for one local index: local top-k emitted as packed 64-bit
(score, global id) keys by the search's own merge kernel, ONE RCCL all-gather of those [nq, k] words, on-device
merge of the gathered buffer."""
from __future__ import annotations

import logging
import os
import warnings
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from .engine import HipIndex, topk_merge_keys
from .utils import list_shards, read_shard, shard_rank

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


def _load_queries(args, rank="own") -> Tuple[np.ndarray, List[str], List[bool]]:
    """-> (reps, ids, mine): the query shards of this process (`rank="own"`: dense_retriever.py:40-46) or of every
    process (`rank=None`), in sorted file order; mine[i] = query i belongs to this process's own shards."""
    parts = list_shards(args.output_dir, "query", args.process_index if rank == "own" else None)
    own = set(list_shards(args.output_dir, "query", args.process_index))
    if rank != "own":
        # corpus-sharded form: a query file belongs to the process its rank field maps to — the same rule as the corpus files
        # (rank % world), so a retrieval world smaller than the encoding world drops nothing
        dist = torch.distributed
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        own = {p for p in parts if _shard_rank(p) % world == int(args.process_index) % world}
    logger.info("query_all_partitions = %s", parts)
    reps, ids, mine = [], [], []
    for p in parts:
        r, i = read_shard(p)
        if len(i) == 0:
            continue
        reps.append(r)
        ids.extend(i)
        mine.extend([p in own] * len(i))
    if not reps:
        raise ValueError("No pre-computed query embeddings found")
    return np.concatenate(reps), ids, mine


_shard_rank = shard_rank


def _sharded_requested(args, sharded: Optional[bool]) -> bool:
    if sharded is None:
        sharded = getattr(args, "sharded_corpus", None)
    if sharded is None:
        sharded = os.environ.get("VISRAG_SHARDED_RETRIEVE", "0") not in ("", "0")
    return bool(sharded) and _collective_wanted()


def _collective_wanted(group=None) -> bool:
    """A process group of two or more ranks exchanges; a group of ONE rank returns its own result without touching the
    backend (a single-rank run that happens to have initialised torch.distributed needs no working all-gather) unless
    VISRAG_SHARDED_RETRIEVE=force asks for the transport anyway (the one-GPU RCCL test)."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("VISRAG_SHARDED_RETRIEVE", "") == "force"


def distributed_parallel_retrieve(args, topk: int, global_topk: bool = False, sharded: Optional[bool] = None,
                                  index_factory: Optional[Callable] = None, merge_keys: Optional[Callable] = None
                                  ) -> Dict[str, Dict[str, float]]:
    """`index_factory(dim, capacity, device)` / `merge_keys` default to HipIndex / vr_topk_merge_keys; the CPU (gloo)
    tests inject host stand-ins so that THIS function is what runs under world_size 2."""
    make_index = index_factory or HipIndex
    if _sharded_requested(args, sharded):
        return _retrieve_corpus_sharded(args, topk, global_topk, make_index, merge_keys)
    queries, qids, _ = _load_queries(args)
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
            ix = make_index(dim, len(ids), dev)
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
    ix = make_index(dim, max(total, 1), dev)
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


def _retrieve_corpus_sharded(args, topk: int, global_topk: bool, make_index: Callable, merge_keys: Optional[Callable]
                             ) -> Dict[str, Dict[str, float]]:
    """The corpus-sharded form (module docstring).  Row ids are global: files in sorted order (the order the
    replicated form walks them), a file's rows in place.  Exchange steps: one all_gather_object of the files' doc-id
    lists (control plane: strings), ONE all_gather_into_tensor of the packed keys (data path)."""
    dist = torch.distributed
    rank, world = dist.get_rank(), dist.get_world_size()
    queries, qids, mine = _load_queries(args, rank=None)                  # every rank searches all queries
    files = list_shards(args.output_dir, "corpus")
    if len(files) == 0:
        raise ValueError("No pre-computed document embeddings found")
    my_files = [p for p in files if _shard_rank(p) % world == rank]
    logger.info("corpus partitions of rank %d = %s", rank, my_files)
    dev = _device_index(args)
    on_gpu = make_index is HipIndex
    local = {os.path.basename(p): read_shard(p) for p in my_files}
    metas: List[Dict[str, List[str]]] = [None] * world
    dist.all_gather_object(metas, {name: list(ids) for name, (_, ids) in local.items()})
    doc_ids: Dict[str, List[str]] = {}
    for m in metas:
        doc_ids.update(m)
    offsets, table, off = {}, [], 0
    for p in files:                                                        # global ids: sorted file order
        name = os.path.basename(p)
        offsets[name] = off
        table.extend(doc_ids[name])
        off += len(doc_ids[name])
    dim, nq = queries.shape[1], queries.shape[0]
    q = torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32))
    if on_gpu:
        q = q.to(f"cuda:{dev}")
    # slot f of the exchange buffer = the f-th file of the sorted list owned by this rank; ranks own different numbers
    # of files: pad with empty keys
    f_max = max(sum(1 for p in files if _shard_rank(p) % world == r) for r in range(world))
    keys = torch.zeros((f_max, nq, topk), dtype=torch.int64, device=q.device)
    for f, p in enumerate(my_files):
        reps, ids = local[os.path.basename(p)]
        if len(ids) == 0:
            continue
        ix = make_index(dim, len(ids), dev)
        ix.add(reps)
        keys[f] = ix.search_keys(q, topk, offsets[os.path.basename(p)])
        ix.close()
    gathered = exchange_keys(keys)                                         # [world, f_max, nq, k]: the one data-path collective
    result: Dict[str, Dict[str, float]] = {qid: {} for qid, m in zip(qids, mine) if m}
    if global_topk:
        sc, gid = (merge_keys or topk_merge_keys)(gathered.view(world * f_max, nq, topk))
        sc, gid = sc.cpu().numpy(), gid.cpu().numpy()
        for qi, qid in enumerate(qids):
            if mine[qi]:
                for s, j in zip(sc[qi], gid[qi]):
                    if j >= 0:
                        result[qid][table[int(j)]] = float(s)
        return result
    # the reference's union of the per-file top-k lists, files in sorted order like the replicated form
    sc, gid = unpack_keys_host(gathered.cpu().numpy())
    owners = [r for r in range(world)]
    slot_of = {}
    for r in owners:
        for f, p in enumerate([p for p in files if _shard_rank(p) % world == r]):
            slot_of[os.path.basename(p)] = (r, f)
    for p in files:
        r, f = slot_of[os.path.basename(p)]
        for qi, qid in enumerate(qids):
            if mine[qi]:
                for s, j in zip(sc[r, f, qi], gid[r, f, qi]):
                    if j >= 0:
                        result[qid][table[int(j)]] = float(s)
    return result


def exchange_keys(keys: torch.Tensor, group=None) -> torch.Tensor:
    """ONE all_gather_into_tensor of this rank's packed keys (same shape on every rank) -> [world, *keys.shape] on the
    keys' device.  RCCL over xGMI when the group's backend is nccl; a gloo group (CPU tests, two ranks sharing one
    GPU) exchanges on the host."""
    dist = torch.distributed
    world = dist.get_world_size(group)
    dev = keys.device
    mine = keys.contiguous()
    if mine.is_cuda and dist.get_backend(group) == "gloo":
        mine = mine.cpu()
    gathered = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    return gathered.to(dev).view((world,) + tuple(keys.shape))


def sharded_search(index, queries, k: int, id_offset: int = 0, group=None,
                   local_search_keys: Optional[Callable] = None, merge_keys: Optional[Callable] = None,
                   local_search: Optional[Callable] = None, merge: Optional[Callable] = None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Every rank holds `index` = its corpus shard (row j has global id id_offset + j) and the
    same `queries` [nq, dim] (a cuda tensor; a cpu tensor or numpy array is moved to the index's device).
    Returns the global (scores, ids) [nq, k] on every rank.

    A world of one rank (no process group, or a group of one) with the library's own calls is ONE call, `index.search`.
    Otherwise the data path is three library calls and one collective, no tensor arithmetic in between:
      1. `index.search_keys(queries, k, id_offset)` — the local fused search, whose merge kernel writes
         each result as ONE 64-bit word, orderable(score) << 32 | ~global_id (vr_index_search_keys);
      2. ONE `all_gather_into_tensor` of those [nq, k] words (80 KB per rank at nq = 1000, k = 10;
         RCCL over xGMI when the group's backend is nccl) — the only exchange step of the path;
      3. `vr_topk_merge_keys` over the gathered buffer as it is.
    `local_search_keys(queries, k, id_offset)` / `merge_keys(keys[world, nq, k])` default to those calls;
    the CPU (gloo) test injects the host statements below so that THIS function runs under world_size 2.
    (`local_search(queries, k) -> (scores, local ids)` / `merge(all_scores, all_ids)` are the round-2 names of the
    two hooks, still accepted: their results are converted through the host statements of the key format.)"""
    n_local = len(index) if index is not None else 0
    if id_offset < 0 or id_offset + n_local >= 2 ** 32 - 1:
        raise ValueError("global row ids must stay below 2^32 - 1 for the packed exchange")
    if local_search is not None and local_search_keys is None:
        warnings.warn("sharded_search(local_search=...) is deprecated: pass local_search_keys", DeprecationWarning, stacklevel=2)

        def local_search_keys(q_, k_, off_):
            s_, i_ = local_search(q_, k_)
            s_ = s_.cpu().numpy() if isinstance(s_, torch.Tensor) else s_
            i_ = i_.cpu().numpy() if isinstance(i_, torch.Tensor) else i_
            return pack_keys_host(s_, i_, off_)
    if merge is not None and merge_keys is None:
        warnings.warn("sharded_search(merge=...) is deprecated: pass merge_keys", DeprecationWarning, stacklevel=2)

        def merge_keys(keys_):
            s_, i_ = unpack_keys_host(keys_.cpu().numpy())
            return merge(torch.from_numpy(s_), torch.from_numpy(i_))
    if local_search_keys is None and index is not None and not (isinstance(queries, torch.Tensor) and queries.is_cuda):
        queries = torch.as_tensor(np.ascontiguousarray(queries, dtype=np.float32) if not isinstance(queries, torch.Tensor)
                                  else queries).to(f"cuda:{index.device}")      # host queries: the exchange buffer lives in HBM
    if local_search_keys is None and merge_keys is None and index is not None and not _collective_wanted(group):
        # a world of ONE rank with the library's own calls: nothing to exchange, so no exchange format either — the local search
        # writes (scores, ids) itself (one launch and two allocations less per search than pack -> merge of a single part)
        sc, ids = index.search(queries, k)
        if id_offset:
            ids += int(id_offset) * (ids >= 0)
        return sc, ids
    keys = (local_search_keys or index.search_keys)(queries, k, id_offset)
    if not isinstance(keys, torch.Tensor):
        keys = torch.from_numpy(np.ascontiguousarray(keys))
    merge_fn = merge_keys or topk_merge_keys
    if not _collective_wanted(group):
        return merge_fn(keys.view((1,) + tuple(keys.shape)))
    return merge_fn(exchange_keys(keys, group))                              # the one collective of the path


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


def pack_topk(scores: torch.Tensor, ids: torch.Tensor, id_offset: int = 0) -> torch.Tensor:
    """Deprecated name (round 2) of the exchange packing — the product path packs inside the search's merge kernel
    (HipIndex.search_keys).  Host statement of the CURRENT key format (include/visrag_hip.h), torch in / torch out."""
    warnings.warn("pack_topk is deprecated: HipIndex.search_keys emits packed keys", DeprecationWarning, stacklevel=2)
    return torch.from_numpy(pack_keys_host(scores.cpu().numpy(), ids.cpu().numpy(), id_offset)).to(scores.device)


def unpack_topk(packed: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Deprecated name (round 2): host statement of the key format's inverse (see unpack_keys_host)."""
    warnings.warn("unpack_topk is deprecated: vr_topk_merge_keys consumes packed keys", DeprecationWarning, stacklevel=2)
    s_, i_ = unpack_keys_host(packed.cpu().numpy())
    return torch.from_numpy(s_).to(packed.device), torch.from_numpy(i_).to(packed.device)
