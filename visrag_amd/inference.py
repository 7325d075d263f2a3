"""Embedding inference loop with the reference's signature and on-disk contract
(src/openmatch/inference/inference.py:53-172): iterate batches of {'id','text','image'},
call `model(passage=batch)` / `model(query=batch)`, write pickle shards
`embeddings.{corpus|query}.rank.{r}[.{lo}-{hi}]`, barrier.

Differences that do not change results: embeddings stay on the device until a shard is
flushed (one D2H copy per shard instead of a `.cpu()` sync per batch, inference.py:98); batches are
loaded ahead of the GPU by background threads — the role of the reference's
`DataLoader(num_workers=args.dataloader_num_workers, pin_memory=...)` (inference.py:66-73): item loading
(`dataset[i]`: image open / decode) runs on `dataloader_num_workers` threads for map-style datasets, on one
thread for plain iterables, `prefetch_factor` batches ahead (PIL's decoders and numpy copies release the GIL;
pinning happens once per batch in gpu_resize.PageUploader)."""
from __future__ import annotations

import logging
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

from .utils import shard_name, write_shard

logger = logging.getLogger(__name__)


def naive_collator(batch_input: List[Dict]) -> Dict[str, list]:
    assert isinstance(batch_input, list) and len(batch_input) > 0
    keys = list(batch_input[0].keys())
    return {k: [item[k] for item in batch_input] for k in keys}


def _batches(dataset: Iterable[Dict], batch_size: int):
    cur = []
    for ex in dataset:
        cur.append(ex)
        if len(cur) == batch_size:
            yield naive_collator(cur)
            cur = []
    if cur:
        yield naive_collator(cur)


_END = object()


def _to_array(ex: Dict, to_u8: bool) -> Dict:
    """Loader-thread part of an item: a PIL page becomes a contiguous u8 HWC array (what PageUploader copies into its
    pinned buffer) — the RGB conversion and the copy leave the thread that drives the GPU."""
    if to_u8 and ex.get("image") is not None and hasattr(ex["image"], "convert"):
        ex = dict(ex)
        ex["image"] = np.ascontiguousarray(np.asarray(ex["image"].convert("RGB"), dtype=np.uint8))
    return ex


def _prefetched_batches(dataset, batch_size: int, num_workers: int, prefetch: int = 2, to_u8: bool = False):
    """Batches of the dataset, loaded ahead of their consumer.  num_workers <= 0: on the calling thread (the reference's
    num_workers=0).  Map-style dataset (`__len__` + `__getitem__`): items of the next `prefetch` batches are fetched by a
    pool of `num_workers` threads, order kept.  Iterable: one background thread walks it into a bounded queue."""
    if num_workers <= 0:
        for b in _batches((_to_array(ex, to_u8) for ex in dataset), batch_size):
            yield b
        return
    if hasattr(dataset, "__getitem__") and hasattr(dataset, "__len__") and not isinstance(dataset, (dict, str, bytes)):
        n = len(dataset)
        with ThreadPoolExecutor(max_workers=num_workers, thread_name_prefix="visrag-loader") as pool:
            pending = []
            starts = iter(range(0, n, batch_size))

            def submit():
                lo = next(starts, None)
                if lo is not None:
                    pending.append([pool.submit(lambda i=i: _to_array(dataset[i], to_u8)) for i in range(lo, min(n, lo + batch_size))])
            for _ in range(max(1, prefetch)):
                submit()
            while pending:
                futs = pending.pop(0)
                submit()
                yield naive_collator([f.result() for f in futs])
        return
    q: "queue.Queue" = queue.Queue(maxsize=max(1, prefetch))
    stop = threading.Event()

    def put(x) -> bool:                  # False: the consumer is gone (never blocks on a queue nobody reads)
        while not stop.is_set():
            try:
                q.put(x, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def work():
        try:
            for b in _batches((_to_array(ex, to_u8) for ex in dataset), batch_size):
                if not put(b):
                    return
            put(_END)
        except BaseException as e:       # surfaces in the consumer
            put(e)

    t = threading.Thread(target=work, name="visrag-loader", daemon=True)
    t.start()
    try:
        while True:
            b = q.get()
            if b is _END:
                return
            if isinstance(b, BaseException):
                raise b
            yield b
    finally:
        stop.set()


def distributed_parallel_embedding_inference(dataset, model, args, dataset_type: str = "corpus",
                                             split_save: bool = True, model_additional_args: Optional[dict] = None):
    """args needs: per_device_eval_batch_size, output_dir, process_index, world_size,
    max_inmem_docs (reference: InferenceArguments); honours args.dataloader_num_workers (default 1, like eval.sh:65)."""
    if dataset is None:
        raise ValueError("No dataset provided")
    if dataset_type not in ("corpus", "query"):
        raise ValueError(f"dataset_type: {dataset_type} is not valid.")
    model_additional_args = model_additional_args or {}
    os.makedirs(args.output_dir, exist_ok=True)
    world = max(1, int(getattr(args, "world_size", 1)))
    limit = int(getattr(args, "max_inmem_docs", 10_000_000)) // world
    encoded: List[torch.Tensor] = []
    lookup: List[str] = []
    idx = prev = 0
    first = True

    def flush(lo=None, hi=None):
        nonlocal encoded, lookup
        reps = torch.cat(encoded).cpu().numpy() if encoded else np.zeros((0, 0), dtype=np.float32)
        write_shard(os.path.join(args.output_dir, shard_name(dataset_type, args.process_index, lo, hi)), reps, lookup)
        encoded, lookup = [], []

    workers = int(getattr(args, "dataloader_num_workers", 1) or 0)
    to_u8 = bool(getattr(model, "gpu_preprocess", False))        # the GPU pre-processing takes u8 arrays as they are
    for batch in _prefetched_batches(dataset, int(args.per_device_eval_batch_size), workers, to_u8=to_u8):
        lookup.extend(batch["id"])
        idx += len(batch["id"])
        if dataset_type == "corpus":
            reps = model(passage=batch, **model_additional_args).p_reps
        else:
            reps = model(query=batch, **model_additional_args).q_reps
        if first:
            first = False
            assert not bool(torch.isnan(reps).any()), "vital error, model output has nan, please check."
        encoded.append(reps)
        if split_save and len(lookup) >= limit:
            flush(prev, idx)
            prev = idx
    if split_save:
        if lookup:
            flush(prev, idx)
    else:
        flush()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
