"""Embedding inference loop with the reference's signature and on-disk contract
(src/openmatch/inference/inference.py:53-172): iterate batches of {'id','text','image'},
call `model(passage=batch)` / `model(query=batch)`, write pickle shards
`embeddings.{corpus|query}.rank.{r}[.{lo}-{hi}]`, barrier.

Differences that do not change results: embeddings stay on the device until a shard is
flushed (one D2H copy per shard instead of a `.cpu()` sync per batch, inference.py:98)."""
from __future__ import annotations

import logging
import os
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


def distributed_parallel_embedding_inference(dataset, model, args, dataset_type: str = "corpus",
                                             split_save: bool = True, model_additional_args: Optional[dict] = None):
    """args needs: per_device_eval_batch_size, output_dir, process_index, world_size,
    max_inmem_docs (reference: InferenceArguments)."""
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

    for batch in _batches(dataset, int(args.per_device_eval_batch_size)):
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
