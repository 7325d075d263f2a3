"""On-disk formats and eval glue of the retrieval path (own formulation of
src/openmatch/utils.py:125-175 (TREC I/O), :285-308 (eval_mrr) and the pickle embedding
shards of inference/inference.py:114-164)."""
from __future__ import annotations

import glob
import os
import pickle
from typing import Any, Dict, List, Tuple

import numpy as np


def save_as_trec(rank_result: Dict[str, Dict[str, float]], output_path: str, run_id: str = "OpenMatch"):
    """`<qid>\\tQ0\\t<docid>\\t<rank>\\t<score>\\t<run_id>` per line, each query's docs by
    descending score (every stored doc is written, not only k — utils.py:136-140)."""
    d = os.path.dirname(output_path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(output_path, "w") as f:
        for qid, docs in rank_result.items():
            ranked = sorted(docs.items(), key=lambda kv: kv[1], reverse=True)
            for rank, (doc_id, score) in enumerate(ranked, start=1):
                f.write("{}\tQ0\t{}\t{}\t{}\t{}\n".format(qid, doc_id, rank, score, run_id))


def load_from_trec(input_path: str, as_list: bool = False, max_len_per_q: int = None):
    """6-column TREC run or 3-column `qid docid score` (tab separated)."""
    res: Dict[str, Any] = {}
    cnt = 0
    with open(input_path) as f:
        for line in f:
            parts = line.strip().split("\t")
            if len(parts) == 6:
                qid, _, doc_id, _, score, _ = parts
            elif len(parts) == 3:
                qid, doc_id, score = parts
            else:
                raise ValueError("Invalid run format")
            if qid not in res:
                res[qid] = [] if as_list else {}
                cnt = 0
            if max_len_per_q is None or cnt < max_len_per_q:
                if as_list:
                    res[qid].append((doc_id, float(score)))
                else:
                    res[qid][doc_id] = float(score)
            cnt += 1
    return res


def eval_mrr(qrel: Dict[str, Dict[str, int]], run: Dict[str, Dict[str, float]], cutoff: int = None) -> Dict[str, float]:
    """MRR@cutoff with the reference's contract (utils.py:285-308): a dict holding the reciprocal rank of the first relevant
    doc of every query of `qrel` that `run` ranks, plus `'all'` = their mean over exactly those queries (driver/eval.py:303
    reads `eval_mrr(qrels, run, 10)['all']`).  Queries of `run` without a qrel entry do not count; docs are ranked by
    descending score in the run's own (stable) order for ties, as `list.sort` leaves them there.  A query `run` ranks no
    doc for scores 0 (the reference reads a stale variable there); no ranked query at all raises ZeroDivisionError like
    the reference's `mrr /= num_ranked_q`."""
    results: Dict[str, float] = {}
    total = 0.0
    for qid, rels in qrel.items():
        if qid not in run:
            continue
        ranked = sorted(run[qid].items(), key=lambda kv: kv[1], reverse=True)
        if cutoff is not None:
            ranked = ranked[:cutoff]
        rr = 0.0
        for i, (doc_id, _) in enumerate(ranked):
            if rels.get(doc_id, 0) > 0:
                rr = 1.0 / (i + 1)
                break
        results[qid] = rr
        total += rr
    n = len(results)
    results["all"] = total / n
    return results


def get_qrels_from_hf_repo(dataset_name: str) -> Dict[str, Dict[str, int]]:
    """qrels of a HuggingFace dataset repository, `{query-id: {corpus-id: score}}` (utils.py:310-325: the `qrels` config's
    `train` split).  Host-side data loading, kept so that `driver/eval.py:24` imports all four names from one module."""
    import datasets
    qrels: Dict[str, Dict[str, int]] = {}
    for row in datasets.load_dataset(dataset_name, "qrels")["train"]:
        qrels.setdefault(row["query-id"], {})[row["corpus-id"]] = row["score"]
    return qrels


def ndcg_recall_at_k(qrel: Dict[str, Dict[str, int]], run: Dict[str, Dict[str, float]], k: int = 10):
    """nDCG@k / Recall@k as driver/eval.py:281-301 prints them (trec_eval's `ndcg_cut` / `recall` through
    `visrag_amd.pytrec_eval`: linear gain, ties by descending doc id), averaged over the queries both dicts hold."""
    from . import pytrec_eval
    ev = pytrec_eval.RelevanceEvaluator(qrel, {f"ndcg_cut.{k}", f"recall.{k}"}).evaluate(run)
    if not ev:
        return 0.0, 0.0
    return (pytrec_eval.compute_aggregated_measure(f"ndcg_cut_{k}", [m[f"ndcg_cut_{k}"] for m in ev.values()]),
            pytrec_eval.compute_aggregated_measure(f"recall_{k}", [m[f"recall_{k}"] for m in ev.values()]))


# ---- pickle embedding shards: (np.float32[n, D], list[str]) protocol 4 ---------------------
def shard_name(dataset_type: str, rank: int, lo: int = None, hi: int = None) -> str:
    base = "embeddings.{}.rank.{}".format(dataset_type, rank)
    return base if lo is None else "{}.{}-{}".format(base, lo, hi)


def write_shard(path: str, reps: np.ndarray, ids: List[str]) -> None:
    with open(path, "wb") as f:
        pickle.dump((np.ascontiguousarray(reps, dtype=np.float32), list(ids)), f, protocol=4)


def read_shard(path: str) -> Tuple[np.ndarray, List[str]]:
    with open(path, "rb") as f:
        reps, ids = pickle.load(f)
    return reps, ids


def list_shards(output_dir: str, dataset_type: str, rank: int = None) -> List[str]:
    pat = "embeddings.{}.rank.{}".format(dataset_type, "*" if rank is None else f"{rank}*")
    files = sorted(glob.glob(os.path.join(output_dir, pat)))
    if rank is not None:
        # the reference's glob `rank.{r}*` (dense_retriever.py:40-46) also matches ranks 10 r .. 10 r + 9 of a larger world:
        # keep the files whose rank field IS r
        files = [f for f in files if shard_rank(f) == int(rank)]
    return files


def shard_rank(path: str) -> int:
    """`embeddings.{type}.rank.{r}[.{lo}-{hi}]` -> r: the field behind the LAST `.rank.` (a dataset_type with dots of its own
    does not shift it)."""
    tail = os.path.basename(path).rsplit(".rank.", 1)[1]
    return int(tail.split(".", 1)[0])
