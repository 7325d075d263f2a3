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


def eval_mrr(qrel: Dict[str, Dict[str, int]], run: Dict[str, Dict[str, float]], cutoff: int = None) -> float:
    """Mean reciprocal rank of the first relevant doc (utils.py:285-308)."""
    total = 0.0
    for qid, docs in run.items():
        ranked = sorted(docs.items(), key=lambda kv: kv[1], reverse=True)
        if cutoff is not None:
            ranked = ranked[:cutoff]
        for i, (doc_id, _) in enumerate(ranked):
            if qid in qrel and qrel[qid].get(doc_id, 0) > 0:
                total += 1.0 / (i + 1)
                break
    return total / max(1, len(run))


def ndcg_recall_at_k(qrel: Dict[str, Dict[str, int]], run: Dict[str, Dict[str, float]], k: int = 10):
    """pytrec_eval-free nDCG@k / Recall@k (eval.py:281-303 uses pytrec_eval's ndcg_cut / recall)."""
    nd, rc, n = 0.0, 0.0, 0
    for qid, rels in qrel.items():
        if qid not in run:
            continue
        n += 1
        ranked = [d for d, _ in sorted(run[qid].items(), key=lambda kv: kv[1], reverse=True)][:k]
        dcg = sum((2 ** rels.get(d, 0) - 1) / np.log2(i + 2) for i, d in enumerate(ranked))
        ideal = sorted(rels.values(), reverse=True)[:k]
        idcg = sum((2 ** r - 1) / np.log2(i + 2) for i, r in enumerate(ideal))
        nd += dcg / idcg if idcg > 0 else 0.0
        npos = sum(1 for r in rels.values() if r > 0)
        rc += sum(1 for d in ranked if rels.get(d, 0) > 0) / npos if npos else 0.0
    return (nd / n, rc / n) if n else (0.0, 0.0)


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
        files = [f for f in files if os.path.basename(f).split(".")[3] == str(rank)]
    return files
