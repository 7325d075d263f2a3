"""A `pytrec_eval`-shaped evaluator for the retrieve phase of driver/eval.py (:281-301), which calls

    evaluator = pytrec_eval.RelevanceEvaluator(qrels, {"ndcg_cut.10", "recall.10"})
    eval_results = evaluator.evaluate(run)
    pytrec_eval.compute_aggregated_measure(measure, [per-query values])

`pytrec_eval` (a C extension around NIST trec_eval 9.0) is a third-party dependency of the reference that is neither
vendored under /root/reference nor installable here (requirements.txt: `pytrec_eval`).  This module restates the
published trec_eval rules for the measures that call (and its neighbours) need, so that the driver runs unchanged after
`import visrag_amd.pytrec_eval as pytrec_eval`:

  * only queries held by BOTH the qrels and the run are evaluated (pytrec_eval skips a run query without judgments);
  * a query's docs are ranked by score descending, ties by doc id DESCENDING (trec_eval `form_res_rels`: `comp_sim_docno`
    compares the sims, then `strcmp(b->docno, a->docno)`); scores pass through C `float` (the `sim` field is a float);
  * a doc without a judgment is non-relevant; `relevance_level` (default 1) is the smallest judgment counted as relevant;
  * `ndcg` / `ndcg_cut_k`: gain = the judgment itself (linear, trec_eval's default without `-m ndcg.gains`), discount
    log2(rank + 1), ideal ranking = all judged docs with a positive judgment by descending judgment, 0 when it is empty;
  * `recall_k` = relevant retrieved in the top k / relevant; `P_k` = relevant in the top k / k (k, not the number
    retrieved); `map`, `map_cut_k`, `recip_rank`, `Rprec`, `success_k`, `num_ret`, `num_rel`, `num_rel_ret` as trec_eval
    defines them.
Measure names follow pytrec_eval: requested as "ndcg_cut.10" or "ndcg_cut" (all default cutoffs), reported as
"ndcg_cut_10".  Values are Python floats.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Sequence

import numpy as np

_CUTS_DOCS = (5, 10, 15, 20, 30, 100, 200, 500, 1000)          # trec_eval's default cutoffs for P / recall / ndcg_cut / map_cut
_CUTS_SUCCESS = (1, 5, 10)
_CUT_MEASURES = {"ndcg_cut": _CUTS_DOCS, "recall": _CUTS_DOCS, "P": _CUTS_DOCS, "map_cut": _CUTS_DOCS, "success": _CUTS_SUCCESS}
_PLAIN_MEASURES = ("map", "ndcg", "recip_rank", "Rprec", "num_ret", "num_rel", "num_rel_ret", "num_q")

supported_measures = set(_CUT_MEASURES) | set(_PLAIN_MEASURES)
supported_nicknames: Dict[str, set] = {"official": {"map", "P", "recall", "recip_rank", "Rprec", "num_ret", "num_rel", "num_rel_ret", "num_q"}}


def _parse_measures(measures: Iterable[str]) -> Dict[str, Sequence[int]]:
    out: Dict[str, List[int]] = {}
    for m in measures:
        name, _, arg = m.partition(".")
        if name not in supported_measures:
            raise ValueError("unsupported measure: {!r}".format(m))
        if name in _CUT_MEASURES:
            cuts = [int(c) for c in arg.split(",")] if arg else list(_CUT_MEASURES[name])
            if any(c <= 0 for c in cuts):
                raise ValueError("cutoffs must be positive: {!r}".format(m))
            out.setdefault(name, [])
            out[name] = sorted(set(out[name]) | set(cuts))
        else:
            if arg:
                raise ValueError("measure {!r} takes no parameter".format(name))
            out[name] = []
    return out


class RelevanceEvaluator:
    """`RelevanceEvaluator(qrels, measures, relevance_level=1).evaluate(run) -> {qid: {measure_name: value}}`."""

    def __init__(self, query_relevance: Dict[str, Dict[str, int]], measures: Iterable[str], relevance_level: int = 1):
        if not isinstance(query_relevance, dict):
            raise TypeError("Argument object qrel should be of type dictionary.")
        for qid, rels in query_relevance.items():
            if not isinstance(qid, str) or not isinstance(rels, dict):
                raise TypeError("Expected dictionary of {str: {str: int}} for the relevance judgments.")
            for d, r in rels.items():
                if not isinstance(d, str) or isinstance(r, bool) or not isinstance(r, (int, np.integer)):
                    raise TypeError("Expected relevance to be integer.")
        self.qrels = query_relevance
        self.measures = _parse_measures(measures)
        self.relevance_level = int(relevance_level)

    def evaluate(self, scores: Dict[str, Dict[str, float]]) -> Dict[str, Dict[str, float]]:
        if not isinstance(scores, dict):
            raise TypeError("Argument object scores should be of type dictionary.")
        out: Dict[str, Dict[str, float]] = {}
        for qid, docs in scores.items():
            if qid not in self.qrels:
                continue
            if not isinstance(docs, dict):
                raise TypeError("Expected dictionary of {str: {str: float}} for the run.")
            out[qid] = self._one(self.qrels[qid], docs)
        return out

    def _one(self, rels: Dict[str, int], docs: Dict[str, float]) -> Dict[str, float]:
        for d, s in docs.items():
            if not isinstance(d, str) or isinstance(s, bool) or not isinstance(s, (int, float, np.floating, np.integer)):
                raise TypeError("Expected run scores to be float.")
        # trec_eval's ranking: sim descending (as C floats), then docno descending
        items = sorted(docs.items(), key=lambda kv: kv[0], reverse=True)
        items.sort(key=lambda kv: float(np.float32(kv[1])), reverse=True)
        judged = [rels.get(d, -1) for d, _ in items]                  # -1: not in the pool = non-relevant
        lvl = self.relevance_level
        is_rel = [j >= lvl for j in judged]
        num_ret = len(items)
        num_rel = sum(1 for r in rels.values() if r >= lvl)
        num_rel_ret = sum(is_rel)
        gains = [float(j) if j > 0 else 0.0 for j in judged]
        ideal = sorted((float(r) for r in rels.values() if r > 0), reverse=True)

        def dcg(g: Sequence[float], k: int) -> float:
            return sum(x / math.log2(i + 2) for i, x in enumerate(g[:k]) if x != 0.0)

        def rel_upto(k: int) -> int:
            return sum(is_rel[:k])

        def ap(k: int) -> float:
            if num_rel == 0:
                return 0.0
            hit, acc = 0, 0.0
            for i, r in enumerate(is_rel[:k]):
                if r:
                    hit += 1
                    acc += hit / (i + 1)
            return acc / num_rel

        res: Dict[str, float] = {}
        for name, cuts in self.measures.items():
            if name == "ndcg_cut":
                for k in cuts:
                    idcg = dcg(ideal, k)
                    res["ndcg_cut_{}".format(k)] = dcg(gains, k) / idcg if idcg > 0 else 0.0
            elif name == "ndcg":
                idcg = dcg(ideal, len(ideal))
                res["ndcg"] = dcg(gains, num_ret) / idcg if idcg > 0 else 0.0
            elif name == "recall":
                for k in cuts:
                    res["recall_{}".format(k)] = rel_upto(k) / num_rel if num_rel else 0.0
            elif name == "P":
                for k in cuts:
                    res["P_{}".format(k)] = rel_upto(k) / k
            elif name == "success":
                for k in cuts:
                    res["success_{}".format(k)] = 1.0 if rel_upto(k) > 0 else 0.0
            elif name == "map_cut":
                for k in cuts:
                    res["map_cut_{}".format(k)] = ap(k)
            elif name == "map":
                res["map"] = ap(num_ret)
            elif name == "recip_rank":
                res["recip_rank"] = next((1.0 / (i + 1) for i, r in enumerate(is_rel) if r), 0.0)
            elif name == "Rprec":
                res["Rprec"] = rel_upto(num_rel) / num_rel if num_rel else 0.0
            elif name == "num_ret":
                res["num_ret"] = float(num_ret)
            elif name == "num_rel":
                res["num_rel"] = float(num_rel)
            elif name == "num_rel_ret":
                res["num_rel_ret"] = float(num_rel_ret)
            elif name == "num_q":
                res["num_q"] = 1.0
        return res


def compute_aggregated_measure(measure: str, values: Sequence[float]) -> float:
    """The summary a trec_eval run prints for `measure` over per-query `values`: counts (`num_*`) add up, `gm_*`
    measures take the geometric mean of their (log-domain) values, everything else the arithmetic mean."""
    vals = [float(v) for v in values]
    if measure.startswith("num_"):
        return float(sum(vals))
    if measure.startswith("gm_"):
        return float(math.exp(sum(vals) / len(vals)))
    return float(sum(vals) / len(vals))
