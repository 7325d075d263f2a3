#!/usr/bin/env python
"""The REFERENCE's own code timed on this container's host cores (build container only: /root/reference does not
exist on the GPU box).  Writes profiles/ref_cpu_baseline.json, which bench.py quotes next to its own CPU leg
(`cpu_baseline.reference_in_build_container`): BASELINE.md section 3's protocol —
  * encode: openmatch DRModelForInference (VisRAG_Ret.forward, fp32, sdpa) at full MiniCPM-V-2.0 dims with the
    seeded synthetic weights, pages 448x448 in batches of 16 (README.md:146), 1 warm-up batch of 4, then N pages;
    16 text queries in one batch;
  * retrieve: openmatch.retriever.distributed_parallel_retrieve over pickle shards (4 shards, 100k x 2304 unit-norm
    rows, 1k queries, top-10), unpickling and the Python merge loop included.

    python tools/ref_cpu_baseline.py [--pages 16] [--rows 100000]
"""
import argparse
import json
import os
import pickle
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=16)
    ap.add_argument("--rows", type=int, default=100_000)
    ap.add_argument("--queries", type=int, default=1000)
    args = ap.parse_args()
    from PIL import Image
    from oracle import ref_harness
    from visrag_amd.config import full_config
    from visrag_amd.synth import synth_pages, synth_queries, synth_state_dict
    from visrag_amd.tokenizer import StandInTokenizer
    if not ref_harness.reference_available():
        raise SystemExit("no /root/reference here: this tool runs in the build container only")
    cfg = full_config()
    t0 = time.time()
    dr = ref_harness.build_reference_dr_model(cfg, synth_state_dict(cfg, 0))
    tok = StandInTokenizer(cfg.vocab_size)
    print(f"reference model built in {time.time() - t0:.0f}s, threads={torch.get_num_threads()}", flush=True)
    pages = synth_pages(args.pages + 4, size=448, seed=0)
    qtexts = ["Represent this query for retrieving relevant documents: " + q for q in synth_queries(16, seed=0)]

    def encode_pages(arrs):
        imgs = [Image.fromarray(a) for a in arrs]
        return dr(passage={"id": [str(i) for i in range(len(imgs))], "text": [""] * len(imgs), "image": imgs}, tokenizer=tok,
                  max_inp_length=2048).p_reps

    with torch.no_grad():
        encode_pages(pages[:4])                                            # warm-up
        t0 = time.perf_counter()
        for lo in range(4, 4 + args.pages, 16):
            encode_pages(pages[lo:min(lo + 16, 4 + args.pages)])
        t_pages = time.perf_counter() - t0
        t0 = time.perf_counter()
        dr(query={"id": [str(i) for i in range(16)], "text": qtexts, "image": [None] * 16}, tokenizer=tok, max_inp_length=512)
        t_q = time.perf_counter() - t0
    print(f"pages: {args.pages / t_pages:.3f}/s   queries: {16 / t_q:.2f}/s", flush=True)
    del dr
    ref_harness.install_shims()
    from openmatch.retriever.dense_retriever import distributed_parallel_retrieve
    rng = np.random.default_rng(0)
    C = rng.standard_normal((args.rows, cfg.hidden_size)).astype(np.float32); C /= np.linalg.norm(C, axis=1, keepdims=True)
    Q = rng.standard_normal((args.queries, cfg.hidden_size)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    with tempfile.TemporaryDirectory() as d:
        per = (args.rows + 3) // 4
        for s in range(4):
            lo, hi = s * per, min(args.rows, (s + 1) * per)
            with open(os.path.join(d, f"embeddings.corpus.rank.0.{lo}-{hi}"), "wb") as f:
                pickle.dump((C[lo:hi], [f"doc{j}" for j in range(lo, hi)]), f, protocol=4)
        with open(os.path.join(d, "embeddings.query.rank.0"), "wb") as f:
            pickle.dump((Q, [f"q{j}" for j in range(len(Q))]), f, protocol=4)
        a = types.SimpleNamespace(output_dir=d, process_index=0, device="cpu")
        t0 = time.perf_counter()
        res = distributed_parallel_retrieve(a, 10)
        t_r = time.perf_counter() - t0
    assert len(res) == args.queries
    out = {"pages_per_sec": round(args.pages / t_pages, 3), "query_encode_per_sec": round(16 / t_q, 2),
           "queries_per_sec_search": round(args.queries / t_r, 1), "cores": torch.get_num_threads(),
           "what": f"the reference's own code (openmatch DRModelForInference + distributed_parallel_retrieve, /root/reference) on "
                   f"the build container's host cores, fp32: {args.pages} pages 448x448 in batches of 16 in {t_pages:.1f}s, 16 text "
                   f"queries in {t_q:.1f}s, {args.queries} queries x {args.rows} rows top-10 over 4 pickle shards in {t_r:.2f}s",
           "source": "tools/ref_cpu_baseline.py"}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "ref_cpu_baseline.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
