#!/usr/bin/env python
"""bench.py — VisRAG-Ret corpus embedding + retrieval on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one batch of synthetic input already resident in
HBM: 32 page images (448x448x3 uint8) -> vr_encode (SigLIP ViT + resampler + MiniCPM-2B
decoder + wmean pool + L2 norm) -> 32 fp32 embeddings appended to the HBM-resident index.
Workload at N=1 = BASELINE.json configs[1] (bf16 encode, batch 32 pages); pages are sharded
across ranks with no data-path collective (weak scaling).  The same JSON line also carries
the retrieval half of the metric (configs[2]/[3]): 1k queries top-10 over a 100k-row index
sharded over the N ranks, local fused search + ONE RCCL all-gather of [nq,k] + device merge.

Full model dims, random-init (deterministic synthetic) weights, synthetic pages: there is no
network for checkpoints or datasets.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md
QUERY_PREFIX = "Represent this query for retrieving relevant documents: "


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--index-rows", type=int, default=100_000)
    ap.add_argument("--corpus-pages", type=int, default=100_000,
                    help="DISTINCT synthetic pages embedded by the model into the retrieval index, whole job (BASELINE configs[2]: "
                         "100k, ~2.5 minutes on one GPU); 0 = skip, search the i.i.d. filler index only")
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--search-steps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipelined", action="store_true",
                    help="measure encode with two batches in flight (vr_model_clone + two HIP streams) even with --no-extras; "
                         "reported under \"pipelined\", never as `value`")
    ap.add_argument("--cpu-pages", type=int, default=16,
                    help="pages of the CPU baseline sample (one reference-sized batch of 16 by default; 64 = all of "
                         "BASELINE config 1, about 3 minutes of host time)")
    ap.add_argument("--no-extras", action="store_true", help="skip the PIL-input and sliced-page measurements")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc measurement of the dominant kernel's HBM traffic")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = torch.distributed
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # one process per GPU (RCCL).  More ranks than visible GPUs (verifying the torchrun path on a 1-GPU box)
    # share devices round-robin and rendezvous over gloo, because RCCL refuses two ranks on one device.
    ndev = torch.cuda.device_count()
    shared = world > ndev
    local_rank %= ndev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from PIL import Image
    from visrag_amd.config import full_config
    from visrag_amd.engine import HipEncoder, HipIndex
    from visrag_amd.preprocess import prepare_batch
    from visrag_amd.retriever import sharded_search
    from visrag_amd.synth import iter_synth_weights, synth_pages, synth_queries
    from visrag_amd.tokenizer import StandInTokenizer

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    cfg = full_config()
    B = args.batch
    t0 = time.time()
    enc = HipEncoder(cfg, device=local_rank, max_images=B, max_tokens=max(4096, B * 80), max_seqs=max(64, B))
    enc.load_state_dict(iter_synth_weights(cfg, 0, device=dev))
    log(f"[rank {rank}] weights ready in {time.time() - t0:.1f}s")
    tok = StandInTokenizer(cfg.vocab_size)

    # ---- synthetic page pool resident in HBM; rank r embeds pages r*pool .. (its shard)
    pool = 2 * B
    pages = synth_pages(pool, size=448, seed=0, first=rank * pool)
    items = prepare_batch([""] * pool, [Image.fromarray(p) for p in pages], tok, cfg, 2048)
    dev_pages = [torch.from_numpy(p).to(dev) for p in pages]
    batches = [(items[i:i + B], dev_pages[i:i + B]) for i in range(0, pool, B)]
    rows_local = (args.index_rows + world - 1) // world
    index = HipIndex(cfg.hidden_size, rows_local + (args.steps + args.warmup + 2) * B, device=local_rank)
    out = torch.empty((B, cfg.hidden_size), dtype=torch.float32, device=dev)

    def step(i):
        it, px = batches[i % len(batches)]
        enc.encode_items(it, device_slices=px, out=out)
        index.add(out)

    for i in range(args.warmup):
        step(i)
    barrier()
    index.reset()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):                       # the timed region of `value`: no events, no profiling inside
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    # per-kernel-class HIP events (roofline, phases) in a SECOND pass of the same steps: outside `value`
    enc.set_profile(True)
    for i in range(args.steps):
        it, px = batches[i % len(batches)]
        enc.encode_items(it, device_slices=px, out=out)
    barrier()
    prof = enc.get_profile()
    # the decoder phase split by events BETWEEN its kernels: a third pass (those events would lengthen "decoder" above)
    enc.set_profile(2)
    for i in range(args.steps):
        it, px = batches[i % len(batches)]
        enc.encode_items(it, device_slices=px, out=out)
    barrier()
    prof.update({k: v for k, v in enc.get_profile().items() if k.startswith("dec_")})
    enc.set_profile(False)
    def max_over_ranks(x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if shared else dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt = max_over_ranks(dt)
    pages_per_s = world * args.steps * B / dt
    ms_per_step = dt / args.steps * 1e3

    # ---- optional: two batches in flight (own workspace + stream each, shared weights); outside `value`
    pipelined = None
    if args.pipelined or not args.no_extras:
        enc2 = enc.clone()
        from visrag_amd.engine import overlapping_streams
        sa, sb = overlapping_streams(dev, 2)            # (two pool streams can share a hardware queue: probed)
        slots = [(enc, sa, out), (enc2, sb, torch.empty_like(out))]

        def pstep(i):
            e, st, o = slots[i & 1]
            it, px = batches[i % len(batches)]
            with torch.cuda.stream(st):
                e.encode_items(it, device_slices=px, out=o)

        for i in range(4):
            pstep(i)
        barrier()
        tp = time.perf_counter()
        for i in range(2 * args.steps):
            pstep(i)
        barrier()
        dp = max_over_ranks(time.perf_counter() - tp)
        pipelined = {"in_flight": 2, "pages_per_sec": round(world * 2 * args.steps * B / dp, 2),
                     "ms_per_step": round(dp / (2 * args.steps) * 1e3, 3)}
        enc2.close()

    # ---- retrieval half of the metric (BASELINE configs[2] / [3]) ------------------------------------------------
    # (1) "model" index — configs[2] AS WORDED: this rank's share of `--corpus-pages` DISTINCT synthetic pages (generated on
    #     the GPU, visrag_amd/synth.py) is embedded by the model and appended to the HBM-resident index, the text queries are
    #     encoded by the model, and the search runs over THOSE embeddings; (2) "filler": i.i.d. Gaussian unit rows — what
    #     rounds 1-3 timed, kept beside it; (3) "templated": 100 families x 1000 near-duplicate pages laid out contiguously
    #     (a deck embedded page after page), built by perturbing model embeddings: the certification's worst case.
    E = cfg.hidden_size
    rows_local = (args.index_rows + world - 1) // world
    corpus_local = (min(args.corpus_pages, args.index_rows) + world - 1) // world if args.corpus_pages > 0 else 0
    emb_m = None
    corpus_embed = None
    if corpus_local > 0:
        from visrag_amd.synth import synth_pages_gpu
        emb_m = torch.empty((corpus_local, E), dtype=torch.float32, device=dev)
        pxbuf = torch.empty((B, 448, 448, 3), dtype=torch.uint8, device=dev)
        tmpl = [items[0]] * B                                      # every 448 x 448 page has the same token layout
        barrier()
        tc0 = time.perf_counter()
        for lo in range(0, corpus_local, B):
            nb = min(B, corpus_local - lo)
            synth_pages_gpu(nb, size=448, seed=0, first=1_000_000 + rank * corpus_local + lo, device=local_rank, out=pxbuf)
            enc.encode_items(tmpl[:nb], device_slices=[pxbuf[i] for i in range(nb)], out=emb_m[lo:lo + nb])
        barrier()
        dc = max_over_ranks(time.perf_counter() - tc0)
        corpus_embed = {"pages": corpus_local * world, "pages_per_sec": round(corpus_local * world / dc, 1), "seconds": round(dc, 1),
                        "what": "distinct synthetic pages generated on the GPU -> vr_encode -> fp32 rows kept in HBM (single stream)"}
        log(f"[rank {rank}] corpus of {corpus_local} pages embedded in {dc:.1f}s")
    qtexts = [QUERY_PREFIX + q for q in synth_queries(args.queries, seed=0)]
    qitems = prepare_batch(qtexts, [None] * len(qtexts), tok, cfg, 512)
    barrier()
    tq0 = time.perf_counter()
    qreps = []
    for lo in range(0, len(qitems), 64):
        qreps.append(enc.encode_items(qitems[lo:lo + 64]))
    Q = torch.cat(qreps)
    barrier()
    q_encode_s = time.perf_counter() - tq0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def ids_vs_fp64(rows: torch.Tensor, ids: torch.Tensor, sc: torch.Tensor):
        """the local search's ids against an fp64 brute force over the same fp32 rows (the checker: torch fp64 matmul)"""
        ref = Q.double() @ rows.double().T
        rv, ri = torch.topk(ref, args.topk, dim=1)
        got = torch.gather(ref, 1, ids)
        differ = ids != ri
        gap = float((rv - got).abs()[differ].max()) if bool(differ.any()) else 0.0
        err = float((got - sc.double()).abs().max())
        return {"queries_with_identical_ids": int((~differ.any(dim=1)).sum()), "queries": int(ids.shape[0]),
                "max_fp64_score_gap_where_ids_differ": gap, "max_returned_score_error": err,
                "ok": bool(gap < 3e-7 and err < 2e-6),
                "what": "ids differ from the fp64 ranking only where two fp64 scores are closer than fp32 summation noise (3e-7)"}

    def search_bench(index, rows, kind, check):
        for _ in range(3):
            sc, ids = sharded_search(index, Q, args.topk, id_offset=rank * rows_local)
        barrier()
        ts0 = time.perf_counter()
        for _ in range(args.search_steps):
            sc, ids = sharded_search(index, Q, args.topk, id_offset=rank * rows_local)
        barrier()
        ds = max_over_ranks(time.perf_counter() - ts0)
        e0.record()
        for _ in range(args.search_steps):
            lsc, lids = index.search(Q, args.topk)
        e1.record()
        torch.cuda.synchronize()
        sweep_ms = e0.elapsed_time(e1) / args.search_steps
        flops = 2.0 * args.queries * len(index) * E
        index.search_stats(reset=True)
        index.set_search_profile(True)
        for _ in range(args.search_steps):
            index.search(Q, args.topk)
        stages = index.get_search_profile()
        index.set_search_profile(False)
        cert = index.search_stats()
        n_cert = max(1, cert["certified"] + cert["certified_extended"] + cert["flagged"] + cert["uncertified"])
        for _ in range(3):
            index.search(Q[:1], args.topk)
        e0.record()
        for _ in range(20):
            index.search(Q[:1], args.topk)
        e1.record()
        torch.cuda.synchronize()
        one_ms = e0.elapsed_time(e1) / 20
        d = {"index_kind": kind, "index_rows": len(index) * world, "rows_per_gpu": len(index), "queries": args.queries, "k": args.topk,
             "queries_per_sec": round(args.queries * args.search_steps / ds, 1),
             "ms_per_search": round(ds / args.search_steps * 1e3, 3),
             "local_sweep_ms": round(sweep_ms, 3),
             "local_sweep_tflops": round(flops / (sweep_ms * 1e-3) / 1e12, 1),
             "local_sweep_frac_of_mfma_peak": round(flops / (sweep_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
             "index_GBps": round(len(index) * E * 2 / (sweep_ms * 1e-3) / 1e9, 1),
             "stages_ms": {k: round(v, 4) for k, v in stages.items() if k != "calls"},
             "sweep_kernel_tflops": round(flops / max(stages["sweep"], 1e-9) / 1e9, 1),
             "error_model": {k: float(f"{v:.4g}") for k, v in index.error_model().items()},
             "plan": index.search_plan(args.queries),      # (prepass_chunks > 0: the pre-pass owns its 16 sampled tiles, the sweep skips them)
             "certification": {"certified_at_once": round(cert["certified"] / n_cert, 4),
                               "certified_after_extended_rescoring": round(cert["certified_extended"] / n_cert, 4),
                               "band_pass": round(cert["band_pass"] / n_cert, 4),
                               "exact_fp32_pass": round(cert["exact_pass"] / n_cert, 4),
                               "gathered_twice": round(cert["regathered"] / n_cert, 4),
                               "what": "fraction of queries; the ids returned are the fp32 ranking's (rigorous bf16 error bound "
                                       "from the data's measured rounding residuals, visrag_hip.h: vr_index_set_search_eps); "
                                       "band_pass / exact_fp32_pass = redone behind the sweep (search_band.hip / search_exact.hip)"},
             "single_query": {"ms": round(one_ms, 4), "bound": "hbm",
                              "index_GBps": round(len(index) * E * 2 / (one_ms * 1e-3) / 1e9, 1),
                              "frac_of_hbm_peak": round(len(index) * E * 2 / (one_ms * 1e-3) / 1e9 / 8000.0, 4)}}
        if check and rows is not None:
            d["ids_vs_fp64"] = ids_vs_fp64(rows, lids, lsc)
        return d, ds

    searches = {}
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    if emb_m is not None:
        index.reset()
        index_m = HipIndex(E, corpus_local, device=local_rank)
        index_m.add(emb_m)
        searches["model"], ds_main = search_bench(index_m, emb_m, "model", True)
        # what the corpus looks like to the search (the filler's page-page cosines are ~0 +- 0.02)
        sub = emb_m[:: max(1, corpus_local // 2000)][:2000]
        cc = (sub @ sub.T)
        iu = torch.triu_indices(cc.shape[0], cc.shape[0], 1, device=dev)
        pc = cc[iu[0], iu[1]]
        qs = (Q @ sub.T)
        searches["model"]["embedding_stats"] = {
            "page_page_cosine": {"median": round(float(pc.median()), 4), "p99": round(float(pc.quantile(0.99)), 4), "max": round(float(pc.max()), 4)},
            "query_page_score": {"mean": round(float(qs.mean()), 4), "std_over_pages": round(float(qs.std(dim=1).mean()), 5)}}
        index_m.close()
    filler = torch.randn((rows_local, E), generator=g, device=dev)
    filler = filler / filler.norm(dim=1, keepdim=True)
    index.reset()
    index.add(filler)
    searches["filler"], ds_f = search_bench(index, filler, "filler", True)
    if emb_m is None:
        ds_main = ds_f
    # the exact fp32 pass and the band pass, priced: 8 / 256 queries forced through them (an error model nothing satisfies
    # flags every query with the whole index as its band -> exact pass; the templated corpus below exercises the band pass)
    def timed(nq_, reps=5):
        index.search(Q[:nq_], args.topk)
        torch.cuda.synchronize()        # (the warm call done: a band beyond 8192 rows met by the streaming search sets the index's
        e0.record()                     # host-visible word there, the calls behind it are routed to the exact pass)
        for _ in range(reps):
            index.search(Q[:nq_], args.topk)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    base8 = timed(8)
    index.set_search_eps(100.0)
    forced8 = timed(8)
    index.set_search_eps(None)
    searches["filler"]["exact_pass_ms_per_8q"] = round(forced8 - base8, 4)
    del filler
    templated = None
    if emb_m is not None and world == 1 and corpus_local >= 100 and not args.no_extras:
        n_fam = 100
        per = rows_local // n_fam
        t_f = torch.logspace(np.log10(0.176), np.log10(0.0316), n_fam, device=dev)       # pairwise cosine 0.97 .. 0.999 inside a family
        centers = emb_m[torch.linspace(0, corpus_local - 1, n_fam, device=dev).long()]
        rows_t = torch.empty((n_fam * per, E), dtype=torch.float32, device=dev)
        for f in range(n_fam):
            noise = torch.randn((per, E), generator=g, device=dev) / (E ** 0.5)
            r_ = centers[f][None, :] + t_f[f] * noise
            rows_t[f * per:(f + 1) * per] = r_ / r_.norm(dim=1, keepdim=True)
        index.reset()
        index.add(rows_t)
        templated, _ = search_bench(index, rows_t, "templated", True)
        templated["what"] = (f"{n_fam} families x {per} pages, contiguous, pairwise cosine 0.97 .. 0.999 inside a family (model embeddings "
                             "perturbed): the pre-pass threshold and the half-lists of a family's chunk sit inside the error band")
        templated["vs_filler_time"] = round(templated["local_sweep_ms"] / searches["filler"]["local_sweep_ms"], 2)
        del rows_t
    main_kind = "model" if emb_m is not None else "filler"
    srch = searches[main_kind]
    search_qps = srch["queries_per_sec"]
    # the exchange step alone: ONE all-gather of the packed [nq, k] keys (8 B each) per search
    gather_us = None
    if world > 1:
        mine = index.search_keys(Q, args.topk, id_offset=rank * rows_local)
        if shared:
            mine = mine.cpu()
        buf = torch.empty((world * mine.shape[0], mine.shape[1]), dtype=torch.int64, device=mine.device)
        for _ in range(3):
            dist.all_gather_into_tensor(buf, mine)
        barrier()
        tg0 = time.perf_counter()
        for _ in range(20):
            dist.all_gather_into_tensor(buf, mine)
        barrier()
        gather_us = max_over_ranks(time.perf_counter() - tg0) / 20 * 1e6

    # who took part in the exchange: every rank's id and the rows of ITS shard, all-gathered (the driver's N > 1 lines show which
    # transport ran and that every GPU held a shard)
    ranks_seen = rows_per_rank = None
    if world > 1:
        me = torch.tensor([[rank, srch["rows_per_gpu"], local_rank]], dtype=torch.int64, device="cpu" if shared else dev)
        allr = torch.empty((world, 3), dtype=torch.int64, device=me.device)
        dist.all_gather_into_tensor(allr, me)
        allr = allr.cpu().tolist()
        ranks_seen = sorted(int(r[0]) for r in allr)
        rows_per_rank = [int(r[1]) for r in sorted(allr)]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel class (by event-bracketed time)
    gemm_like = {k: v for k, v in prof.items() if k.startswith("vit_") and v["launches"] > 0}
    dom = max(gemm_like, key=lambda k: gemm_like[k]["ms"])
    d = prof[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    kernel_names = {"vit_qkv": "vr::gemm256w_bf16_kernel<0, true, 8> (EPI_BF16; ViT qkv, 256x256 tile, one wave per SIMD)",
                    "vit_attn": "vr::attention72w_kernel (ViT self-attention: persistent, one wave per SIMD, hand-ordered stream)",
                    "vit_proj": "vr::gemm256w_bf16_kernel<3, false, 6> (EPI_RESID; ViT attn proj, 256x192 tile, one wave per SIMD)",
                    "vit_fc1": "vr::gemm256w_bf16_kernel<1, true, 8> (EPI_GELU; ViT MLP fc1, 256x256 tile, one wave per SIMD)",
                    "vit_fc2": "vr::gemm256w_bf16_kernel<3, false, 6> (EPI_RESID; ViT MLP fc2, 256x192 tile, one wave per SIMD)"}
    roofline = {"bound": "mfma", "kernel": kernel_names[dom], "achieved": round(achieved, 2),
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                "avg_launch_ms": round(d["ms"] / d["launches"], 4), "launches": d["launches"],
                "flops_per_launch": d["flops"] / d["launches"], "traffic": None}
    # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process, so rank 0 runs the SAME encode
    # step under `rocprofv3 --pmc` in two child processes (FETCH_SIZE, then WRITE_SIZE: separate passes, kernel trace only,
    # folded by tools/pmc_traffic.py with the guide's gfx950 correction FETCH x 2) — measured in THIS run, on this box.  Only when
    # that fails (no rocprofv3 on the box) the committed measurement of an earlier run (profiles/rNN_traffic.json) is quoted instead.
    def traffic_live(key):
        import shutil
        import subprocess
        import tempfile
        if args.no_pmc or world != 1 or shutil.which("rocprofv3") is None:
            return None
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_traffic
        td = tempfile.mkdtemp(prefix="vr_pmc_", dir="/tmp")
        vals = {}
        try:
            for cn in ("FETCH_SIZE", "WRITE_SIZE"):
                d_ = os.path.join(td, cn)
                r_ = subprocess.run(["rocprofv3", "--pmc", cn, "--kernel-trace", "-d", d_, "-o", "enc", "--", sys.executable,
                                     os.path.join(ROOT, "tools", "encode_only.py"), "1"], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                                    capture_output=True, text=True, timeout=180)
                if r_.returncode != 0:
                    return None
                hit_ = [v for k, v in pmc_traffic.per_kernel(d_).items() if key in k]
                if not hit_:
                    return None
                vals[cn] = hit_[0]
            return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
        except Exception as e_:
            log("live PMC traffic failed:", repr(e_)[:200])
            return None
        finally:
            shutil.rmtree(td, ignore_errors=True)

    key = kernel_names[dom].split(" (")[0].replace("vr::", "")
    live = traffic_live(key)
    if live is not None:
        roofline["traffic"] = round(live)
        roofline["traffic_source"] = "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two child processes of tools/encode_only.py 1), 2 x FETCH + WRITE, per launch"
        algo = {"vit_qkv": 2.0 * (B * 1024 * 1152 + 3456 * 1152 + B * 1024 * 3456), "vit_proj": 2.0 * (B * 1024 * 1152 + 1152 * 1152) + 8.0 * B * 1024 * 1152,
                "vit_fc1": 2.0 * (B * 1024 * 1152 + 4304 * 1152 + B * 1024 * 4304), "vit_fc2": 2.0 * (B * 1024 * 4304 + 1152 * 4304) + 8.0 * B * 1024 * 1152,
                "vit_attn": 2.0 * (B * 1024 * 3456 + B * 1024 * 1152)}.get(dom)
        if algo:
            roofline["algorithmic_bytes"] = round(algo)
            roofline["traffic_over_algorithmic"] = round(live / algo, 3)
    else:
      try:
          import glob
          tfiles = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_traffic.json")))
          tj = json.load(open(tfiles[-1]))
          hit = [v for k, v in tj.items() if key in k]
          if hit:
              roofline["traffic"] = round(hit[0])
              roofline["traffic_source"] = os.path.basename(tfiles[-1]) + " (a COMMITTED earlier measurement: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch)"
      except Exception:
        pass
    def box_id():
        """which box, at which clocks: the pool's boxes differ by 7 % on the same library"""
        import socket
        import subprocess
        info = {"host": socket.gethostname(), "device": torch.cuda.get_device_name(local_rank)}
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            info.update(cus=pr.multi_processor_count, hbm_gb=round(pr.total_memory / 2 ** 30, 1))
            for attr in ("uuid", "pci_bus_id"):
                if hasattr(pr, attr):
                    info[attr] = str(getattr(pr, attr))
        except Exception:
            pass
        try:
            out_ = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
            js = json.loads(out_[out_.index("{"):])
            card = js.get(f"card{local_rank}", next(iter(js.values())))
            info["rocm_smi_after_the_timed_steps"] = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "power", "temperature (sensor junction)"))}
        except Exception as e:
            info["rocm_smi"] = repr(e)[:120]
        return info

    phases = {k: {"ms_per_step": round(v["ms"] / args.steps, 3),
                  "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)} for k, v in prof.items()}
    # (dec_* = the decoder phase split by events BETWEEN its kernels, taken in a pass of their own; "decoder" is the undisturbed total)
    f_page = cfg.flops_page(1024, len(items[0].input_ids))
    result = {
        "metric": "page-images embedded/sec (VisRAG-Ret encode, 448x448, bf16 MFMA) "
                  "+ queries/sec@top-10 over 100k-page index",
        "value": round(pages_per_s, 2), "unit": "pages/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "VisRAG-Ret (SigLIP-so400m 26 blk + resampler + MiniCPM-2B 40 layers) bf16 "
                               f"encode batch={B} page images 448x448 per GPU; index {args.index_rows} rows x 2304 "
                               f"sharded {world} way(s), {args.queries} queries top-{args.topk}",
                   "batch_per_gpu": B, "global_batch": B * world, "tokens_per_page": len(items[0].input_ids),
                   "flops_per_page": f_page, "parallelism": f"dp{world} (pages sharded, no data-path collective)"},
        "model_tflops": round(pages_per_s * f_page / 1e12 / world, 1),
        "model_frac_of_mfma_peak": round(pages_per_s * f_page / 1e12 / world / PEAK_BF16_TFLOPS, 4),
        "queries_per_sec": round(search_qps, 1),
        # the retrieval half on the index BASELINE configs[2] words: model-embedded pages, model-encoded queries
        # (index_kind "filler" only when the corpus embed was skipped with --corpus-pages 0)
        "search": dict(srch, exchange=None if world == 1 else {
                           "collective": "all_gather_into_tensor of the packed [nq, k] 64-bit keys, one per search",
                           "backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_seen": ranks_seen,
                           "rows_per_gpu": rows_per_rank, "devices_shared": bool(shared),
                           "bytes_per_rank": args.queries * args.topk * 8, "all_gather_us": round(gather_us, 1)},
                       query_encode_per_sec=round(args.queries / q_encode_s, 1)),
        "search_filler": searches["filler"] if main_kind != "filler" else None,
        "search_templated": templated,
        "corpus_embed": corpus_embed,
        "roofline": roofline,
        "phases": phases,
        "pipelined": pipelined,
        "box": box_id(),
    }

    # ---- extras (outside `value`): what the reference's own entry point sees, and real-document pages
    if world == 1 and not args.no_extras:
        try:
            import tempfile
            import types
            from visrag_amd.inference import distributed_parallel_embedding_inference
            from visrag_amd.modeling import DRModelForInference
            model = DRModelForInference(cfg, enc)
            model.set_pipeline(2)
            # (a) PIL pages in -> pickle shards out through distributed_parallel_embedding_inference: host
            #     prompt/tokenise, H2D of the pixels, GPU resize (identity for 448x448), encode, D2H, pickle
            n_pil = 128 * B                                   # (128 batches, ~6 s: pipeline fill and the last flush are ~1 % of the run)
            pil_pool = [Image.fromarray(p_) for p_ in pages]            # (the pool's 64 pages over and over: host RAM, not a shortcut —
            corpus = [{"id": str(i), "text": "", "image": pil_pool[i % pool]} for i in range(n_pil)]    # every page is converted and uploaded each time)
            with tempfile.TemporaryDirectory() as td:
                # split_save: a shard is flushed every 1024 pages (D2H + pickle on the calling thread while the two batches
                # in flight keep the GPU busy), like the reference's max_inmem_docs
                a_ = types.SimpleNamespace(output_dir=td, per_device_eval_batch_size=B, process_index=0, world_size=1,
                                           max_inmem_docs=1024, device=str(dev), dataloader_num_workers=1)
                distributed_parallel_embedding_inference(corpus[:2 * B], model, a_, "corpus", False,
                                                         {"tokenizer": tok, "max_inp_length": 2048})
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                distributed_parallel_embedding_inference(corpus, model, a_, "corpus", True,
                                                         {"tokenizer": tok, "max_inp_length": 2048})
                torch.cuda.synchronize()
                pil_s = time.perf_counter() - t0
            result["pil_pipeline"] = {"pages_per_sec": round(n_pil / pil_s, 1), "pages": n_pil, "batch": B,
                                      "vs_pipelined": round(n_pil / pil_s / pipelined["pages_per_sec"], 4) if pipelined else None,
                                      "what": "PIL images -> distributed_parallel_embedding_inference (loader thread, host prepare + H2D + "
                                              "encode, two batches in flight) -> a pickle shard per 1024 pages"}
            # (b) A4 pages rasterised at 200 dpi (1654x2339): 1 source + 3x3 slices of ~1026 patches, ~662 tokens
            from visrag_amd.gpu_resize import prepare_item_gpu
            a4 = np.ascontiguousarray(np.tile(pages[0], (6, 4, 1))[:2339, :1654])
            a4_dev = torch.from_numpy(a4).to(dev)
            one = prepare_item_gpu("", a4_dev, tok, cfg, 2048, local_rank)[0]
            n_sl, n_tok = len(one.slices), len(one.input_ids)
            nb = max(1, min(8, enc.max_tokens // n_tok))          # pages per call: the packed decoder tokens must fit
            its = [prepare_item_gpu("", a4_dev, tok, cfg, 2048, local_rank)[0] for _ in range(nb)]
            patches = sum((int(s_.shape[0]) // cfg.patch_size) * (int(s_.shape[1]) // cfg.patch_size) for s_ in its[0].slices)
            f_a4 = sum(cfg.flops_vit((int(s_.shape[0]) // cfg.patch_size) * (int(s_.shape[1]) // cfg.patch_size)) +
                       cfg.flops_resampler((int(s_.shape[0]) // cfg.patch_size) * (int(s_.shape[1]) // cfg.patch_size))
                       for s_ in its[0].slices) + cfg.flops_decoder(n_tok)
            for _ in range(2):
                enc.encode_items(its)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                its = [prepare_item_gpu("", a4_dev, tok, cfg, 2048, local_rank)[0] for _ in range(nb)]   # GPU resize + slicing
                enc.encode_items(its)
            torch.cuda.synchronize()
            a4_s = time.perf_counter() - t0
            result["sliced_pages"] = {"pages_per_sec": round(4 * nb / a4_s, 2), "page": "A4 @ 200 dpi, 1654x2339",
                                      "slices_per_page": n_sl, "patches_per_page": patches, "tokens_per_page": n_tok,
                                      "tflop_per_page": round(f_a4 / 1e12, 2),
                                      "model_tflops": round(4 * nb / a4_s * f_a4 / 1e12, 1),
                                      "frac_of_mfma_peak": round(4 * nb / a4_s * f_a4 / 1e12 / PEAK_BF16_TFLOPS, 4),
                                      "what": f"device-resident page -> PIL-exact GPU bicubic resize + slicing -> encode, {nb} pages per call"}
            for e_, _s in model._slots[1:]:
                e_.close()
        except Exception as e:   # informational
            result["extras_error"] = repr(e)
        # BASELINE config 5: EVisRAG-7B-shaped generation over the top-5 pages (vision tower + language model)
        try:
            from visrag_amd.evisrag import SamplingParams, bench_generate

            def chain(llm):
                """BASELINE config 5 as worded: a text query -> VisRAG-Ret query embedding -> top-5 over the HBM index ->
                the five pages fetched from the (host) page store -> EVisRAG generation (GPU image processing + tower +
                prefill + 64 answer tokens).  One query at a time like predict.py:128-149; wall clock per query."""
                sp_ = SamplingParams(temperature=0.1, repetition_penalty=1.05, max_tokens=64, stop_token_ids=())
                rng_ = np.random.default_rng(5)
                prompt = [int(t) for t in rng_.integers(1000, 50000, 60)] + [llm.cfg.image_token_id, 198] * 5 + \
                         [int(t) for t in rng_.integers(1000, 50000, 60)]
                ts, parts = [], []
                for qi_ in range(5):
                    torch.cuda.synchronize(); a0 = time.perf_counter()
                    qrep = enc.encode_items(prepare_batch([qtexts[qi_]], [None], tok, cfg, 512))
                    _, top = index.search(qrep, 5)
                    top = top.cpu().numpy()[0]
                    a1 = time.perf_counter()
                    fetched = [pages[int(j) % pool] for j in top]              # the page store: u8 arrays on the host
                    out_ = llm.generate([{"prompt_token_ids": prompt, "multi_modal_data": {"image": fetched}}], sp_)
                    torch.cuda.synchronize(); a2 = time.perf_counter()
                    assert len(out_[0].outputs[0].token_ids) == 64
                    ts.append(a2 - a0); parts.append((a1 - a0, a2 - a1))
                ts, parts = ts[1:], parts[1:]                                   # the first query warms the caches up
                return {"queries_per_s": round(1.0 / float(np.median(ts)), 3), "ms_per_query": round(float(np.median(ts)) * 1e3, 1),
                        "retrieve_ms": round(float(np.median([p_[0] for p_ in parts])) * 1e3, 2),
                        "generate_ms": round(float(np.median([p_[1] for p_ in parts])) * 1e3, 1),
                        "what": "query text -> vr_encode (split-precision text pass) -> vr_index_search top-5 over the 100k-row index "
                                "-> 5 pages (448 x 448 u8, host store) -> GPU resize/normalise/patchify -> tower -> prefill (1405 "
                                "tokens) -> 64 answer tokens; one query at a time, wall clock"}

            result["evisrag_generate"] = bench_generate(5, 64, 2, local_rank, chain=chain, a4_pages=True)
        except Exception as e:   # informational
            result["evisrag_error"] = repr(e)

    # ---- CPU baseline: the oracle (fp32 restatement of the reference) on the host cores, rank 0, N=1.
    #      kind "port": /root/reference does not exist on the GPU box, so the timed code is oracle/ (pinned to
    #      the reference by tests/test_oracle_golden.py).
    if world == 1 and not args.no_cpu_baseline:
        # the reference ITSELF cannot run on the GPU box (no /root/reference there): its timing in the build container is
        # re-measured by the committed tools/ref_cpu_baseline.py and read from the file that tool writes
        try:
            ref_in_container = json.load(open(os.path.join(ROOT, "profiles", "ref_cpu_baseline.json")))
        except Exception:
            ref_in_container = None
        try:
            from oracle import visrag_ret_oracle as O
            # 16 threads is the fastest setting for this fp32 forward on the many-core host
            # (measured: 16 -> 2.4 s/page, 32 -> 2.6, 64 -> 4.4, 128 -> 11 s/page)
            torch.set_num_threads(min(16, os.cpu_count() or 16))
            tc = time.time()
            W = {k: v.cpu() for k, v in iter_synth_weights(cfg, 0, device=dev)}
            log(f"cpu weights in {time.time() - tc:.1f}s; threads={torch.get_num_threads()}")
            n, bs = args.cpu_pages, 16
            it = (items * ((n + pool - 1) // pool))[:n]
            O.encode(W, cfg, [i.input_ids for i in it[:1]], [i.image_bound for i in it[:1]], [i.slices for i in it[:1]])
            tc = time.perf_counter()
            refs = [O.encode(W, cfg, [i.input_ids for i in it[lo:lo + bs]], [i.image_bound for i in it[lo:lo + bs]],
                             [i.slices for i in it[lo:lo + bs]]) for lo in range(0, n, bs)]      # batches of 16 (README.md:146)
            cpu_s = time.perf_counter() - tc
            ref = torch.cat(refs)
            dpx = (dev_pages * ((n + pool - 1) // pool))[:n]
            got = torch.cat([enc.encode_items(it[lo:lo + B], device_slices=dpx[lo:lo + B]).cpu() for lo in range(0, n, B)])
            cos = float((got * ref).sum(1).min())
            nq_cpu = 16
            tc = time.perf_counter()
            qref = O.encode(W, cfg, [i.input_ids for i in qitems[:nq_cpu]], [[]] * nq_cpu, [[]] * nq_cpu)
            cpu_q_s = time.perf_counter() - tc
            qcos = float((Q[:nq_cpu].cpu() * qref).sum(1).min())
            # north_star's bar is on SCORES: query x page scores of the GPU embeddings against the CPU restatement's
            s_ref = qref @ ref.T
            err_q = float((Q[:nq_cpu].cpu() @ ref.T - s_ref).abs().max())       # the queries' side alone
            err_p = float((qref @ got.T - s_ref).abs().max())                   # the pages' side alone
            err_joint = float((Q[:nq_cpu].cpu() @ got.T - s_ref).abs().max())
            # retrieval baseline: fp32 matmul + topk over the FULL index size (dense_retriever.py:28-30)
            gC = torch.Generator().manual_seed(7)
            Cc = torch.randn((args.index_rows, cfg.hidden_size), generator=gC); Cc = Cc / Cc.norm(dim=1, keepdim=True)
            Qc = Q.cpu()
            tc = time.perf_counter()
            torch.topk(torch.matmul(Qc, Cc.T), k=args.topk)        # the reference's two ops (dense_retriever.py:28-30)
            cpu_search_s = time.perf_counter() - tc
            result["cpu_baseline"] = {
                "value": round(n / cpu_s, 3), "unit": "pages/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{n} pages 448x448 in batches of {bs} through oracle/visrag_ret_oracle.py (torch-CPU fp32 restatement "
                          f"of the reference forward, full dims) in {cpu_s:.1f}s; {nq_cpu} text queries in {cpu_q_s:.1f}s; search: "
                          f"{args.queries} queries x {args.index_rows} rows fp32 matmul+topk in {cpu_search_s:.2f}s (no extrapolation)",
                "query_encode_per_sec": round(nq_cpu / cpu_q_s, 2),
                "queries_per_sec_search": round(args.queries / cpu_search_s, 1),
                "parity_min_cosine_vs_gpu": round(cos, 6), "parity_min_cosine_queries": round(qcos, 6),
                "parity_max_score_err": {"queries_side": round(err_q, 7), "pages_side": round(err_p, 7), "joint": round(err_joint, 7),
                                         "pairs": f"{nq_cpu} queries x {n} pages", "bar": 1e-3},
                "reference_in_build_container": ref_in_container}
        except Exception as e:   # the baseline is informational; never lose the GPU numbers
            result["cpu_baseline"] = {"value": None, "error": repr(e)}
        # north_star's "identical top-k doc IDs, cosine scores within 1e-3" against the REFERENCE ITSELF (not the port): the
        # committed fixtures of oracle/gen_golden.py — openmatch's DRModelForInference + distributed_parallel_retrieve (top-10)
        # over (`reference_parity`) 51 slide decks x 10 pages + the reference's own four input images x 1024 queries, where the
        # cut falls between decks and 610 queries must return the reference's id set, and (`reference_parity_unrelated_pages`)
        # 512 unrelated pages + the same four images x 514 queries — re-encoded here
        def ref_parity(name, min_strict):
            from tests import config1xl_util as X
            g_, man_ = X.load_fixture(name)
            corpus_, queries_ = X.corpus_and_queries(g_, man_)
            P_ = []
            n_syn = int(g_["n_pages"])
            for lo in range(0, n_syn, B):
                chunk_ = corpus_[lo:min(lo + B, n_syn)]
                its_ = prepare_batch([""] * len(chunk_), [c_["image"] for c_ in chunk_], tok, cfg, 2048)
                P_.append(enc.encode_items(its_).cpu())
            for c_ in corpus_[n_syn:]:                                  # the reference's sliced pages, one per call (demo.py:44-58)
                P_.append(enc.encode_items(prepare_batch([""], [c_["image"]], tok, cfg, 2048)).cpu())
            P_ = torch.cat(P_)
            qi_ = prepare_batch([q_["text"] for q_ in queries_], [None] * len(queries_), tok, cfg, 512)
            Q_ = torch.cat([enc.encode_items(qi_[lo:lo + 64]).cpu() for lo in range(0, len(qi_), 64)])
            ix_ = HipIndex(cfg.hidden_size, len(P_), device=local_rank)
            ix_.add(P_.to(dev))
            sc_, id_ = ix_.search(Q_.to(dev), int(g_["k"]))
            ix_.close()
            doc_ids_ = [str(x_) for x_ in g_["doc_ids"]]
            sc_, id_ = sc_.cpu().numpy(), id_.cpu().numpy()
            run_ = {f"q{i}": {doc_ids_[int(j)]: float(v) for v, j in zip(sc_[i], id_[i])} for i in range(len(Q_))}
            st_ = X.parity_stats(g_, P_.numpy(), Q_.numpy(), run_, fixture=name)
            if "gap_hist" in g_.files:
                st_["reference_gap_histogram"] = {"upper_edges": [float(x_) for x_ in g_["gap_hist_edges"][1:]], "queries": [int(x_) for x_ in g_["gap_hist"]]}
            try:
                X.assert_bars(st_, min_strict=min_strict)
                st_["bars_met"] = True
            except AssertionError:
                st_["bars_met"] = False
            return st_
        for key_, name_, ms_ in (("reference_parity", "config1sep", 400), ("reference_parity_unrelated_pages", "config1xl", 30)):
            try:
                st_ = ref_parity(name_, ms_)
                if isinstance(result.get("cpu_baseline"), dict):
                    result["cpu_baseline"][key_] = st_
                else:
                    result[key_] = st_
            except Exception as e:
                result[key_ + "_error"] = repr(e)
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
