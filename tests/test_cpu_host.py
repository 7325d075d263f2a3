"""CPU-only host-logic tests: C-ABI symbols, on-disk formats, slicing policy, loud failure
without a GPU, and the world_size-2 (gloo) sharded-retrieval merge."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from visrag_amd import _lib
from visrag_amd.preprocess import slice_image, choose_grid, find_best_resize
from visrag_amd.retriever import merge_topk_host
from visrag_amd import utils as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """include/visrag_hip.h is the contract: every `vr_*` it declares must be exported by the
    built library and bound by the ctypes table (no compute calls here: no GPU)."""
    hdr = open(os.path.join(ROOT, "include", "visrag_hip.h")).read()
    declared = set(re.findall(r"\b(vr_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    # ... and the same for the generator's header
    ghdr = open(os.path.join(ROOT, "include", "visrag_gen.h")).read()
    gdecl = set(re.findall(r"\b(vg_[a-z0-9_]+)\s*\(", ghdr))
    for name in sorted(gdecl):
        assert hasattr(lib, name), f"{name} declared in visrag_gen.h but not exported"
    assert gdecl == set(_lib.GEN_SIGNATURES), gdecl ^ set(_lib.GEN_SIGNATURES)
    assert lib.vr_version().startswith(b"visrag_hip")


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_silent_cpu_fallback():
    from visrag_amd.config import tiny_config
    from visrag_amd.engine import HipEncoder, HipIndex
    with pytest.raises(_lib.VisragHipError):
        HipEncoder(tiny_config())
    with pytest.raises(_lib.VisragHipError):
        HipIndex(64, 10)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under visrag_amd/ may import or exec it."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|importlib[^\n]*oracle|open\([^\n]*oracle/", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "visrag_amd")):
        for f in files:
            if f.endswith(".py"):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f


def test_slicing_policy_examples():
    """Sizes the survey probed on the reference's slice_image."""
    from PIL import Image
    def plan(w, h):
        src, patches, grid = slice_image(Image.new("RGB", (w, h)), 9, 448, 14)
        return src.size, grid, (patches[0][0].size if patches else None), sum(len(r) for r in patches)
    assert plan(448, 448) == ((448, 448), None, None, 0)
    assert plan(1114, 1670) == ((364, 546), [2, 4], (518, 392), 8)
    assert plan(1654, 2339) == ((378, 532), [3, 3], (378, 532), 9)
    assert plan(2160, 1790)[1] == [3, 3] and plan(1072, 670)[1] == [2, 2] and plan(564, 3040)[1] == [1, 8]
    assert choose_grid((448, 448), 9, 448) is None


def test_trec_and_shard_formats(tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, "retrieve.npz"))
    qids = [str(q) for q in g["qids"]]
    run = {q: {str(d): float(s) for d, s in zip(g["docs"][i], g["scores"][i])} for i, q in enumerate(qids)}
    p = tmp_path / "out" / "test.0.trec"
    U.save_as_trec(run, str(p))
    assert open(p).read() == str(g["trec"])            # byte-identical to the reference's writer
    back = U.load_from_trec(str(p))
    assert back.keys() == run.keys() and all(abs(back[q][d] - run[q][d]) < 1e-9 for q in run for d in run[q])
    reps = np.arange(12, dtype=np.float32).reshape(3, 4)
    sp = tmp_path / U.shard_name("corpus", 1, 0, 3)
    U.write_shard(str(sp), reps, ["a", "b", "c"])
    r2, ids = U.read_shard(str(sp))
    assert np.array_equal(r2, reps) and ids == ["a", "b", "c"] and sp.name == "embeddings.corpus.rank.1.0-3"
    qrel = {qids[0]: {str(g["docs"][0][1]): 1}}
    assert abs(U.eval_mrr(qrel, {qids[0]: run[qids[0]]}, 10) - 0.5) < 1e-9
    nd, rc = U.ndcg_recall_at_k(qrel, run, 10)
    assert rc == 1.0 and abs(nd - 1.0 / np.log2(3)) < 1e-9


def test_merge_rule_equals_global_topk():
    """Sharding rows over P parts, taking local top-k and merging == global top-k (the
    invariant the multi-GPU path relies on)."""
    from oracle import visrag_ret_oracle as O
    rng = np.random.default_rng(0)
    C = rng.standard_normal((1000, 32)).astype(np.float32)
    Q = rng.standard_normal((9, 32)).astype(np.float32)
    C[500] = C[20]                                      # cross-shard exact tie
    k, P = 10, 4
    per = 250
    parts_s, parts_i = [], []
    for r in range(P):
        s, i = O.search_topk(Q, C[r * per:(r + 1) * per], k)
        parts_s.append(s); parts_i.append(i + r * per)
    ms, mi = merge_topk_host(np.stack(parts_s), np.stack(parts_i), k)
    gs, gi = O.search_topk(Q, C, k)
    assert np.array_equal(mi, gi) and np.allclose(ms, gs)


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["VR_ROOT"])
import numpy as np, torch, torch.distributed as dist
from oracle import visrag_ret_oracle as O
from visrag_amd.retriever import merge_keys_host, pack_keys_host, sharded_search
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["VR_PORT"],
                        rank=int(os.environ["VR_RANK"]), world_size=2)
rank = dist.get_rank()
rng = np.random.default_rng(0)
C = rng.standard_normal((400, 16)).astype(np.float32); Q = rng.standard_normal((5, 16)).astype(np.float32)
C[300] = C[20]                                          # exact tie across the two shards
Q[4] = -Q[0]                                            # negative scores: the key order must hold below zero too
lo = rank * 200
calls = {"n": 0}
real_gather = dist.all_gather_into_tensor
def counting_gather(*a, **k):
    calls["n"] += 1
    return real_gather(*a, **k)
dist.all_gather_into_tensor = counting_gather
# the PRODUCT's multi-rank function; only the device pieces are swapped for CPU stand-ins:
# HipIndex.search_keys -> the oracle's matmul + top-k on this rank's shard packed by the host statement of the
# key format, vr_topk_merge_keys -> merge_keys_host
def local_search_keys(q, k, id_offset):
    s, i = O.search_topk(q.numpy(), C[lo:lo + 200], k)
    return pack_keys_host(s, i, id_offset)
def merge_keys(keys):
    s, i = merge_keys_host(keys.numpy(), keys.shape[2])
    return torch.from_numpy(s), torch.from_numpy(i)
ms, mi = sharded_search(None, torch.from_numpy(Q), 7, id_offset=lo, local_search_keys=local_search_keys, merge_keys=merge_keys)
assert calls["n"] == 1, calls                           # ONE exchange step
rs, ri = O.search_topk(Q, C, 7)
assert np.array_equal(mi.numpy(), ri) and np.array_equal(ms.numpy(), rs), rank
# fewer rows than k on one rank: its tail (empty keys) must survive the packed exchange
def short_search_keys(q, k, id_offset):                 # (HipIndex pads a short shard with empty slots)
    n = 3 if rank == 1 else 200
    s, i = O.search_topk(q.numpy(), C[lo:lo + n], min(k, n))
    pad = k - s.shape[1]
    return pack_keys_host(np.pad(s, ((0, 0), (0, pad)), constant_values=-np.inf), np.pad(i, ((0, 0), (0, pad)), constant_values=-1),
                          id_offset)
ms, mi = sharded_search(None, torch.from_numpy(Q), 7, id_offset=lo, local_search_keys=short_search_keys, merge_keys=merge_keys)
rs, ri = O.search_topk(Q, np.concatenate([C[:200], C[200:203]]), 7)
assert np.array_equal(mi.numpy(), ri) and np.array_equal(ms.numpy(), rs), rank
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_retrieval_world2_gloo(tmp_path):
    """world_size 2 on CPU (gloo): the product's retriever.sharded_search (shard -> local top-k ->
    ONE packed all_gather -> merge) == global search."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, VR_ROOT=ROOT, VR_PORT=str(port), VR_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_pil_resize_oracle_is_bit_exact():
    """oracle/pil_resize_oracle.py (the arithmetic resize.hip implements) == Pillow, bit for bit."""
    from PIL import Image
    from oracle.pil_resize_oracle import resize_bicubic
    rng = np.random.default_rng(0)
    for (h, w), (ow, oh) in [((300, 200), (140, 84)), ((200, 300), (364, 546)), ((64, 50), (518, 392)),
                             ((333, 517), (112, 112)), ((100, 100), (100, 37)), ((97, 131), (210, 131))]:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.Resampling.BICUBIC))
        assert np.array_equal(resize_bicubic(img, (ow, oh)), ref), ((h, w), (ow, oh))


def test_checkpoint_config_and_iteration(tmp_path):
    """config_from_checkpoint / iter_checkpoint (the loader half of DRModelForInference.build,
    dense_retrieval_model.py:233-318): config.json fields, safetensors shards in sorted order, dtypes kept."""
    import json
    import torch
    from safetensors.torch import save_file
    from visrag_amd.modeling import config_from_checkpoint, iter_checkpoint
    d = tmp_path / "ckpt"; d.mkdir()
    (d / "config.json").write_text(json.dumps({
        "hidden_size": 2304, "num_hidden_layers": 40, "num_attention_heads": 36, "intermediate_size": 5760,
        "vocab_size": 122753, "rms_norm_eps": 1e-5, "scale_emb": 12, "scale_depth": 1.4, "query_num": 64,
        "patch_size": 14, "max_slice_nums": 9, "scale_resolution": 448, "slice_mode": False, "dim_model_base": 256}))
    c = config_from_checkpoint(str(d))
    assert (c.hidden_size, c.num_layers, c.num_heads, c.intermediate_size, c.vocab_size) == (2304, 40, 36, 5760, 122753)
    assert c.slice_mode is False and c.scale_emb == 12 and abs(c.residual_scale - 1.4 / 40 ** 0.5) < 1e-12
    save_file({"b.weight": torch.ones(2, 3, dtype=torch.float16), "a.weight": torch.zeros(4, dtype=torch.bfloat16)},
              str(d / "model-00002-of-00002.safetensors"))
    save_file({"c.bias": torch.arange(3, dtype=torch.float32)}, str(d / "model-00001-of-00002.safetensors"))
    got = list(iter_checkpoint(str(d)))
    assert [k for k, _ in got] == ["c.bias", "a.weight", "b.weight"]
    assert [t.dtype for _, t in got] == [torch.float32, torch.bfloat16, torch.float16]
    e = tmp_path / "empty"; e.mkdir()
    with pytest.raises(FileNotFoundError):
        list(iter_checkpoint(str(e)))


def test_device_resolution(monkeypatch):
    """build(device=None) follows LOCAL_RANK (torchrun) like the reference's encoding_args.device; explicit
    cuda specs are parsed; non-cuda devices are refused (no CPU fallback)."""
    import torch
    from visrag_amd.modeling import _device_index, default_device
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert default_device() == 0
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert default_device() in (3, 3 % max(1, torch.cuda.device_count() or 4))
    assert _device_index("cuda:2") == 2 and _device_index(5) == 5 and _device_index(torch.float16) is None
    with pytest.raises(RuntimeError):
        _device_index("cpu")


def test_build_flags_compiler_use_of_accumulation_registers():
    """build.py disassembles the one-wave-per-SIMD kernels (their accumulators are hand-allocated in asm text, the
    compiler only sees clobbers): any accumulation-register instruction outside the asm blocks must fail the build."""
    from visrag_amd.build import AGPR_CHECKED, SOURCES, agpr_violations
    ok = """
        v_add_f32 v1, v2, v3
        ;;#ASMSTART
        v_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]
        v_accvgpr_read_b32 v9, a3
        ;;#ASMEND
        s_endpgm
    """
    assert agpr_violations(ok) == []
    spill = ok + "\n\tv_accvgpr_write_b32 a17, v1 ; the compiler parking a VGPR\n"
    assert agpr_violations(spill) == ["v_accvgpr_write_b32 a17, v1"]
    # gfx950 can address accumulation registers directly: any a-register OPERAND outside the asm blocks counts, a
    # comment that mentions one does not, and neither does a spilled VGPR go unnoticed
    direct = ok + "\n\tds_read_b128 a[0:3], v5\n\tscratch_store_dword off, a5, s0\n\tv_mov_b32 v1, v2 ; not a5\n"
    assert agpr_violations(direct) == ["ds_read_b128 a[0:3], v5", "scratch_store_dword off, a5, s0"]
    assert len(agpr_violations(ok + "\n\t.vgpr_spill_count: 2\n")) == 1 and agpr_violations(ok + "\n\t.vgpr_spill_count: 0\n") == []
    assert AGPR_CHECKED <= set(SOURCES) and {"gemm256w.hip", "search256w.hip"} <= AGPR_CHECKED


def test_evisrag_rope_index_layout():
    """positions of a prompt with two images (Qwen2.5-VL get_rope_index for stills): text runs all three axes
    together, an image pins the temporal axis and walks its grid, text resumes past the largest position."""
    from visrag_amd.evisrag import rope_index
    IMG = 9
    ids = [1, 2, 3] + [IMG] * 6 + [4] + [IMG] * 4 + [5, 6]
    pos = rope_index(ids, IMG, [(2, 3), (2, 2)])
    assert pos[:, :3].tolist() == [[0, 1, 2]] * 3
    assert pos[0, 3:9].tolist() == [3] * 6 and pos[1, 3:9].tolist() == [3, 3, 3, 4, 4, 4] and pos[2, 3:9].tolist() == [3, 4, 5, 3, 4, 5]
    assert pos[:, 9].tolist() == [6, 6, 6]                       # 3 + max(2, 3)
    assert pos[0, 10:14].tolist() == [7] * 4 and pos[1, 10:14].tolist() == [7, 7, 8, 8] and pos[2, 10:14].tolist() == [7, 8, 7, 8]
    assert pos[:, 14:].tolist() == [[9, 10]] * 3                 # 7 + max(2, 2)
    import pytest
    with pytest.raises(ValueError):
        rope_index(ids, IMG, [(2, 3)])


def test_evisrag_token_loop_and_placeholder_expansion():
    """The host logic of evisrag.LLM that needs no GPU: one-placeholder-per-image expansion, and the token loop that keeps
    one captured step enqueued ahead of the token it inspects (driven here by a stub in place of the library)."""
    from types import SimpleNamespace

    from visrag_amd.evisrag import LLM, GenConfig, SamplingParams

    me = SimpleNamespace(cfg=GenConfig(image_token_id=5))
    assert LLM.expand_image_tokens(me, [1, 5, 2, 5, 3], [2, 3]) == [1, 5, 5, 2, 5, 5, 5, 3]
    assert LLM.expand_image_tokens(me, [1, 5, 5, 2, 5, 5, 5, 3], [2, 3]) == [1, 5, 5, 2, 5, 5, 5, 3]      # already expanded
    with pytest.raises(ValueError):
        LLM.expand_image_tokens(me, [1, 5, 2], [2, 3])

    class Stub:
        """Produces the token stream 100, 101, ...; counts what the loop enqueues."""
        def __init__(self):
            self.issued = self.ended = 0
            self.begun = None
            self.max_ahead = 0
            self.collected = 0

        def run_begin(self, position, sp, first_step):
            self.begun = (position, first_step)

        def run_step(self):
            self.issued += 1
            self.max_ahead = max(self.max_ahead, self.issued - self.collected)

        def run_token(self, i):
            assert i == self.collected and i < self.issued
            self.collected += 1
            return 101 + i

        def run_end(self):
            self.ended += 1

        def decode(self, tok, pos):
            self.issued += 1
            self.last = (tok, pos)

        def sample(self, sp, step):
            return 100 + step

    sp = SamplingParams(temperature=0.0)
    for pipelined in (True, False):
        s = Stub()
        assert LLM._continue(s, 100, 40, 6, sp, set(), pipelined) == [100, 101, 102, 103, 104, 105]
        assert s.issued == 5                                        # limit 6 tokens = the first one + five steps, never more
        if pipelined:
            assert s.begun == (40, 1) and s.ended == 1 and s.max_ahead == 2
        else:
            assert s.last == (104, 44)
        s = Stub()
        assert LLM._continue(s, 100, 40, 50, sp, {103}, pipelined) == [100, 101, 102, 103]
        assert s.issued == (4 if pipelined else 3)                  # a stop token costs at most one step already in flight
        s = Stub()
        assert LLM._continue(s, 100, 40, 50, sp, {100}, pipelined) == [100] and s.issued == 0
        s = Stub()
        assert LLM._continue(s, 100, 40, 1, sp, set(), pipelined) == [100] and s.issued == 0
