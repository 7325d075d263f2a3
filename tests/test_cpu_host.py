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
    assert U.shard_rank("/x/embeddings.corpus.rank.12.0-3") == 12 and U.shard_rank("embeddings.corpus.v1.5.rank.3") == 3     # dots in the type
    for r in (1, 10):
        U.write_shard(str(tmp_path / U.shard_name("corpus.v1.5", r)), reps, ["a", "b", "c"])
    assert [os.path.basename(f) for f in U.list_shards(str(tmp_path), "corpus.v1.5", 1)] == ["embeddings.corpus.v1.5.rank.1"]
    qrel = {qids[0]: {str(g["docs"][0][1]): 1}}
    assert U.eval_mrr(qrel, {qids[0]: run[qids[0]]}, 10) == {qids[0]: 0.5, 'all': 0.5}
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


def test_build_flags_mfma_operand_hazards():
    """attention_w.hip issues its MFMAs as asm statements on arch VGPRs, so hipcc neither pads nor orders around them.  The two
    hazards met on gfx950 must fail the build: a VALU write of an operand right in front of the MFMA that
    reads it, and an instruction reusing an MFMA's result registers while it is still in the pipe.  The accumulate chain
    (the next MFMA takes the whole result as its C) and the same code with enough distance are fine."""
    from visrag_amd.build import MFMA_HAZARD_CHECKED, SOURCES, mfma_operand_hazards
    mfma = "v_mfma_f32_16x16x32_bf16 v[20:23], v[0:3], v[4:7], v[20:23]"
    assert mfma_operand_hazards(f"\n\t{mfma}\n\t{mfma}\n\ts_endpgm\n") == []                  # accumulate chain
    zero_c = f"\n\tv_mov_b32 v20, 0\n\t{mfma}\n"
    hz = mfma_operand_hazards(zero_c)
    assert len(hz) == 1 and "v_mov_b32 v20, 0" in hz[0]
    assert mfma_operand_hazards(f"\n\tv_mov_b32 v20, 0\n\ts_nop 1\n\t{mfma}\n") == []        # two wait states between
    reuse = f"\n\t{mfma}\n\tv_add_f32 v21, v8, v9\n"
    hz = mfma_operand_hazards(reuse)
    assert len(hz) == 1 and "touches the result" in hz[0]
    assert mfma_operand_hazards(f"\n\t{mfma}\n\ts_nop 15\n\tv_add_f32 v21, v8, v9\n") == []
    other = "v_mfma_f32_16x16x32_bf16 v[24:27], v[20:23], v[4:7], v[24:27]"                       # result read as A too early
    assert len(mfma_operand_hazards(f"\n\t{mfma}\n\t{other}\n")) == 1
    assert mfma_operand_hazards(f"\n\tv_mov_b32 v20, 0\n.LBB0_1:\n\t{mfma}\n") == []            # (not modelled across labels)
    assert MFMA_HAZARD_CHECKED <= set(SOURCES)


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


RETRIEVE_WORKER = r'''
import io, os, sys, types
sys.path.insert(0, os.environ["VR_ROOT"])
import numpy as np, torch, torch.distributed as dist
from oracle import visrag_ret_oracle as O
from visrag_amd.retriever import distributed_parallel_retrieve, merge_keys_host, pack_keys_host
from visrag_amd.utils import save_as_trec
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["VR_PORT"],
                        rank=int(os.environ["VR_RANK"]), world_size=2)
rank = dist.get_rank()
calls = {"n": 0}
real_gather = dist.all_gather_into_tensor
def counting_gather(*a, **k):
    calls["n"] += 1
    return real_gather(*a, **k)
dist.all_gather_into_tensor = counting_gather


class HostIndex:                      # CPU stand-in for HipIndex: the oracle's matmul + top-k (fixed tie rule)
    def __init__(self, dim, capacity, device=0):
        self.rows = np.zeros((0, dim), np.float32)
    def add(self, reps):
        self.rows = np.concatenate([self.rows, np.asarray(reps, np.float32)])
    def __len__(self):
        return len(self.rows)
    def _search(self, q, k):
        q = q.numpy() if isinstance(q, torch.Tensor) else q
        kk = min(k, len(self.rows))
        s, i = O.search_topk(q, self.rows, kk)
        pad = k - kk
        return (np.pad(s, ((0, 0), (0, pad)), constant_values=-np.inf), np.pad(i, ((0, 0), (0, pad)), constant_values=-1))
    def search(self, q, k):
        return self._search(q, k)
    def search_keys(self, q, k, id_offset=0):
        s, i = self._search(q, k)
        return torch.from_numpy(pack_keys_host(s, i, id_offset))
    def close(self):
        pass


def merge_keys(keys):
    s, i = merge_keys_host(keys.numpy(), keys.shape[2])
    return torch.from_numpy(s), torch.from_numpy(i)


def trec(res, path):
    save_as_trec(res, path)
    return open(path, "rb").read()


out = os.environ["VR_OUT"]
args = types.SimpleNamespace(output_dir=out, process_index=rank)
for k in (3, 7):
    for gt in (False, True):
        ref = distributed_parallel_retrieve(args, k, global_topk=gt, sharded=False, index_factory=HostIndex)
        n0 = calls["n"]
        got = distributed_parallel_retrieve(args, k, global_topk=gt, sharded=True, index_factory=HostIndex, merge_keys=merge_keys)
        assert calls["n"] == n0 + 1, calls                    # ONE data-path collective
        assert list(got.keys()) == list(ref.keys()) and got == ref, (rank, k, gt)
        a = trec(ref, os.path.join(out, f"ref.{rank}.{k}.{gt}.trec")); b = trec(got, os.path.join(out, f"got.{rank}.{k}.{gt}.trec"))
        assert a == b and len(a) > 0, (rank, k, gt)
# the switches: args.sharded_corpus / VISRAG_SHARDED_RETRIEVE route the unchanged call into the sharded form
n0 = calls["n"]
args.sharded_corpus = True
got = distributed_parallel_retrieve(args, 5, index_factory=HostIndex, merge_keys=merge_keys)
assert calls["n"] == n0 + 1
del args.sharded_corpus
os.environ["VISRAG_SHARDED_RETRIEVE"] = "1"
got2 = distributed_parallel_retrieve(args, 5, index_factory=HostIndex, merge_keys=merge_keys)
assert calls["n"] == n0 + 2 and got2 == got
os.environ["VISRAG_SHARDED_RETRIEVE"] = "0"
got3 = distributed_parallel_retrieve(args, 5, index_factory=HostIndex)
assert calls["n"] == n0 + 2 and got3 == got
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_corpus_sharded_retrieve_behind_the_kept_api_world2_gloo(tmp_path):
    """Row g: `distributed_parallel_retrieve` in its corpus-sharded form (rank r loads only the shards rank r wrote, all
    queries searched everywhere, ONE all-gather of packed keys, each rank returns ITS queries) == the replicated form of
    the reference, dict for dict and TREC file byte for byte — union-of-per-file-top-k and global top-k, a rank with two
    split files and one with a file shorter than k, exact ties across ranks."""
    import socket
    from visrag_amd.utils import shard_name, write_shard
    rng = np.random.default_rng(7)
    dim = 16
    C = rng.standard_normal((260, dim)).astype(np.float32)
    C[250] = C[20]; C[130] = C[20]                         # exact ties across files and ranks
    Q = rng.standard_normal((9, dim)).astype(np.float32)
    out = tmp_path / "emb"; out.mkdir()
    docs = [f"doc{i}" for i in range(260)]
    write_shard(str(out / shard_name("corpus", 0, 0, 120)), C[:120], docs[:120])
    write_shard(str(out / shard_name("corpus", 0, 120, 255)), C[120:255], docs[120:255])
    write_shard(str(out / shard_name("corpus", 1)), C[255:260], docs[255:260])       # 5 rows: shorter than k = 7
    write_shard(str(out / shard_name("query", 0)), Q[:5], [f"q{i}" for i in range(5)])
    write_shard(str(out / shard_name("query", 1)), Q[5:], [f"q{i}" for i in range(5, 9)])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(RETRIEVE_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, VR_ROOT=ROOT, VR_PORT=str(port), VR_RANK=str(r), VR_OUT=str(out))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_sharded_search_accepts_the_round2_hooks():
    """the `local_search` / `merge` kwargs and pack_topk / unpack_topk of round 2 still work (deprecated)."""
    import warnings
    import torch
    from oracle import visrag_ret_oracle as O
    from visrag_amd.retriever import pack_keys_host, pack_topk, sharded_search, unpack_topk
    rng = np.random.default_rng(3)
    C = rng.standard_normal((50, 8)).astype(np.float32); Q = rng.standard_normal((4, 8)).astype(np.float32)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sc, ids = sharded_search(None, torch.from_numpy(Q), 5, id_offset=100,
                                 local_search=lambda q, k: O.search_topk(q.numpy(), C, k),
                                 merge=lambda s, i: (s[0], i[0]))
        assert any(issubclass(x.category, DeprecationWarning) for x in w)
        rs, ri = O.search_topk(Q, C, 5)
        assert np.array_equal(ids.numpy(), ri + 100) and np.array_equal(sc.numpy(), rs)
        p = pack_topk(torch.from_numpy(rs), torch.from_numpy(ri), 100)
        assert np.array_equal(p.numpy(), pack_keys_host(rs, ri, 100))
        us, ui = unpack_topk(p)
        assert np.array_equal(us.numpy(), rs) and np.array_equal(ui.numpy(), ri + 100)


def test_sentencepiece_tokenizer_wrapper(tmp_path):
    """`SentencePieceTokenizer`: the attribute set the path reads from the reference's LlamaTokenizerWrapper
    (modeling_minicpmv.py:404-438) over a real sentencepiece model — trained here, llama-style (identity normaliser,
    dummy prefix, byte fallback, the MiniCPM-V markers as user-defined symbols): marker ids, `<unk>` placeholders,
    per-piece encoding with the legacy prefix, and the image_bound the host side derives from those ids."""
    import random
    import sentencepiece as spm
    from visrag_amd.config import tiny_config
    from visrag_amd.preprocess import prepare_batch
    from visrag_amd.synth import synth_pages
    from visrag_amd.tokenizer import SentencePieceTokenizer
    from PIL import Image
    words = ("revenue table chart figure growth annual report market share total net income page section summary results "
             "Represent this query for retrieving relevant documents").split()
    random.seed(0)
    sents = [" ".join(random.choice(words) for _ in range(random.randint(3, 12))) for _ in range(1500)]
    prefix = str(tmp_path / "tokenizer")
    spm.SentencePieceTrainer.train(sentence_iterator=iter(sents), model_prefix=prefix, vocab_size=400, model_type="bpe",
                                   unk_id=0, bos_id=1, eos_id=2, pad_id=-1, byte_fallback=True, character_coverage=1.0,
                                   normalization_rule_name="identity", remove_extra_whitespaces=False, add_dummy_prefix=True,
                                   user_defined_symbols=["<image>", "</image>", "<slice>", "</slice>"], minloglevel=2)
    (tmp_path / "tokenizer_config.json").write_text('{"add_bos_token": true, "add_eos_token": false}')
    tok = SentencePieceTokenizer.from_pretrained(str(tmp_path))
    sp = tok.sp_model
    for attr in ("im_start", "im_end", "unk_token", "slice_start", "slice_end", "add_bos_token", "bos_id", "eos_id", "unk_id",
                 "im_start_id", "im_end_id", "encode"):
        assert hasattr(tok, attr), attr
    assert (tok.unk_id, tok.bos_id, tok.eos_id) == (0, 1, 2) and tok.unk_token == "<unk>"
    assert tok.im_start_id == sp.piece_to_id("<image>") != 0 and tok.im_end_id == sp.piece_to_id("</image>") != 0
    # the placeholder of a page: bos, <image>, 64 x unk, </image> — exactly query_num ids between the markers (the reference's
    # scatter needs that) — then "\n" as its byte piece behind the dummy prefix
    ph = tok.im_start + tok.unk_token * 64 + tok.im_end
    ids = tok.encode(ph + "\n")
    assert ids[0] == tok.bos_id
    body = ids[1:]
    i0 = body.index(tok.im_start_id)
    assert i0 == 0
    assert body[i0 + 1:i0 + 65] == [0] * 64 and body[i0 + 65] == tok.im_end_id
    assert sp.piece_to_id("<0x0A>") in body[i0 + 66:]                     # identity normaliser: the newline survives as a byte
    # text pieces are encoded on their own: the legacy behaviour re-inserts the dummy prefix after a special token
    text = "revenue table"
    assert tok.encode(text) == [1] + sp.encode(text)
    assert tok.encode("<unk>" + text) == [1, 0] + sp.encode(text)
    assert tok.encode(text + "</s>" + text) == [1] + sp.encode(text) + [2] + sp.encode(text)
    assert tok.decode(tok.encode(text)[1:]) == text
    # through the host side of the path: ids / image_bound agree with the marker positions
    cfg = tiny_config()
    cfg.vocab_size = tok.vocab_size
    page = Image.fromarray(synth_pages(1, size=cfg.scale_resolution, seed=0)[0])
    items = prepare_batch(["a caption about revenue", "Represent this query for retrieving relevant documents: annual report"],
                          [page, None], tok, cfg, 2048)
    it = items[0]
    starts = [i for i, t in enumerate(it.input_ids) if t == tok.im_start_id]
    ends = [i for i, t in enumerate(it.input_ids) if t == tok.im_end_id]
    assert len(starts) == len(ends) == len(it.image_bound) == len(it.slices) == 1
    assert [tuple(b) for b in it.image_bound] == [(s + 1, e) for s, e in zip(starts, ends)]
    assert all(t == tok.unk_id for t in it.input_ids[starts[0] + 1:ends[0]]) and ends[0] - starts[0] - 1 == cfg.query_num
    assert items[1].image_bound == [] or len(items[1].image_bound) == 0
    assert max(items[1].input_ids) < tok.vocab_size and items[1].input_ids[0] == tok.bos_id
    with pytest.raises(ValueError):                                           # a model without the markers is refused
        spm.SentencePieceTrainer.train(sentence_iterator=iter(sents), model_prefix=prefix + "_plain", vocab_size=300, model_type="bpe",
                                       byte_fallback=True, character_coverage=1.0, minloglevel=2)
        SentencePieceTokenizer(prefix + "_plain.model")


def test_tokenizer_is_pinned_to_the_references_wrapper(golden_dir):
    """ADVICE r4 / verdict r5 item 8: `SentencePieceTokenizer.encode` against ids recorded from the REFERENCE's own
    `LlamaTokenizerWrapper` class (modeling_minicpmv.py:404-438) running on transformers' slow-tokenizer machinery
    (oracle/gen_golden_tokenizer.py) over the committed llama-style sentencepiece model (byte fallback, dummy prefix,
    identity normaliser, the MiniCPM-V markers as special tokens): the page placeholder `<image>` + 64 x `<unk>` + `</image>`,
    sliced-page grid placeholders (2 x 3 and 3 x 3), a captioned page, plain / non-ASCII (byte fallback) / special-token-bearing
    queries, whitespace, the empty string — and the image bounds the host side derives from those ids."""
    import json
    from PIL import Image
    from visrag_amd.config import full_config
    from visrag_amd.preprocess import get_grid_placeholder, image_placeholder, prepare_item
    from visrag_amd.tokenizer import SentencePieceTokenizer
    d = os.path.join(golden_dir, "tokenizer")
    tok = SentencePieceTokenizer.from_pretrained(d)
    exp = json.load(open(os.path.join(d, "expected.json")))
    meta = exp["meta"]
    assert (tok.bos_id, tok.eos_id, tok.unk_id, tok.im_start_id, tok.im_end_id) == (meta["bos_id"], meta["eos_id"], meta["unk_id"],
                                                                                   meta["im_start_id"], meta["im_end_id"])
    assert len(exp["prompts"]) >= 9
    for name, case in exp["prompts"].items():
        assert tok.encode(case["text"]) == case["ids"], name
    # the placeholder strings themselves are the reference's (get_grid_placeholder, modeling_minicpmv.py:595-609)
    ph = image_placeholder(tok, 64)
    assert exp["prompts"]["page_single"]["text"] == ph + "\n"
    assert exp["prompts"]["page_sliced_2x3"]["text"] == ph + get_grid_placeholder(tok, [2, 3], 64) + "\n"
    # image_bound lands on the <unk> runs: through prepare_item for a 448 x 448 page (one image) and a 1072 x 670 page (2 x 2 grid + source)
    cfg = full_config()
    for size, n_img in (((448, 448), 1), ((1072, 670), 5)):
        it = prepare_item("", Image.new("RGB", size, (200, 200, 200)), tok, cfg, 2048)
        assert len(it.image_bound) == n_img == len(it.slices)
        for s, e in it.image_bound:
            assert e - s == cfg.query_num and all(t == tok.unk_id for t in it.input_ids[s:e])
            assert it.input_ids[s - 1] == tok.im_start_id and it.input_ids[e] == tok.im_end_id
        if n_img == 1:
            assert it.input_ids == exp["prompts"]["page_single"]["ids"]
        else:
            assert it.input_ids.count(meta["slice_start_id"]) == 1 and it.input_ids.count(meta["slice_end_id"]) == 1


def test_prefetched_batches_order_errors_and_worker_modes():
    """inference._prefetched_batches (the DataLoader's role: inference.py:66-73): same batches in the same order on the
    calling thread, from a map-style dataset through a thread pool and from a plain iterable through a loader thread; an
    error in the loader surfaces in the consumer; PIL pages become u8 arrays in the loader when asked."""
    import threading
    from PIL import Image
    from visrag_amd.inference import _prefetched_batches

    class MapDS:
        def __init__(self): self.threads = set()
        def __len__(self): return 23
        def __getitem__(self, i):
            if i >= 23:
                raise IndexError(i)
            self.threads.add(threading.current_thread().name)
            return {"id": str(i), "text": f"t{i}", "image": Image.new("L", (4, 3), i) if i % 2 else None}

    def gen():
        for i in range(23):
            yield {"id": str(i), "text": f"t{i}", "image": None}

    ref = [b["id"] for b in _prefetched_batches(MapDS(), 5, 0)]
    assert ref == [[str(i) for i in range(lo, min(23, lo + 5))] for lo in range(0, 23, 5)]
    ds = MapDS()
    got = list(_prefetched_batches(ds, 5, 3, to_u8=True))
    assert [b["id"] for b in got] == ref and all(t.startswith("visrag-loader") for t in ds.threads)
    im = got[0]["image"][1]
    assert isinstance(im, np.ndarray) and im.shape == (3, 4, 3) and im.dtype == np.uint8 and (im == 1).all() and got[0]["image"][0] is None
    assert [b["id"] for b in _prefetched_batches(gen(), 5, 2)] == ref

    def bad():
        yield {"id": "0", "text": "", "image": None}
        raise RuntimeError("decode failed")
    with pytest.raises(RuntimeError, match="decode failed"):
        list(_prefetched_batches(bad(), 1, 1))
