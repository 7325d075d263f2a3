"""-m gpu: the EVisRAG generator at the reference's OPERATING POINT (src/evisrag/predict.py:112-149: five full-resolution
pages per query, max_tokens=2048) at the 7B widths with two layers / blocks and seeded weights, against the oracles on
the host cores:

  * the tower on five pages of 72 x 56 patches (20 160 patch rows -> 5 040 image tokens: 1 008 per page; 315 windows in
    the window block, five 4 032-row full-attention segments in the other);
  * prefill of an 8 300-token prompt (five 1 008-token pages + text): the flash-attention kernel and the tile GEMMs at
    8 k rows, 3-D rope positions from rope_index;
  * 16 teacher-forced decode steps over the 8 300-row cache (16 KV ranges of ~520 rows merged by their log-sum-exps);
  * 2 048 free-running tokens (captured decode + sample steps, state advancing on the device) == the host-driven loop,
    token for token, greedy and with temperature.
The 256-wide fixtures and tests/test_gpu_evisrag_7b.py stop at 1 405 prompt tokens and 24 answer tokens."""
import numpy as np
import pytest
import torch

from oracle.qwen_gen_oracle import QwenGenConfig, QwenGenOracle, apply_repetition_penalty, synth_weights
from oracle.qwen_vision_oracle import QwenVisionConfig, QwenVisionOracle, synth_vision_weights

pytestmark = pytest.mark.gpu

N_IMG, PGRID = 5, (72, 56)                  # patch grid of a page; 36 x 28 = 1008 image tokens
TOK_GRID = (PGRID[0] // 2, PGRID[1] // 2)
N_TOK = TOK_GRID[0] * TOK_GRID[1]
PROMPT_LEN = 8300


def _gen_cfg(cfg):
    from visrag_amd.evisrag import GenConfig
    return GenConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
                     rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, mrope_section=tuple(cfg.mrope_section),
                     image_token_id=5, eos_token_ids=())


@pytest.fixture(scope="module")
def setup():
    from visrag_amd.evisrag import LLM, VisionConfig
    cfg = QwenGenConfig(num_hidden_layers=2)
    vcfg = QwenVisionConfig(depth=2, fullatt_block_indexes=(1,))
    wg = synth_weights(cfg, seed=0, device="cuda")
    wv = synth_vision_weights(vcfg, seed=1)
    vc = VisionConfig(depth=vcfg.depth, hidden_size=vcfg.hidden_size, num_heads=vcfg.num_heads, intermediate_size=vcfg.intermediate_size,
                      out_hidden_size=vcfg.out_hidden_size, window_size=vcfg.window_size,
                      fullatt_block_indexes=tuple(vcfg.fullatt_block_indexes))
    w = dict(wg)
    w.update(wv)
    llm = LLM(_gen_cfg(cfg), max_model_len=PROMPT_LEN + 2048 + 64, max_prefill=PROMPT_LEN + 64, vision=vc,
              max_vision_rows=N_IMG * PGRID[0] * PGRID[1] + 256, weights=w)
    host = {k: v.cpu() for k, v in wg.items()}
    del w, wg
    torch.cuda.empty_cache()
    rng = np.random.default_rng(0)
    ids = rng.integers(16, cfg.vocab_size, PROMPT_LEN)
    at = 60
    for _ in range(N_IMG):
        ids[at:at + N_TOK] = 5
        at += N_TOK + 3
    assert at < PROMPT_LEN - 1000 and int((ids == 5).sum()) == N_IMG * N_TOK
    yield cfg, vcfg, host, wv, llm, ids.tolist()
    llm.close()


def test_tower_on_five_full_pages_at_7b_widths(setup):
    cfg, vcfg, host, wv, llm, ids = setup
    grids = np.array([[1, PGRID[0], PGRID[1]]] * N_IMG, dtype=np.int32)
    rows = int((grids[:, 1] * grids[:, 2]).sum())
    assert rows >= 20000
    rng = np.random.default_rng(3)
    px = torch.from_numpy(rng.standard_normal((rows, vcfg.patch_dim)).astype(np.float32)).to(torch.bfloat16).float().numpy()
    emb = llm.encode_images(px, grids)
    ref = QwenVisionOracle(vcfg, wv).forward(torch.from_numpy(px), [tuple(int(x) for x in g) for g in grids]).numpy()
    assert emb.shape == ref.shape == (rows // 4, cfg.hidden_size)
    scale = np.abs(ref).max()
    assert np.abs(emb - ref).max() < 2e-2 * scale, (np.abs(emb - ref).max(), scale)
    cos = (emb * ref).sum(-1) / (np.linalg.norm(emb, axis=-1) * np.linalg.norm(ref, axis=-1))
    assert cos.min() > 1 - 1e-3, cos.min()


def test_prefill_8k_and_decode_over_an_8k_cache_at_7b_widths(setup):
    from visrag_amd.evisrag import SamplingParams
    cfg, vcfg, host, wv, llm, ids = setup
    rng = np.random.default_rng(1)
    embs = [(0.05 * rng.standard_normal((N_TOK, cfg.hidden_size))).astype(np.float32) for _ in range(N_IMG)]
    pos3 = llm.prefill(ids, embs, [TOK_GRID] * N_IMG)
    assert pos3.shape == (3, PROMPT_LEN)
    o = QwenGenOracle(cfg, host)
    idt = torch.tensor(ids)
    emb = o.embed(idt).clone()
    emb[idt == 5] = torch.from_numpy(np.concatenate(embs))
    o_logits = o.forward(emb, torch.from_numpy(pos3).long(), last_only=True)[-1]
    ours, ref = llm.logits(), o_logits.numpy()
    scale = np.abs(ref).max()
    assert np.abs(ours - ref).max() < 1.5e-2 * scale, (np.abs(ours - ref).max(), scale)
    cos = float(ours @ ref / (np.linalg.norm(ours) * np.linalg.norm(ref)))
    assert cos > 1 - 2e-4, cos
    seen, nxt, exact = torch.tensor(ids), int(pos3.max()) + 1, 0
    sp = SamplingParams(temperature=0.0, repetition_penalty=1.05)
    for k in range(16):
        ours = torch.from_numpy(llm.logits())
        assert float((ours - o_logits).abs().max()) < 2e-2 * float(o_logits.abs().max()), k
        pick = llm.sample(sp, k)
        pen = apply_repetition_penalty(o_logits, seen, 1.05)
        want = int(torch.argmax(pen))
        if pick == want:
            exact += 1
        else:
            gap = float(pen[want] - pen[pick])
            assert 0 <= gap < 1e-2 * float(pen.abs().max()), (k, pick, want, gap)
        seen = torch.cat([seen, torch.tensor([want])])
        llm.decode(want, nxt)
        o_logits = o.forward(o.embed(torch.tensor([want])), torch.full((3, 1), nxt), last_only=True)[-1]
        nxt += 1
    assert exact >= 13, exact


def test_2048_free_running_tokens_equal_host_driven_tokens_after_an_8k_prompt(setup):
    from visrag_amd.evisrag import SamplingParams
    cfg, vcfg, host, wv, llm, ids = setup
    rng = np.random.default_rng(2)
    embs = [(0.05 * rng.standard_normal((N_TOK, cfg.hidden_size))).astype(np.float32) for _ in range(N_IMG)]
    pr = [{"prompt_token_ids": ids, "multi_modal_data": {"image_embeds": embs, "image_grids": [TOK_GRID] * N_IMG}}]
    for temperature in (0.0, 0.7):
        sp = SamplingParams(temperature=temperature, repetition_penalty=1.05, max_tokens=2048, seed=5, stop_token_ids=())
        a = llm.generate(pr, sp, pipelined=False)[0].outputs[0].token_ids
        b = llm.generate(pr, sp, pipelined=True)[0].outputs[0].token_ids
        assert len(a) == 2048 and a == b, temperature
