"""Diagnostic: error budget of the config-1 fixture (pages / queries cosine vs the reference, score errors)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from visrag_amd.config import full_config
from visrag_amd.modeling import DRModelForInference
from visrag_amd.synth import iter_synth_weights, synth_pages, synth_queries
from visrag_amd.tokenizer import StandInTokenizer
g = np.load("tests/golden/config1_full.npz")
cfg = full_config()
model = DRModelForInference.build(cfg=cfg, state_dict=iter_synth_weights(cfg, 0, device="cuda"), pipeline=1)
tok = StandInTokenizer(cfg.vocab_size)
pages = synth_pages(64, size=448, seed=0)
P = []
for lo in range(0, 64, 32):
    P.append(model(passage={"id": [""] * 32, "text": [""] * 32, "image": [Image.fromarray(p) for p in pages[lo:lo + 32]]}, tokenizer=tok).p_reps.cpu().numpy())
P = np.concatenate(P)
q = ["Represent this query for retrieving relevant documents: " + t for t in synth_queries(16, seed=0)]
Q = model(query={"id": [""] * 16, "text": q, "image": [None] * 16}, tokenizer=tok, max_inp_length=512).q_reps.cpu().numpy()
cp, cq = (P * g["p_reps"]).sum(1), (Q * g["q_reps"]).sum(1)
print("pages  1-cos: max %.2e mean %.2e" % ((1 - cp).max(), (1 - cp).mean()))
print("query  1-cos: max %.2e mean %.2e" % ((1 - cq).max(), (1 - cq).mean()))
S, R = Q @ P.T, g["scores"]
E = S - R
print("score err: max %.2e rms %.2e mean %.2e" % (np.abs(E).max(), np.sqrt((E ** 2).mean()), E.mean()))
# decomposition: error from pages only (reference queries) and from queries only (reference pages)
Ep, Eq = g["q_reps"] @ P.T - R, Q @ g["p_reps"].T - R
print("pages-only err: max %.2e rms %.2e ; queries-only err: max %.2e rms %.2e" % (np.abs(Ep).max(), np.sqrt((Ep ** 2).mean()), np.abs(Eq).max(), np.sqrt((Eq ** 2).mean())))
dp = P - g["p_reps"]; dq = Q - g["q_reps"]
mp = g["p_reps"].mean(0); mp /= np.linalg.norm(mp)
print("page err norm: mean %.2e ; component along the mean page direction: %.2e ; page-to-page err correlation: %.3f" % (
    np.linalg.norm(dp, axis=1).mean(), np.abs(dp @ mp).mean(), np.corrcoef(dp[:8])[np.triu_indices(8, 1)].mean()))
print("query err norm: mean %.2e ; query-to-query err correlation %.3f" % (np.linalg.norm(dq, axis=1).mean(), np.corrcoef(dq[:8])[np.triu_indices(8, 1)].mean()))
print("score range", R.min(), R.max(), "page-page cos mean", (g["p_reps"] @ g["p_reps"].T).mean())
