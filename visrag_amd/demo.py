"""Drop-in for the demo pipeline of the reference (visrag_scripts/demo/visrag_pipeline/):

  * `encode(model, tokenizer, text_or_image_list)`            utils.py:12-32   (re-exported from modeling)
  * `add_pages(...)` / `add_pdfs(...)`  -> `reps.npy`, `index2img_filename.txt`, cached page PNGs
                                                               build_index.py:14-58
  * `retrieve(knowledge_base_path, query, topk, ...)` -> paths of the top-k page images
                                                               answer.py:14-40

Same on-disk knowledge base (`reps.npy` float32 [n_pages, 2304], `index2img_filename.txt` one file name per
row, `<pdf>_<idx>.png`), so a base built by either side can be queried by the other.  Differences that do
not change results: pages are embedded in batches (the reference encodes one page per forward,
build_index.py:40-41) and the query x corpus matmul + topk (answer.py:31-33) runs on the HBM-resident
HipIndex.  The generation step of answer.py (MiniCPM-V-2.6 chat) is outside this path.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from .engine import HipIndex
from .modeling import encode  # noqa: F401  (demo/visrag_pipeline/utils.py:12-32)

# answer.py:27 (singular "document": the demo's own instruction string, kept verbatim)
QUERY_INSTRUCTION = "Represent this query for retrieving relevant document: "


@torch.no_grad()
def add_pages(model, tokenizer, pages: Sequence, knowledge_base_path: str, names: Optional[Sequence[str]] = None,
              batch_size: int = 32, save_images: bool = True, append: bool = False) -> np.ndarray:
    """Embed PIL page images into `knowledge_base_path` (build_index.py:37-55 without the PDF rasteriser).
    `names[i]` is the cache file name of page i (default `page_<i>.png`).  Returns the [n, D] float32 reps."""
    os.makedirs(knowledge_base_path, exist_ok=True)
    names = list(names) if names is not None else [f"page_{i}.png" for i in range(len(pages))]
    if len(names) != len(pages):
        raise ValueError("names and pages must have the same length")
    reps: List[np.ndarray] = []
    for lo in range(0, len(pages), batch_size):
        reps.append(encode(model, tokenizer, list(pages[lo:lo + batch_size])))
    out = np.concatenate(reps).astype(np.float32) if reps else np.zeros((0, model.cfg.hidden_size), np.float32)
    rp, ip = os.path.join(knowledge_base_path, "reps.npy"), os.path.join(knowledge_base_path, "index2img_filename.txt")
    if append and os.path.exists(rp):
        out = np.concatenate([np.load(rp), out])
        with open(ip) as f:
            names = [n for n in f.read().split("\n") if n] + names
    if save_images:
        for img, name in zip(pages, names[-len(pages):] if len(pages) else []):
            img.save(os.path.join(knowledge_base_path, name))
    np.save(rp, out)
    with open(ip, "w") as f:
        f.write("\n".join(names))
    return out


def add_pdfs(model, tokenizer, pdf_dir: str, knowledge_base_path: str, dpi: int = 200, batch_size: int = 32) -> np.ndarray:
    """build_index.py:14-58: rasterise every PDF under `pdf_dir` at 200 dpi (PyMuPDF, like the reference) and
    embed the pages.  PyMuPDF is an optional dependency of the demo, not of the library."""
    try:
        import fitz  # PyMuPDF
    except ImportError as e:  # pragma: no cover - not installed in the build image
        raise ImportError("add_pdfs needs PyMuPDF (`fitz`), as the reference demo does; "
                          "rasterise the pages yourself and call add_pages()") from e
    from PIL import Image
    pages, names = [], []
    for fn in sorted(f for f in os.listdir(pdf_dir) if f.endswith(".pdf")):
        doc = fitz.open(os.path.join(pdf_dir, fn))
        for idx, page in enumerate(doc):
            pix = page.get_pixmap(dpi=dpi)
            pages.append(Image.frombytes("RGB", [pix.width, pix.height], pix.samples))
            names.append(f"{fn}_{idx}.png")
    return add_pages(model, tokenizer, pages, knowledge_base_path, names, batch_size)


def load_knowledge_base(knowledge_base_path: str, device: Optional[int] = None):
    """-> (HipIndex holding reps.npy in HBM, list of image file names)."""
    with open(os.path.join(knowledge_base_path, "index2img_filename.txt")) as f:
        names = f.read().split("\n")
    reps = np.load(os.path.join(knowledge_base_path, "reps.npy")).astype(np.float32)
    from .modeling import default_device
    ix = HipIndex(reps.shape[1], max(len(reps), 1), default_device() if device is None else device)
    if len(reps):
        ix.add(reps)
    return ix, names


@torch.no_grad()
def retrieve(knowledge_base_path: str, query: str, topk: int, model, tokenizer, index=None, names=None,
             return_scores: bool = False):
    """answer.py:14-40: paths of the `topk` most similar page images (None if the base does not exist).
    Pass `index, names = load_knowledge_base(path)` to keep the index resident between questions."""
    if not os.path.exists(knowledge_base_path):
        return None
    own = index is None
    if own:
        index, names = load_knowledge_base(knowledge_base_path, model.encoder.device)
    q = encode(model, tokenizer, [QUERY_INSTRUCTION + query])
    sc, ids = index.search(q, min(topk, max(len(index), 1)))
    if own:
        index.close()
    keep = [int(i) for i in ids[0] if i >= 0]
    paths = [os.path.join(knowledge_base_path, names[i]) for i in keep]
    return (paths, [float(s) for s in sc[0][: len(keep)]]) if return_scores else paths
