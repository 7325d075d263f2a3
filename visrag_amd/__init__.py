"""visrag_amd: MI355X-native VisRAG-Ret corpus-embedding + retrieval hot path.

Python host code mirrors the reference's src/openmatch encode()/retrieve() surface and
calls hand-written gfx950 HIP kernels through the C-ABI of libvisrag_hip.so
(include/visrag_hip.h).  PyTorch tensors are containers only.
"""
from .config import VisRAGRetConfig, full_config, tiny_config  # noqa: F401

__version__ = "0.1.0"
