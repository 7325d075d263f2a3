#!/bin/bash
# first GPU run of the one-wave-per-SIMD ViT attention: numerics, then op-level A/B against the shipped kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/attn1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" > gpurun_out/attn1/tests.log 2>&1
tail -15 gpurun_out/attn1/tests.log
timeout 600 python tools/ab_attention.py visrag_amd/libvisrag_hip.so visrag_amd/libvisrag_hip_attnold.so > gpurun_out/attn1/ab.log 2>&1
cat gpurun_out/attn1/ab.log | tail -12
