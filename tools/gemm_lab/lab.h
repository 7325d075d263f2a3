// Declarations shared by the GEMM experiment kernels (NOT part of libvisrag_hip.so; built by
// tools/gemm_lab/build.py into libvisrag_gemm_lab.so, entry point vr_lab_gemm).
#pragma once
#include "kernels.h"

namespace vr {
// variant ids of the round-1 experiments (0 / 3 / 7 / 9 / 12 / 13 are the product's, see kernels.h)
enum LabVariant { LAB_REG = 1, LAB_256 = 2, LAB_256P4 = 4, LAB_256MID = 5, LAB_256STAG = 6, LAB_32 = 8, LAB_256P = 10,
                  LAB_256T = 11, LAB_256W4 = 14 };    // (12 / 13 are the product's one-wave-per-SIMD kernels since round 2)
hipError_t launch_gemm_lab(const GemmArgs& a, int epilogue, int variant, hipStream_t s);
hipError_t launch_gemm32(const GemmArgs& a, int epilogue, hipStream_t s);      // 256^2 on v_mfma_f32_32x32x16_bf16
hipError_t launch_gemm256p(const GemmArgs& a, int epilogue, hipStream_t s);    // persistent, one workgroup per CU
hipError_t launch_gemm256t(const GemmArgs& a, int epilogue, hipStream_t s);    // 256^2 + L2 touch-ahead
hipError_t launch_gemm256w4(const GemmArgs& a, int epilogue, hipStream_t s);   // 256^2, 4 waves x (128 x 128)
hipError_t launch_gemm_ablate(const GemmArgs& a, int ablation, hipStream_t s); // variants 20..29: timing only, INVALID results
}  // namespace vr
