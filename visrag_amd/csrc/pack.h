#pragma once
#include <hip/hip_runtime.h>
namespace vr {
// dst[(r/blk)*blk_stride + blk_off + r%blk][c] = (transpose ? src[c][r] : src[r][c]) as bf16
// lo_part: store bf16(v - bf16(v)) instead — the low half of a hi + lo split of an fp32 weight (hp_text.hip)
hipError_t launch_pack_weight(const void* src, int src_is_bf16, int rows, int cols, int src_ld,
                              int transpose, void* dst, int dst_ld, int blk, int blk_stride, int blk_off,
                              hipStream_t s, int lo_part = 0);
// dst[(r/rblk)*rstride + r%rblk][(c/cblk)*cstride + c%cblk] = src[r][c] as bf16
hipError_t launch_pack_weight_blocks(const void* src, int src_is_bf16, int rows, int cols, int src_ld, void* dst, int dst_ld,
                                     int rblk, int rstride, int cblk, int cstride, hipStream_t s);
// conv weight [D][3][P][P] -> bf16 [D][ld] with k = ky*3P + kx*3 + c (the image's byte order inside a patch)
hipError_t launch_pack_patch_weight(const void* src, int src_is_bf16, int D, int P, void* dst, int dst_ld, hipStream_t s);
hipError_t launch_to_f32(const void* src, int src_is_bf16, float* dst, size_t n, hipStream_t s);
// dst[(i/blk)*stride + off + i%blk] = src[i]
hipError_t launch_to_f32_blocks(const void* src, int src_is_bf16, float* dst, size_t n, int blk, int stride, int off, hipStream_t s);
}  // namespace vr
