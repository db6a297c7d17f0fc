// WStreamT: the weight stream of the 16- / 32-row kernels (petr_attn_out_ln, petr_ln_proj, petr_ln_bwd_proj, petr_ffn_fwd / bwd,
// petr_branch_fwd): a 256-deep product of LDS-resident rows with a k-major weight that is read straight from L2.
#ifndef PETR_WSTREAM_H_
#define PETR_WSTREAM_H_
#include "common.h"

namespace {

constexpr int AO_C = 256, AO_ROWS = 16, AO_PITCH = AO_C + 4;

// The weight is read k-major (wT [K][N]: row k holds the N outputs' weights of input k): lane
// (n = lane & 15, kq = lane >> 4) takes ONE float2 wT[k][col0 + 2 n, + 1] per k - the 16 lanes of a k share one 128-byte line, an
// instruction touches 4 full lines.  (With W [N][K] every lane of an instruction sits in a line of its own: 64 tag lookups per
// 1 KB, and the 16-row kernels ran at the L1's lookup rate - 14 bytes / clock / CU measured - not at the matrix cores'.)  The
// wave's two column tiles are the even and the odd columns of its 32.  Group g = 16 k: lane kq holds k = 16 g + 4 kq + j,
// j = 0..3, against one float4 of the A row; eight groups (half of K = 256) in flight.
template <int RG>
struct WStreamT {
  float2 f[8][4];
  // w: wave-uniform pointer to wT[0][first column of the wave's 32]; lo = byte offset of the lane = (4 (lane >> 4) * ld + 2 (lane & 15)) * 4
  static __device__ __forceinline__ uint32_t lane_off(int lane, long ld) { return (uint32_t)((4 * (lane >> 4) * ld + 2 * (lane & 15)) * 4); }
  // buffer loads: resource = the wave's column base, VGPR offset = the lane's, SGPR offset = the row's (scalar multiplies) - no
  // vector instruction computes an address (global loads took a 64-bit VALU add per load, on the port the f32 MFMAs issue on)
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ float2 ld2(__amdgpu_buffer_rsrc_t r, uint32_t lo, long row_floats) {
    const i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)lo, (int)(row_floats * 4), 0);
    return make_float2(__int_as_float(v[0]), __int_as_float(v[1]));
  }
  static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* w) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), (short)0, 0x7fffffff, 0x00020000);
  }
  __device__ __forceinline__ void first(const float* w, long ld, uint32_t lo) {
    const __amdgpu_buffer_rsrc_t r = rsrc(w);
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) f[g][j] = ld2(r, lo, (16 * g + j) * ld);
  }
  // acc[rg][0 / 1] += A row group rg (16 x 256, LDS rows at arow + rg * 16 * pitch, arow = As + (lane & 15) * pitch + 4 (lane >> 4))
  // x wT[:, even / odd columns].  RG row groups share every weight fragment (RG x the MFMAs per byte streamed).  The stream does
  // not drain between products: once the second half of this weight is in flight, the slots that free up take the first half
  // of the NEXT product's weight (wn, null: none) - each product otherwise starts with a full memory latency in the open.
  __device__ __forceinline__ void run(const float* arow, int pitch, const float* w, long ld, uint32_t lo, const float* wn, long ldn, uint32_t lon,
                                      f32x4 (&acc)[RG][2]) {
    const __amdgpu_buffer_rsrc_t r = rsrc(w), rn = rsrc(wn ? wn : w);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      float4 a[RG];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) a[rg] = *reinterpret_cast<const float4*>(arow + rg * 16 * pitch + 16 * g);
      const float2 b0 = f[g & 7][0], b1 = f[g & 7][1], b2 = f[g & 7][2], b3 = f[g & 7][3];
      if (g + 8 < 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[g & 7][j] = ld2(r, lo, (16 * (g + 8) + j) * ld);
      } else if (wn) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[g & 7][j] = ld2(rn, lon, (16 * (g - 8) + j) * ldn);
      }
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        acc[rg][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rg].x, b0.x, acc[rg][0], 0, 0, 0);
        acc[rg][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rg].x, b0.y, acc[rg][1], 0, 0, 0);
      }
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        acc[rg][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rg].y, b1.x, acc[rg][0], 0, 0, 0);
        acc[rg][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rg].y, b1.y, acc[rg][1], 0, 0, 0);
      }
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        acc[rg][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rg].z, b2.x, acc[rg][0], 0, 0, 0);
        acc[rg][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rg].z, b2.y, acc[rg][1], 0, 0, 0);
      }
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        acc[rg][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rg].w, b3.x, acc[rg][0], 0, 0, 0);
        acc[rg][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rg].w, b3.y, acc[rg][1], 0, 0, 0);
      }
    }
  }
};

}  // namespace
#endif
