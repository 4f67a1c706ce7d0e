// pdegym_mlp_tile.h -- the MFMA reduction of one 16 x 16 output tile of a dense layer (v_mfma_f32_16x16x4_f32, float32 in, float32
// accumulate), shared by pdegym_mlp_forward (pdegym_mlp.hip) and by the policies of more than 64 units evaluated inside the 1D
// rollout kernels (pdegym_policy.h): the same operand order and the same accumulator chain, hence the same bits.
//
// Operands: lane l holds (row or neuron li = l % 16, k-slot lg = l / 16).  Per 16 reduction indices a lane supplies ONE float4 of the
// layer input (row li, inputs 16 kb + 4 lg .. + 3, from LDS) and ONE float4 of weights per tile (blocked layout
// wq[k / 4][neuron][k % 4]: a wave's load is four contiguous 256-byte pieces) for four MFMAs; weight loads run a software pipeline
// of kStage k-blocks.  Accumulation order is that of the MFMA (k in groups of four, ascending).
#ifndef PDEGYM_MLP_TILE_H
#define PDEGYM_MLP_TILE_H

#include <hip/hip_runtime.h>

#include "pdegym.h"

namespace pdegym_mlp_tile {

typedef float v4f __attribute__((ext_vector_type(4)));

// LDS row strides are (a multiple of 64) + 4 floats: 16-byte aligned rows, and 16 rows x one float4 hit 64 distinct banks.
// The staging area is sized by the launch for the layer widths at hand (dynamic LDS): a 257-64-64-1 policy takes 25 KB per
// workgroup instead of the 50 KB of the largest shapes, so four workgroups share a CU when the batch is large.
__host__ __device__ constexpr int lds_stride(int width) { return ((width + 63) / 64) * 64 + 4; }
constexpr int kStage = 4;                   // k-blocks (of 16 inputs) per software-pipeline stage

__device__ __forceinline__ float activate(float v, int act) {
  if (act == PDEGYM_MLP_TANH) return tanhf(v);
  if (act == PDEGYM_MLP_RELU) return v > 0.f ? v : 0.f;
  return v;
}

// weights of k-blocks kb0 .. kb0 + kStage - 1 for this lane: group g = 4 kb + l / 16 of the blocked matrix, neuron n (already
// clamped to the layer width); groups past the end of the (zero-padded) matrix read as zero
template <int NT, int ST = kStage>
__device__ __forceinline__ void load_w(v4f (&w)[ST][NT], const v4f* __restrict__ wq, int H, int ngroups, int kb0, int lg,
                                       const int (&col)[NT]) {
#pragma unroll
  for (int s = 0; s < ST; ++s) {
    const int g = 4 * (kb0 + s) + lg;
    const bool ok = g < ngroups;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const v4f v = wq[(ok ? g : ngroups - 1) * H + col[t]];
      w[s][t] = ok ? v : (v4f){0.f, 0.f, 0.f, 0.f};
    }
  }
}

template <int NT, int ST = kStage>
__device__ __forceinline__ void mma_stage(v4f (&acc)[NT], const v4f (&w)[ST][NT], const float* in_row, int kb0, int nblk, int lg) {
#pragma unroll
  for (int s = 0; s < ST; ++s) {
    if (kb0 + s < nblk) {     // wave-uniform
      const v4f a = *reinterpret_cast<const v4f*>(in_row + 16 * (kb0 + s) + 4 * lg);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[s][t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[s][t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[s][t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[s][t].w, acc[t], 0, 0, 0);
      }
    }
  }
}

// acc[t] += in[16 rows, nblk k-blocks] x W^T for this wave's NT tiles; in_row = this lane's input row in LDS (index 0 = first
// input of the chunk), gofs = k-block offset of the chunk within the weight matrix.  wa: the first stage's weights, loaded
// by the caller BEFORE it waited for the inputs (the barrier after staging / after the previous layer's epilogue).
template <int NT, int ST = kStage>
__device__ __forceinline__ void reduce_blocks(v4f (&acc)[NT], v4f (&wa)[ST][NT], const v4f* __restrict__ wq, int H, int ngroups,
                                              int gofs, int nblk, const float* in_row, int lg, const int (&col)[NT]) {
  v4f wb[ST][NT];
  int kb = 0;
  while (true) {
    const bool more_b = kb + ST < nblk;
    if (more_b) load_w<NT, ST>(wb, wq, H, ngroups, gofs + kb + ST, lg, col);
    mma_stage<NT, ST>(acc, wa, in_row, kb, nblk, lg);
    kb += ST;
    if (!more_b) break;
    const bool more_a = kb + ST < nblk;
    if (more_a) load_w<NT, ST>(wa, wq, H, ngroups, gofs + kb + ST, lg, col);
    mma_stage<NT, ST>(acc, wb, in_row, kb, nblk, lg);
    kb += ST;
    if (!more_a) break;
  }
}

}  // namespace pdegym_mlp_tile
#endif
