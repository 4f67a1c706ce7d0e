// pdegym_policy.h -- an MLP policy evaluated INSIDE a rollout kernel (pdegym_*_rollout with pdegym_rollout*.policy).
//
// A workgroup of 16 waves (16 instances, one CU's worth at four waves per SIMD) keeps ONE copy of the network in LDS for the
// whole launch; every wave evaluates it for its own instance at the start of each env-step, on its own clock -- no barrier
// after the weights are staged, so the waves of a SIMD keep drifting apart.  Lane j owns neuron j (layers of up to 64 units):
// per four inputs one 16-byte broadcast read of the layer input and one 16-byte read of the lane's weights, four fused
// multiply-adds; k ascending in ONE chain from zero, bias added last -- the summation order pdegym_mlp_forward documents,
// up to the MFMA's order inside a group of 16 inputs.
#ifndef PDEGYM_POLICY_H
#define PDEGYM_POLICY_H

#include <hip/hip_runtime.h>

#include "pdegym.h"
#include "pdegym_common.h"

namespace pdegym_policy {

constexpr int kWave = 64;
constexpr int kWaves = 16;             // waves (= instances) per workgroup
constexpr int kMaxWidth = 64;          // widest layer: one neuron per lane
constexpr int kMaxLdsBytes = 160 * 1024;

__host__ __device__ inline int xpad(int n) { return (n + 3) & ~3; }
// In LDS a neuron's weights are contiguous -- [neuron][group of four inputs][4], the group count rounded up to an odd number
// so that the 16-byte reads of 16 consecutive lanes (1 KB apart for 257 inputs) fall into distinct banks -- and every read of
// the reduction loop is base + immediate offset.
__host__ __device__ inline int groups(int in_dim) { return ((in_dim + 3) >> 2) | 1; }
// floats of LDS: per layer its weights and a bias row of 64, then per wave the padded observation row and two hidden rows of 64
__host__ __device__ inline int lds_floats(const pdegym_mlp& N, int n_in) {
  int f = 0;
  for (int l = 0; l < N.n_layers; ++l) f += groups(N.layer[l].in_dim) * 4 * N.layer[l].out_dim + kMaxWidth;
  return f + kWaves * (xpad(n_in) + 2 * kMaxWidth);
}

// Host-side check of a descriptor for the in-kernel evaluation: n_in inputs, n_out outputs.  Returns nullptr or the reason.
inline const char* check(const pdegym_mlp& N, int n_in, int n_out) {
  if (N.n_layers < 1 || N.n_layers > PDEGYM_MLP_MAX_LAYERS) return "policy: n_layers must be 1..4";
  for (int l = 0; l < N.n_layers; ++l) {
    const pdegym_mlp_layer& L = N.layer[l];
    if (!L.w) return "policy: null weight pointer";
    if (L.in_dim != (l ? N.layer[l - 1].out_dim : n_in)) return "policy: layer input size must match the observation row / the previous layer";
    if (L.out_dim < 1 || L.out_dim > kMaxWidth) return "policy inside the rollout kernel: layers of 1..64 units";
    if (L.act < PDEGYM_MLP_IDENTITY || L.act > PDEGYM_MLP_RELU) return "policy: unknown activation";
  }
  if (N.layer[N.n_layers - 1].out_dim != n_out) return "policy: the last layer must produce one command per actuator";
  if (N.clamp && !(N.lo <= N.hi)) return "policy: clamp bounds must satisfy lo <= hi";
  if (N.noise && N.noise_stride < n_out) return "policy: noise row stride shorter than the command";
  if (lds_floats(N, n_in) * (int)sizeof(float) > kMaxLdsBytes) return "policy inside the rollout kernel: the network does not fit into 160 KB of LDS";
  return nullptr;
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ float activate(float v, int act) {
  if (act == PDEGYM_MLP_TANH) return tanhf(v);
  if (act == PDEGYM_MLP_RELU) return v > 0.f ? v : 0.f;
  return v;
}

struct Staged {
  int woff[PDEGYM_MLP_MAX_LAYERS], boff[PDEGYM_MLP_MAX_LAYERS];
  int end;      // first float after the network: the per-wave rows start here
};

// Copy the network into LDS (all threads of the workgroup; ends with the launch's only barrier).
// ABI layout [group][neuron][4] -> [neuron][group][4]; the padding group reads as zero.
__device__ __forceinline__ Staged stage(const pdegym_mlp& N, float* smem) {
  Staged S;
  int off = 0;
#pragma unroll
  for (int l = 0; l < PDEGYM_MLP_MAX_LAYERS; ++l) {
    S.woff[l] = S.boff[l] = 0;
    if (l < N.n_layers) {
      const int H = N.layer[l].out_dim, ng = (N.layer[l].in_dim + 3) >> 2, ngo = groups(N.layer[l].in_dim);
      const int nw = ngo * 4 * H;
      S.woff[l] = off;
      S.boff[l] = off + nw;
      for (int i = threadIdx.x; i < nw; i += kWave * kWaves) {
        const int e = i & 3, g = (i >> 2) % ngo, j = (i >> 2) / ngo;
        smem[off + i] = g < ng ? N.layer[l].w[((size_t)g * H + j) * 4 + e] : 0.f;
      }
      for (int i = threadIdx.x; i < kMaxWidth; i += kWave * kWaves)
        smem[off + nw + i] = (N.layer[l].b && i < N.layer[l].out_dim) ? N.layer[l].b[i] : 0.f;
      off += nw + kMaxWidth;
    }
  }
  S.end = off;
  __syncthreads();
  return S;
}

// Evaluate the network for this wave's observation row xw (LDS, n_in floats zero-padded to a multiple of four, already
// visible to the wave); hw: two rows of 64 floats of this wave.  Returns the last layer's value of neuron `lane` (garbage
// in lanes beyond its width); noise and clamp are the caller's.
__device__ __forceinline__ float eval(const pdegym_mlp& N, const Staged& S, const float* smem, const float* xw, float* hw, int n_in,
                                      int lane) {
  const float* in = xw;
  int K = n_in;
  float out = 0.f;
#pragma unroll
  for (int l = 0; l < PDEGYM_MLP_MAX_LAYERS; ++l) {
    if (l < N.n_layers) {
      const int H = N.layer[l].out_dim, ng = (K + 3) >> 2;
      const int jj = lane < H ? lane : H - 1;
      const float4* W = reinterpret_cast<const float4*>(smem + S.woff[l]) + (size_t)jj * groups(K);
      const float4* X = reinterpret_cast<const float4*>(in);
      float acc = 0.f;
      int kb = 0;
      for (; kb + 4 <= ng; kb += 4) {          // eight reads in flight, then their sixteen fused multiply-adds
        float4 xv[4], wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { xv[i] = X[kb + i]; wv[i] = W[kb + i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc = __builtin_fmaf(xv[i].x, wv[i].x, acc);
          acc = __builtin_fmaf(xv[i].y, wv[i].y, acc);
          acc = __builtin_fmaf(xv[i].z, wv[i].z, acc);
          acc = __builtin_fmaf(xv[i].w, wv[i].w, acc);
        }
      }
      for (; kb < ng; ++kb) {
        const float4 xv = X[kb], wv = W[kb];
        acc = __builtin_fmaf(xv.x, wv.x, acc);
        acc = __builtin_fmaf(xv.y, wv.y, acc);
        acc = __builtin_fmaf(xv.z, wv.z, acc);
        acc = __builtin_fmaf(xv.w, wv.w, acc);
      }
      const float o = activate(acc + smem[S.boff[l] + jj], N.layer[l].act);
      if (l == N.n_layers - 1) {
        out = o;
      } else {
        float* hl = hw + (l & 1) * kMaxWidth;
        hl[lane] = lane < H ? o : 0.f;      // zero beyond the layer width: the next layer reads whole groups of four
        wave_lds_sync();
        in = hl;
        K = H;
      }
    }
  }
  return out;
}

__device__ __forceinline__ float lane_value(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

}  // namespace pdegym_policy
#endif
