// pdegym_policy.h -- an MLP policy evaluated INSIDE a rollout kernel (pdegym_*_rollout with pdegym_rollout*.policy).
//
// A workgroup of 16 waves (16 instances, one CU's worth at four waves per SIMD) keeps ONE copy of the network in LDS for the
// whole launch; every wave evaluates it for its own instance at the start of each env-step, on its own clock -- no barrier
// after the weights are staged, so the waves of a SIMD keep drifting apart.  Lane j owns neuron j (layers of up to 64 units):
// per four inputs one 16-byte broadcast read of the layer input and one 16-byte read of the lane's weights, four fused
// multiply-adds; k ascending in ONE chain from zero, bias added last -- the summation order pdegym_mlp_forward documents,
// up to the MFMA's order inside a group of 16 inputs.
//
// Networks with a layer of MORE than 64 units (up to 256: the 256-256 actors of SB3's SAC / a PPO with net_arch [256, 256]) do not
// fit into LDS (0.5 MB of weights) and have no neuron-per-lane form: they are evaluated COOPERATIVELY (eval_wide) -- the 16 waves
// put their observation rows into one LDS matrix, wave w computes the 16 x 16 output tile w of every layer with the MFMA reduction
// of pdegym_mlp_forward (pdegym_mlp_tile.h: weights streamed from L2 in the ABI's blocked layout, same operand order, same bits),
// one workgroup barrier per layer.
#ifndef PDEGYM_POLICY_H
#define PDEGYM_POLICY_H

#include <hip/hip_runtime.h>

#include "pdegym.h"
#include "pdegym_common.h"
#include "pdegym_mlp_tile.h"

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
constexpr int kWideStage = 1;          // k-blocks per weight-pipeline stage of the cooperative evaluation (registers: 3 stages live)
constexpr int kWideMax = 256;          // widest layer of the cooperative (MFMA) evaluation: 16 tiles of 16 neurons, one per wave
__host__ __device__ inline bool is_wide(const pdegym_mlp& N) {
  bool w = false;
  for (int l = 0; l < N.n_layers && l < PDEGYM_MLP_MAX_LAYERS; ++l) w = w || N.layer[l].out_dim > kMaxWidth;
  return w;
}
// cooperative evaluation: 16 observation rows (zero-padded to a multiple of 16 inputs), two ping-pong matrices of 16 hidden rows,
// 16 commands; row strides as in pdegym_mlp_forward
__host__ __device__ inline int wide_ldx(int n_in) { return pdegym_mlp_tile::lds_stride((n_in + 15) & ~15); }
constexpr int kWideLdh = pdegym_mlp_tile::lds_stride(kWideMax);
constexpr int kWideOut = 2;            // outputs kept per row (neurons 0 and 1 of the last layer: the traffic engine's two commands)
__host__ __device__ inline int wide_lds_floats(int n_in) { return kWaves * (wide_ldx(n_in) + 2 * kWideLdh) + kWideOut * kWaves; }
// floats of LDS: per layer its weights and a bias row of 64, then per wave the padded observation row and two hidden rows of 64
__host__ __device__ inline int lds_floats(const pdegym_mlp& N, int n_in) {
  if (is_wide(N)) return wide_lds_floats(n_in);
  int f = 0;
  for (int l = 0; l < N.n_layers; ++l) f += groups(N.layer[l].in_dim) * 4 * N.layer[l].out_dim + kMaxWidth;
  return f + kWaves * (xpad(n_in) + 2 * kMaxWidth);
}

// Host-side check of a descriptor for the in-kernel evaluation: n_in inputs, n_out outputs.  Returns nullptr or the reason.
// allow_wide: the caller has the cooperative evaluation for layers of 65..256 units (the 1D rollout kernels).
inline const char* check(const pdegym_mlp& N, int n_in, int n_out, bool allow_wide = false) {
  if (N.n_layers < 1 || N.n_layers > PDEGYM_MLP_MAX_LAYERS) return "policy: n_layers must be 1..4";
  for (int l = 0; l < N.n_layers; ++l) {
    const pdegym_mlp_layer& L = N.layer[l];
    if (!L.w) return "policy: null weight pointer";
    if (L.in_dim != (l ? N.layer[l - 1].out_dim : n_in)) return "policy: layer input size must match the observation row / the previous layer";
    if (L.out_dim < 1 || L.out_dim > (allow_wide ? kWideMax : kMaxWidth))
      return allow_wide ? "policy inside the rollout kernel: layers of 1..256 units" : "policy inside the rollout kernel: layers of 1..64 units";
    if (L.act < PDEGYM_MLP_IDENTITY || L.act > PDEGYM_MLP_RELU) return "policy: unknown activation";
  }
  if (N.layer[N.n_layers - 1].out_dim != n_out) return "policy: the last layer must produce one command per actuator";
  if (N.clamp && !(N.lo <= N.hi)) return "policy: clamp bounds must satisfy lo <= hi";
  if (N.noise && N.noise_stride < n_out) return "policy: noise row stride shorter than the command";
  if (lds_floats(N, n_in) * (int)sizeof(float) > kMaxLdsBytes)
    return is_wide(N) ? "policy inside the rollout kernel: 16 observation rows do not fit into 160 KB of LDS"
                      : "policy inside the rollout kernel: the network does not fit into 160 KB of LDS";
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

// ---- cooperative evaluation (layers of up to 256 units) ------------------------------------------------------------------------
struct Wide {
  float* X;       // [16][ldx] observation rows (row = wave), zero beyond n_in up to a multiple of 16
  float* H[2];    // [16][kWideLdh] hidden activations, ping-pong
  float* act;     // [16][kWideOut] neurons 0 .. kWideOut - 1 of the last layer per row
  int ldx;
};

__device__ __forceinline__ Wide wide_setup(const pdegym_mlp& N, float* smem, int n_in, int wave, int lane) {
  Wide W;
  W.ldx = wide_ldx(n_in);
  W.X = smem;
  W.H[0] = smem + kWaves * W.ldx;
  W.H[1] = W.H[0] + kWaves * kWideLdh;
  W.act = W.H[1] + kWaves * kWideLdh;
  // the row's padding up to a multiple of 16 inputs is written once (every wave, active or not, owns row `wave`)
  float* xr = W.X + wave * W.ldx;
  for (int j = lane; j < ((n_in + 15) & ~15); j += kWave) xr[j] = 0.f;
  return W;
}

// ALL 16 waves of the workgroup call this in every env-step (waves without an instance too: their row stays zero), having written
// their observation row into W.X[wave]; returns the value of neuron 0 of the last layer for row `wave` (wide_out: the others).
// Barriers: one ahead of the first layer, one per layer.
__device__ __forceinline__ float eval_wide(const pdegym_mlp& N, const Wide& W, int n_in, int wave, int lane) {
  namespace mt = pdegym_mlp_tile;
  const int li = lane & 15, lg = lane >> 4;
  auto first_stage = [&](int l, mt::v4f (&w)[kWideStage][1]) {
    const pdegym_mlp_layer& L = N.layer[l];
    if (16 * wave < L.out_dim) {
      const int n = 16 * wave + li;
      const int col[1] = {n < L.out_dim ? n : L.out_dim - 1};
      mt::load_w<1, kWideStage>(w, reinterpret_cast<const mt::v4f*>(L.w), L.out_dim, (L.in_dim + 3) >> 2, 0, lg, col);
    }
  };
  mt::v4f wfirst[kWideStage][1];
  first_stage(0, wfirst);      // requested before the barrier that publishes the observation rows
  __syncthreads();
#pragma unroll 1
  for (int l = 0; l < N.n_layers; ++l) {      // not unrolled: one copy of the reduction, the descriptor read with scalar loads
    const pdegym_mlp_layer L = N.layer[l];
    const int K = L.in_dim, H = L.out_dim;
    const bool last = l == N.n_layers - 1, any_tile = 16 * wave < H;
    const int n = 16 * wave + li;
    const int col[1] = {n < H ? n : H - 1};
    const float bias = (L.b && n < H) ? L.b[n] : 0.f;
    mt::v4f acc[1] = {(mt::v4f){0.f, 0.f, 0.f, 0.f}};
    mt::v4f wnext[kWideStage][1];
    if (!last) first_stage(l + 1, wnext);     // travels while this layer is reduced
    const float* in_row = l == 0 ? W.X + li * W.ldx : W.H[(l + 1) & 1] + li * kWideLdh;
    if (any_tile)
      mt::reduce_blocks<1, kWideStage>(acc, wfirst, reinterpret_cast<const mt::v4f*>(L.w), H, (K + 3) >> 2, 0, (K + 15) >> 4, in_row, lg, col);
    if (!last) {
#pragma unroll
      for (int s2 = 0; s2 < kWideStage; ++s2) wfirst[s2][0] = wnext[s2][0];
    }
    // bias, activation: D[row = 4 lg + v][neuron = li] of tile `wave`; hidden layers zero-padded to a multiple of 16 columns
    const int Hpad = (H + 15) & ~15;
    if (n < Hpad) {
      const float av[4] = {acc[0].x, acc[0].y, acc[0].z, acc[0].w};
      float* hout = W.H[l & 1];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 4 * lg + v;
        const float o = n < H ? mt::activate(av[v] + bias, L.act) : 0.f;
        if (last) {
          if (n < kWideOut) W.act[r * kWideOut + n] = o;
        } else {
          hout[r * kWideLdh + n] = o;
        }
      }
    }
    __syncthreads();
  }
  return W.act[wave * kWideOut];
}

__device__ __forceinline__ float wide_out(const Wide& W, int wave, int k) { return W.act[wave * kWideOut + k]; }

__device__ __forceinline__ float lane_value(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

}  // namespace pdegym_policy
#endif
