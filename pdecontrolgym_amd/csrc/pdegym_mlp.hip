// pdegym_mlp.hip -- fused forward pass of a small multi-layer perceptron policy (see include/pdegym.h: pdegym_mlp_forward).
//
// The caller on the other side of env.step(): the reference trains SB3 "MlpPolicy" networks (two hidden layers of 64 tanh
// units by default, examples/transportPDE/transport1Dppo.py:88-90) and evaluates them once per env-step.  Inside an
// on-device rollout (pde_control_gym.DeviceRollout) that evaluation was 3 GEMM + 3 tanh + clamp + copy launches -- 29 us
// per step next to a 20 us environment step at C2 -- so the whole forward pass is ONE launch here.  The arithmetic is
// tiny (85 MFLOP at B = 4096); what decides the time is operand delivery and the number of dependent memory round trips.
//
// Mapping (the one GEMM-shaped piece of the project -> MFMA): a workgroup of four waves (8 / 16 for layers wider than 64 /
// 128 units) owns 16 observation rows for all layers; wave w computes the 16 x 16 output tiles w, w + WV, ... of a layer with v_mfma_f32_16x16x4_f32 (float32 in,
// float32 accumulate).  Per 16 reduction indices a lane supplies ONE float4 of the layer input (row l % 16, inputs
// 4 (l / 16) .. + 3, read from LDS: the observation rows are staged there once, hidden activations are written there by
// the previous layer) and ONE float4 of weights per tile (blocked layout wq[k / 4][neuron][k % 4]: a wave's load is four
// contiguous 256-byte pieces) for four MFMAs -- 1/8 float per FMA, against 1.25 for a VALU formulation whose wave-uniform
// inputs have to be broadcast (the first versions: LDS broadcast reads and the texture addresser bound them at 12-17 us).
// Weight loads run a software pipeline of 64 reduction indices (a wave is alone on its SIMD).  Accumulation order is that
// of the MFMA (k in groups of four, ascending), bias added last: deterministic, but not the summation order of a BLAS GEMM
// -- results match torch within float32 rounding (tests: rtol 2e-5, atol 4e-6 on O(1) values).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pdegym.h"
#include "pdegym_common.h"
#include "pdegym_mlp_tile.h"

namespace {

using namespace pdegym_mlp_tile;     // v4f, lds_stride, kStage, activate, load_w, mma_stage, reduce_blocks

constexpr int kRows = 16;                   // observation rows per workgroup (the M of the MFMA tile)
constexpr int kMaxWidth = PDEGYM_MLP_MAX_WIDTH;
constexpr int kXChunk = 512;                // observation entries per row staged in LDS at a time

// NT = 16-neuron tiles per wave, WV = waves per workgroup: one tile per wave throughout -- 4 waves for layers of up to 64
// units, 8 up to 128, 16 up to 256.  The reduction is bound by the latency of the weight stream, and more waves keep more
// loads in flight per CU than fewer waves with more tiles each (257-256-256-1 at B = 4096: 27.7 us with 4 waves x 4 tiles,
// 21.4 with 8 x 2, 19.2 with 16 x 1).
template <int NT, int WV, typename TX, typename TY>
__global__ __launch_bounds__(64 * WV) void mlp_forward_kernel(pdegym_mlp N, const TX* __restrict__ x, long long x_stride,
                                                          TY* __restrict__ y, long long y_stride, int B, int ldx) {
  // LDS: activations ping-pong between hb0 and hb1; the staged observation chunk shares its space with hb1 (first written
  // by the second layer, when the observations are no longer needed)
  extern __shared__ __attribute__((aligned(16))) float mlp_smem[];
  constexpr int kLdh = lds_stride(16 * WV * NT);
  constexpr int kRowsPerWave = kRows / WV;        // observation rows a wave stages
  const int kLdx = ldx;                              // >= kLdh (the launch takes the larger of the two)
  float* const hb0 = mlp_smem;
  float* const xs = mlp_smem + kRows * kLdh;
  auto hbuf = [&](int i) -> float* { return i ? xs : hb0; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;      // MFMA operand lane = (row or neuron li, k-slot lg)
  const int row0 = blockIdx.x * kRows;

  // weights of the first pipeline stage of the coming layer: requested before the barrier that publishes its inputs
  v4f wfirst[kStage][NT];
  auto tile_cols = [&](int H, int (&col)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * (wave + WV * t) + li;
      col[t] = n < H ? n : H - 1;
    }
  };
  {
    int col0[NT];
    tile_cols(N.layer[0].out_dim, col0);
    if (16 * wave < N.layer[0].out_dim)
      load_w<NT>(wfirst, reinterpret_cast<const v4f*>(N.layer[0].w), N.layer[0].out_dim, (N.layer[0].in_dim + 3) >> 2, 0, lg, col0);
  }

  // biases of every layer up front (their pointers and the loads are off the critical path of the layers); the layer loop
  // is unrolled over the at most four layers so that the descriptors are read with constant offsets at kernel start
  float bias_all[PDEGYM_MLP_MAX_LAYERS][NT];
#pragma unroll
  for (int l = 0; l < PDEGYM_MLP_MAX_LAYERS; ++l) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * (wave + WV * t) + li;
      bias_all[l][t] = (l < N.n_layers && N.layer[l].b && n < N.layer[l].out_dim) ? N.layer[l].b[n] : 0.f;
    }
  }
#pragma unroll
  for (int l = 0; l < PDEGYM_MLP_MAX_LAYERS; ++l) {
    if (l >= N.n_layers) break;
    const pdegym_mlp_layer L = N.layer[l];
    const int K = L.in_dim, H = L.out_dim;
    const int ngroups = (K + 3) >> 2;
    const v4f* __restrict__ wq = reinterpret_cast<const v4f*>(L.w);
    int col[NT];
    tile_cols(H, col);
    v4f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    const bool any_tile = 16 * wave < H;      // this wave has at least one tile of the layer
    const bool last = l == N.n_layers - 1;
    // the NEXT layer's first weights are requested now: they travel while this layer is reduced
    v4f wnext[kStage][NT];
    if (!last) {
      const pdegym_mlp_layer Ln = N.layer[l + 1];
      int coln[NT];
      tile_cols(Ln.out_dim, coln);
      if (16 * wave < Ln.out_dim)
        load_w<NT>(wnext, reinterpret_cast<const v4f*>(Ln.w), Ln.out_dim, (Ln.in_dim + 3) >> 2, 0, lg, coln);
    }
    if (l == 0) {
      for (int c0 = 0; c0 < K; c0 += kXChunk) {
        const int clen = (K - c0) < kXChunk ? (K - c0) : kXChunk;
        const int cpad = (clen + 15) & ~15;
        // stage the chunk: wave w copies rows kRowsPerWave w .. + kRowsPerWave - 1, a wave-wide load = 64 consecutive entries of one row; all
        // the loads of a thread are issued before the first LDS store.  Rows past the batch and entries past K read as zero.
        {
          float v[kRowsPerWave][kXChunk / 64];
#pragma unroll
          for (int rr = 0; rr < kRowsPerWave; ++rr) {
            const int r = kRowsPerWave * wave + rr;
            const bool row_ok = row0 + r < B;
            const TX* src = x + (long long)(row_ok ? row0 + r : 0) * x_stride + c0;
#pragma unroll
            for (int i = 0; i < kXChunk / 64; ++i) {
              const int c = lane + 64 * i;
              v[rr][i] = (row_ok && c < clen) ? (float)src[c] : 0.f;
            }
          }
          if (c0 > 0 && any_tile) load_w<NT>(wfirst, wq, H, ngroups, c0 / 16, lg, col);
#pragma unroll
          for (int rr = 0; rr < kRowsPerWave; ++rr)
#pragma unroll
            for (int i = 0; i < kXChunk / 64; ++i) {
              const int c = lane + 64 * i;
              if (c < cpad) xs[(kRowsPerWave * wave + rr) * kLdx + c] = v[rr][i];
            }
        }
        __syncthreads();
        if (any_tile) reduce_blocks<NT>(acc, wfirst, wq, H, ngroups, c0 / 16, cpad / 16, xs + li * kLdx, lg, col);
        __syncthreads();
      }
    } else {
      const float* hin = hbuf((l + 1) & 1);       // written by layer l - 1, zero beyond its width up to a multiple of 16
      if (any_tile) reduce_blocks<NT>(acc, wfirst, wq, H, ngroups, 0, (K + 15) >> 4, hin + li * kLdh, lg, col);
    }
    if (!last) {
#pragma unroll
      for (int s2 = 0; s2 < kStage; ++s2)
#pragma unroll
        for (int t = 0; t < NT; ++t) wfirst[s2][t] = wnext[s2][t];
    }
    // bias, activation; D[i = 4 lg + v][j = li] of tile t.  Hidden layers go to LDS, zero-padded to a multiple of 16 columns.
    float* hout = hbuf(l & 1);
    const int Hpad = (H + 15) & ~15;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * (wave + WV * t) + li;
      if (n < Hpad) {
        const float av[4] = {acc[t].x, acc[t].y, acc[t].z, acc[t].w};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 4 * lg + v;
          float o = n < H ? activate(av[v] + bias_all[l][t], L.act) : 0.f;
          if (last) {
            if (N.noise && n < H && row0 + r < B) o += N.noise[(long long)(row0 + r) * N.noise_stride + n];
            if (N.clamp) o = fminf(fmaxf(o, N.lo), N.hi);
            if (n < H && row0 + r < B) y[(long long)(row0 + r) * y_stride + n] = (TY)o;
          } else {
            hout[r * kLdh + n] = o;
          }
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

template <int NT, int WV, typename TX, typename TY>
static void launch_mlp_nt(const pdegym_mlp* net, const TX* xp, long long xs, TY* yp, long long ys, int B, hipStream_t st) {
  const int kin = net->layer[0].in_dim < kXChunk ? net->layer[0].in_dim : kXChunk;
  const int ldh = lds_stride(16 * WV * NT);
  const int ldx = lds_stride(kin) > ldh ? lds_stride(kin) : ldh;
  const size_t lds_bytes = (size_t)kRows * (ldh + ldx) * sizeof(float);         // <= 50 KB: below the 64 KB that need no opt-in
  hipLaunchKernelGGL((mlp_forward_kernel<NT, WV, TX, TY>), dim3((B + kRows - 1) / kRows), dim3(64 * WV), lds_bytes, st, *net, xp, xs, yp, ys,
                     B, ldx);
}

template <typename TX, typename TY>
static void launch_mlp(const pdegym_mlp* net, const void* x, long long xs, void* y, long long ys, int B, int width, hipStream_t st) {
  const TX* xp = static_cast<const TX*>(x);
  TY* yp = static_cast<TY*>(y);
  if (width <= 64) launch_mlp_nt<1, 4, TX, TY>(net, xp, xs, yp, ys, B, st);
  else if (width <= 128) launch_mlp_nt<1, 8, TX, TY>(net, xp, xs, yp, ys, B, st);
  else launch_mlp_nt<1, 16, TX, TY>(net, xp, xs, yp, ys, B, st);
}

extern "C" int pdegym_mlp_forward(const pdegym_mlp* net, const void* x, int64_t x_stride, void* y, int64_t y_stride, int32_t B,
                                  void* stream) {
  if (!net || !x || !y) return pdegym::fail(-3, "null pointer");
  if (B < 0) return pdegym::fail(-2, "B must be >= 0");
  if (net->n_layers < 1 || net->n_layers > PDEGYM_MLP_MAX_LAYERS) return pdegym::fail(-2, "n_layers must be 1..4");
  for (int l = 0; l < net->n_layers; ++l) {
    const pdegym_mlp_layer& L = net->layer[l];
    if (!L.w) return pdegym::fail(-3, "null weight pointer");
    if (L.in_dim < 1 || L.out_dim < 1) return pdegym::fail(-2, "layer dimensions must be >= 1");
    if (L.out_dim > PDEGYM_MLP_MAX_WIDTH) return pdegym::fail(-2, "layer wider than PDEGYM_MLP_MAX_WIDTH");
    if (l == 0 && L.in_dim > PDEGYM_MLP_MAX_INPUT) return pdegym::fail(-2, "observation wider than PDEGYM_MLP_MAX_INPUT");
    if (l > 0 && L.in_dim != net->layer[l - 1].out_dim) return pdegym::fail(-2, "layer input size does not match the previous layer");
    if (L.act < PDEGYM_MLP_IDENTITY || L.act > PDEGYM_MLP_RELU) return pdegym::fail(-2, "unknown activation");
  }
  if (x_stride < net->layer[0].in_dim || y_stride < net->layer[net->n_layers - 1].out_dim)
    return pdegym::fail(-2, "row stride shorter than the row");
  if (net->clamp && !(net->lo <= net->hi)) return pdegym::fail(-2, "clamp bounds must satisfy lo <= hi");
  if (net->noise && net->noise_stride < net->layer[net->n_layers - 1].out_dim) return pdegym::fail(-2, "noise row stride shorter than the row");
  if (B == 0) return 0;
  int width = 0;
  for (int l = 0; l < net->n_layers; ++l) width = net->layer[l].out_dim > width ? net->layer[l].out_dim : width;
  hipStream_t st = (hipStream_t)stream;
  const long long xs = x_stride, ys = y_stride;
  if (net->x_f64 && net->y_f64) launch_mlp<double, double>(net, x, xs, y, ys, B, width, st);
  else if (net->x_f64) launch_mlp<double, float>(net, x, xs, y, ys, B, width, st);
  else if (net->y_f64) launch_mlp<float, double>(net, x, xs, y, ys, B, width, st);
  else launch_mlp<float, float>(net, x, xs, y, ys, B, width, st);
  return pdegym::check_launch("mlp_forward");
}
