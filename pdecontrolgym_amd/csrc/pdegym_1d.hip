// pdegym_1d.hip -- gfx950 kernels for the 1D transport / reaction-diffusion environment steppers.
//
// Design (MI355X-first, not a translation of the reference's NumPy slicing):
//   * one 64-lane wavefront owns one environment instance; lane l holds EPL consecutive grid nodes
//     [l*EPL, (l+1)*EPL) of the row in VGPRs for ALL S sub-steps of an env-step (temporal fusion: the row
//     is read from HBM once and written once per env-step instead of once per sub-step);
//   * halo values cross lanes with one wave shift per side per sub-step; the non-local transport term
//     u(0,t)*beta(x) is a readfirstlane broadcast;
//   * L2-norm reductions for truncate()/TunedReward1D are butterfly shuffles inside the wavefront;
//   * arithmetic follows the reference's float32 operation order exactly (built with -ffp-contract=off,
//     true IEEE division) so fields are bit-identical to NumPy:
//       transport  environments1d/hyperbolic.py:143-155
//       parabolic  environments1d/parabolic.py:138-150
//       reward     rewards/tuned_reward_1d.py:25-40 (streaming form)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "pdegym.h"
#include "pdegym_common.h"

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ float from_left_lane(float v) { return __shfl_up(v, 1); }
__device__ __forceinline__ float from_right_lane(float v) { return __shfl_down(v, 1); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// value of row element j when lane l holds elements [l*EPL, l*EPL+EPL)
template <int EPL>
__device__ __forceinline__ float row_get(const float (&x)[EPL], int j) {
  const int src = j / EPL, e = j - src * EPL;
  float sel = x[0];
#pragma unroll
  for (int k = 1; k < EPL; ++k) sel = (e == k) ? x[k] : sel;
  return __shfl(sel, src);
}

template <int EPL>
__device__ __forceinline__ float row_sumsq(const float (&x)[EPL], int j0, int n) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) s += (j0 + e < n) ? x[e] * x[e] : 0.f;
  return wave_sum(s);
}

// (a+1)*m-m, base_env_1d.py:36-39
__device__ __forceinline__ float normalize_ctrl(float a, float m, int on) { return on ? (a + 1.0f) * m - m : a; }

template <int EPL, bool PARABOLIC, bool NEUMANN>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void step1d_kernel(pdegym_params1d P, pdegym_bufs1d Bf, int B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int inst = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (inst >= B) return;  // wave-uniform
  const int n = P.n;
  const int j0 = lane * EPL;
  float* urow = Bf.u + (size_t)inst * n;
  const float* brow = Bf.beta + (size_t)inst * Bf.beta_stride;

  float x[EPL], c[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int j = j0 + e;
    const bool ok = j < n;
    x[e] = ok ? urow[j] : 0.f;
    const float b = ok ? brow[j] : 0.f;
    c[e] = PARABOLIC ? P.dt * b : b;  // parabolic.py:144: (dt*beta) is formed first, then *u
  }

  int t = __builtin_amdgcn_readfirstlane(Bf.time_index[inst]);
  const int S = P.substeps > 0 ? P.substeps : 1;
  int nsub = P.nt - 1 - t;  // hyperbolic.py:140: while i < sample_rate and time_index < nt-1
  nsub = nsub < P.substeps ? nsub : P.substeps;
  nsub = nsub > 0 ? nsub : 0;

  const float a = Bf.action[inst];
  const float dx = P.dx, dt = P.dt, F = P.F;
  // control_update (hyperbolic.py:68,95). Transport/Neumann reads u[t][-2] of the NEW row, which is still
  // zero (hyperbolic.py:144), so its boundary value is constant over the sub-steps.
  const float cdx = a * dx;
  float bval = NEUMANN ? normalize_ctrl(cdx + 0.0f, P.max_control, P.normalize) : normalize_ctrl(a, P.max_control, P.normalize);

  double bsum = Bf.bsum[inst];
  const bool rec_all = P.nt <= PDEGYM_RING;
  int k = (t + PDEGYM_LOOKBACK) % S;
  float* ring = Bf.ring + (size_t)inst * PDEGYM_RING;
  float* hist = Bf.history ? Bf.history + (size_t)inst * P.nt * n : nullptr;

  for (int s = 0; s < nsub; ++s) {
    const float xl = from_left_lane(x[EPL - 1]);  // p[j0-1]
    const float xr = from_right_lane(x[0]);       // p[j0+EPL]
    float p0 = 0.f;
    if constexpr (!PARABOLIC) p0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x[0])));
    if constexpr (PARABOLIC && NEUMANN) {
      // parabolic.py:148-150: previous row's neighbour u[t-1][-2]
      bval = normalize_ctrl(cdx + row_get<EPL>(x, n - 2), P.max_control, P.normalize);
    }
    float y[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int j = j0 + e;
      const float p = x[e];
      const float pm = (e == 0) ? xl : x[e - 1];
      const float pp = (e == EPL - 1) ? xr : x[e + 1];
      float v;
      if constexpr (PARABOLIC) {
        // parabolic.py:143-144   u + F*(um - 2*u + up) + (dt*beta)*u
        const float t1 = 2.0f * p;
        const float t2 = pm - t1;
        const float t3 = t2 + pp;
        const float t4 = F * t3;
        const float t5 = p + t4;
        const float t7 = c[e] * p;
        v = t5 + t7;
        if (e == 0) v = (lane == 0) ? 0.0f : v;  // parabolic.py:146  u(0,t) = 0
      } else {
        // hyperbolic.py:146-155   u + dt*((up - u)/dx + u[0]*beta)
        const float d1 = pp - p;
        const float d2 = d1 / dx;
        const float r = p0 * c[e];
        const float d3 = d2 + r;
        const float d4 = dt * d3;
        v = p + d4;
        (void)pm;
      }
      v = (j >= n - 1) ? ((j == n - 1) ? bval : 0.0f) : v;  // controlled boundary node; padding stays 0
      y[e] = v;
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) x[e] = y[e];
    ++t;
    k = (k + 1 == S) ? 0 : k + 1;
    if constexpr (NEUMANN) bsum += (double)fabsf(bval);
    if (hist) {
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        if (j0 + e < n) hist[(size_t)t * n + j0 + e] = x[e];
    }
    // rows whose norm a later reward call looks back at (tuned_reward_1d.py:40): r+100 is a step end
    if (s + 1 < nsub && (rec_all || k == 0 || t + PDEGYM_LOOKBACK == P.nt - 1)) {
      const float nr = sqrtf(row_sumsq<EPL>(x, j0, n));
      if (lane == 0) ring[t & (PDEGYM_RING - 1)] = nr;
    }
  }
  if constexpr (!NEUMANN) bsum += (double)nsub * (double)fabsf(bval);

  // ---- epilogue: norms, flags, reward, observation ------------------------------------------------
  const float norm_now = sqrtf(row_sumsq<EPL>(x, j0, n));
  if (nsub > 0 && (rec_all || k == 0 || t + PDEGYM_LOOKBACK == P.nt - 1)) {
    if (lane == 0) ring[t & (PDEGYM_RING - 1)] = norm_now;
  }
  const bool terminate = t >= P.nt - 1;                                 // hyperbolic.py:171-180
  const bool truncate = P.limit_state && (norm_now >= P.max_state);     // hyperbolic.py:182-194
  // NormReward variants need wave-wide reductions: do them before the single-lane tail
  float nr_alt = norm_now;
  if (P.reward_kind == PDEGYM_REWARD_NORM_L1) {
    float s1 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s1 += (j0 + e < n) ? fabsf(x[e]) : 0.f;
    nr_alt = wave_sum(s1);
  } else if (P.reward_kind == PDEGYM_REWARD_NORM_LINF) {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) m = fmaxf(m, (j0 + e < n) ? fabsf(x[e]) : 0.f);
    nr_alt = wave_max(m);
  }
  // look-back row t-100 (Python negative index wraps into the zero-filled tail of the history).
  // Only lane 0 ever touches the ring, so its own earlier stores are visible to this load.
  float norm_back = 0.f;
  float reward = 0.f;
  if (lane == 0) {
    const int tb = t - PDEGYM_LOOKBACK;
    const int src = tb < 0 ? P.nt + tb : tb;
    const bool zero_row = (tb < 0 && src > t) || src < 0;
    if (!zero_row) norm_back = ring[src & (PDEGYM_RING - 1)];
    if (P.reward_kind == PDEGYM_REWARD_TUNED1D) {
      if (terminate && norm_now < 20.0f) {
        reward = (P.terminate_reward - ((float)bsum) / 1000.0f) - norm_now;  // tuned_reward_1d.py:36-37
      } else if (truncate) {
        reward = (float)((double)P.truncate_penalty * (double)(P.reward_nt - t));  // tuned_reward_1d.py:38-39
      } else {
        reward = norm_back - norm_now;  // tuned_reward_1d.py:40
      }
    } else if (P.reward_kind >= PDEGYM_REWARD_NORM_L1) {
      // documented intent of norm_reward.py:48-54 ("temporal" horizon)
      reward = terminate ? P.terminate_reward
                         : (truncate ? (float)((double)P.truncate_penalty * (double)(P.reward_nt - t)) : -nr_alt);
    }
  }

  // sensing_update (hyperbolic.py:72-116)
  const bool auto_reset = (Bf.reset_init != nullptr) && (terminate || truncate);  // wave-uniform
  auto emit_obs = [&](float* obs_base) {
    if (P.sensing == PDEGYM_SENSE_FULL) {
      float* orow = obs_base + (size_t)inst * n;
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        if (j0 + e < n) orow[j0 + e] = x[e];
    } else {
      float o;
      if (P.sensing == PDEGYM_SENSE_LAST) o = row_get<EPL>(x, n - 1);
      else if (P.sensing == PDEGYM_SENSE_LAST_DERIV) o = (row_get<EPL>(x, n - 1) - row_get<EPL>(x, n - 2)) / dx;
      else if (P.sensing == PDEGYM_SENSE_FIRST_DERIV) o = (row_get<EPL>(x, 1) - row_get<EPL>(x, 0)) / dx;
      else o = row_get<EPL>(x, 0);
      if (lane == 0) obs_base[inst] = o;
    }
  };
  if (lane == 0) {
    if (P.reward_kind != PDEGYM_REWARD_NONE) Bf.reward[inst] = reward;
    Bf.norm_now[inst] = norm_now;
    Bf.norm_back[inst] = norm_back;
    Bf.terminated[inst] = terminate ? 1 : 0;
    Bf.truncated[inst] = truncate ? 1 : 0;
  }
  if (!auto_reset) {
    if (nsub > 0) {
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        if (j0 + e < n) urow[j0 + e] = x[e];
    }
    emit_obs(Bf.obs);
    if (lane == 0) {
      Bf.time_index[inst] = t;
      Bf.bsum[inst] = bsum;
    }
  } else {
    // fused VecEnv auto-reset: keep the terminal observation, restart from the pool row (hyperbolic.py:214-227)
    if (Bf.final_obs) emit_obs(Bf.final_obs);
    const float* irow = Bf.reset_init + (size_t)inst * n;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      x[e] = (j0 + e < n) ? irow[j0 + e] : 0.f;
      if (j0 + e < n) urow[j0 + e] = x[e];
    }
    if (hist) {
      for (size_t q = lane; q < (size_t)P.nt * n; q += kWave) hist[q] = (q < (size_t)n) ? irow[q] : 0.f;
    }
    const float n0 = sqrtf(row_sumsq<EPL>(x, j0, n));
    const float last = row_get<EPL>(x, n - 1);
    emit_obs(Bf.obs);
    if (lane == 0) {
      Bf.time_index[inst] = 0;
      Bf.bsum[inst] = (double)fabsf(last);
      ring[0] = n0;
    }
  }
}

// ---- reset (state part of hyperbolic.py:214-227 / parabolic.py:208-221) -----------------------------
__global__ __launch_bounds__(kWave* kWavesPerBlock) void reset1d_kernel(pdegym_params1d P, pdegym_bufs1d Bf,
                                                                         const float* init, const uint8_t* mask, int B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int inst = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (inst >= B) return;
  if (mask && !mask[inst]) return;
  const int n = P.n;
  const float* src = init + (size_t)inst * n;
  float* urow = Bf.u + (size_t)inst * n;
  float ss = 0.f;
  for (int j = lane; j < n; j += kWave) {
    const float v = src[j];
    urow[j] = v;
    if (P.sensing == PDEGYM_SENSE_FULL) Bf.obs[(size_t)inst * n + j] = v;
    ss += v * v;
  }
  ss = wave_sum(ss);
  if (lane == 0) {
    const float last = src[n - 1];
    Bf.time_index[inst] = 0;
    Bf.bsum[inst] = (double)fabsf(last);
    Bf.ring[(size_t)inst * PDEGYM_RING] = sqrtf(ss);
    Bf.norm_now[inst] = sqrtf(ss);
    Bf.norm_back[inst] = 0.f;
    Bf.terminated[inst] = 0;
    Bf.truncated[inst] = 0;
    if (P.sensing != PDEGYM_SENSE_FULL) {
      float o;
      if (P.sensing == PDEGYM_SENSE_LAST) o = last;
      else if (P.sensing == PDEGYM_SENSE_LAST_DERIV) o = (last - src[n - 2]) / P.dx;
      else if (P.sensing == PDEGYM_SENSE_FIRST_DERIV) o = (src[1] - src[0]) / P.dx;
      else o = src[0];
      Bf.obs[inst] = o;
    }
  }
}

// history[b] = zeros((nt, n)); history[b, 0] = init[b]   (hyperbolic.py:214-217), spread over blockIdx.x chunks
__global__ __launch_bounds__(256) void reset_history_kernel(pdegym_params1d P, float* history, const float* init,
                                                            const uint8_t* mask, int B) {
  const int inst = blockIdx.y;
  if (inst >= B || (mask && !mask[inst])) return;
  const size_t total = (size_t)P.nt * P.n;
  float* h = history + (size_t)inst * total;
  const float* src = init + (size_t)inst * P.n;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x)
    h[q] = (q < (size_t)P.n) ? src[q] : 0.f;
}

__global__ __launch_bounds__(kWave* kWavesPerBlock) void rownorm2_kernel(const float* rows, float* out, int n, int B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int inst = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (inst >= B) return;
  const float* r = rows + (size_t)inst * n;
  float ss = 0.f;
  for (int j = lane; j < n; j += kWave) ss += r[j] * r[j];
  ss = wave_sum(ss);
  if (lane == 0) out[inst] = sqrtf(ss);
}

template <int EPL, bool PARABOLIC>
int launch_epl(const pdegym_params1d& P, const pdegym_bufs1d& Bf, int B, hipStream_t st) {
  const dim3 grid((B + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
  if (P.control_type == PDEGYM_CONTROL_NEUMANN)
    hipLaunchKernelGGL((step1d_kernel<EPL, PARABOLIC, true>), grid, block, 0, st, P, Bf, B);
  else
    hipLaunchKernelGGL((step1d_kernel<EPL, PARABOLIC, false>), grid, block, 0, st, P, Bf, B);
  return pdegym::check_launch("step1d");
}

template <bool PARABOLIC>
int launch_step(const pdegym_params1d* prm, const pdegym_bufs1d* buf, int B, void* stream) {
  if (!prm || !buf) return pdegym::fail(-1, "null params/bufs");
  if (B <= 0) return 0;
  const pdegym_params1d& P = *prm;
  if (P.n < 3 || P.n > PDEGYM_MAX_N1D) return pdegym::fail(-2, "n must be in [3, 1024] for the wave-per-instance 1D kernels");
  if (P.nt < 2) return pdegym::fail(-2, "nt must be >= 2");
  if (!buf->u || !buf->beta || !buf->action || !buf->time_index || !buf->bsum || !buf->ring || !buf->obs ||
      !buf->norm_now || !buf->norm_back || !buf->terminated || !buf->truncated)
    return pdegym::fail(-3, "null device buffer");
  if (P.reward_kind != PDEGYM_REWARD_NONE && !buf->reward) return pdegym::fail(-3, "null reward buffer");
  hipStream_t st = (hipStream_t)stream;
  const int epl = (P.n + kWave - 1) / kWave;
  switch (epl) {
    case 1: return launch_epl<1, PARABOLIC>(P, *buf, B, st);
    case 2: return launch_epl<2, PARABOLIC>(P, *buf, B, st);
    case 3: return launch_epl<3, PARABOLIC>(P, *buf, B, st);
    case 4: return launch_epl<4, PARABOLIC>(P, *buf, B, st);
    case 5: return launch_epl<5, PARABOLIC>(P, *buf, B, st);
    case 6: return launch_epl<6, PARABOLIC>(P, *buf, B, st);
    case 7: case 8: return launch_epl<8, PARABOLIC>(P, *buf, B, st);
    case 9: case 10: case 11: case 12: return launch_epl<12, PARABOLIC>(P, *buf, B, st);
    default: return launch_epl<16, PARABOLIC>(P, *buf, B, st);
  }
}

}  // namespace

extern "C" {

int pdegym_transport_step(const pdegym_params1d* prm, const pdegym_bufs1d* buf, int32_t B, void* stream) {
  return launch_step<false>(prm, buf, B, stream);
}

int pdegym_parabolic_step(const pdegym_params1d* prm, const pdegym_bufs1d* buf, int32_t B, void* stream) {
  return launch_step<true>(prm, buf, B, stream);
}

int pdegym_reset1d_masked(const pdegym_params1d* prm, const pdegym_bufs1d* buf, const float* init, const uint8_t* mask,
                          int32_t B, void* stream) {
  if (!prm || !buf || !init) return pdegym::fail(-1, "null params/bufs/init");
  if (B <= 0) return 0;
  if (prm->n < 3) return pdegym::fail(-2, "n must be >= 3");
  const dim3 grid((B + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
  hipLaunchKernelGGL(reset1d_kernel, grid, block, 0, (hipStream_t)stream, *prm, *buf, init, mask, B);
  if (buf->history) {
    const size_t total = (size_t)prm->nt * prm->n;
    const int gx = (int)((total + 255) / 256 > 512 ? 512 : (total + 255) / 256);
    hipLaunchKernelGGL(reset_history_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, *prm, buf->history, init,
                       mask, B);
  }
  return pdegym::check_launch("reset1d");
}

int pdegym_rownorm2_f32(const float* rows, float* out, int32_t n, int32_t B, void* stream) {
  if (!rows || !out) return pdegym::fail(-1, "null pointer");
  if (B <= 0) return 0;
  const dim3 grid((B + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
  hipLaunchKernelGGL(rownorm2_kernel, grid, block, 0, (hipStream_t)stream, rows, out, n, B);
  return pdegym::check_launch("rownorm2");
}

}  // extern "C"
