// pdegym_1d.hip -- gfx950 kernels for the 1D transport / reaction-diffusion environment steppers.
//
// Design (MI355X-first, not a translation of the reference's NumPy slicing):
//   * one 64-lane wavefront owns one environment instance.  The row minus its fixed left node (parabolic
//     u(0,t)=0) is spread over "slots": lane l keeps EPL consecutive slots in VGPRs for ALL S sub-steps of an
//     env-step (temporal fusion: the row is read from HBM once and written once per env-step instead of once
//     per sub-step).  nx=256 (257 nodes) -> 256 slots = 4 per lane with every lane busy; nx=512 -> 8 per lane;
//   * halo values cross lanes with ONE DPP wave shift per side per sub-step (v_mov_b32 wave_shr/wave_shl, no
//     LDS traffic); the lane that has no neighbour receives the boundary value through the DPP "old" operand;
//     the non-local transport term u(0,t)*beta(x) is a readfirstlane broadcast;
//   * the controlled boundary node and the padding slots carry zero stencil coefficients, so the inner loop has
//     no selects: y = (p + F_e*lap) + c_e*p leaves them unchanged.  0*inf would poison them, so a step whose
//     result norm is not finite is recomputed from the untouched HBM row with explicit selects (EXACT mode):
//     results are bit-identical to the reference in every case, finite or not;
//   * L2-norm reductions for truncate()/TunedReward1D are butterfly shuffles inside the wavefront;
//   * arithmetic follows the reference's float32 operation order exactly (built with -ffp-contract=off,
//     IEEE division and sqrt) so fields are bit-identical to NumPy:
//       transport  environments1d/hyperbolic.py:143-155
//       parabolic  environments1d/parabolic.py:138-150
//       reward     rewards/tuned_reward_1d.py:25-40 (streaming form, see docs/HISTORY.md section 3.3)
#include "pdegym_1d_body.h"

namespace {

// FULL (round 5): the row fills the wave exactly (n - J0 == 64 EPL; the BASELINE shapes nx = 256 and nx = 512 do) -- the slot
// masks of prologue / epilogue fold away (pdegym_1d_body.h: run_substeps).  The float32 Dirichlet fast path only.
// HFAST (round 6): the fast loop + one trajectory-row store per sub-step -- what a single environment with record_history (the
// reference's env.u) runs: 0.066 us per sub-step of one instance against 0.228 for the select form.
template <int EPL, bool PARABOLIC, bool NEUMANN, bool HIST, bool BURGERS = false, bool M64 = false, bool FULL = false, bool HFAST = false>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void step1d_kernel(pdegym_params1d P, pdegym_bufs1d Bf, int B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int inst = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (inst >= B) return;  // wave-uniform
  step1d_body<EPL, PARABOLIC, NEUMANN, HIST, BURGERS, M64, false, false, FULL, HFAST>(P, Bf, B, inst, lane);
}

// Rows of more than 2048 nodes: the register-resident layout would not fit, so the row ping-pongs between two LDS copies
// owned by the wave (node j lives in lane j % 64; wave-level ordering only, no workgroup barrier).  This is the plain
// select form of the reference arithmetic -- same expressions and order as run_substeps<..., FAST = false> -- with the
// same bookkeeping (reward ring, look-back, history, sensing, fused auto-reset).  n <= PDEGYM_MAX_N1D_WIDE.
// ================================================================================================
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// M64 = the reference's mixed-precision arithmetic for a float64 beta and/or a float64 / Python-float control input
// (pdegym_params1d.beta_f64 / action_kind): the parity mode of the docs quickstart (beta = np.ones(nx)).  Rows of ANY length
// take this kernel in that mode; with M64 = false it is the float32 kernel for rows beyond the register-resident limit.
template <bool PARABOLIC, bool BURGERS, bool M64 = false>
__global__ __launch_bounds__(kWave) void step1d_wide_kernel(pdegym_params1d P, pdegym_bufs1d Bf, int B) {
  extern __shared__ float wl[];
  constexpr int J0 = PARABOLIC ? 1 : 0;
  const int lane = threadIdx.x;
  const int inst = blockIdx.x;
  if (inst >= B) return;
  const int n = P.n;
  float* cur = wl;
  float* nxt = wl + n;
  const float* urow_in = (Bf.state_in ? Bf.state_in : Bf.u) + (size_t)inst * n;      // include/pdegym.h: state_in
  float* urow = Bf.state_in ? nullptr : Bf.u + (size_t)inst * n;
  const bool beta64 = M64 && P.beta_f64;
  const float* brow = static_cast<const float*>(Bf.beta) + (beta64 ? 0 : (size_t)inst * Bf.beta_stride);
  const double* brow64 = static_cast<const double*>(Bf.beta) + (beta64 ? (size_t)inst * Bf.beta_stride : 0);
  float* ring = Bf.ring + (size_t)inst * PDEGYM_RING;
  float* hist = Bf.history ? Bf.history + (size_t)inst * P.nt * n : nullptr;
  const bool neumann = P.control_type == PDEGYM_CONTROL_NEUMANN;
  for (int j = lane; j < n; j += kWave) cur[j] = urow_in[j];
  const int t_in = Bf.time_index[inst];
  const int S = P.substeps > 0 ? P.substeps : 1;
  int nsub = P.nt - 1 - t_in;
  nsub = nsub < P.substeps ? nsub : P.substeps;
  nsub = nsub > 0 ? nsub : 0;
  const bool act64 = M64 && P.action_kind != PDEGYM_ACTION_F32;
  const float a = act64 ? 0.f : static_cast<const float*>(Bf.action)[inst];
  const double a64 = act64 ? static_cast<const double*>(Bf.action)[inst] : 0.0;
  int t = t_in, k = (t_in + PDEGYM_LOOKBACK) % S;
  double bsum = Bf.bsum[inst];
  const int t_end = t_in + nsub;
  const int tb = t_end - PDEGYM_LOOKBACK;
  const int src_row = tb < 0 ? P.nt + tb : tb;
  const bool zero_row = (tb < 0 && src_row > t_end) || src_row < 0;
  const bool from_ring = !zero_row && src_row <= t_in;
  const float norm_back_pre = from_ring ? ring[src_row & (PDEGYM_RING - 1)] : 0.f;
  const int back_row = (!zero_row && !from_ring) ? src_row : -1;
  float back_norm = 0.f;
  const bool rec_all = P.nt <= PDEGYM_RING;
  const float dx = P.dx, dt = P.dt, F = P.F;
  // transport/Neumann reads u[t][-2] of the NEW (still zero) row, hyperbolic.py:144
  float bval = boundary_value<M64>(P, a, a64, 0.0f, neumann);
  auto row_norm = [&](const float* row) {
    float ss = 0.f;
    for (int j = lane; j < n; j += kWave) ss += row[j] * row[j];
    return sqrtf(wave_sum(ss));
  };
  // the reward's own norm of a row (NormReward kinds), for the "t-horizon" reward (see step1d_body)
  auto kind_norm_row = [&](const float* row) {
    if (P.reward_kind == PDEGYM_REWARD_NORM_L1) {
      float s1 = 0.f;
      for (int j = lane; j < n; j += kWave) s1 += fabsf(row[j]);
      return wave_sum(s1);
    }
    if (P.reward_kind == PDEGYM_REWARD_NORM_LINF) {
      float m = 0.f;
      for (int j = lane; j < n; j += kWave) m = mag_max(m, fabsf(row[j]));
      return wave_mag_max(m);
    }
    return row_norm(row);
  };
  const int thor_k = (P.reward_horizon == PDEGYM_HORIZON_T && P.reward_kind >= PDEGYM_REWARD_NORM_L1) ? P.reward_t_horizon : 0;
  wave_lds_sync();
  if (thor_k > 0) {
    const float nk0 = kind_norm_row(cur);
    if (lane == 0) ring[t_in & (PDEGYM_RING - 1)] = nk0;
  }
  for (int s = 0; s < nsub; ++s) {
    if (PARABOLIC && neumann) bval = boundary_value<M64>(P, a, a64, cur[n - 2], true);   // parabolic.py:148-150
    const float p0 = cur[0];
    for (int j = lane; j < n; j += kWave) {
      const float p = cur[j];
      float v;
      if (j == n - 1) {
        v = bval;                                                       // controlled boundary node
      } else if (PARABOLIC && j == 0) {
        v = 0.0f;                                                       // parabolic.py:146
      } else if constexpr (PARABOLIC) {
        const float pm = cur[j - 1], pp = cur[j + 1];
        const float t1 = 2.0f * p;                                      // parabolic.py:143-144
        const float t2 = pm - t1;
        const float t3 = t2 + pp;
        const float t4 = F * t3;
        const float t5 = p + t4;
        if (beta64) {                                                   // float64 beta: dt*beta and its product with u are double
          const double t7 = (P.dt64 * brow64[j]) * (double)p;
          v = (float)((double)t5 + t7);
        } else {
          const float t7 = (dt * brow[j]) * p;
          v = t5 + t7;
        }
      } else {
        const float pp = cur[j + 1];
        const float d1 = pp - p;                                        // hyperbolic.py:146-155
        const float d2 = d1 / dx;
        if (beta64) {                                                   // float64 beta: u[0]*beta, the sum, dt*(...) and u + ... are double
          const double r = (double)p0 * brow64[j];
          const double d3 = (double)(BURGERS ? p * d2 : d2) + r;
          const double d4 = P.dt64 * d3;
          v = (float)((double)p + d4);
        } else {
          const float r = p0 * brow[j];
          const float d3 = (BURGERS ? p * d2 : d2) + r;
          const float d4 = dt * d3;
          v = p + d4;
        }
      }
      nxt[j] = v;
    }
    {
      float* sw = cur;
      cur = nxt;
      nxt = sw;
    }
    ++t;
    k = (k + 1 == S) ? 0 : k + 1;
    if (neumann) bsum += (double)fabsf(bval);
    wave_lds_sync();
    if (hist) {
      float* hrow = hist + (size_t)t * n;
      for (int j = lane; j < n; j += kWave) hrow[j] = cur[j];
    }
    if (s + 1 < nsub && (rec_all || k == 0 || t + PDEGYM_LOOKBACK == P.nt - 1)) {   // tuned_reward_1d.py:40 look-back rows
      const float nr = row_norm(cur);
      if (lane == 0) ring[t & (PDEGYM_RING - 1)] = nr;
      if (t == back_row) back_norm = nr;
    }
    if (thor_k > 0 && s + 1 < nsub && nsub - (s + 1) < thor_k) {
      const float nk = kind_norm_row(cur);
      if (lane == 0) ring[t & (PDEGYM_RING - 1)] = nk;
    }
  }
  if (!neumann) bsum += (double)nsub * (double)fabsf(bval);
  const float norm_now = row_norm(cur);
  if (nsub > 0 && (rec_all || k == 0 || t + PDEGYM_LOOKBACK == P.nt - 1)) {
    if (lane == 0) ring[t & (PDEGYM_RING - 1)] = norm_now;
  }
  const bool terminate = t >= P.nt - 1;
  const bool truncate = P.limit_state && (norm_now >= P.max_state);
  float nr_alt = norm_now;
  // "differential" horizon (norm_reward.py:55-59): after the last swap nxt still holds the row before the last sub-step
  const bool nr_diff = P.reward_horizon == PDEGYM_HORIZON_DIFFERENTIAL && P.reward_kind >= PDEGYM_REWARD_NORM_L1 && nsub > 0;
  if (nr_diff) {
    float acc = 0.f;
    for (int j = lane; j < n; j += kWave) {
      const float d = fabsf(cur[j] - nxt[j]);
      if (P.reward_kind == PDEGYM_REWARD_NORM_L1) acc += d;
      else if (P.reward_kind == PDEGYM_REWARD_NORM_L2) acc += d * d;
      else acc = mag_max(acc, d);
    }
    nr_alt = P.reward_kind == PDEGYM_REWARD_NORM_LINF ? wave_mag_max(acc)
                                                      : (P.reward_kind == PDEGYM_REWARD_NORM_L2 ? sqrtf(wave_sum(acc)) : wave_sum(acc));
  } else if (P.reward_kind == PDEGYM_REWARD_NORM_L1) {
    float s1 = 0.f;
    for (int j = lane; j < n; j += kWave) s1 += fabsf(cur[j]);
    nr_alt = wave_sum(s1);
  } else if (P.reward_kind == PDEGYM_REWARD_NORM_LINF) {
    float m = 0.f;
    for (int j = lane; j < n; j += kWave) m = mag_max(m, fabsf(cur[j]));
    nr_alt = wave_mag_max(m);
  }
  const float norm_back = from_ring ? norm_back_pre : ((back_row >= 0) ? ((back_row == t) ? norm_now : back_norm) : 0.f);
  float thor_mean = 0.f;
  if (thor_k > 0 && lane == 0) {
    if (nsub > 0) ring[t & (PDEGYM_RING - 1)] = nr_alt;
    const int kk = thor_k < t + 1 ? thor_k : t + 1;
    float acc = nr_alt;
    for (int i = 1; i < kk; ++i) acc += ring[(t - i) & (PDEGYM_RING - 1)];
    thor_mean = acc / (float)kk;
  }
  float reward = 0.f;
  if (P.reward_kind == PDEGYM_REWARD_TUNED1D) {
    if (terminate && norm_now < 20.0f) reward = (P.terminate_reward - ((float)bsum) / 1000.0f) - norm_now;
    else if (truncate) reward = (float)((double)P.truncate_penalty * (double)(P.reward_nt - t));
    else reward = norm_back - norm_now;
  } else if (P.reward_kind >= PDEGYM_REWARD_NORM_L1) {
    reward = terminate ? P.terminate_reward
                       : (truncate ? (float)((double)P.truncate_penalty * (double)(P.reward_nt - t))
                                   : (nr_diff ? nr_alt : (thor_k > 0 ? -thor_mean : -nr_alt)));
  }
  const bool auto_reset = (Bf.reset_init != nullptr) && (terminate || truncate);
  auto emit_obs = [&](float* obs_base, const float* row) {
    if (P.sensing == PDEGYM_SENSE_FULL) {
      float* orow = obs_base + (size_t)inst * n;
      for (int j = lane; j < n; j += kWave) orow[j] = row[j];
    } else if (lane == 0) {
      float o;
      if (P.sensing == PDEGYM_SENSE_LAST) o = row[n - 1];
      else if (P.sensing == PDEGYM_SENSE_LAST_DERIV) o = (row[n - 1] - row[n - 2]) / P.dx;
      else if (P.sensing == PDEGYM_SENSE_FIRST_DERIV) o = (row[1] - row[0]) / P.dx;
      else o = row[0];
      obs_base[inst] = o;
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
    if (nsub > 0 && urow)
      for (int j = lane; j < n; j += kWave) urow[j] = cur[j];
    emit_obs(Bf.obs, cur);
    if (lane == 0) {
      Bf.time_index[inst] = t;
      Bf.bsum[inst] = bsum;
    }
  } else {
    if (Bf.final_obs) emit_obs(Bf.final_obs, cur);
    const int prow = pool_row(Bf, inst, B);
    const float* irow = Bf.reset_init + (size_t)prow * n;
    if (Bf.reset_beta && Bf.beta_stride != 0) {      // the reference redraws beta at every reset (hyperbolic.py:208)
      if (beta64) {
        double* bdst = const_cast<double*>(brow64);
        const double* bsrc = static_cast<const double*>(Bf.reset_beta) + (size_t)prow * n;
        for (int j = lane; j < n; j += kWave) bdst[j] = bsrc[j];
      } else {
        float* bdst = const_cast<float*>(brow);
        const float* bsrc = static_cast<const float*>(Bf.reset_beta) + (size_t)prow * n;
        for (int j = lane; j < n; j += kWave) bdst[j] = bsrc[j];
      }
    }
    if (Bf.reset_count && lane == 0) Bf.reset_count[inst] += 1;
    wave_lds_sync();
    for (int j = lane; j < n; j += kWave) {
      const float v = irow[j];
      nxt[j] = v;
      if (urow) urow[j] = v;
    }
    if (hist)
      for (size_t q = lane; q < (size_t)P.nt * n; q += kWave) hist[q] = (q < (size_t)n) ? irow[q] : 0.f;
    wave_lds_sync();
    const float n0 = row_norm(nxt);
    emit_obs(Bf.obs, nxt);
    if (lane == 0) {
      Bf.time_index[inst] = 0;
      Bf.bsum[inst] = (double)fabsf(nxt[n - 1]);
      ring[0] = n0;
    }
  }
  (void)J0;
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
  float* urow = Bf.u ? Bf.u + (size_t)inst * n : nullptr;      // NULL: the state lives in the observation buffers
  float ss = 0.f;
  for (int j = lane; j < n; j += kWave) {
    const float v = src[j];
    if (urow) urow[j] = v;
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

__global__ void selftest_quotient_kernel(const float* a, float dx, double rdx, unsigned int* mismatches, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float x = a[i];
    const float q_fast = (float)((double)x * rdx);
    const float q_ieee = x / dx;
    const bool same = (__builtin_bit_cast(unsigned int, q_fast) == __builtin_bit_cast(unsigned int, q_ieee)) ||
                      (q_fast != q_fast && q_ieee != q_ieee);
    if (!same) atomicAdd(mismatches, 1u);
  }
}

template <int EPL, bool PARABOLIC, bool BURGERS = false>
int launch_epl(const pdegym_params1d& P, const pdegym_bufs1d& Bf, int B, hipStream_t st) {
  const dim3 grid((B + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
  // the select-form (HIST) instantiation also serves NormReward's "differential" horizon: it keeps the row before the last sub-step
  // ... and its "t-horizon" (norms of the last rows of the sub-step loop)
  const bool differential = P.reward_horizon != PDEGYM_HORIZON_TEMPORAL && P.reward_kind >= PDEGYM_REWARD_NORM_L1;
  const bool neu = P.control_type == PDEGYM_CONTROL_NEUMANN, hist = Bf.history != nullptr || differential;
  // (a dummy dynamic-LDS request that capped resident workgroups per CU was A/B-tested and removed: profiles/r02_ab_lds_balance.txt)
  constexpr int lds = 0;
  if (neu && hist)
    hipLaunchKernelGGL((step1d_kernel<EPL, PARABOLIC, true, true, BURGERS>), grid, block, lds, st, P, Bf, B);
  else if (neu)
    hipLaunchKernelGGL((step1d_kernel<EPL, PARABOLIC, true, false, BURGERS>), grid, block, lds, st, P, Bf, B);
  else if (hist) {
    // a trajectory buffer and a reward the fast loop serves, rows of up to 512 slots, the reference's flux: fast arithmetic + row stores
    constexpr bool kHasHFast = !BURGERS && EPL <= 8;
    if (kHasHFast && Bf.history != nullptr && !differential)
      hipLaunchKernelGGL((step1d_kernel<EPL, PARABOLIC, false, true, BURGERS, false, false, kHasHFast>), grid, block, lds, st, P, Bf, B);
    else
      hipLaunchKernelGGL((step1d_kernel<EPL, PARABOLIC, false, true, BURGERS>), grid, block, lds, st, P, Bf, B);
  } else {
    constexpr bool kHasFull = EPL == 1 || EPL == 2 || EPL == 4 || EPL == 8;      // rows of 64 / 128 / 256 / 512 slots
    if (kHasFull && P.n - (PARABOLIC ? 1 : 0) == kWave * EPL)
      hipLaunchKernelGGL((step1d_kernel<EPL, PARABOLIC, false, false, BURGERS, false, kHasFull>), grid, block, lds, st, P, Bf, B);
    else
      hipLaunchKernelGGL((step1d_kernel<EPL, PARABOLIC, false, false, BURGERS>), grid, block, lds, st, P, Bf, B);
  }
  return pdegym::check_launch("step1d");
}

template <bool PARABOLIC, bool BURGERS = false>
int launch_step(const pdegym_params1d* prm, const pdegym_bufs1d* buf, int B, void* stream) {
  if (!prm || !buf) return pdegym::fail(-1, "null params/bufs");
  if (B <= 0) return 0;
  const pdegym_params1d& P = *prm;
  if (P.n < 3 || P.n > PDEGYM_MAX_N1D_WIDE) return pdegym::fail(-2, "n must be in [3, 8192] for the 1D kernels");
  if (P.nt < 2) return pdegym::fail(-2, "nt must be >= 2");
  if ((!buf->u && !buf->state_in) || !buf->beta || !buf->action || !buf->time_index || !buf->bsum || !buf->ring || !buf->obs ||
      !buf->norm_now || !buf->norm_back || !buf->terminated || !buf->truncated)
    return pdegym::fail(-3, "null device buffer");
  if (buf->state_in) {
    if (P.sensing != PDEGYM_SENSE_FULL) return pdegym::fail(-2, "state_in needs full-state sensing (the observation is the row)");
    if (buf->history) return pdegym::fail(-2, "state_in cannot be combined with a history buffer");
    if (buf->state_in == buf->obs) return pdegym::fail(-3, "state_in must not alias obs (double-buffer the observations)");
  }
  if (P.reward_kind != PDEGYM_REWARD_NONE && !buf->reward) return pdegym::fail(-3, "null reward buffer");
  hipStream_t st = (hipStream_t)stream;
  if (P.action_kind < PDEGYM_ACTION_F32 || P.action_kind > PDEGYM_ACTION_WEAK) return pdegym::fail(-2, "bad action_kind");
  if (P.reward_horizon < PDEGYM_HORIZON_TEMPORAL || P.reward_horizon > PDEGYM_HORIZON_T) return pdegym::fail(-2, "bad reward_horizon");
  if (P.reward_horizon == PDEGYM_HORIZON_T && P.reward_kind >= PDEGYM_REWARD_NORM_L1 &&
      (P.reward_t_horizon < 1 || P.reward_t_horizon > PDEGYM_RING))
    return pdegym::fail(-2, "reward_t_horizon must be in [1, 128] (the ring of row norms)");
  if (P.beta_f64 || P.action_kind != PDEGYM_ACTION_F32) {   // the reference's float64-operand arithmetic (parity mode)
    const int slots = P.n - (PARABOLIC ? 1 : 0), epl64 = (slots + kWave - 1) / kWave;
    if (!BURGERS && epl64 <= 8) {      // rows of up to 512 nodes stay in registers in this mode too
      const dim3 grid((B + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
      const bool neu = P.control_type == PDEGYM_CONTROL_NEUMANN,
                 hist = buf->history != nullptr || (P.reward_horizon == PDEGYM_HORIZON_DIFFERENTIAL && P.reward_kind >= PDEGYM_REWARD_NORM_L1);
      auto go = [&](auto epl_tag) {
        constexpr int E = decltype(epl_tag)::value;
        if (neu && hist) hipLaunchKernelGGL((step1d_kernel<E, PARABOLIC, true, true, false, true>), grid, block, 0, st, P, *buf, B);
        else if (neu) hipLaunchKernelGGL((step1d_kernel<E, PARABOLIC, true, false, false, true>), grid, block, 0, st, P, *buf, B);
        else if (hist) hipLaunchKernelGGL((step1d_kernel<E, PARABOLIC, false, true, false, true>), grid, block, 0, st, P, *buf, B);
        else hipLaunchKernelGGL((step1d_kernel<E, PARABOLIC, false, false, false, true>), grid, block, 0, st, P, *buf, B);
      };
      switch (epl64) {
        case 1: go(std::integral_constant<int, 1>{}); break;
        case 2: go(std::integral_constant<int, 2>{}); break;
        case 3: case 4: go(std::integral_constant<int, 4>{}); break;
        default: go(std::integral_constant<int, 8>{}); break;
      }
      return pdegym::check_launch("step1d_m64");
    }
    hipLaunchKernelGGL((step1d_wide_kernel<PARABOLIC, BURGERS, true>), dim3(B), dim3(kWave), 2 * (size_t)P.n * sizeof(float), st, P, *buf, B);
    return pdegym::check_launch("step1d_m64_wide");
  }
  if (P.n > PDEGYM_MAX_N1D) {    // LDS-resident rows (one wave per instance, two row copies)
    hipLaunchKernelGGL((step1d_wide_kernel<PARABOLIC, BURGERS, false>), dim3(B), dim3(kWave), 2 * (size_t)P.n * sizeof(float), st, P, *buf, B);
    return pdegym::check_launch("step1d_wide");
  }
  const int nslots = P.n - (PARABOLIC ? 1 : 0);  // parabolic node 0 lives in a wave-uniform register
  const int epl = (nslots + kWave - 1) / kWave;
  switch (epl) {
    case 1: return launch_epl<1, PARABOLIC, BURGERS>(P, *buf, B, st);
    case 2: return launch_epl<2, PARABOLIC, BURGERS>(P, *buf, B, st);
    case 3: return launch_epl<3, PARABOLIC, BURGERS>(P, *buf, B, st);
    case 4: return launch_epl<4, PARABOLIC, BURGERS>(P, *buf, B, st);
    case 5: return launch_epl<5, PARABOLIC, BURGERS>(P, *buf, B, st);
    case 6: return launch_epl<6, PARABOLIC, BURGERS>(P, *buf, B, st);
    case 7: case 8: return launch_epl<8, PARABOLIC, BURGERS>(P, *buf, B, st);
    case 9: case 10: case 11: case 12: return launch_epl<12, PARABOLIC, BURGERS>(P, *buf, B, st);
    case 13: case 14: case 15: case 16: return launch_epl<16, PARABOLIC, BURGERS>(P, *buf, B, st);
    case 17: case 18: case 19: case 20: case 21: case 22: case 23: case 24: return launch_epl<24, PARABOLIC, BURGERS>(P, *buf, B, st);
    default: return launch_epl<32, PARABOLIC, BURGERS>(P, *buf, B, st);
  }
}

}  // namespace

extern "C" {

int pdegym_transport_step(const pdegym_params1d* prm, const pdegym_bufs1d* buf, int32_t B, void* stream) {
  // flux = PDEGYM_FLUX_BURGERS is an extension that the reference does not have
  if (prm && prm->flux == PDEGYM_FLUX_BURGERS) return launch_step<false, true>(prm, buf, B, stream);
  return launch_step<false, false>(prm, buf, B, stream);
}

int pdegym_parabolic_step(const pdegym_params1d* prm, const pdegym_bufs1d* buf, int32_t B, void* stream) {
  return launch_step<true>(prm, buf, B, stream);
}

int pdegym_reset1d_masked(const pdegym_params1d* prm, const pdegym_bufs1d* buf, const float* init, const uint8_t* mask,
                          int32_t B, void* stream) {
  if (!prm || !buf || !init) return pdegym::fail(-1, "null params/bufs/init");
  if (B <= 0) return 0;
  if (prm->n < 3) return pdegym::fail(-2, "n must be >= 3");
  if (!buf->u && prm->sensing != PDEGYM_SENSE_FULL) return pdegym::fail(-3, "u may only be NULL with full-state sensing (rows go to obs)");
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

int pdegym_selftest_quotient(const float* a, float dx, double rdx, uint32_t* mismatches, int32_t n, void* stream) {
  if (!a || !mismatches) return pdegym::fail(-1, "null pointer");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(selftest_quotient_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, a, dx, rdx, mismatches, n);
  return pdegym::check_launch("selftest_quotient");
}

int pdegym_rownorm2_f32(const float* rows, float* out, int32_t n, int32_t B, void* stream) {
  if (!rows || !out) return pdegym::fail(-1, "null pointer");
  if (B <= 0) return 0;
  const dim3 grid((B + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
  hipLaunchKernelGGL(rownorm2_kernel, grid, block, 0, (hipStream_t)stream, rows, out, n, B);
  return pdegym::check_launch("rownorm2");
}

}  // extern "C"
