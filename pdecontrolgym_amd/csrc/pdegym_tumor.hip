// pdegym_tumor.hip -- gfx950 kernel for the 1D brain-tumour radiotherapy environment (float64).
//
// One 64-lane wavefront owns one patient (for one day, or for a whole run of days: pdegym_tumor_advance).  The live density row (nx = 201 in the shipped example) is staged once in
// LDS so that every lane reads its two neighbours from there; lane l updates nodes l, l+64, l+128, ... (coalesced
// global loads and stores).  One launch is one simulated day: finite-difference update with the radiotherapy kill
// term, MRI radii (wave max-reductions of the rightmost node above a threshold), the Growth/Therapy/Post-Therapy
// stage machine with its day counters, terminate/truncate and the reward -- i.e. everything
// environments1d/brain_tumor_env.py:123-352 does per step() call.
// Operation order follows brain_tumor_env.py:221-245 exactly (-ffp-contract=off): the density rows are bit-identical
// to NumPy.  exp/pow are libm calls in the reference: the caller may pass the kill fraction precomputed (bit parity);
// the in-kernel exp / pow are within 1 ulp of libm's.
#include <hip/hip_runtime.h>

#include "pdegym.h"
#include "pdegym_common.h"

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kMaxNx = 4096;

// index of the highest lane set in a chunk's ballot (chunks are visited left to right, so a later hit wins)
__device__ __forceinline__ int rightmost(unsigned long long ballot, int chunk_base, int so_far) {
  return ballot ? chunk_base + 63 - __builtin_clzll(ballot) : so_far;
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// brain_tumor_env.py:221-245, one interior node.
__device__ __forceinline__ double fd_node(const pdegym_params_tumor& P, double ul, double uc, double ur, double R, bool rad) {
  double lap = (ur - 2.0 * uc) + ul;
  if (P.dx2 != 1.0) lap = lap / P.dx2;                                 // x / 1.0 == x: skip the f64 division when dx = 1
  const double diffusion = P.D * lap;
  const double logistic = 1.0 - (uc / P.k);
  const double proliferation = (P.rho * uc) * logistic;
  double s = diffusion + proliferation;
  if (rad) s = s - (R * uc) * logistic;
  return uc + P.dt * s;
}

__device__ __forceinline__ double clip0k(double x, double k) { return fmin(fmax(x, 0.0), k); }

// mode: PDEGYM_TUMOR_RUN_*; every participating instance simulates at most max_days days.  The row ping-pongs between two
// LDS copies owned by the wave (no workgroup barrier: waves of a block run different numbers of days), the scalars of the
// stage machine live in registers (all lanes carry the same values), global memory sees the row and the scalars once
// at the end -- plus one history row per day if the caller asked for the trajectory.
template <bool SINGLE_DAY>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void tumor_step_kernel(pdegym_params_tumor P, pdegym_bufs_tumor Bf, int mode,
                                                                           int max_days_arg, int B) {
  const int max_days = SINGLE_DAY ? 1 : max_days_arg;
  extern __shared__ double lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int w = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  // readfirstlane: the patient index is wave-uniform, which lets the compiler keep the whole stage machine (day counters,
  // doses, flags) in scalar registers and scalar loads
  const int inst = __builtin_amdgcn_readfirstlane(blockIdx.x * wpb + w);
  if (inst >= B) return;                                               // wave-uniform; no workgroup barriers below
  if (Bf.active && !Bf.active[inst]) return;
  const int nx = P.nx;
  double* cur = lds + (size_t)(2 * w) * nx;
  double* nxt = cur + nx;
  int t = Bf.time_index[inst];
  int stage = Bf.stage[inst];
  const bool one_day = mode == PDEGYM_TUMOR_RUN_ONE_DAY;
  // which instances take part (TherapyWrapper.reset :409-428 / .step :437-446 / .benchmark :488-503)
  const bool takes_part = one_day || mode == PDEGYM_TUMOR_RUN_TO_END || (mode == PDEGYM_TUMOR_RUN_GROWTH && stage == PDEGYM_TUMOR_GROWTH) ||
                          (mode == PDEGYM_TUMOR_RUN_POST && stage == PDEGYM_TUMOR_POST);
  const bool live0 = t < P.nt - 1;                                     // :136
  if (!takes_part || !live0) {
    if (one_day && lane == 0) {
      Bf.reward[inst] = 0.0;
      Bf.terminated[inst] = 0;
      Bf.truncated[inst] = 0;
    }
    return;
  }
  double* g = Bf.u + (size_t)inst * nx;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  // ---- stage the live row; T2 radius of the row BEFORE the first update (:255 reads time_index-1)
  // "rightmost node at or above a threshold" (:106-121) = highest set bit of a wave ballot, chunk of 64 nodes by chunk:
  // scalar results without a shuffle reduction
  int t2_idx = -1;
  if (nx <= 4 * kWave) {       // rows of up to 256 nodes (the shipped 201): the four loads of a lane are issued together
    double v4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = c * kWave + lane;
      v4[c] = i < nx ? g[i] : -1.0;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = c * kWave + lane;
      if (i < nx) cur[i] = v4[c];
      if (c * kWave < nx) t2_idx = rightmost(__ballot(i < nx && v4[c] >= P.thr_t2), c * kWave, t2_idx);
    }
  } else {
    for (int i0 = 0; i0 < nx; i0 += kWave) {
      const int i = i0 + lane;
      double v = -1.0;
      if (i < nx) {
        v = g[i];
        cur[i] = v;
      }
      t2_idx = rightmost(__ballot(i < nx && v >= P.thr_t2), i0, t2_idx);
    }
  }
  double remaining = Bf.remaining[inst];
  int32_t* days = Bf.days + (size_t)inst * PDEGYM_TUMOR_DAYS;
  int growth = days[PDEGYM_TUMOR_DAY_GROWTH], therapyDays = days[PDEGYM_TUMOR_DAY_THERAPY];
  int post = days[PDEGYM_TUMOR_DAY_POST], sim = days[PDEGYM_TUMOR_DAY_SIM], death = days[PDEGYM_TUMOR_DAY_DEATH];
  const double tb = Bf.t_benchmark ? Bf.t_benchmark[inst] : nan;
  const bool has_tb = tb == tb;
  const double control = one_day ? Bf.control[inst] : 0.0;             // the wrapper's loops call env.step(0)
  double* hist = Bf.history ? Bf.history + (size_t)inst * P.nt * nx : nullptr;
  double* t1log = Bf.t1_log ? Bf.t1_log + (size_t)inst * P.nt : nullptr;
  double reward = 0.0, T1 = nan, T2 = nan, treat_r = 0.0, applied = 0.0;
  bool term = false, lethal = false;
  for (int day = 0; day < max_days; ++day) {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the row written by the previous day is visible
    t += 1;
    const int stage0 = stage;
    const bool therapy = stage0 == PDEGYM_TUMOR_THERAPY;
    applied = 0.0;
    treat_r = 0.0;
    double kill = 0.0;
    if (therapy) {                                                     // :158-168, :247-263
      const double want = control * P.total_dosage;
      applied = remaining < want ? remaining : want;                   // Python min(want, remaining)
      remaining = remaining - applied;
      treat_r = t2_idx < 0 ? 0.0 : (double)t2_idx * P.dx + P.margin;
      if (one_day && Bf.kill) {
        kill = Bf.kill[inst];
      } else {
        const double bed = applied + ((applied * applied) / P.alpha_beta_ratio);
        kill = 1.0 - exp(-P.alpha * bed);
      }
    }
    // ---- finite-difference update, Neumann ends, clip to [0, k]; T1/T2 radii of the new row
    int t1_idx = -1, t2n_idx = -1;
    double* hrow = hist ? hist + (size_t)t * nx : nullptr;
    // four chunks of 64 nodes per trip, fully unrolled: their LDS reads and f64 division chains are independent, so one
    // wave overlaps them (a lone wave issues an instruction only every ~6.5 cycles; the day loop is latency-bound)
    for (int ib = 0; ib < nx; ib += 4 * kWave) {
      double vv[4];
      bool inn[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = ib + q * kWave + lane;
        inn[q] = i < nx;
        vv[q] = -1.0;
        if (inn[q]) {
          const int c = i == 0 ? 1 : (i == nx - 1 ? nx - 2 : i);       // :241-242 copy the neighbour's new value
          const bool rad = therapy && Bf.xscale[c] <= treat_r;         // outside: BED = 0 -> R = 1 - exp(-0) = 0
          vv[q] = clip0k(fd_node(P, cur[c - 1], cur[c], cur[c + 1], kill, rad), P.k);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i0 = ib + q * kWave, i = i0 + lane;
        if (inn[q]) {
          if constexpr (SINGLE_DAY) g[i] = vv[q];   // the only day of this launch: straight to global memory
          else nxt[i] = vv[q];
          if (hrow) hrow[i] = vv[q];
        }
        t1_idx = rightmost(__ballot(inn[q] && vv[q] >= P.thr_t1), i0, t1_idx);
        t2n_idx = rightmost(__ballot(inn[q] && vv[q] >= P.thr_t2), i0, t2n_idx);
      }
    }
    t2_idx = t2n_idx;
    {
      double* sw = cur;
      cur = nxt;
      nxt = sw;
    }
    // ---- stage machine :146-177, terminate/truncate :280-352, reward (every lane carries the same scalars)
    T1 = t1_idx < 0 ? nan : (double)t1_idx * P.dx;
    T2 = t2n_idx < 0 ? nan : (double)t2n_idx * P.dx;
    if (t1log && lane == 0) t1log[t] = t1_idx < 0 ? nan : T1 / P.dx;   // _log_radii :271-273
    if (stage0 == PDEGYM_TUMOR_GROWTH) {
      growth = t;
      if (t1_idx >= 0 && T1 >= P.detect_radius) stage = PDEGYM_TUMOR_THERAPY;
    } else if (therapy && remaining < P.dose_end) {
      therapyDays = t - growth;
      stage = PDEGYM_TUMOR_POST;
    }
    term = t >= P.nt - 1;
    lethal = t1_idx >= 0 && T1 >= P.death_radius;
    if (term || (lethal && death < 0)) {
      if (stage == PDEGYM_TUMOR_THERAPY) {
        therapyDays = t - growth;
        sim = growth + therapyDays;
      } else if (stage == PDEGYM_TUMOR_POST) {
        post = t - therapyDays - growth;
        sim = growth + therapyDays + post;
      }
    }
    if (lethal && death < 0) death = t;
    reward = 0.0;
    if (therapy) {
      if (!has_tb) reward = 0.0;
      else if (term || lethal) reward = (double)t - tb;
      else {                                                           // brain_tumor_reward.py:59-73
        const double maxsafe = 116.0 * pow(treat_r, -0.685);
        const double ratio = (applied - maxsafe) / (P.total_dosage - maxsafe);
        const double r = fmin(fmax(ratio, 0.0), 1.0);
        reward = -50.0 * pow(r, 1.0 / 3.0);
      }
    } else if (stage == PDEGYM_TUMOR_POST && (term || lethal)) {
      reward = has_tb ? (double)t - tb : 0.0;
    }
    // ---- loop control of the wrapper's three loops
    if (term || lethal) break;
    if (mode == PDEGYM_TUMOR_RUN_GROWTH && stage != PDEGYM_TUMOR_GROWTH) break;
  }
  if constexpr (!SINGLE_DAY) {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = lane; i < nx; i += kWave) g[i] = cur[i];
  }
  if (lane != 0) return;
  Bf.time_index[inst] = t;
  Bf.stage[inst] = stage;
  Bf.remaining[inst] = remaining;
  days[PDEGYM_TUMOR_DAY_GROWTH] = growth;
  days[PDEGYM_TUMOR_DAY_THERAPY] = therapyDays;
  days[PDEGYM_TUMOR_DAY_POST] = post;
  days[PDEGYM_TUMOR_DAY_SIM] = sim;
  days[PDEGYM_TUMOR_DAY_DEATH] = death;
  Bf.reward[inst] = reward;
  Bf.terminated[inst] = term;
  Bf.truncated[inst] = lethal;
  double* o = Bf.out + (size_t)inst * PDEGYM_TUMOR_OUTS;
  o[PDEGYM_TUMOR_OUT_T1] = T1;
  o[PDEGYM_TUMOR_OUT_T2] = T2;
  o[PDEGYM_TUMOR_OUT_TREAT] = treat_r;
  o[PDEGYM_TUMOR_OUT_DOSE] = applied;
}

__global__ __launch_bounds__(kWave* kWavesPerBlock) void tumor_reset_kernel(pdegym_params_tumor P, pdegym_bufs_tumor Bf,
                                                                            const double* init, long long init_stride,
                                                                            const uint8_t* mask, int B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int inst = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (inst >= B) return;
  if (mask && !mask[inst]) return;
  const double* src = init + (size_t)inst * init_stride;
  double* g = Bf.u + (size_t)inst * P.nx;
  for (int i = lane; i < P.nx; i += kWave) g[i] = src[i];
  if (lane == 0) {
    Bf.time_index[inst] = 0;
    Bf.stage[inst] = PDEGYM_TUMOR_GROWTH;
    Bf.remaining[inst] = P.total_dosage;
    int32_t* days = Bf.days + (size_t)inst * PDEGYM_TUMOR_DAYS;
    days[0] = days[1] = days[2] = days[3] = 0;
    days[PDEGYM_TUMOR_DAY_DEATH] = -1;
  }
}

int check(const pdegym_params_tumor* prm, const pdegym_bufs_tumor* buf) {
  if (!prm || !buf) return pdegym::fail(-1, "null params/bufs");
  if (prm->nx < 3 || prm->nx > kMaxNx) return pdegym::fail(-2, "tumor: nx must be in [3, 4096]");
  if (prm->nt < 2) return pdegym::fail(-2, "tumor: nt must be >= 2");
  if (!buf->u || !buf->time_index || !buf->stage || !buf->remaining || !buf->days)
    return pdegym::fail(-3, "null device buffer");
  return 0;
}

}  // namespace

extern "C" {

static int tumor_launch(const pdegym_params_tumor* prm, const pdegym_bufs_tumor* buf, int mode, int max_days, int32_t B,
                        void* stream) {
  if (int rc = check(prm, buf)) return rc;
  if (!buf->xscale || !buf->control || !buf->reward || !buf->terminated || !buf->truncated || !buf->out)
    return pdegym::fail(-3, "null device buffer");
  if (mode < PDEGYM_TUMOR_RUN_ONE_DAY || mode > PDEGYM_TUMOR_RUN_TO_END) return pdegym::fail(-2, "tumor: unknown run mode");
  if (B <= 0 || max_days <= 0) return 0;
  // two LDS rows per wave, at most 64 KB per workgroup
  const int wpb = prm->nx <= 1024 ? 4 : (prm->nx <= 2048 ? 2 : 1);
  const dim3 grid((B + wpb - 1) / wpb), block(kWave * wpb);
  const size_t lds = (size_t)2 * wpb * prm->nx * sizeof(double);
  if (max_days == 1)
    hipLaunchKernelGGL(tumor_step_kernel<true>, grid, block, lds, (hipStream_t)stream, *prm, *buf, mode, max_days, B);
  else
    hipLaunchKernelGGL(tumor_step_kernel<false>, grid, block, lds, (hipStream_t)stream, *prm, *buf, mode, max_days, B);
  return pdegym::check_launch("tumor_step");
}

int pdegym_tumor_step(const pdegym_params_tumor* prm, const pdegym_bufs_tumor* buf, int32_t B, void* stream) {
  return tumor_launch(prm, buf, PDEGYM_TUMOR_RUN_ONE_DAY, 1, B, stream);
}

int pdegym_tumor_advance(const pdegym_params_tumor* prm, const pdegym_bufs_tumor* buf, int32_t mode, int32_t max_days,
                         int32_t B, void* stream) {
  return tumor_launch(prm, buf, mode, max_days, B, stream);
}

int pdegym_tumor_reset_masked(const pdegym_params_tumor* prm, const pdegym_bufs_tumor* buf, const double* init,
                              int64_t init_stride, const uint8_t* mask, int32_t B, void* stream) {
  if (int rc = check(prm, buf)) return rc;
  if (!init) return pdegym::fail(-1, "null init");
  if (B <= 0) return 0;
  const dim3 grid((B + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
  hipLaunchKernelGGL(tumor_reset_kernel, grid, block, 0, (hipStream_t)stream, *prm, *buf, init, (long long)init_stride, mask, B);
  return pdegym::check_launch("tumor_reset");
}

}  // extern "C"
