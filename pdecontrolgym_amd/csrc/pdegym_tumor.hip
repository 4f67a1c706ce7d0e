// pdegym_tumor.hip -- gfx950 kernel for the 1D brain-tumour radiotherapy environment (float64).
//
// One 64-lane wavefront owns one patient.  The live density row (nx = 201 in the shipped example) is staged once in
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

__global__ __launch_bounds__(kWave* kWavesPerBlock) void tumor_step_kernel(pdegym_params_tumor P, pdegym_bufs_tumor Bf, int B) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int w = threadIdx.x >> 6;
  const int inst = blockIdx.x * kWavesPerBlock + w;
  const int nx = P.nx;
  double* row = lds + (size_t)w * nx;
  const bool present = inst < B;
  int t = 0;
  bool live = false;
  if (present) {
    t = Bf.time_index[inst];
    live = t < P.nt - 1;                                             // :136
  }
  double* g = Bf.u + (size_t)(present ? inst : 0) * nx;
  // ---- stage the live row, T2 radius of the row BEFORE the update (:255 reads time_index-1)
  int t2_idx = -1;
  if (live) {
    for (int i = lane; i < nx; i += kWave) {
      const double v = g[i];
      row[i] = v;
      if (v >= P.thr_t2) t2_idx = i;                                 // i grows with the loop: keeps the largest
    }
    t2_idx = wave_max_i(t2_idx);
  }
  __syncthreads();
  if (!live) {
    if (present && lane == 0) {
      Bf.reward[inst] = 0.0;
      Bf.terminated[inst] = 0;
      Bf.truncated[inst] = 0;
    }
    return;
  }
  t += 1;
  const int stage0 = Bf.stage[inst];
  double remaining = Bf.remaining[inst];
  double applied = 0.0, treat_r = 0.0, kill = 0.0;
  const bool therapy = stage0 == PDEGYM_TUMOR_THERAPY;
  if (therapy) {                                                     // :158-168, :247-263
    const double want = Bf.control[inst] * P.total_dosage;
    applied = remaining < want ? remaining : want;                   // Python min(want, remaining)
    remaining = remaining - applied;
    treat_r = t2_idx < 0 ? 0.0 : (double)t2_idx * P.dx + P.margin;
    if (Bf.kill) {
      kill = Bf.kill[inst];
    } else {
      const double bed = applied + ((applied * applied) / P.alpha_beta_ratio);
      kill = 1.0 - exp(-P.alpha * bed);
    }
  }
  // ---- finite-difference update, Neumann ends, clip to [0, k]; T1/T2 radii of the new row
  int t1_idx = -1, t2n_idx = -1;
  for (int i = lane; i < nx; i += kWave) {
    int c = i == 0 ? 1 : (i == nx - 1 ? nx - 2 : i);                 // :241-242 copy the neighbour's new value
    const bool rad = therapy && Bf.xscale[c] <= treat_r;             // outside: BED = 0 -> R = 1 - exp(-0) = 0
    const double v = clip0k(fd_node(P, row[c - 1], row[c], row[c + 1], kill, rad), P.k);
    g[i] = v;
    if (v >= P.thr_t1) t1_idx = i;
    if (v >= P.thr_t2) t2n_idx = i;
  }
  t1_idx = wave_max_i(t1_idx);
  t2n_idx = wave_max_i(t2n_idx);
  if (lane != 0) return;
  // ---- scalar bookkeeping (lane 0): stage machine :146-177, terminate/truncate :280-352, reward
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  const double T1 = t1_idx < 0 ? nan : (double)t1_idx * P.dx;
  int32_t* days = Bf.days + (size_t)inst * PDEGYM_TUMOR_DAYS;
  int growth = days[PDEGYM_TUMOR_DAY_GROWTH], therapyDays = days[PDEGYM_TUMOR_DAY_THERAPY];
  int post = days[PDEGYM_TUMOR_DAY_POST], sim = days[PDEGYM_TUMOR_DAY_SIM], death = days[PDEGYM_TUMOR_DAY_DEATH];
  int stage = stage0;
  if (stage0 == PDEGYM_TUMOR_GROWTH) {
    growth = t;
    if (t1_idx >= 0 && T1 >= P.detect_radius) stage = PDEGYM_TUMOR_THERAPY;
  } else if (therapy && remaining < P.dose_end) {
    therapyDays = t - growth;
    stage = PDEGYM_TUMOR_POST;
  }
  const bool term = t >= P.nt - 1;
  const bool lethal = t1_idx >= 0 && T1 >= P.death_radius;
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
  double reward = 0.0;
  const double tb = Bf.t_benchmark ? Bf.t_benchmark[inst] : nan;
  const bool has_tb = tb == tb;
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
  o[PDEGYM_TUMOR_OUT_T2] = t2n_idx < 0 ? nan : (double)t2n_idx * P.dx;
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

int pdegym_tumor_step(const pdegym_params_tumor* prm, const pdegym_bufs_tumor* buf, int32_t B, void* stream) {
  if (int rc = check(prm, buf)) return rc;
  if (!buf->xscale || !buf->control || !buf->reward || !buf->terminated || !buf->truncated || !buf->out)
    return pdegym::fail(-3, "null device buffer");
  if (B <= 0) return 0;
  const dim3 grid((B + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
  const size_t lds = (size_t)kWavesPerBlock * prm->nx * sizeof(double);
  hipLaunchKernelGGL(tumor_step_kernel, grid, block, lds, (hipStream_t)stream, *prm, *buf, B);
  return pdegym::check_launch("tumor_step");
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
