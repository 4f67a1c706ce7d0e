// pdegym_traffic.hip -- gfx950 kernel for the Aw-Rascle-Zhang traffic environment (float64).
//
// One 64-lane wavefront owns one freeway instance, lane j holds node j (the reference's M = 51 nodes), both fields
// (r, y) stay in registers for the control_freq sub-steps of a step() call.  Neighbour values cross lanes with
// wave shuffles; the "minus" midpoint of node j IS the "plus" midpoint of node j-1 (same floating-point expression,
// addition is commutative), so each lane evaluates one midpoint flux and the other arrives from lane j-1.
// Operation order follows environments1d/traffic_arz_env.py:172-227 exactly (-ffp-contract=off, IEEE division):
// fields are bit-identical to NumPy; the reward norms are wave reductions (rtol 1e-14 vs BLAS ddot).
#include <hip/hip_runtime.h>

#include "pdegym.h"
#include "pdegym_common.h"
#include "pdegym_policy.h"

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;

// Wave-wide float64 sum on DPP (result in every lane).  Round 5: __shfl_xor compiles to ds_bpermute_b32 -- two per double and
// step, 24 LDS round trips for the two reward norms of an env-step, each ~25 cycles of the wave's SIMD (docs/HISTORY.md section 4:
// tools/attic/ab_ns_col.py).  Same scheme as wave_reduce of the 1D kernels: lane pair, quad (quad_perm), half row, row (row_half_mirror /
// row_mirror), lane 15 of rows 0 and 2 into rows 1 and 3 (row_bcast:15), lane 31 into rows 2, 3 (row_bcast:31); lane 63 holds the
// total and v_readlane hands it to everybody.  A fixed order (deterministic): the reward norms were never bitwise against BLAS
// ddot (tests: rtol 1e-12); step and rollout kernels share it, so they stay bit-identical to each other.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_d(double v) {      // lanes without a source (or masked rows) read 0.0
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
  v += dpp_move_d<0xB1, 0xf>(v);     // quad_perm:[1,0,3,2]
  v += dpp_move_d<0x4E, 0xf>(v);     // quad_perm:[2,3,0,1]
  v += dpp_move_d<0x141, 0xf>(v);    // row_half_mirror
  v += dpp_move_d<0x140, 0xf>(v);    // row_mirror
  v += dpp_move_d<0x142, 0xa>(v);    // row_bcast:15 -> rows 1, 3
  v += dpp_move_d<0x143, 0xc>(v);    // row_bcast:31 -> rows 2, 3
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Neighbouring lanes of a double: two v_mov_b32_dpp each (wave_shl / wave_shr; the lane without a source reads 0, which no
// valid node ever uses).  The empty asm pins the shift where it is written: folded into a branch that masks lanes off (the
// inner update runs on lanes 1 .. M-2 only), a DPP read from a masked-off neighbour would return 0.  ds_bpermute
// (__shfl_up / __shfl_down, what this kernel used first) costs an LDS round trip per shuffle.
__device__ __forceinline__ double lane_next(double v) {        // lane i <- lane i + 1
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xf, 0xf, true);
  double r = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
  asm volatile("" : "+v"(r));
  return r;
}
__device__ __forceinline__ double lane_prev(double v) {        // lane i <- lane i - 1
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xf, 0xf, true);
  double r = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
  asm volatile("" : "+v"(r));
  return r;
}
__device__ __forceinline__ double lane_value(double v, int l) {   // l: wave-uniform
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double Veq(double vm, double rm, double rho) { return vm * (1 - rho / rm); }  // :270-272
__device__ __forceinline__ double F_r(double vm, double rm, double rho, double y) { return y + rho * Veq(vm, rm, rho); }
__device__ __forceinline__ double F_y(double vm, double rm, double rho, double y) { return y * (y / rho + Veq(vm, rm, rho)); }

// Scalar sub-expressions that Python evaluates before they meet an array (traffic_arz_env.py:201-222, :106): formed ONCE on the
// host in the same double arithmetic (IEEE division is correctly rounded on both sides, so the values are bit-identical to
// forming them per wave on the device -- which cost five float64 divisions, a third of the step kernel's instructions).
struct TrafficConsts {
  double c1, c2, c3, c4;   // dt/(2 dx), 0.25 dt/tau, dt/dx, 0.5 dt/tau
  double t_end;            // T/dt  (:106: seconds compared with a step count -- kept)
};
inline TrafficConsts traffic_consts(const pdegym_params_traffic& P) {
  TrafficConsts K;
  K.c1 = P.dt / (2 * P.dx);
  K.c2 = 0.25 * P.dt / P.tau;
  K.c3 = P.dt / P.dx;
  K.c4 = 0.5 * P.dt / P.tau;
  K.t_end = P.T / P.dt;
  return K;
}

// One env-step of one freeway held by one wave (M <= 64: node j in lane j): action clip, control_freq Lax-Wendroff sub-steps,
// speed, reward, flags.  Shared by traffic_step_kernel and every iteration of traffic_rollout_kernel.
struct TrafficStepOut {
  double v, reward, vs;
  bool done, trunc;
};

__device__ __forceinline__ TrafficStepOut traffic_step_wave(const pdegym_params_traffic& P, const TrafficConsts& K, double& r, double& y,
                                                            double& time, const double rs, const double qc, double a0, double a1,
                                                            const int lane) {
  const int M = P.M;
  const bool in = lane < M;
  const double vm = P.vm, rm = P.rm, dt = P.dt, dx = P.dx;
  const double vs = Veq(vm, rm, rs);              // :66-72
  const double qs = rs * vs;
  // action clip (:151-156) -- np.clip(a, low, high) = min(max(a, low), high)
  const double lo = qc * 0.8, hi = 1.2 * qc;
  a0 = fmin(fmax(a0, lo), hi);
  a1 = fmin(fmax(a1, lo), hi);
  double q_in, q_out;
  if (P.sim == PDEGYM_TRAFFIC_BOTH) { q_in = a0; q_out = a1; }
  else if (P.sim == PDEGYM_TRAFFIC_INLET) { q_in = a0; q_out = qs; }
  else { q_in = qs; q_out = a0; }

  time = time + dt;                               // :146
  const double c1 = K.c1, c2 = K.c2, c3 = K.c3, c4 = K.c4;
  if (time < P.T) {                               // :172  (time does not change inside the loop)
    for (int s = 0; s < P.control_freq; ++s) {
      // boundary conditions :174-190: the end nodes take their neighbour's density, then y = q - r Veq(r).  Round 5: that Veq(r)
      // IS the Veq(r) the nodal fluxes below evaluate for the same lane (same expression, same operand), so the density is set
      // first, Veq is formed ONCE for the whole wave and the two end lanes pick their y by select -- the round-4 form ran two more
      // float64 division sequences per sub-step under single-lane branches (a third of the loop's divisions).
      const double r1 = lane_value(r, 1), rm2 = lane_value(r, M - 2);
      const bool first = lane == 0, last = lane == M - 1;
      r = first ? r1 : (last ? rm2 : r);
      const double ve = Veq(vm, rm, r);
      const double yb = (first ? q_in : q_out) - r * ve;
      y = (first || last) ? yb : y;
      // nodal fluxes and the "plus" midpoint (:201-216)
      const double fr = y + r * ve, fy = y * (y / r + ve);      // F_r, F_y with the shared Veq(r)
      const double r_p = lane_next(r), y_p = lane_next(y), fr_p = lane_next(fr), fy_p = lane_next(fy);
      const double r_pm = 0.5 * (r_p + r) - c1 * (fr_p - fr);
      const double y_pm = (0.5 * (y_p + y) - c1 * (fy_p - fy)) - c2 * (y_p + y);
      const double Frp = F_r(vm, rm, r_pm, y_pm), Fyp = F_y(vm, rm, r_pm, y_pm);
      // the "minus" midpoint of node j is the "plus" midpoint of node j-1
      const double Frm = lane_prev(Frp), Fym = lane_prev(Fyp), y_mm = lane_prev(y_pm);
      if (lane >= 1 && lane <= M - 2) {           // inner update :219-223
        r = r - c3 * (Frp - Frm);
        y = y - (c3 * (Fyp - Fym) + c4 * (y_pm + y_mm));
      }
    }
  }
  TrafficStepOut o;
  o.vs = vs;
  o.v = y / r + Veq(vm, rm, r);                    // :227
  // reward (traffic_arz_reward.py:22)
  const double dv = in ? o.v - vs : 0.0, dr = in ? r - rs : 0.0;
  // ||v - vs|| / vs + ||r - rs|| / rs: the two square roots and the two divisions act on wave-uniform values, so they share ONE
  // float64 sqrt and ONE division sequence -- lane 0 takes the speed term, every other lane the density term (same operands, same
  // operations: the same bits as two scalar evaluations)
  const double sv = wave_sum_d(dv * dv), sr = wave_sum_d(dr * dr);
  const double term_ = sqrt(lane == 0 ? sv : sr) / (lane == 0 ? vs : rs);
  o.reward = -(lane_value(term_, 0) + lane_value(term_, 1));
  const bool term = time >= K.t_end;               // :106 (seconds compared with a step count -- kept)
  if (term) time = 0.0;
  bool trunc = false;
  if (P.limit) trunc = __any(in && (o.v > vm || r > rm));
  o.trunc = trunc || !__any(in && (dr != 0.0 || dv != 0.0));   // exact steady state (:127-128)
  o.done = (P.sim == PDEGYM_TRAFFIC_OUTLET_TRAIN) ? term : (term || o.reward > -0.00023);
  return o;
}

// observation row of one freeway: (r, v), or ((r - rs)/rs, (v - vs)/vs) for outlet-train (:227-230)
__device__ __forceinline__ void traffic_emit_obs(const pdegym_params_traffic& P, double* o, double r, double v, double rs, double vs,
                                                 int lane) {
  const int M = P.M;
  if (lane < M) {
    if (P.sim == PDEGYM_TRAFFIC_OUTLET_TRAIN) {
      o[lane] = (r - rs) / rs;
      o[M + lane] = (v - vs) / vs;
    } else {
      o[lane] = r;
      o[M + lane] = v;
    }
  }
}

// pool row of the k-th restart of instance b (include/pdegym.h)
__device__ __forceinline__ int traffic_pool_row(const pdegym_bufs_traffic& Bf, int inst, int B) {
  const int rows = Bf.reset_pool_rows > 0 ? Bf.reset_pool_rows : B;
  const long long k = Bf.reset_count ? (long long)Bf.reset_count[inst] : 0;
  return (int)(((long long)inst + k * B) % rows);
}

// TrafficPDE1D.reset of one node (traffic_arz_env.py:256-258; the expressions of traffic_reset_kernel)
__device__ __forceinline__ void traffic_restart_node(const pdegym_params_traffic& P, double rs, double prof, double& r, double& y,
                                                     double& v) {
  const double vs = Veq(P.vm, P.rm, rs), qs = rs * vs;
  r = rs * prof;
  y = (qs * 1.0 - P.vm * r) + (P.vm / P.rm) * (r * r);
  v = y / r + Veq(P.vm, P.rm, r);
}

__global__ __launch_bounds__(kWave* kWavesPerBlock) void traffic_step_kernel(pdegym_params_traffic P, pdegym_bufs_traffic Bf, TrafficConsts K, int B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int inst = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (inst >= B) return;
  const int M = P.M;
  const bool in = lane < M;
  double r = in ? Bf.r[(size_t)inst * M + lane] : 1.0;
  double y = in ? Bf.y[(size_t)inst * M + lane] : 0.0;
  const double rs = Bf.rs[inst];
  const int astr = Bf.action_stride > 0 ? Bf.action_stride : 2;      // 1: one command per freeway (no second column to read)
  const double a0 = Bf.action[(size_t)inst * astr], a1 = astr > 1 ? Bf.action[(size_t)inst * astr + 1] : 0.0;
  double time = Bf.time[inst];
  const TrafficStepOut o = traffic_step_wave(P, K, r, y, time, rs, Bf.qs_clip[inst], a0, a1, lane);
  if (Bf.reset_rs && (o.done || o.trunc)) {       // fused auto-reset (wave-uniform)
    if (Bf.final_obs) traffic_emit_obs(P, Bf.final_obs + (size_t)inst * 2 * M, r, o.v, rs, o.vs, lane);
    const double rs_new = Bf.reset_rs[traffic_pool_row(Bf, inst, B)];
    double v0 = 0.0;
    if (in) {
      traffic_restart_node(P, rs_new, Bf.reset_profile[lane], r, y, v0);
      Bf.r[(size_t)inst * M + lane] = r;
      Bf.y[(size_t)inst * M + lane] = y;
      double* ob = Bf.obs + (size_t)inst * 2 * M;
      ob[lane] = r;
      ob[M + lane] = v0;
    }
    if (lane == 0) {
      Bf.rs[inst] = rs_new;
      if (Bf.reset_count) Bf.reset_count[inst] += 1;
      Bf.time[inst] = 0.0;
      Bf.reward[inst] = o.reward;
      Bf.done[inst] = o.done ? 1 : 0;
      Bf.truncated[inst] = o.trunc ? 1 : 0;
    }
    return;
  }
  if (in) {
    Bf.r[(size_t)inst * M + lane] = r;
    Bf.y[(size_t)inst * M + lane] = y;
  }
  traffic_emit_obs(P, Bf.obs + (size_t)inst * 2 * M, r, o.v, rs, o.vs, lane);
  if (lane == 0) {
    Bf.time[inst] = time;
    Bf.reward[inst] = o.reward;
    Bf.done[inst] = o.done ? 1 : 0;
    Bf.truncated[inst] = o.trunc ? 1 : 0;
  }
}

// T env-steps of one freeway by one wave in ONE launch (pdegym_traffic_rollout): (r, y) stay in registers across env-steps,
// iteration t is traffic_step_wave with the command of row t, and writes observation slot t + 1, reward / done / truncated row
// t -- bit-identical to T pdegym_traffic_step calls (a finished episode keeps running, as it does there: restarting is the
// caller's business in this engine).  With a policy (pdegym_policy.h) the command of step t is computed inside the launch
// from observation slot t rounded to float32 -- the cast SB3 makes in front of its float32 network -- widened back to
// float64, plus noise, clamped, and stored to actions row t.
// WIDE: a policy with a layer of more than 64 units -- the 16 waves of the workgroup evaluate it together (pdegym_policy.h: eval_wide),
// so every wave, with or without a freeway, runs all T iterations (barriers).
// POLICY: compiled apart, so that the commands-given-ahead form carries none of the policy's loads (a load under a run-time branch
// leaves the compiler's wait-count pass with "may be pending" registers at every join of the loop).
template <bool WIDE, bool POLICY>
__global__ __launch_bounds__(kWave* pdegym_policy::kWaves) void traffic_rollout_kernel(pdegym_params_traffic P, pdegym_bufs_traffic Bf,
                                                                                       pdegym_rollout_traffic Ro, pdegym_mlp N, TrafficConsts K,
                                                                                       int B) {
  static_assert(POLICY || !WIDE, "the cooperative evaluation belongs to a policy");
  constexpr bool has_policy = POLICY;
  namespace pol = pdegym_policy;
  extern __shared__ __attribute__((aligned(16))) float pol_smem[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int inst_raw = blockIdx.x * pol::kWaves + wave;
  const int M = P.M, D = 2 * M, xpad = pol::xpad(D);
  pol::Staged St = {};
  pol::Wide Wd;
  if constexpr (WIDE) Wd = pol::wide_setup(N, pol_smem, D, wave, lane);
  else if constexpr (has_policy) St = pol::stage(N, pol_smem);        // the launch's only barrier (uniform over the grid)
  const bool active = inst_raw < B;      // wave-uniform
  if (!WIDE && !active) return;
  const int inst = active ? inst_raw : 0;      // a wave without a freeway only attends the barriers: it reads instance 0, stores nothing
  float* const xw = WIDE ? Wd.X + wave * Wd.ldx : pol_smem + St.end + wave * (xpad + 2 * pol::kMaxWidth);
  float* const hw = xw + xpad;
  const bool in = lane < M;
  double r = in ? Bf.r[(size_t)inst * M + lane] : 1.0;
  double y = in ? Bf.y[(size_t)inst * M + lane] : 0.0;
  double rs = Bf.rs[inst];
  const double qc = Bf.qs_clip[inst];
  double time = Bf.time[inst];
  int restarts = 0;
  const double prof = (Bf.reset_rs && in) ? Bf.reset_profile[lane] : 1.0;
  const int A = Bf.action_stride > 0 ? Bf.action_stride : 2;
  const size_t slot = (size_t)B * D;
  // commands given ahead (no policy): 64 env-steps per load -- lane l holds the command(s) of step t0 + l, handed out by v_readlane --
  // and no fence between iterations (nothing stored by an iteration is read by the next): as in rollout1d_kernel, a load or a
  // fence per env-step makes every step wait for the stores of the step before (gfx9 counts both in vmcnt)
  auto command_batch = [&](int t0, int col) {
    return (!has_policy && t0 + lane < Ro.T && col < A) ? Ro.actions[((size_t)(t0 + lane) * B + inst) * A + col] : 0.0;
  };
  double b0 = command_batch(0, 0), b1 = command_batch(0, 1);
  // vmcnt(0) as a real instruction wherever the open-loop form loads (here, at a batch hand-over, after a restart): the wait-count pass
  // then knows that nothing is pending at the joins of the loop and puts no static wait into its steady state
  if constexpr (!has_policy) __builtin_amdgcn_s_waitcnt(0x0F70);
  for (int t = 0; t < Ro.T; ++t) {
    double* arow = Ro.actions + ((size_t)t * B + inst) * A;
    double a0, a1 = 0.0;
    const int tj = t & (kWave - 1);
    if (!has_policy && tj == 0 && t) {      // wave-uniform, once per 64 env-steps; the loop's only wait for memory
      b0 = command_batch(t, 0);
      b1 = command_batch(t, 1);
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    if constexpr (has_policy) {
      const double* xrow = Ro.obs + (size_t)t * slot + (size_t)inst * D;
      float c0, c1;
      if constexpr (WIDE) {
        if (active)
          for (int j = lane; j < D; j += kWave) xw[j] = (float)xrow[j];
        pol::eval_wide(N, Wd, D, wave, lane);
        if (!active) continue;
        c0 = pol::wide_out(Wd, wave, 0);
        c1 = A > 1 ? pol::wide_out(Wd, wave, 1) : 0.f;
      } else {
        for (int j = lane; j < xpad; j += kWave) xw[j] = j < D ? (float)xrow[j] : 0.f;
        pol::wave_lds_sync();
        const float o = pol::eval(N, St, pol_smem, xw, hw, D, lane);
        c0 = pol::lane_value(o, 0);
        c1 = A > 1 ? pol::lane_value(o, 1) : 0.f;
      }
      if (N.noise) {
        const float* nz = N.noise + ((size_t)t * B + inst) * N.noise_stride;
        c0 += nz[0];
        if (A > 1) c1 += nz[1];
      }
      if (N.clamp) {
        c0 = fminf(fmaxf(c0, N.lo), N.hi);
        c1 = fminf(fmaxf(c1, N.lo), N.hi);
      }
      a0 = (double)c0;
      a1 = (double)c1;
      if (lane == 0) {
        arow[0] = a0;
        if (A > 1) arow[1] = a1;
      }
    } else {
      a0 = lane_value(b0, tj);
      a1 = lane_value(b1, tj);
    }
    const TrafficStepOut o = traffic_step_wave(P, K, r, y, time, rs, qc, a0, a1, lane);
    double* onext = Ro.obs + (size_t)(t + 1) * slot + (size_t)inst * D;
    if (Bf.reset_rs && (o.done || o.trunc)) {     // fused auto-reset, as in traffic_step_kernel
      if (Bf.final_obs) traffic_emit_obs(P, Bf.final_obs + (size_t)inst * D, r, o.v, rs, o.vs, lane);
      const int rows = Bf.reset_pool_rows > 0 ? Bf.reset_pool_rows : B;
      const long long k = (Bf.reset_count ? (long long)Bf.reset_count[inst] : 0) + restarts;
      rs = Bf.reset_rs[(int)(((long long)inst + k * B) % rows)];
      ++restarts;
      double v0 = 0.0;
      if (in) {
        traffic_restart_node(P, rs, prof, r, y, v0);
        onext[lane] = r;
        onext[M + lane] = v0;
      }
      time = 0.0;
      if constexpr (!has_policy) __builtin_amdgcn_s_waitcnt(0x0F70);
    } else {
      traffic_emit_obs(P, onext, r, o.v, rs, o.vs, lane);
    }
    if (lane == 0) {
      Ro.rewards[(size_t)t * B + inst] = o.reward;
      Ro.done[(size_t)t * B + inst] = o.done ? 1 : 0;
      Ro.truncated[(size_t)t * B + inst] = o.trunc ? 1 : 0;
    }
    if constexpr (has_policy) {      // the observation row just stored is read back (by other lanes) at the start of the next iteration
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
  if (!active) return;
  if (in) {
    Bf.r[(size_t)inst * M + lane] = r;
    Bf.y[(size_t)inst * M + lane] = y;
  }
  if (lane == 0) {
    Bf.time[inst] = time;
    if (restarts) {
      Bf.rs[inst] = rs;
      if (Bf.reset_count) Bf.reset_count[inst] += restarts;
    }
  }
}

// ---- rows of more than 64 nodes (finer grids than the reference notebook's M = 51): one wave still owns one freeway,
// node j lives in lane j % 64; the fields and the per-sub-step intermediates go through a wave-private LDS region so that
// every node reaches its neighbours (wave-level ordering only: no workgroup barrier).  Same expressions, same order.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(kWave* kWavesPerBlock) void traffic_step_wide_kernel(pdegym_params_traffic P, pdegym_bufs_traffic Bf, TrafficConsts K, int B) {
  extern __shared__ double tl[];
  const int lane = threadIdx.x & (kWave - 1);
  const int w = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int inst = __builtin_amdgcn_readfirstlane(blockIdx.x * wpb + w);
  if (inst >= B) return;
  const int M = P.M;
  double* R = tl + (size_t)w * 7 * M;      // r, y, fr, fy | y_pm, Frp, Fyp
  double* Y = R + M;
  double* FR = Y + M;
  double* FY = FR + M;
  double* YPM = FY + M;
  double* FRP = YPM + M;
  double* FYP = FRP + M;
  const double vm = P.vm, rm = P.rm, dt = P.dt, dx = P.dx;
  for (int j = lane; j < M; j += kWave) {
    R[j] = Bf.r[(size_t)inst * M + j];
    Y[j] = Bf.y[(size_t)inst * M + j];
  }
  const double rs = Bf.rs[inst];
  const double vs = Veq(vm, rm, rs);
  const double qs = rs * vs;
  const double qc = Bf.qs_clip[inst];
  const int astr = Bf.action_stride > 0 ? Bf.action_stride : 2;      // 1: one command per freeway (no second column to read)
  double a0 = Bf.action[(size_t)inst * astr], a1 = astr > 1 ? Bf.action[(size_t)inst * astr + 1] : 0.0;
  const double lo = qc * 0.8, hi = 1.2 * qc;
  a0 = fmin(fmax(a0, lo), hi);
  a1 = fmin(fmax(a1, lo), hi);
  double q_in, q_out;
  if (P.sim == PDEGYM_TRAFFIC_BOTH) { q_in = a0; q_out = a1; }
  else if (P.sim == PDEGYM_TRAFFIC_INLET) { q_in = a0; q_out = qs; }
  else { q_in = qs; q_out = a0; }
  double time = Bf.time[inst] + dt;
  const double c1 = K.c1, c2 = K.c2, c3 = K.c3, c4 = K.c4;
  wave_lds_sync();
  if (time < P.T) {
    for (int s = 0; s < P.control_freq; ++s) {
      // boundary conditions :174-190 (read the neighbours first, then overwrite the two end nodes)
      const double r1 = R[1], rm2 = R[M - 2];
      wave_lds_sync();
      if (lane == 0) {
        R[0] = r1;
        Y[0] = q_in - r1 * Veq(vm, rm, r1);
        R[M - 1] = rm2;
        Y[M - 1] = q_out - rm2 * Veq(vm, rm, rm2);
      }
      wave_lds_sync();
      for (int j = lane; j < M; j += kWave) {      // nodal fluxes :201-204
        const double r = R[j], y = Y[j];
        FR[j] = F_r(vm, rm, r, y);
        FY[j] = F_y(vm, rm, r, y);
      }
      wave_lds_sync();
      for (int j = lane; j < M - 1; j += kWave) {  // "plus" midpoint of node j and its fluxes :205-216
        const double r = R[j], y = Y[j], r_p = R[j + 1], y_p = Y[j + 1];
        const double r_pm = 0.5 * (r_p + r) - c1 * (FR[j + 1] - FR[j]);
        const double y_pm = (0.5 * (y_p + y) - c1 * (FY[j + 1] - FY[j])) - c2 * (y_p + y);
        YPM[j] = y_pm;
        FRP[j] = F_r(vm, rm, r_pm, y_pm);
        FYP[j] = F_y(vm, rm, r_pm, y_pm);
      }
      wave_lds_sync();
      for (int j = lane; j < M; j += kWave) {      // inner update :219-223 ("minus" midpoint of j = "plus" of j-1)
        if (j >= 1 && j <= M - 2) {
          const double r = R[j], y = Y[j];
          R[j] = r - c3 * (FRP[j] - FRP[j - 1]);
          Y[j] = y - (c3 * (FYP[j] - FYP[j - 1]) + c4 * (YPM[j] + YPM[j - 1]));
        }
      }
      wave_lds_sync();
    }
  }
  double sv = 0.0, sr = 0.0;
  bool over = false, moved = false;
  double* o = Bf.obs + (size_t)inst * 2 * M;
  for (int j = lane; j < M; j += kWave) {
    const double r = R[j], y = Y[j];
    const double v = y / r + Veq(vm, rm, r);       // :227
    const double dv = v - vs, dr = r - rs;
    sv += dv * dv;
    sr += dr * dr;
    over = over || v > vm || r > rm;
    moved = moved || dr != 0.0 || dv != 0.0;
    Bf.r[(size_t)inst * M + j] = r;
    Bf.y[(size_t)inst * M + j] = y;
    if (P.sim == PDEGYM_TRAFFIC_OUTLET_TRAIN) {
      o[j] = (r - rs) / rs;
      o[M + j] = (v - vs) / vs;
    } else {
      o[j] = r;
      o[M + j] = v;
    }
  }
  const double nv = sqrt(wave_sum_d(sv)), nr = sqrt(wave_sum_d(sr));
  const double reward = -(nv / vs + nr / rs);
  const bool term = time >= K.t_end;
  if (term) time = 0.0;
  bool trunc = false;
  if (P.limit) trunc = __any(over);
  trunc = trunc || !__any(moved);
  const bool done = (P.sim == PDEGYM_TRAFFIC_OUTLET_TRAIN) ? term : (term || reward > -0.00023);
  if (Bf.reset_rs && (done || trunc)) {           // fused auto-reset, as in traffic_step_kernel (each lane re-reads its own stores)
    const double rs_new = Bf.reset_rs[traffic_pool_row(Bf, inst, B)];
    double* fo = Bf.final_obs ? Bf.final_obs + (size_t)inst * 2 * M : nullptr;
    for (int j = lane; j < M; j += kWave) {
      if (fo) {
        fo[j] = o[j];
        fo[M + j] = o[M + j];
      }
      double r, y, v;
      traffic_restart_node(P, rs_new, Bf.reset_profile[j], r, y, v);
      Bf.r[(size_t)inst * M + j] = r;
      Bf.y[(size_t)inst * M + j] = y;
      o[j] = r;
      o[M + j] = v;
    }
    time = 0.0;
    if (lane == 0) {
      Bf.rs[inst] = rs_new;
      if (Bf.reset_count) Bf.reset_count[inst] += 1;
    }
  }
  if (lane == 0) {
    Bf.time[inst] = time;
    Bf.reward[inst] = reward;
    Bf.done[inst] = done ? 1 : 0;
    Bf.truncated[inst] = trunc ? 1 : 0;
  }
}

__global__ __launch_bounds__(kWave* kWavesPerBlock) void traffic_reset_kernel(pdegym_params_traffic P, pdegym_bufs_traffic Bf,
                                                                               const double* profile, const uint8_t* mask, int B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int inst = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (inst >= B || (mask && !mask[inst])) return;
  const int M = P.M;
  for (int j = lane; j < M; j += kWave) {
    const double rs = Bf.rs[inst];
    const double vs = Veq(P.vm, P.rm, rs), qs = rs * vs;
    // :256-258   r = rs*(sin(3x/L pi)*0.1 + 1) ; y = qs - vm r + vm/rm r^2 ; v = y/r + Veq(r)
    const double r = rs * profile[j];
    const double y = (qs * 1.0 - P.vm * r) + (P.vm / P.rm) * (r * r);
    const double v = y / r + Veq(P.vm, P.rm, r);
    Bf.r[(size_t)inst * M + j] = r;
    Bf.y[(size_t)inst * M + j] = y;
    Bf.obs[(size_t)inst * 2 * M + j] = r;
    Bf.obs[(size_t)inst * 2 * M + M + j] = v;
  }
  if (lane == 0) {
    Bf.time[inst] = 0.0;
    Bf.done[inst] = 0;
    Bf.truncated[inst] = 0;
  }
}

int check(const pdegym_params_traffic* prm, const pdegym_bufs_traffic* buf) {
  if (!prm || !buf) return pdegym::fail(-1, "null params/bufs");
  if (prm->M < 4 || prm->M > PDEGYM_TRAFFIC_MAX_M) return pdegym::fail(-2, "traffic: M must be in [4, 1024]");
  if (prm->control_freq < 1) return pdegym::fail(-2, "control_freq must be >= 1");
  if (prm->sim < 0 || prm->sim > 3) return pdegym::fail(-2, "bad simulation type");
  if (!buf->r || !buf->y || !buf->time || !buf->rs || !buf->qs_clip || !buf->obs || !buf->done || !buf->truncated)
    return pdegym::fail(-3, "null device buffer");
  if (buf->reset_rs && !buf->reset_profile) return pdegym::fail(-3, "reset_rs needs reset_profile");
  if (buf->reset_pool_rows < 0) return pdegym::fail(-2, "reset_pool_rows must be >= 0");
  return 0;
}

}  // namespace

extern "C" {

int pdegym_traffic_step(const pdegym_params_traffic* prm, const pdegym_bufs_traffic* buf, int32_t B, void* stream) {
  if (int rc = check(prm, buf)) return rc;
  if (!buf->action || !buf->reward) return pdegym::fail(-3, "null device buffer");
  if (B <= 0) return 0;
  if (prm->M <= kWave) {       // the reference's grid (M = 51): one node per lane, fields in registers
    const dim3 grid((B + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
    hipLaunchKernelGGL(traffic_step_kernel, grid, block, 0, (hipStream_t)stream, *prm, *buf, traffic_consts(*prm), B);
  } else {                     // finer grids: wave-private LDS rows, at most 56 KB per workgroup
    const int wpb = prm->M <= 256 ? 4 : (prm->M <= 512 ? 2 : 1);
    const dim3 grid((B + wpb - 1) / wpb), block(kWave * wpb);
    hipLaunchKernelGGL(traffic_step_wide_kernel, grid, block, (size_t)wpb * 7 * prm->M * sizeof(double), (hipStream_t)stream,
                       *prm, *buf, traffic_consts(*prm), B);
  }
  return pdegym::check_launch("traffic_step");
}

int pdegym_traffic_rollout(const pdegym_params_traffic* prm, const pdegym_bufs_traffic* buf, const pdegym_rollout_traffic* ro, int32_t B,
                           void* stream) {
  if (int rc = check(prm, buf)) return rc;
  if (!ro) return pdegym::fail(-1, "null rollout descriptor");
  if (B <= 0 || ro->T <= 0) return 0;
  if (prm->M > kWave) return pdegym::fail(-2, "traffic rollout: freeways of up to 64 nodes (the register-resident kernel)");
  if (!ro->obs || !ro->actions || !ro->rewards || !ro->done || !ro->truncated) return pdegym::fail(-3, "null rollout buffer");
  const int A = buf->action_stride > 0 ? buf->action_stride : 2;
  if (A > 2) return pdegym::fail(-2, "action_stride must be 1 or 2");
  pdegym_mlp net = {};
  size_t lds_bytes = 0;
  bool wide = false;
  if (ro->policy) {
    net = *ro->policy;
    if (A > pdegym_policy::kWideOut) return pdegym::fail(-2, "policy: at most two commands");
    if (const char* why = pdegym_policy::check(net, 2 * prm->M, A, true)) return pdegym::fail(-2, why);
    wide = pdegym_policy::is_wide(net);
    lds_bytes = (size_t)pdegym_policy::lds_floats(net, 2 * prm->M) * sizeof(float);
    static signed char attr[2][pdegym::kMaxDevices] = {};
    const void* fn = wide ? reinterpret_cast<const void*>(&traffic_rollout_kernel<true, true>) : reinterpret_cast<const void*>(&traffic_rollout_kernel<false, true>);
    if (!pdegym::raise_dynamic_lds_limit(fn, pdegym_policy::kMaxLdsBytes, attr[wide ? 1 : 0]))
      return pdegym::fail(-4, "cannot raise the dynamic LDS limit of traffic_rollout_kernel");
  }
  const dim3 grid((B + pdegym_policy::kWaves - 1) / pdegym_policy::kWaves), block(kWave * pdegym_policy::kWaves);
  if (wide)
    hipLaunchKernelGGL((traffic_rollout_kernel<true, true>), grid, block, lds_bytes, (hipStream_t)stream, *prm, *buf, *ro, net, traffic_consts(*prm), B);
  else if (ro->policy)
    hipLaunchKernelGGL((traffic_rollout_kernel<false, true>), grid, block, lds_bytes, (hipStream_t)stream, *prm, *buf, *ro, net, traffic_consts(*prm), B);
  else
    hipLaunchKernelGGL((traffic_rollout_kernel<false, false>), grid, block, lds_bytes, (hipStream_t)stream, *prm, *buf, *ro, net, traffic_consts(*prm), B);
  return pdegym::check_launch("traffic_rollout");
}

int pdegym_traffic_reset_masked(const pdegym_params_traffic* prm, const pdegym_bufs_traffic* buf, const double* profile,
                                const uint8_t* mask, int32_t B, void* stream) {
  if (int rc = check(prm, buf)) return rc;
  if (!profile) return pdegym::fail(-1, "null profile");
  if (B <= 0) return 0;
  const dim3 grid((B + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
  hipLaunchKernelGGL(traffic_reset_kernel, grid, block, 0, (hipStream_t)stream, *prm, *buf, profile, mask, B);
  return pdegym::check_launch("traffic_reset");
}

}  // extern "C"
