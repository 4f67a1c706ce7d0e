// pdegym_1d_body.h -- the device-side body shared by the 1D step kernels (pdegym_1d.hip) and the 1D rollout kernels
// (pdegym_1d_rollout.hip): lane shifts and wave reductions, the sub-step loops, one env-step of one instance by one wave
// (step1d_body) and the state a rollout carries in registers.  Two translation units so that they compile in parallel and so
// that a change of the rollout kernels leaves the fingerprint of the step kernels (and the counters committed for them) alone.
// Design notes: at the top of pdegym_1d.hip.
#ifndef PDEGYM_1D_BODY_H
#define PDEGYM_1D_BODY_H

#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pdegym.h"
#include "pdegym_common.h"

namespace {


constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;

// lane i <- lane i-1 ; lane 0 <- `edge`      (DPP wave_shr:1, gfx9 wave-wide shift)
__device__ __forceinline__ float from_left_lane(float v, float edge) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v),
                                                               0x138, 0xf, 0xf, false));
}
// lane i <- lane i+1 ; lane 63 <- `edge`     (DPP wave_shl:1)
__device__ __forceinline__ float from_right_lane(float v, float edge) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v),
                                                               0x130, 0xf, 0xf, false));
}

// Wave-wide reductions on DPP (result in every lane).  __shfl_xor compiles to ds_bpermute_b32, an LDS round trip per step:
// six dependent ones per reduction, four reductions per env-step on the critical path of a wave's prologue / epilogue.
// Steps: the lane pair, the quad (quad_perm), the half row and the row (row_half_mirror / row_mirror: lane i pairs with
// lane 7-i / 15-i), then lane 15 of rows 0 and 2 into rows 1 and 3 (row_bcast:15) and lane 31 into rows 2, 3
// (row_bcast:31): lane 63 holds the total, v_readlane hands it to everybody.  A fixed order (deterministic), not the
// butterfly's -- norms and rewards were never bitwise against a BLAS dot product anyway (tests: rtol 1e-6).
template <typename Op>
__device__ __forceinline__ float wave_reduce(float v, float identity, Op op) {
  auto dpp = [&](float x, const int ctrl_tag) {
    const int xi = __builtin_bit_cast(int, x), idn = __builtin_bit_cast(int, identity);
    int r;
    switch (ctrl_tag) {
      case 0: r = __builtin_amdgcn_update_dpp(idn, xi, 0xB1, 0xf, 0xf, false); break;    // quad_perm:[1,0,3,2]
      case 1: r = __builtin_amdgcn_update_dpp(idn, xi, 0x4E, 0xf, 0xf, false); break;    // quad_perm:[2,3,0,1]
      case 2: r = __builtin_amdgcn_update_dpp(idn, xi, 0x141, 0xf, 0xf, false); break;   // row_half_mirror
      case 3: r = __builtin_amdgcn_update_dpp(idn, xi, 0x140, 0xf, 0xf, false); break;   // row_mirror
      case 4: r = __builtin_amdgcn_update_dpp(idn, xi, 0x142, 0xa, 0xf, false); break;   // row_bcast:15 -> rows 1, 3
      default: r = __builtin_amdgcn_update_dpp(idn, xi, 0x143, 0xc, 0xf, false); break;  // row_bcast:31 -> rows 2, 3
    }
    return __builtin_bit_cast(float, r);
  };
  v = op(v, dpp(v, 0));
  v = op(v, dpp(v, 1));
  v = op(v, dpp(v, 2));
  v = op(v, dpp(v, 3));
  v = op(v, dpp(v, 4));
  v = op(v, dpp(v, 5));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum(float v) {
  return wave_reduce(v, 0.0f, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max(float v) {      // callers pass magnitudes: 0 is the identity
  return wave_reduce(v, 0.0f, [](float a, float b) { return fmaxf(a, b); });
}
// max of two magnitudes (sign bit clear) that propagates NaN like np.max: non-negative floats order as their bit patterns,
// and every NaN pattern lies above +inf.  The Linf reward norms use it (np.linalg.norm(row, ord=inf) of a row holding NaN is NaN).
__device__ __forceinline__ float mag_max(float a, float b) {
  const unsigned int ua = __float_as_uint(a), ub = __float_as_uint(b);
  return __uint_as_float(ua > ub ? ua : ub);
}
__device__ __forceinline__ float wave_mag_max(float v) {
  return wave_reduce(v, 0.0f, [](float a, float b) { return mag_max(a, b); });
}

// value of slot s when lane l holds slots [l*EPL, l*EPL+EPL)
template <int EPL>
__device__ __forceinline__ float slot_get(const float (&x)[EPL], int s) {
  const int src = s / EPL, e = s - src * EPL;
  float sel = x[0];
#pragma unroll
  for (int k = 1; k < EPL; ++k) sel = (e == k) ? x[k] : sel;
  return __shfl(sel, src);
}

template <int EPL>
__device__ __forceinline__ float slots_sumsq(const float (&x)[EPL], int s0, int ns) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) s += (s0 + e < ns) ? x[e] * x[e] : 0.f;
  return wave_sum(s);
}

// (a+1)*m-m, base_env_1d.py:36-39
__device__ __forceinline__ float normalize_ctrl(float a, float m, int on) { return on ? (a + 1.0f) * m - m : a; }

// s_waitcnt vmcnt(0) as a real instruction (the compiler's wait-count pass sees it): the carried rollout loop ends every RARE
// path that loads (exact redo, auto-reset) with it, so that no register is "possibly still being loaded" at the loop's back edge --
// otherwise the pass puts a static vmcnt(0) in front of the first use of each such register in EVERY iteration, and on gfx9 that
// also waits for all the stores in flight (vmcnt counts both).
__device__ __forceinline__ void drain_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0), expcnt / lgkmcnt untouched

// The norm ring of one instance (PDEGYM_RING slots, slot = fine time index mod PDEGYM_RING) as the step body sees it.
// RingMem: the ring in memory, written and read by lane 0 alone (program order of one lane).
// RingReg (the carried rollout kernels, round 5): the 128 slots live in TWO registers per lane for the whole launch (lane l holds
// slots l and 64 + l) -- a rollout step neither loads from nor stores to the ring, so nothing in its steady state waits on vmcnt
// (gfx9 counts loads AND stores in vmcnt: the one ring load per env-step made every step wait for the 13 stores of the step before;
// SQ_WAIT_ANY was 57 % of the wave cycles of parabolic_c2_s1_open_loop_rollout, profiles/r04p_summary.json).  The memory ring is
// read once at the head of the launch and written back once at its end (rollout1d_kernel).  Values are wave-uniform.
struct RingMem {
  float* mem;
  __device__ __forceinline__ void put(int idx, float v, int lane) {
    if (lane == 0) mem[idx] = v;
  }
  __device__ __forceinline__ float get(int idx, int lane) const { return lane == 0 ? mem[idx] : 0.f; }   // lane 0's value
};
struct RingReg {
  float lo, hi;
  __device__ __forceinline__ void put(int idx, float v, int lane) {
    lo = (lane == idx) ? v : lo;
    hi = (lane + kWave == idx) ? v : hi;
  }
  __device__ __forceinline__ float get(int idx, int) const {
    const float src = (idx & kWave) ? hi : lo;
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, src), idx & (kWave - 1)));
  }
};
static_assert(PDEGYM_RING == 2 * kWave, "RingReg keeps the ring in two registers per lane");

// EPL consecutive floats of one lane from / to a row (`base` = the row's first slot: 4-byte aligned, rows of 257 floats are not
// 16-byte aligned): a lane whose slots are all inside the row moves them as dwordx4 / x2 pieces (gfx950 global accesses need no
// alignment beyond the dword; the type says so to the compiler), the row's last lanes fall back to guarded dwords.
typedef float pdegym_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float pdegym_f2u __attribute__((ext_vector_type(2), aligned(4)));
template <int EPL>
__device__ __forceinline__ void store_slots(float* base, const float (&x)[EPL], int s0, int ns) {
  if (s0 + EPL <= ns) {
    float* p = base + s0;
    constexpr int Q = EPL / 4 * 4, D = (EPL - Q) / 2 * 2;
#pragma unroll
    for (int e = 0; e < Q; e += 4) *reinterpret_cast<pdegym_f4u*>(p + e) = pdegym_f4u{x[e], x[e + 1], x[e + 2], x[e + 3]};
    if constexpr (D == 2) *reinterpret_cast<pdegym_f2u*>(p + Q) = pdegym_f2u{x[Q], x[Q + 1]};
    if constexpr (Q + D < EPL) p[EPL - 1] = x[EPL - 1];
  } else {
#pragma unroll
    for (int e = 0; e < EPL; ++e)
      if (s0 + e < ns) base[s0 + e] = x[e];
  }
}
template <int EPL>
__device__ __forceinline__ void load_slots(const float* base, float (&x)[EPL], int s0, int ns) {
  if (s0 + EPL <= ns) {
    const float* p = base + s0;
    constexpr int Q = EPL / 4 * 4, D = (EPL - Q) / 2 * 2;
#pragma unroll
    for (int e = 0; e < Q; e += 4) {
      const pdegym_f4u v = *reinterpret_cast<const pdegym_f4u*>(p + e);
      x[e] = v.x; x[e + 1] = v.y; x[e + 2] = v.z; x[e + 3] = v.w;
    }
    if constexpr (D == 2) {
      const pdegym_f2u v = *reinterpret_cast<const pdegym_f2u*>(p + Q);
      x[Q] = v.x; x[Q + 1] = v.y;
    }
    if constexpr (Q + D < EPL) x[EPL - 1] = p[EPL - 1];
  } else {
#pragma unroll
    for (int e = 0; e < EPL; ++e) x[e] = (s0 + e < ns) ? base[s0 + e] : 0.f;
  }
}

// Row of the reset pools that the next restart of instance `inst` takes (see pdegym_bufs1d.reset_pool_rows).
__device__ __forceinline__ int pool_row(const pdegym_bufs1d& Bf, int inst, int B) {
  const int rows = Bf.reset_pool_rows > 0 ? Bf.reset_pool_rows : B;
  const long long k = Bf.reset_count ? (long long)Bf.reset_count[inst] : 0;
  return (int)(((long long)inst + k * (long long)B) % rows);
}

// normalize(control_update(control, neighbour, dx), max_control_value) as NumPy evaluates it for the given kind of `control`
// (hyperbolic.py:143-145, parabolic.py:148-150, base_env_1d.py:36-39); the result is what lands in the float32 row.
template <bool M64>
__device__ __forceinline__ float boundary_value(const pdegym_params1d& P, float a32, double a64, float neighbour, bool neumann) {
  if (!M64 || P.action_kind == PDEGYM_ACTION_F32) {
    const float v = neumann ? a32 * P.dx + neighbour : a32;
    return normalize_ctrl(v, P.max_control, P.normalize);
  }
  if (P.action_kind == PDEGYM_ACTION_F64 || !neumann) {
    double v = neumann ? a64 * P.dx64 + (double)neighbour : a64;
    if (P.normalize) v = (v + 1.0) * P.max_control64 - P.max_control64;
    return (float)v;
  }
  // NEP 50 weak Python scalar: control*dx is a Python float product, then adopts the float32 of the neighbour
  const float v = (float)(a64 * P.dx64) + neighbour;
  return normalize_ctrl(v, P.max_control, P.normalize);
}

// Per-wave state of one instance while it is stepped.
template <int EPL>
struct Row {
  float x[EPL];   // slots (row nodes J0 .. n-1)
  float bl;       // parabolic: node 0 (u(0,t)); unused for transport
  int t;          // time_index
  int k;          // (t + LOOKBACK) mod S
  double bsum;    // running sum of |u[tau,-1]|
  int back_row;   // row index the reward of THIS call looks back at (t_end - 100), if it is produced in this call
  float back_norm;  // its norm, captured in the loop (lane-uniform)
};

// State of one instance kept in registers ACROSS the env-steps of a rollout launch (rollout1d_kernel): the row, its plant
// parameter, the time index, the running |u[-1]| sum and the wave maximum of |dt*beta| (constant between resets).
template <int EPL>
struct Carry {
  float x[EPL];
  float beta[EPL];
  float bl;
  float cm;       // wave_max |dt * beta[j]| (the fast-loop overflow pre-check)
  float norm;     // ||row||_2 as the previous step (or the prologue) computed it: an upper bound of max |row[j]|
  int t;
  int k;          // (t + LOOKBACK) mod S, advanced with t (no division per env-step)
  double bsum;
  RingReg ring;   // the instance's norm ring, register-resident for the launch
  float glog;     // log2 of the growth bound g(cm): constant between resets (follows beta)
};

// growth bound of one parabolic sub-step (see the overflow pre-check in step1d_body)
__device__ __forceinline__ float growth_bound(const pdegym_params1d& P, float cm) {
  return ((P.F >= 0.0f && P.F <= 0.5f) ? 1.0f : 1.0f + 4.0f * fabsf(P.F)) + cm + 9.5367431640625e-7f;
}
// what follows a (re)draw of the carried beta: wave maximum of |dt*beta| and log2 of the growth bound.  (Carrying the fast loop's
// coefficients as well saved ~12 instructions per env-step of the open-loop kernel and cost 8 registers: the policy kernels -- 16 waves
// per workgroup, 128 registers per lane -- spilled 78 of them; with FULL rows the coefficients are four multiplications and one select.)
template <int EPL, bool PARABOLIC>
__device__ __forceinline__ void carry_refresh_beta(Carry<EPL>& C, const pdegym_params1d& P, int s0, int ns) {
  float cm = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) cm = fmaxf(cm, fabsf(P.dt * C.beta[e]));
  C.cm = wave_max(cm);
  C.glog = __log2f(growth_bound(P, C.cm));
  (void)s0;
  (void)ns;
}

// The reward's own norm of a register-resident row (NormReward kinds; same expressions as the epilogue of step1d_body), for the
// "t-horizon" reward: -(mean of the norms of the last k fine-time rows), norm_reward.py:60-73.
template <int EPL>
__device__ __forceinline__ float kind_norm(const float (&x)[EPL], float bl, int s0, int ns, int kind) {
  if (kind == PDEGYM_REWARD_NORM_L1) {
    float s1 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s1 += (s0 + e < ns) ? fabsf(x[e]) : 0.f;
    return wave_sum(s1) + fabsf(bl);
  }
  if (kind == PDEGYM_REWARD_NORM_LINF) {
    float m = fabsf(bl);
#pragma unroll
    for (int e = 0; e < EPL; ++e) m = mag_max(m, (s0 + e < ns) ? fabsf(x[e]) : 0.f);
    return wave_mag_max(m);
  }
  return sqrtf(slots_sumsq<EPL>(x, s0, ns) + bl * bl);
}

// S sub-steps of one instance.  FAST: boundary/padding slots are frozen by zero coefficients (no selects);
// otherwise explicit selects (exact for non-finite states, and required when the boundary value changes every
// sub-step, i.e. parabolic Neumann control).
// M64: float64 beta and/or float64 / Python-float control (pdegym_params1d.beta_f64 / action_kind): the select form with the
// reference's mixed-precision expressions (see step1d_wide_kernel for the same arithmetic on LDS-resident rows).
template <int EPL, bool PARABOLIC, bool NEUMANN, bool FAST, bool HIST, bool BURGERS = false, bool M64 = false, bool ROLL = false,
          typename RingT = RingMem, bool FULL = false>
__device__ __forceinline__ void run_substeps(Row<EPL>& R, const float (&beta)[EPL], const pdegym_params1d& P, int nsub,
                                             float a, RingT& ring, float* hist, int lane, const double* b64 = nullptr, double a64 = 0.0,
                                             float* xprev = nullptr, float* blprev = nullptr, int thor_k = 0) {
  // xprev/blprev (select form only): the row BEFORE the last sub-step, for NormReward's "differential" horizon
  // thor_k (select form only): NormReward "t-horizon" of length thor_k -- the reward's norm of each of the last thor_k - 1 rows
  // before the final one goes into the ring (the final row's is the epilogue's)
  // FAST && HIST (round 6): the fast arithmetic with one row store per sub-step into the trajectory -- the single environments'
  // default configuration (record_history: env.u is the whole trajectory, as in the reference)
  static_assert(!(FAST && NEUMANN), "fast mode is the Dirichlet path");
  static_assert(!(FAST && M64), "the mixed-precision mode uses the select form");
  constexpr int J0 = PARABOLIC ? 1 : 0;
  // FULL (round 5): the row fills the wave exactly (n - J0 == 64 EPL, checked by the launcher) -- with the slot count a compile-time
  // constant every "is this slot inside the row" select folds away and "is this the boundary slot" becomes lane 63's last slot
  const int n = FULL ? kWave * EPL + J0 : P.n, ns = n - J0, s0 = lane * EPL;
  const float dx = P.dx, dt = P.dt, F = P.F;
  const int S = P.substeps > 0 ? P.substeps : 1;
  const bool rec_all = P.nt <= PDEGYM_RING;
  // control_update (hyperbolic.py:68,95). Transport/Neumann reads u[t][-2] of the NEW row, which is still zero
  // (hyperbolic.py:144), so its boundary value is constant over the sub-steps.
  const float cdx = a * dx;
  const float rdxf = 1.0f / dx;   // exact when dx is a power of two (the only case it is used in)
  const bool pow2_dx = !PARABOLIC && dx > 0.0f && (__float_as_uint(dx) & 0x7fffffu) == 0u && rdxf * dx == 1.0f &&
                       rdxf < 3.0e38f;
  float bval = M64 ? boundary_value<true>(P, a, a64, 0.0f, NEUMANN)
                   : (NEUMANN ? normalize_ctrl(cdx + 0.0f, P.max_control, P.normalize) : normalize_ctrl(a, P.max_control, P.normalize));
  const bool beta64 = M64 && P.beta_f64;
  double c64[M64 ? EPL : 1];        // parabolic: dt*beta in double (parabolic.py:144 forms dt*beta first); transport: beta
  if constexpr (M64) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) c64[e] = beta64 ? (PARABOLIC ? P.dt64 * b64[e] : b64[e]) : 0.0;
  }

  // per-slot coefficients: parabolic c = dt*beta (parabolic.py:144 forms dt*beta first), transport c = beta
  float c[EPL], fe[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const bool interior = (s0 + e < ns - 1);
    const float cc = PARABOLIC ? dt * beta[e] : beta[e];
    if constexpr (FAST) {
      c[e] = interior ? cc : 0.0f;
      fe[e] = interior ? (PARABOLIC ? F : dt) : 0.0f;
    } else {
      c[e] = cc;
      fe[e] = PARABOLIC ? F : dt;
    }
  }

  // one PDE sub-step on the register-resident row (no bookkeeping).  GENERAL_EDGE: node 0 may be non-zero (first
  // sub-step after a reset) and, on the fast path, the frozen boundary slot takes the new control value.
  auto pde_substep = [&](auto general_edge, auto pow2_tag) {
    constexpr bool GENERAL_EDGE = decltype(general_edge)::value;
    constexpr bool POW2_DX = decltype(pow2_tag)::value;
    // p[first slot - 1]; after the first sub-step node 0 is identically 0, so the shift needs no fill operand
    const float xl = PARABOLIC ? from_left_lane(R.x[EPL - 1], GENERAL_EDGE ? R.bl : 0.0f) : 0.0f;
    const float xr = from_right_lane(R.x[0], 0.0f);                          // p[last slot + 1]
    float p0 = 0.f;
    if constexpr (!PARABOLIC) p0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, R.x[0])));
    if constexpr (PARABOLIC && NEUMANN) {
      // parabolic.py:148-150: previous row's neighbour u[t-1][-2]
      if constexpr (M64) bval = boundary_value<true>(P, a, a64, slot_get<EPL>(R.x, ns - 2), true);
      else bval = normalize_ctrl(cdx + slot_get<EPL>(R.x, ns - 2), P.max_control, P.normalize);
    }
    float y[EPL];
    if constexpr (FAST && PARABOLIC) {
      // Stage-major form of the same expression tree: the EPL independent chains advance together, so consecutive
      // instructions of a wave do not depend on each other (in-order issue stalls on back-to-back dependent VALU ops).
      // parabolic.py:143-144   u + F*(um - 2*u + up) + (dt*beta)*u
      float t2[EPL], t3[EPL], t4[EPL], t5[EPL], t7[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        // pm - 2p: 2p is exact, so fma(-2, p, pm) == pm + (-2p).  The first slot takes the product form so that the
        // lane shift folds into the add (v_add_f32_dpp); VOP3 fma cannot carry a DPP operand.
        // ROLL (the T-steps-per-launch kernels, whose waves drift apart) with EPL > 1: the two DPP operands are written out with
        // an s_nop 0 ahead of each (the wave yields its issue slot before the DPP operation: open-loop rollout 368 -> 350 us per
        // 25 env-steps; in the lock-step per-env-step launch the same form is 2 % slower, so it stays on the compiler's).  Each
        // statement opens with a plain VALU instruction of its own, so that together with the s_nop two wait states separate
        // the DPP read from whatever the compiler scheduled in front (VALU write -> DPP read hazard).
        if (ROLL && EPL > 1 && e == 0 && !GENERAL_EDGE) {
          asm volatile("v_add_f32 %0, %2, %2\n\ts_nop 0\n\tv_sub_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                       : "=&v"(t2[0]) : "v"(R.x[EPL - 1]), "v"(R.x[0]));
        } else if (ROLL && EPL > 1 && e == EPL - 1 && !GENERAL_EDGE) {
          t2[e] = 0.f;     // formed inside the statement below
        } else if (e == 0 && !GENERAL_EDGE) t2[e] = xl + (-2.0f * R.x[e]);
        else t2[e] = __builtin_fmaf(-2.0f, R.x[e], (e == 0) ? xl : R.x[e - 1]);
      }
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        if (ROLL && EPL > 1 && e == EPL - 1 && !GENERAL_EDGE)
          asm volatile("v_fma_f32 %0, -2.0, %2, %3\n\ts_nop 0\n\tv_add_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                       : "=&v"(t3[e]) : "v"(R.x[0]), "v"(R.x[e]), "v"(R.x[e > 0 ? e - 1 : 0]));
        else
          t3[e] = t2[e] + ((e == EPL - 1) ? xr : R.x[e + 1]);
      }
#pragma unroll
      for (int e = 0; e < EPL; ++e) t7[e] = c[e] * R.x[e];
#pragma unroll
      for (int e = 0; e < EPL; ++e) t4[e] = fe[e] * t3[e];
#pragma unroll
      for (int e = 0; e < EPL; ++e) t5[e] = R.x[e] + t4[e];
#pragma unroll
      for (int e = 0; e < EPL; ++e) y[e] = t5[e] + t7[e];
    } else if constexpr (FAST && !PARABOLIC) {
      // hyperbolic.py:146-155   u + dt*((up - u)/dx + u[0]*beta), stage-major like the parabolic form above.
      // Quotient: RN32(RN64(d1 * RN64(1/dx))) == RN32(d1/dx): the double product is within 2^-52 (relative) of the
      // true quotient, while a quotient of two 24-bit floats is never closer than 2^-50 to a float rounding boundary
      // (checked on the device by pdegym_selftest_quotient).
      float d1[EPL], d2[EPL], r[EPL], d3[EPL], d4[EPL];
      double q[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) d1[e] = ((e == EPL - 1) ? xr : R.x[e + 1]) - R.x[e];
#pragma unroll
      for (int e = 0; e < EPL; ++e) r[e] = p0 * c[e];
      if constexpr (POW2_DX) {
        // dx = 2^k: d/dx == d * 2^-k exactly (same rounding into the denormal range, same overflow), one multiply
#pragma unroll
        for (int e = 0; e < EPL; ++e) d2[e] = d1[e] * rdxf;
        (void)q;
      } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[e] = (double)d1[e];
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[e] = q[e] * P.rdx;
#pragma unroll
        for (int e = 0; e < EPL; ++e) d2[e] = (float)q[e];
      }
      if constexpr (BURGERS) {   // extension: u_t = u u_x + beta(x) u(0,t)  ->  p*((pp - p)/dx) replaces (pp - p)/dx
#pragma unroll
        for (int e = 0; e < EPL; ++e) d2[e] = R.x[e] * d2[e];
      }
#pragma unroll
      for (int e = 0; e < EPL; ++e) d3[e] = d2[e] + r[e];
#pragma unroll
      for (int e = 0; e < EPL; ++e) d4[e] = fe[e] * d3[e];
#pragma unroll
      for (int e = 0; e < EPL; ++e) y[e] = R.x[e] + d4[e];
      (void)xl;
    } else {
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const float p = R.x[e];
      const float pm = (e == 0) ? xl : R.x[e - 1];
      const float pp = (e == EPL - 1) ? xr : R.x[e + 1];
      float v;
      if constexpr (PARABOLIC) {
        // parabolic.py:143-144   u + F*(um - 2*u + up) + (dt*beta)*u
        float t2;
        if constexpr (FAST) {
          t2 = __builtin_fmaf(-2.0f, p, pm);  // == pm - RN(2p): 2p is exact unless it overflows (-> EXACT redo)
        } else {
          const float t1 = 2.0f * p;
          t2 = pm - t1;
        }
        const float t3 = t2 + pp;
        const float t4 = fe[e] * t3;
        const float t5 = p + t4;
        if (M64 && beta64) {          // float64 beta: (dt*beta)*u and the final sum are double, rounded once
          v = (float)((double)t5 + c64[M64 ? e : 0] * (double)p);
        } else {
          const float t7 = c[e] * p;
          v = t5 + t7;
        }
      } else {
        // hyperbolic.py:146-155   u + dt*((up - u)/dx + u[0]*beta)
        const float d1 = pp - p;
        float d2;
        if constexpr (FAST) {
          // RN32(RN64(d1 * RN64(1/dx))) == RN32(d1/dx): the double product is within 2^-52 (relative) of the true
          // quotient, while a quotient of two 24-bit floats is never closer than 2^-50 to a float rounding boundary.
          d2 = (float)((double)d1 * P.rdx);
        } else {
          d2 = d1 / dx;
        }
        if (M64 && beta64) {          // float64 beta: u[0]*beta, the sum, dt*(...) and u + ... are double, rounded once
          const double r = (double)p0 * c64[M64 ? e : 0];
          const double d3 = (double)(BURGERS ? p * d2 : d2) + r;
          v = (float)((double)p + P.dt64 * d3);
        } else {
          const float r = p0 * c[e];
          const float d3 = (BURGERS ? p * d2 : d2) + r;
          const float d4 = fe[e] * d3;
          v = p + d4;
        }
        (void)pm;
      }
      if constexpr (!FAST) {
        const int sl = s0 + e;
        v = (sl >= ns - 1) ? ((sl == ns - 1) ? bval : 0.0f) : v;  // controlled boundary node; padding stays 0
      }
      y[e] = v;
    }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) R.x[e] = y[e];
    if constexpr (FAST && GENERAL_EDGE) {  // the frozen boundary slot takes the new control value once (parabolic.py:148-150)
#pragma unroll
      for (int e = 0; e < EPL; ++e) R.x[e] = (s0 + e == ns - 1) ? bval : R.x[e];
    }
    R.bl = 0.0f;  // parabolic.py:146  u(0,t) = 0
  };
  // rows whose norm a later reward call looks back at (tuned_reward_1d.py:40): r+100 is a step end
  auto record_norm = [&](int done) {
    if (done < nsub && (rec_all || R.k == 0 || R.t + PDEGYM_LOOKBACK == P.nt - 1)) {
      const float nr = sqrtf(slots_sumsq<EPL>(R.x, s0, ns));  // node 0 is 0 after any sub-step
      ring.put(R.t & (PDEGYM_RING - 1), nr, lane);
      if (R.t == R.back_row) R.back_norm = nr;                // this call's own look-back row: no memory round trip
    }
  };
  if constexpr (FAST) {
    // Every instruction of the loop -- scalar bookkeeping included -- takes an issue slot of the SIMD (about one per
    // 2.8 cycles with four resident waves), so the sub-steps between two norm records run in a bare inner loop and the
    // time index / phase counters advance once per run.
    int s = 0;
    // HIST: row R.t of the trajectory after every sub-step (frozen boundary slot = the commanded value, node 0 = 0: the row
    // as the reference stores it); the stores are fire-and-forget, nothing in the loop waits for them
    // `hist` is wave-uniform (step1d_body forms it from a scalar instance index): the row address is a scalar base, advanced by one
    // scalar add per sub-step, + the lane's slot offset.  A lane whose EPL slots all lie inside the row stores them as one vector; the
    // at most one lane that straddles the row's end stores slot by slot -- and only when there is such a lane (hoisted, wave-uniform).
    float* hrow = nullptr;
    const bool hfull = s0 + EPL <= ns, hpart = !hfull && s0 < ns;
    bool any_part = false;
    if constexpr (HIST) {
      if (hist) {
        hrow = hist + (size_t)R.t * n + J0;
        any_part = __builtin_amdgcn_ballot_w64(hpart) != 0ull;
      }
    }
    // PART: some lane straddles the row's end (decided once per call: the loops below exist in both forms, so the common form has no
    // test and no skipped block inside).  Node 0 of a parabolic row is not stored: it is 0 after any sub-step and every row >= 1 of the
    // trajectory is zero-filled by the reset (reset_history_kernel, the fused auto-reset), so the word already holds it.
    auto store_row = [&](auto part_tag) {
      if constexpr (HIST) {
        hrow += n;
        if (hfull) {
          float* q = hrow + s0;
          constexpr int Q = EPL / 4 * 4, D = (EPL - Q) / 2 * 2;
#pragma unroll
          for (int e = 0; e < Q; e += 4) *reinterpret_cast<pdegym_f4u*>(q + e) = pdegym_f4u{R.x[e], R.x[e + 1], R.x[e + 2], R.x[e + 3]};
          if constexpr (D == 2) *reinterpret_cast<pdegym_f2u*>(q + Q) = pdegym_f2u{R.x[Q], R.x[Q + 1]};
          if constexpr (Q + D < EPL) q[EPL - 1] = R.x[EPL - 1];
        }
        if constexpr (decltype(part_tag)::value) {
          if (hpart) {
#pragma unroll
            for (int e = 0; e < EPL; ++e)
              if (s0 + e < ns) hrow[s0 + e] = R.x[e];
          }
        }
      }
    };
    auto run_of = [&](int run, auto pow2_tag, auto part_tag) {
      for (int i = 0; i < run; ++i) {
        pde_substep(std::false_type{}, pow2_tag);
        store_row(part_tag);
      }
    };
    if (nsub > 0) {
      pde_substep(std::true_type{}, std::false_type{});
      ++R.t;
      R.k = (R.k + 1 == S) ? 0 : R.k + 1;
      s = 1;
      if (HIST && hrow) store_row(std::true_type{});
      record_norm(s);
    }
    while (s < nsub) {
      int run = nsub - s;
      if (rec_all) {
        run = 1;
      } else {
        const int to_phase0 = S - R.k;                                   // sub-steps until R.k wraps to 0
        const int to_lookback = (P.nt - 1 - PDEGYM_LOOKBACK) - R.t;      // ... until R.t + LOOKBACK == nt - 1
        run = run < to_phase0 ? run : to_phase0;
        if (to_lookback > 0) run = run < to_lookback ? run : to_lookback;
      }
      if constexpr (HIST) {
        if (hrow == nullptr) {        // (never with HFAST: the launcher takes it only with a trajectory buffer)
          if (pow2_dx) for (int i = 0; i < run; ++i) pde_substep(std::false_type{}, std::true_type{});
          else for (int i = 0; i < run; ++i) pde_substep(std::false_type{}, std::false_type{});
        } else if (pow2_dx) {
          if (any_part) run_of(run, std::true_type{}, std::true_type{});
          else run_of(run, std::true_type{}, std::false_type{});
        } else {
          if (any_part) run_of(run, std::false_type{}, std::true_type{});
          else run_of(run, std::false_type{}, std::false_type{});
        }
      } else {
        if (pow2_dx) for (int i = 0; i < run; ++i) pde_substep(std::false_type{}, std::true_type{});
        else for (int i = 0; i < run; ++i) pde_substep(std::false_type{}, std::false_type{});
      }
      R.t += run;
      R.k += run;
      if (R.k >= S) R.k -= S;
      s += run;
      record_norm(s);
    }
  } else {
    for (int s = 0; s < nsub; ++s) {
      if (xprev && s == nsub - 1) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) xprev[e] = R.x[e];
        *blprev = R.bl;
      }
      pde_substep(std::true_type{}, std::false_type{});
      ++R.t;
      R.k = (R.k + 1 == S) ? 0 : R.k + 1;
      if constexpr (NEUMANN) R.bsum += (double)fabsf(bval);
      if constexpr (HIST) {
        if (hist) {     // HIST without a buffer: the select-form kernel taken for the "differential" reward
          float* hrow = hist + (size_t)R.t * n;
          if (PARABOLIC && lane == 0) hrow[0] = 0.0f;
#pragma unroll
          for (int e = 0; e < EPL; ++e)
            if (s0 + e < ns) hrow[J0 + s0 + e] = R.x[e];
        }
      }
      record_norm(s + 1);
      if (thor_k > 0 && s + 1 < nsub && nsub - (s + 1) < thor_k) {    // wave-uniform; after record_norm: this slot holds the reward's norm
        const float nk = kind_norm<EPL>(R.x, R.bl, s0, ns, P.reward_kind);
        ring.put(R.t & (PDEGYM_RING - 1), nk, lane);
      }
    }
  }
  if constexpr (!NEUMANN) R.bsum += (double)nsub * (double)fabsf(bval);
}

template <int EPL, bool PARABOLIC>
__device__ __forceinline__ void load_row(Row<EPL>& R, float (&beta)[EPL], const float* urow, const float* brow, const int n,
                                         int lane) {
  constexpr int J0 = PARABOLIC ? 1 : 0;
  const int ns = n - J0, s0 = lane * EPL;
  load_slots<EPL>(urow + J0, R.x, s0, ns);
  load_slots<EPL>(brow + J0, beta, s0, ns);
  R.bl = PARABOLIC ? urow[0] : 0.f;
}

// One env-step of one instance by one wave: the body of step1d_kernel, and of every iteration of rollout1d_kernel.
// CARRY (rollout1d_kernel): the state enters and leaves through *carry instead of memory -- no row / beta / time-index / sum /
// ring loads at the head of the step.  What a step stores is what a caller can see of it afterwards: observation slot t + 1 and
// row t of the reward / flag arrays always; the per-instance state words that every step would overwrite (norm_now, norm_back,
// time_index, bsum -- and the ring, kept in registers) only when store_state says this is the launch's last step.
// HFAST (round 6, with HIST): the fast sub-step loop storing every row into the trajectory buffer (bufs.history must be given; the
// launcher keeps the NormReward "differential" / "t-horizon" requests on the select form).
template <int EPL, bool PARABOLIC, bool NEUMANN, bool HIST, bool BURGERS = false, bool M64 = false, bool ROLL = false,
          bool CARRY = false, bool FULL = false, bool HFAST = false>
__device__ __forceinline__ void step1d_body(const pdegym_params1d& P, const pdegym_bufs1d& Bf, const int B, const int inst,
                                            const int lane, const float* command = nullptr, Carry<EPL>* carry = nullptr,
                                            const bool store_state = true) {
  constexpr int J0 = PARABOLIC ? 1 : 0;
  static_assert(!HFAST || HIST, "HFAST is a history mode");
  constexpr bool kFast = !NEUMANN && !M64 && (!HIST || HFAST);
  static_assert(!CARRY || kFast, "the carried state is the float32 Dirichlet rollout path");
#ifdef PDEGYM_TIMING
  const unsigned long long tm0 = __builtin_amdgcn_s_memtime();
  const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
#endif
  static_assert(!FULL || kFast, "FULL rows: the float32 Dirichlet fast path only");
  const int n = FULL ? kWave * EPL + J0 : P.n, ns = n - J0, s0 = lane * EPL;
#ifdef PDEGYM_TIMING
  const unsigned long long tmk = __builtin_amdgcn_s_memtime() + (unsigned long long)(n == 0x7fffffff);  // kernarg arrived
#endif
  // state_in given: the row comes from the previous call's observation and goes to obs only (include/pdegym.h)
  const float* urow_in = (Bf.state_in ? Bf.state_in : Bf.u) + (size_t)inst * n;
  float* urow = Bf.state_in ? nullptr : Bf.u + (size_t)inst * n;
  const bool beta64 = M64 && P.beta_f64;
  // float32 beta row; in the mixed-precision mode with a float64 beta it is read as double below (beta then stays zero)
  const float* brow = beta64 ? urow_in : static_cast<const float*>(Bf.beta) + (size_t)inst * Bf.beta_stride;
  float* const ring_mem = Bf.ring + (size_t)inst * PDEGYM_RING;
  using RingT = std::conditional_t<CARRY, RingReg, RingMem>;
  RingT ring;
  if constexpr (CARRY) ring = carry->ring;
  else ring.mem = ring_mem;
  // (HFAST: a scalar instance index, so that the trajectory pointer is wave-uniform -- scalar base + lane offset addressing in the loop)
  float* hist = (HIST && Bf.history) ? Bf.history + (size_t)(HFAST ? __builtin_amdgcn_readfirstlane(inst) : inst) * P.nt * n : nullptr;

  Row<EPL> R;
  float beta[EPL];
  if constexpr (CARRY) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      R.x[e] = carry->x[e];
      beta[e] = carry->beta[e];
    }
    R.bl = carry->bl;
  } else {
    load_row<EPL, PARABOLIC>(R, beta, urow_in, brow, n, lane);
  }
  double b64[M64 ? EPL : 1];
  if constexpr (M64) {
    const double* brow64 = static_cast<const double*>(Bf.beta) + (size_t)inst * Bf.beta_stride;
#pragma unroll
    for (int e = 0; e < EPL; ++e) b64[e] = (beta64 && s0 + e < ns) ? brow64[J0 + s0 + e] : 0.0;
  }
  int t_in;
  double bsum_in;
  if constexpr (CARRY) {
    t_in = __builtin_amdgcn_readfirstlane(carry->t);     // wave-uniform by construction; keeps the loop counters scalar
    bsum_in = carry->bsum;
  } else {
    t_in = __builtin_amdgcn_readfirstlane(Bf.time_index[inst]);
    bsum_in = Bf.bsum[inst];
  }
  const int S = P.substeps > 0 ? P.substeps : 1;
  int nsub = P.nt - 1 - t_in;  // hyperbolic.py:140: while i < sample_rate and time_index < nt-1
  nsub = nsub < P.substeps ? nsub : P.substeps;
  nsub = nsub > 0 ? nsub : 0;
  const bool act64 = M64 && P.action_kind != PDEGYM_ACTION_F32;
  // command: the control input computed inside the launch (rollout kernel with its policy) instead of bufs.action
  const float a = command ? *command : (act64 ? 0.f : static_cast<const float*>(Bf.action)[inst]);
  const double a64 = act64 ? static_cast<const double*>(Bf.action)[inst] : 0.0;
  R.t = t_in;
  if constexpr (CARRY) R.k = __builtin_amdgcn_readfirstlane(carry->k);
  else R.k = (t_in + PDEGYM_LOOKBACK) % S;
  R.bsum = bsum_in;
  // look-back row of this call's reward (tuned_reward_1d.py:40): t_end - 100, Python negative index wraps into the
  // zero-filled tail of the history.  Rows that predate this call are fetched NOW (latency hidden by the loop).
  const int t_end = t_in + nsub;
  const int tb = t_end - PDEGYM_LOOKBACK;
  const int src_row = tb < 0 ? P.nt + tb : tb;
  const bool zero_row = (tb < 0 && src_row > t_end) || src_row < 0;
  const bool from_ring = !zero_row && src_row <= t_in;
  float norm_back_pre = 0.f;
  if (from_ring) norm_back_pre = ring.get(src_row & (PDEGYM_RING - 1), lane);
  R.back_row = (!zero_row && !from_ring) ? src_row : -1;
  R.back_norm = 0.f;

#ifdef PDEGYM_TIMING
  const unsigned long long tm1 = __builtin_amdgcn_s_memtime() + (unsigned long long)(R.x[0] != R.x[0]);
#endif
  // NormReward "differential" (norm_reward.py:55-59): ||u[t] - u[t-1]|| over FINE rows, so the row before the last sub-step is
  // kept.  Only the select-form instantiations evaluate it (launch_epl routes the request there; rollouts refuse it).
  const bool differential = !kFast && P.reward_horizon == PDEGYM_HORIZON_DIFFERENTIAL && P.reward_kind >= PDEGYM_REWARD_NORM_L1;
  float xprev[kFast ? 1 : EPL], blprev = 0.f;
#pragma unroll
  for (int e = 0; e < (kFast ? 1 : EPL); ++e) xprev[e] = 0.f;
  // NormReward "t-horizon" (norm_reward.py:60-73): the ring holds the reward's own norm of the rows the mean looks back at --
  // the row this call starts from (recorded here: independent of how the previous call or the reset left the slot), the last
  // rows of the sub-step loop, the final row (epilogue).  Select-form instantiations only, like "differential".
  const int thor_k = (!kFast && P.reward_horizon == PDEGYM_HORIZON_T && P.reward_kind >= PDEGYM_REWARD_NORM_L1) ? P.reward_t_horizon : 0;
  if constexpr (!kFast) {
    if (thor_k > 0) {
      const float nk0 = kind_norm<EPL>(R.x, R.bl, s0, ns, P.reward_kind);
      ring.put(t_in & (PDEGYM_RING - 1), nk0, lane);
    }
  }
  float norm_now;
  if constexpr (kFast) {
    // The fast loop freezes the controlled boundary slot with zero coefficients: x + 0*t keeps every x except -0.0
    // (-0.0 + +0.0 = +0.0), so a commanded boundary value of exactly -0.0 takes the exact loop (wave-uniform test).
    bool exact = __float_as_uint(normalize_ctrl(a, P.max_control, P.normalize)) == 0x80000000u;
    if constexpr (PARABOLIC) {
      // The fast stencil forms um - 2u as ONE fma, which stays finite where the reference's 2*u overflows (|u| >= 2^127)
      // and could then decay back to a finite row that the non-finite test below never sees.  Rule it out up front: for
      // 0 <= F <= 1/2 the diffusion part is a convex combination (max-norm contraction), so one sub-step grows max|u| by at
      // most g = 1 + max|dt*beta| (+ rounding); otherwise by 1 + 4|F| + max|dt*beta|.  If max|u| * g^nsub can reach 2^126 the
      // exact loop runs instead (wave-uniform; costs two wave reductions per launch).
      float mx = fmaxf(fabsf(R.bl), fabsf(normalize_ctrl(a, P.max_control, P.normalize))), cm = 0.f;
      if constexpr (CARRY) {
        // no reductions here: max|dt*beta| is carried, and max|u| <= ||u||_2, the norm the previous step ended with.  That norm
        // is a float sum of squares: any |u[j]| >= 2^-60 has a normal square, so norm*(1 + 2^-10) bounds it; smaller rows are
        // covered by the 2^-60 floor.  A non-finite norm (NaN would be dropped by fmaxf) takes the exact loop.
        mx = fmaxf(fabsf(normalize_ctrl(a, P.max_control, P.normalize)), fmaxf(carry->norm * 1.0009765625f, 8.673617379884035e-19f));
        cm = carry->cm;
        exact = exact || !(carry->norm <= 3.4028234663852886e38f);
      } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          mx = fmaxf(mx, fabsf(R.x[e]));
          cm = fmaxf(cm, fabsf(P.dt * beta[e]));
        }
        mx = wave_max(mx);
        cm = wave_max(cm);
      }
      if constexpr (CARRY) {
        // log2 g is carried, and log2 mx is only formed when the bound could matter: mx <= 2^40 and nsub log2 g <= 80 give
        // log2 mx + nsub log2 g <= 120 < 126 in the formula below as well -- the same decision, two v_log_f32 fewer per env-step
        const float budget = (float)nsub * carry->glog;
        if (!(mx <= 1.099511627776e12f && budget <= 80.0f)) exact = exact || !(__log2f(mx) + budget < 126.0f);
        (void)cm;
      } else {
        exact = exact || !(__log2f(mx) + (float)nsub * __log2f(growth_bound(P, cm)) < 126.0f);   // NaN / inf anywhere -> exact
      }
    }
    norm_now = 0.f;
    if (!exact) {
      run_substeps<EPL, PARABOLIC, false, true, HFAST, BURGERS, false, ROLL, RingT, FULL>(R, beta, P, nsub, a, ring, hist, lane);
      norm_now = sqrtf(slots_sumsq<EPL>(R.x, s0, ns) + R.bl * R.bl);
      // inf/NaN somewhere (or a squared overflow): 0*inf may have leaked into a frozen slot -> redo exactly
      exact = !(fabsf(norm_now) <= 3.4028234663852886e38f);
      if (exact) {
        if constexpr (CARRY) {
          // the carried input was overwritten: observation slot t holds the row (each lane re-reads the slots it stored),
          // beta may have been redrawn by another lane's stores of an earlier auto-reset -> order them first
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        load_row<EPL, PARABOLIC>(R, beta, urow_in, brow, n, lane);
        if constexpr (CARRY) {
          R.bl = carry->bl;
          drain_vmem();
        }
        R.t = t_in;
        R.k = (t_in + PDEGYM_LOOKBACK) % S;
        R.bsum = bsum_in;
        R.back_norm = 0.f;
      }
    }
    if (exact) {
      // (HFAST: the exact loop rewrites the same trajectory rows)
      run_substeps<EPL, PARABOLIC, false, false, HFAST, BURGERS, false, false, RingT, FULL>(R, beta, P, nsub, a, ring, hist, lane);
      norm_now = sqrtf(slots_sumsq<EPL>(R.x, s0, ns) + R.bl * R.bl);
    }
  } else {
    run_substeps<EPL, PARABOLIC, NEUMANN, false, HIST, BURGERS, M64>(R, beta, P, nsub, a, ring, hist, lane, b64, a64,
                                                                     differential ? xprev : nullptr, &blprev, thor_k);
    norm_now = sqrtf(slots_sumsq<EPL>(R.x, s0, ns) + R.bl * R.bl);
  }
  const int t = R.t;
#ifdef PDEGYM_TIMING
  const unsigned long long tm2 = __builtin_amdgcn_s_memtime() + (unsigned long long)(norm_now != norm_now);
#endif

  // ---- epilogue: norms, flags, reward, observation ------------------------------------------------
  const bool rec_all = P.nt <= PDEGYM_RING;
  if (nsub > 0 && (rec_all || R.k == 0 || t + PDEGYM_LOOKBACK == P.nt - 1)) {
    ring.put(t & (PDEGYM_RING - 1), norm_now, lane);
  }
  const bool terminate = t >= P.nt - 1;                                 // hyperbolic.py:171-180
  const bool truncate = P.limit_state && (norm_now >= P.max_state);     // hyperbolic.py:182-194
  // NormReward variants need wave-wide reductions: do them before the single-lane tail
  float nr_alt = norm_now;
  bool nr_diff = false;
  if constexpr (!kFast) {
    if (differential && t > 0 && nsub > 0) {    // the row minus the one before it; the sign flips (norm_reward.py:56-58)
      nr_diff = true;
      const float d0 = fabsf(R.bl - blprev);      // node 0 is wave-uniform: joined after the reduction
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const float d = (s0 + e < ns) ? fabsf(R.x[e] - xprev[e]) : 0.f;
        if (P.reward_kind == PDEGYM_REWARD_NORM_L1) acc += d;
        else if (P.reward_kind == PDEGYM_REWARD_NORM_L2) acc += d * d;
        else acc = mag_max(acc, d);
      }
      if (P.reward_kind == PDEGYM_REWARD_NORM_L1) nr_alt = wave_sum(acc) + d0;
      else if (P.reward_kind == PDEGYM_REWARD_NORM_L2) nr_alt = sqrtf(wave_sum(acc) + d0 * d0);
      else nr_alt = mag_max(wave_mag_max(acc), d0);
    }
  }
  if (!nr_diff && P.reward_kind == PDEGYM_REWARD_NORM_L1) {
    float s1 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s1 += (s0 + e < ns) ? fabsf(R.x[e]) : 0.f;
    nr_alt = wave_sum(s1) + fabsf(R.bl);
  } else if (!nr_diff && P.reward_kind == PDEGYM_REWARD_NORM_LINF) {
    float m = fabsf(R.bl);
#pragma unroll
    for (int e = 0; e < EPL; ++e) m = mag_max(m, (s0 + e < ns) ? fabsf(R.x[e]) : 0.f);
    nr_alt = wave_mag_max(m);
  }
  float thor_mean = 0.f;
  if constexpr (!kFast) {
    if (thor_k > 0 && lane == 0) {      // -sum(norm(u[t - i]) for i in range(k)) / k, k = min(t_horizon_length, t + 1): one float32 chain
      if (nsub > 0) ring.mem[t & (PDEGYM_RING - 1)] = nr_alt;
      const int kk = thor_k < t + 1 ? thor_k : t + 1;
      float acc = nr_alt;
      for (int i = 1; i < kk; ++i) acc += ring.mem[(t - i) & (PDEGYM_RING - 1)];
      thor_mean = acc / (float)kk;
    }
  }
  // look-back norm: fetched before the loop, captured inside it, the final row itself (nsub == 100 ends on it only
  // when LOOKBACK == 0, never), or 0 for an unwritten row
  float norm_back = 0.f;
  float reward = 0.f;
  if (lane == 0) {
    norm_back = from_ring ? norm_back_pre : ((R.back_row >= 0) ? ((R.back_row == t) ? norm_now : R.back_norm) : 0.f);
    if (P.reward_kind == PDEGYM_REWARD_TUNED1D) {
      if (terminate && norm_now < 20.0f) {
        reward = (P.terminate_reward - ((float)R.bsum) / 1000.0f) - norm_now;  // tuned_reward_1d.py:36-37
      } else if (truncate) {
        reward = (float)((double)P.truncate_penalty * (double)(P.reward_nt - t));  // tuned_reward_1d.py:38-39
      } else {
        reward = norm_back - norm_now;  // tuned_reward_1d.py:40
      }
    } else if (P.reward_kind >= PDEGYM_REWARD_NORM_L1) {
      // documented intent of norm_reward.py:48-54 ("temporal" horizon)
      reward = terminate ? P.terminate_reward
                         : (truncate ? (float)((double)P.truncate_penalty * (double)(P.reward_nt - t))
                                     : (nr_diff ? nr_alt : (thor_k > 0 ? -thor_mean : -nr_alt)));
    }
  }

  // sensing_update (hyperbolic.py:72-116)
  const bool auto_reset = (Bf.reset_init != nullptr) && (terminate || truncate);  // wave-uniform
  auto node = [&](int j) -> float { return (PARABOLIC && j == 0) ? R.bl : slot_get<EPL>(R.x, j - J0); };
  auto emit_obs = [&](float* obs_base) {
    if (P.sensing == PDEGYM_SENSE_FULL) {
      float* orow = obs_base + (size_t)inst * n;
      if (PARABOLIC && lane == 0) orow[0] = R.bl;
      store_slots<EPL>(orow + J0, R.x, s0, ns);
    } else {
      float o;
      if (P.sensing == PDEGYM_SENSE_LAST) o = node(n - 1);
      else if (P.sensing == PDEGYM_SENSE_LAST_DERIV) o = (node(n - 1) - node(n - 2)) / P.dx;
      else if (P.sensing == PDEGYM_SENSE_FIRST_DERIV) o = (node(1) - node(0)) / P.dx;
      else o = node(0);
      if (lane == 0) obs_base[inst] = o;
    }
  };
  if (lane == 0) {
    if (P.reward_kind != PDEGYM_REWARD_NONE) Bf.reward[inst] = reward;
    if (store_state) {
      Bf.norm_now[inst] = norm_now;
      Bf.norm_back[inst] = norm_back;
    }
    Bf.terminated[inst] = terminate ? 1 : 0;
    Bf.truncated[inst] = truncate ? 1 : 0;
  }
  if (!auto_reset) {
    if (nsub > 0 && urow) {
      if (PARABOLIC && lane == 0) urow[0] = R.bl;
      store_slots<EPL>(urow + J0, R.x, s0, ns);
    }
    emit_obs(Bf.obs);
    if (lane == 0 && store_state) {
      Bf.time_index[inst] = t;
      Bf.bsum[inst] = R.bsum;
    }
    if constexpr (CARRY) {
      carry->t = t;
      carry->k = R.k;
      carry->bsum = R.bsum;
      carry->norm = norm_now;
    }
  } else {
    // fused VecEnv auto-reset: keep the terminal observation, restart from the pool row (hyperbolic.py:214-227)
    if (Bf.final_obs) emit_obs(Bf.final_obs);
    const int prow = pool_row(Bf, inst, B);
    const float* irow = Bf.reset_init + (size_t)prow * n;
    if (Bf.reset_beta && Bf.beta_stride != 0) {      // the reference redraws beta at every reset (hyperbolic.py:208)
      if (beta64) {
        double* bdst = const_cast<double*>(static_cast<const double*>(Bf.beta)) + (size_t)inst * Bf.beta_stride;
        const double* bsrc = static_cast<const double*>(Bf.reset_beta) + (size_t)prow * n;
        for (int j = lane; j < n; j += kWave) bdst[j] = bsrc[j];
      } else {
        float* bdst = const_cast<float*>(brow);
        const float* bsrc = static_cast<const float*>(Bf.reset_beta) + (size_t)prow * n;
        for (int j = lane; j < n; j += kWave) bdst[j] = bsrc[j];
        if constexpr (CARRY) {      // the carried copy follows the redraw
#pragma unroll
          for (int e = 0; e < EPL; ++e) carry->beta[e] = (s0 + e < ns) ? bsrc[J0 + s0 + e] : 0.f;
          carry_refresh_beta<EPL, PARABOLIC>(*carry, P, s0, ns);
        }
      }
    }
    if (Bf.reset_count && lane == 0) Bf.reset_count[inst] += 1;
    R.bl = PARABOLIC ? irow[0] : 0.f;
    if (PARABOLIC && lane == 0 && urow) urow[0] = R.bl;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      R.x[e] = (s0 + e < ns) ? irow[J0 + s0 + e] : 0.f;
      if (s0 + e < ns && urow) urow[J0 + s0 + e] = R.x[e];
    }
    if constexpr (HIST) {
      if (hist)
        for (size_t q = lane; q < (size_t)P.nt * n; q += kWave) hist[q] = (q < (size_t)n) ? irow[q] : 0.f;
    }
    const float n0 = sqrtf(slots_sumsq<EPL>(R.x, s0, ns) + R.bl * R.bl);
    const float last = node(n - 1);
    emit_obs(Bf.obs);
    if (lane == 0 && store_state) {
      Bf.time_index[inst] = 0;
      Bf.bsum[inst] = (double)fabsf(last);
    }
    ring.put(0, n0, lane);
    if constexpr (CARRY) {
      carry->t = 0;
      carry->k = PDEGYM_LOOKBACK % S;
      carry->bsum = (double)fabsf(last);
      carry->norm = n0;
      drain_vmem();
    }
  }
  if constexpr (CARRY) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) carry->x[e] = R.x[e];
    carry->bl = R.bl;
    carry->ring = ring;
  }
#ifdef PDEGYM_TIMING
  if (lane == 0) {
    const unsigned long long tm3 = __builtin_amdgcn_s_memtime();
    unsigned int* dbg = reinterpret_cast<unsigned int*>(ring_mem) + 116;
    dbg[0] = (unsigned int)tm0; dbg[1] = (unsigned int)(tm0 >> 32);
    dbg[2] = (unsigned int)(tm1 - tm0); dbg[3] = (unsigned int)(tm2 - tm1); dbg[4] = (unsigned int)(tm3 - tm2);
    dbg[5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
    dbg[6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
    const unsigned long long tr3 = __builtin_amdgcn_s_memrealtime();          // 100 MHz constant clock
    dbg[7] = (unsigned int)tr0; dbg[8] = (unsigned int)(tr3 - tr0);
    dbg[9] = (unsigned int)(tmk - tm0);
  }
#endif
}

// The state a rollout launch starts from, read ONCE: row (observation slot 0), beta, time index and |u[-1]| sum.
template <int EPL, bool PARABOLIC, bool FULL = false>
__device__ __forceinline__ void carry_load(Carry<EPL>& C, const pdegym_params1d& P, const pdegym_bufs1d& Bf, const float* row0, int inst,
                                           int lane) {
  constexpr int J0 = PARABOLIC ? 1 : 0;
  const int n = FULL ? kWave * EPL + J0 : P.n, ns = n - J0, s0 = lane * EPL;
  Row<EPL> R0;
  load_row<EPL, PARABOLIC>(R0, C.beta, row0 + (size_t)inst * n, static_cast<const float*>(Bf.beta) + (size_t)inst * Bf.beta_stride, n, lane);
#pragma unroll
  for (int e = 0; e < EPL; ++e) C.x[e] = R0.x[e];
  C.bl = R0.bl;
  carry_refresh_beta<EPL, PARABOLIC>(C, P, s0, ns);
  C.norm = sqrtf(slots_sumsq<EPL>(C.x, s0, ns) + C.bl * C.bl);
  C.t = __builtin_amdgcn_readfirstlane(Bf.time_index[inst]);
  C.k = (C.t + PDEGYM_LOOKBACK) % (P.substeps > 0 ? P.substeps : 1);
  C.bsum = Bf.bsum[inst];
  const float* ring = Bf.ring + (size_t)inst * PDEGYM_RING;
  C.ring.lo = ring[lane];
  C.ring.hi = ring[kWave + lane];
  drain_vmem();
}

// ... and the ring written back at the end of the launch (the other state words were stored by the last step, store_state).
template <int EPL>
__device__ __forceinline__ void carry_store_ring(const Carry<EPL>& C, const pdegym_bufs1d& Bf, int inst, int lane) {
  float* ring = Bf.ring + (size_t)inst * PDEGYM_RING;
  ring[lane] = C.ring.lo;
  ring[kWave + lane] = C.ring.hi;
}

}  // namespace
#endif
