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
//       reward     rewards/tuned_reward_1d.py:25-40 (streaming form)
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pdegym.h"
#include "pdegym_common.h"
#include "pdegym_policy.h"

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
  double bsum;
};

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
template <int EPL, bool PARABOLIC, bool NEUMANN, bool FAST, bool HIST, bool BURGERS = false, bool M64 = false, bool ROLL = false>
__device__ __forceinline__ void run_substeps(Row<EPL>& R, const float (&beta)[EPL], const pdegym_params1d& P, int nsub,
                                             float a, float* ring, float* hist, int lane, const double* b64 = nullptr, double a64 = 0.0,
                                             float* xprev = nullptr, float* blprev = nullptr, int thor_k = 0) {
  // xprev/blprev (select form only): the row BEFORE the last sub-step, for NormReward's "differential" horizon
  // thor_k (select form only): NormReward "t-horizon" of length thor_k -- the reward's norm of each of the last thor_k - 1 rows
  // before the final one goes into the ring (the final row's is the epilogue's)
  static_assert(!(FAST && (NEUMANN || HIST)), "fast mode is the Dirichlet, history-free path");
  static_assert(!(FAST && M64), "the mixed-precision mode uses the select form");
  constexpr int J0 = PARABOLIC ? 1 : 0;
  const int n = P.n, ns = n - J0, s0 = lane * EPL;
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
      if (lane == 0) ring[R.t & (PDEGYM_RING - 1)] = nr;
      if (R.t == R.back_row) R.back_norm = nr;                // this call's own look-back row: no memory round trip
    }
  };
  if constexpr (FAST) {
    // Every instruction of the loop -- scalar bookkeeping included -- takes an issue slot of the SIMD (about one per
    // 2.8 cycles with four resident waves), so the sub-steps between two norm records run in a bare inner loop and the
    // time index / phase counters advance once per run.
    int s = 0;
    if (nsub > 0) {
      pde_substep(std::true_type{}, std::false_type{});
      ++R.t;
      R.k = (R.k + 1 == S) ? 0 : R.k + 1;
      s = 1;
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
      if (pow2_dx) {
        for (int i = 0; i < run; ++i) pde_substep(std::false_type{}, std::true_type{});
      } else {
        for (int i = 0; i < run; ++i) pde_substep(std::false_type{}, std::false_type{});
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
        if (lane == 0) ring[R.t & (PDEGYM_RING - 1)] = nk;
      }
    }
  }
  if constexpr (!NEUMANN) R.bsum += (double)nsub * (double)fabsf(bval);
}

template <int EPL, bool PARABOLIC>
__device__ __forceinline__ void load_row(Row<EPL>& R, float (&beta)[EPL], const float* urow, const float* brow, int n,
                                         int lane) {
  constexpr int J0 = PARABOLIC ? 1 : 0;
  const int ns = n - J0, s0 = lane * EPL;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const bool ok = s0 + e < ns;
    R.x[e] = ok ? urow[J0 + s0 + e] : 0.f;
    beta[e] = ok ? brow[J0 + s0 + e] : 0.f;
  }
  R.bl = PARABOLIC ? urow[0] : 0.f;
}

// One env-step of one instance by one wave: the body of step1d_kernel, and of every iteration of rollout1d_kernel.
// CARRY (rollout1d_kernel): the state enters and leaves through *carry instead of memory -- no row / beta / time-index / sum
// loads at the head of the step; the stores stay (observation slot t + 1, scalars), nothing waits for them.
template <int EPL, bool PARABOLIC, bool NEUMANN, bool HIST, bool BURGERS = false, bool M64 = false, bool ROLL = false,
          bool CARRY = false>
__device__ __forceinline__ void step1d_body(const pdegym_params1d& P, const pdegym_bufs1d& Bf, const int B, const int inst,
                                            const int lane, const float* command = nullptr, Carry<EPL>* carry = nullptr) {
  constexpr int J0 = PARABOLIC ? 1 : 0;
  constexpr bool kFast = !NEUMANN && !HIST && !M64;
  static_assert(!CARRY || kFast, "the carried state is the float32 Dirichlet rollout path");
#ifdef PDEGYM_TIMING
  const unsigned long long tm0 = __builtin_amdgcn_s_memtime();
  const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int n = P.n, ns = n - J0, s0 = lane * EPL;
#ifdef PDEGYM_TIMING
  const unsigned long long tmk = __builtin_amdgcn_s_memtime() + (unsigned long long)(n == 0x7fffffff);  // kernarg arrived
#endif
  // state_in given: the row comes from the previous call's observation and goes to obs only (include/pdegym.h)
  const float* urow_in = (Bf.state_in ? Bf.state_in : Bf.u) + (size_t)inst * n;
  float* urow = Bf.state_in ? nullptr : Bf.u + (size_t)inst * n;
  const bool beta64 = M64 && P.beta_f64;
  // float32 beta row; in the mixed-precision mode with a float64 beta it is read as double below (beta then stays zero)
  const float* brow = beta64 ? urow_in : static_cast<const float*>(Bf.beta) + (size_t)inst * Bf.beta_stride;
  float* ring = Bf.ring + (size_t)inst * PDEGYM_RING;
  float* hist = (HIST && Bf.history) ? Bf.history + (size_t)inst * P.nt * n : nullptr;

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
  R.k = (t_in + PDEGYM_LOOKBACK) % S;
  R.bsum = bsum_in;
  // look-back row of this call's reward (tuned_reward_1d.py:40): t_end - 100, Python negative index wraps into the
  // zero-filled tail of the history.  Rows that predate this call are fetched NOW (latency hidden by the loop).
  const int t_end = t_in + nsub;
  const int tb = t_end - PDEGYM_LOOKBACK;
  const int src_row = tb < 0 ? P.nt + tb : tb;
  const bool zero_row = (tb < 0 && src_row > t_end) || src_row < 0;
  const bool from_ring = !zero_row && src_row <= t_in;
  float norm_back_pre = 0.f;
  if (from_ring && lane == 0) norm_back_pre = ring[src_row & (PDEGYM_RING - 1)];
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
      if (lane == 0) ring[t_in & (PDEGYM_RING - 1)] = nk0;
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
      const float g = ((P.F >= 0.0f && P.F <= 0.5f) ? 1.0f : 1.0f + 4.0f * fabsf(P.F)) + cm + 9.5367431640625e-7f;
      exact = exact || !(__log2f(mx) + (float)nsub * __log2f(g) < 126.0f);   // NaN / inf anywhere -> exact
    }
    norm_now = 0.f;
    if (!exact) {
      run_substeps<EPL, PARABOLIC, false, true, false, BURGERS, false, ROLL>(R, beta, P, nsub, a, ring, nullptr, lane);
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
        if constexpr (CARRY) R.bl = carry->bl;
        R.t = t_in;
        R.k = (t_in + PDEGYM_LOOKBACK) % S;
        R.bsum = bsum_in;
        R.back_norm = 0.f;
      }
    }
    if (exact) {
      run_substeps<EPL, PARABOLIC, false, false, false, BURGERS>(R, beta, P, nsub, a, ring, nullptr, lane);
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
    if (lane == 0) ring[t & (PDEGYM_RING - 1)] = norm_now;
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
      if (nsub > 0) ring[t & (PDEGYM_RING - 1)] = nr_alt;
      const int kk = thor_k < t + 1 ? thor_k : t + 1;
      float acc = nr_alt;
      for (int i = 1; i < kk; ++i) acc += ring[(t - i) & (PDEGYM_RING - 1)];
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
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        if (s0 + e < ns) orow[J0 + s0 + e] = R.x[e];
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
    Bf.norm_now[inst] = norm_now;
    Bf.norm_back[inst] = norm_back;
    Bf.terminated[inst] = terminate ? 1 : 0;
    Bf.truncated[inst] = truncate ? 1 : 0;
  }
  if (!auto_reset) {
    if (nsub > 0 && urow) {
      if (PARABOLIC && lane == 0) urow[0] = R.bl;
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        if (s0 + e < ns) urow[J0 + s0 + e] = R.x[e];
    }
    emit_obs(Bf.obs);
    if (lane == 0) {
      Bf.time_index[inst] = t;
      Bf.bsum[inst] = R.bsum;
    }
    if constexpr (CARRY) {
      carry->t = t;
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
          float cm = 0.f;
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            carry->beta[e] = (s0 + e < ns) ? bsrc[J0 + s0 + e] : 0.f;
            cm = fmaxf(cm, fabsf(P.dt * carry->beta[e]));
          }
          carry->cm = wave_max(cm);
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
    if (lane == 0) {
      Bf.time_index[inst] = 0;
      Bf.bsum[inst] = (double)fabsf(last);
      ring[0] = n0;
    }
    if constexpr (CARRY) {
      carry->t = 0;
      carry->bsum = (double)fabsf(last);
      carry->norm = n0;
    }
  }
  if constexpr (CARRY) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) carry->x[e] = R.x[e];
    carry->bl = R.bl;
  }
#ifdef PDEGYM_TIMING
  if (lane == 0) {
    const unsigned long long tm3 = __builtin_amdgcn_s_memtime();
    unsigned int* dbg = reinterpret_cast<unsigned int*>(ring) + 116;
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

template <int EPL, bool PARABOLIC, bool NEUMANN, bool HIST, bool BURGERS = false, bool M64 = false>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void step1d_kernel(pdegym_params1d P, pdegym_bufs1d Bf, int B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int inst = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (inst >= B) return;  // wave-uniform
  step1d_body<EPL, PARABOLIC, NEUMANN, HIST, BURGERS, M64>(P, Bf, B, inst, lane);
}

// The state a rollout launch starts from, read ONCE: row (observation slot 0), beta, time index and |u[-1]| sum.
template <int EPL, bool PARABOLIC>
__device__ __forceinline__ void carry_load(Carry<EPL>& C, const pdegym_params1d& P, const pdegym_bufs1d& Bf, const float* row0, int inst,
                                           int lane) {
  constexpr int J0 = PARABOLIC ? 1 : 0;
  const int ns = P.n - J0, s0 = lane * EPL;
  Row<EPL> R0;
  load_row<EPL, PARABOLIC>(R0, C.beta, row0 + (size_t)inst * P.n, static_cast<const float*>(Bf.beta) + (size_t)inst * Bf.beta_stride, P.n, lane);
  float cm = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    C.x[e] = R0.x[e];
    cm = fmaxf(cm, fabsf(P.dt * C.beta[e]));
  }
  C.bl = R0.bl;
  C.cm = wave_max(cm);
  C.norm = sqrtf(slots_sumsq<EPL>(C.x, s0, ns) + C.bl * C.bl);
  C.t = __builtin_amdgcn_readfirstlane(Bf.time_index[inst]);
  C.bsum = Bf.bsum[inst];
}

// T env-steps of one instance by one wave in ONE launch (pdegym_*_rollout): iteration t is the step kernel's body with the
// row written to observation slot t + 1, action / reward / flags taken from / written to row t of the [T, B] rollout arrays
// -- every value equals what T separate step calls produce, bit for bit.  What it removes is the kernel boundary between
// env-steps: no dispatch gap, no load phase (the state stays in registers, Carry), and the waves of a SIMD drift apart
// instead of finishing in two generations.
template <int EPL, bool PARABOLIC, bool BURGERS>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void rollout1d_kernel(pdegym_params1d P, pdegym_bufs1d Bf, pdegym_rollout1d Ro,
                                                                         int B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int inst = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (inst >= B) return;  // wave-uniform
  const size_t slot = (size_t)B * P.n;
  // The state stays in registers over the T env-steps (Carry); each step still stores its observation slot and scalars, but
  // no step waits for memory at its head -- the next command is fetched one step ahead, and the norm ring is read and written
  // by lane 0 alone (program order of one lane).
  Carry<EPL> C;
  carry_load<EPL, PARABOLIC>(C, P, Bf, Ro.obs, inst, lane);
  float a_next = Ro.actions[inst];
  for (int t = 0; t < Ro.T; ++t) {
    pdegym_bufs1d S = Bf;
    S.u = nullptr;
    S.history = nullptr;
    S.state_in = Ro.obs + (size_t)t * slot;
    S.obs = Ro.obs + (size_t)(t + 1) * slot;
    S.action = Ro.actions + (size_t)t * B;
    S.reward = Ro.rewards + (size_t)t * B;
    S.terminated = Ro.terminated + (size_t)t * B;
    S.truncated = Ro.truncated + (size_t)t * B;
    const float a = a_next;
    if (t + 1 < Ro.T) a_next = Ro.actions[(size_t)(t + 1) * B + inst];
    step1d_body<EPL, PARABOLIC, false, false, BURGERS, false, true, true>(P, S, B, inst, lane, &a, &C);
  }
}


// The general form (round 4): Neumann actuation and / or scalar sensing -- what the reference's control / sensing table offers
// beyond the Dirichlet / full-state corner (hyperbolic.py:66-124, parabolic.py:66-122).  Iteration t is the step kernel's body
// exactly as step1d_kernel instantiates it (select form for Neumann, the fast form otherwise), with the state going through
// memory between iterations: full-state sensing keeps it in the observation slots (slot t in, slot t + 1 out), scalar sensing in
// bufs.u (in place) while the observation slots receive the sensed value.  A wave re-reads what its own lanes stored (rows, time
// index, sums, ring), so iterations are separated by a workgroup-scope release / acquire pair, nothing more.
template <int EPL, bool PARABOLIC, bool NEUMANN, bool BURGERS>
__device__ __forceinline__ void rollout1d_general_step(const pdegym_params1d& P, const pdegym_bufs1d& Bf, const pdegym_rollout1d& Ro, int B,
                                                        int inst, int lane, int t, const float* command) {
  const bool full = P.sensing == PDEGYM_SENSE_FULL;
  const size_t slot = (size_t)B * (full ? P.n : 1);
  pdegym_bufs1d S = Bf;
  S.history = nullptr;
  if (full) {
    S.u = nullptr;
    S.state_in = Ro.obs + (size_t)t * slot;
  } else {
    S.state_in = nullptr;
  }
  S.obs = Ro.obs + (size_t)(t + 1) * slot;
  S.action = Ro.actions + (size_t)t * B;
  S.reward = Ro.rewards + (size_t)t * B;
  S.terminated = Ro.terminated + (size_t)t * B;
  S.truncated = Ro.truncated + (size_t)t * B;
  step1d_body<EPL, PARABOLIC, NEUMANN, false, BURGERS>(P, S, B, inst, lane, command);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int EPL, bool PARABOLIC, bool NEUMANN, bool BURGERS>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void rollout1d_general_kernel(pdegym_params1d P, pdegym_bufs1d Bf, pdegym_rollout1d Ro,
                                                                                 int B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int inst = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (inst >= B) return;  // wave-uniform
  for (int t = 0; t < Ro.T; ++t) rollout1d_general_step<EPL, PARABOLIC, NEUMANN, BURGERS>(P, Bf, Ro, B, inst, lane, t, nullptr);
}


// ================================================================================================
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
  else if (hist)
    hipLaunchKernelGGL((step1d_kernel<EPL, PARABOLIC, false, true, BURGERS>), grid, block, lds, st, P, Bf, B);
  else
    hipLaunchKernelGGL((step1d_kernel<EPL, PARABOLIC, false, false, BURGERS>), grid, block, lds, st, P, Bf, B);
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

// ---- the policy inside the rollout kernel (pdegym_policy.h) -----------------------------------------------------------------
// The sensing-noise hook of the reference (hyperbolic.py:160-164: the agent sees sensing_noise_func(observation)) as pre-drawn
// additive noise: the wave's LDS copy of observation t (od values) becomes obs + obs_noise[t], which is what the policy reads
// and what obs_seen[t] receives; the observation slots themselves -- the plant state with full-state sensing -- stay clean.
__device__ __forceinline__ void sense_noise(const float* obs_noise, float* obs_seen, float* xw, int od, int B, int inst, int lane, int t) {
  if (!obs_noise && !obs_seen) return;     // wave-uniform
  const size_t base = ((size_t)t * B + inst) * od;
  for (int j = lane; j < od; j += kWave) {
    float v = xw[j];
    if (obs_noise) v += obs_noise[base + j];
    xw[j] = v;
    if (obs_seen) obs_seen[base + j] = v;
  }
  pdegym_policy::wave_lds_sync();
}

// WIDE: a network with a layer of more than 64 units, evaluated by the 16 waves together (pdegym_policy.h: eval_wide) -- every wave of
// the workgroup, with or without an instance, runs all T iterations because of its barriers.
template <int EPL, bool PARABOLIC, bool BURGERS, bool WIDE>
__global__ __launch_bounds__(kWave* pdegym_policy::kWaves) void rollout1d_policy_kernel(pdegym_params1d P, pdegym_bufs1d Bf,
                                                                                        pdegym_rollout1d Ro, pdegym_mlp N, int B) {
  namespace pol = pdegym_policy;
  extern __shared__ __attribute__((aligned(16))) float pol_smem[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int inst = blockIdx.x * pol::kWaves + wave;
  const int n = P.n, xpad = pol::xpad(n);
  pol::Staged St;
  pol::Wide Wd;
  if constexpr (WIDE) Wd = pol::wide_setup(N, pol_smem, n, wave, lane);
  else St = pol::stage(N, pol_smem);      // the launch's only barrier
  const bool active = inst < B;           // wave-uniform
  if (!WIDE && !active) return;
  float* const xw = WIDE ? Wd.X + wave * Wd.ldx : pol_smem + St.end + wave * (xpad + 2 * pol::kMaxWidth);
  float* const hw = xw + xpad;
  const size_t slot = (size_t)B * n;
  constexpr int J0 = PARABOLIC ? 1 : 0;
  const int ns = n - J0, s0 = lane * EPL;
  Carry<EPL> C;       // the state stays in registers over the T env-steps (see rollout1d_kernel)
  if (active) carry_load<EPL, PARABOLIC>(C, P, Bf, Ro.obs, inst, lane);
  if (!WIDE)
    for (int j = n + lane; j < xpad; j += kWave) xw[j] = 0.f;     // zero padding to a multiple of four: written once
  for (int t = 0; t < Ro.T; ++t) {
    if (active) {
      // observation of this instance -> LDS, straight from the carried row (slot t of Ro.obs holds the same values)
      if (PARABOLIC && lane == 0) xw[0] = C.bl;
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        if (s0 + e < ns) xw[J0 + s0 + e] = C.x[e];
      pol::wave_lds_sync();
      sense_noise(Ro.obs_noise, Ro.obs_seen, xw, n, B, inst, lane, t);
    }
    float a;
    if constexpr (WIDE) a = pol::eval_wide(N, Wd, n, wave, lane);
    else a = pol::lane_value(pol::eval(N, St, pol_smem, xw, hw, n, lane), 0);      // neuron 0 of the last layer
    if (!active) continue;
    if (N.noise) a += N.noise[((size_t)t * B + inst) * N.noise_stride];
    if (N.clamp) a = fminf(fmaxf(a, N.lo), N.hi);
    if (lane == 0) Ro.actions[(size_t)t * B + inst] = a;

    pdegym_bufs1d S = Bf;
    S.u = nullptr;
    S.history = nullptr;
    S.state_in = Ro.obs + (size_t)t * slot;
    S.obs = Ro.obs + (size_t)(t + 1) * slot;
    S.action = Ro.actions + (size_t)t * B;
    S.reward = Ro.rewards + (size_t)t * B;
    S.terminated = Ro.terminated + (size_t)t * B;
    S.truncated = Ro.truncated + (size_t)t * B;
    step1d_body<EPL, PARABOLIC, false, false, BURGERS, false, true, true>(P, S, B, inst, lane, &a, &C);
  }
}

// The policy in front of the general step (Neumann actuation / scalar sensing): its input is observation slot t as stored -- od = n
// values, or the one sensed value -- read back from memory after the previous iteration's fence.
template <int EPL, bool PARABOLIC, bool NEUMANN, bool BURGERS, bool WIDE>
__global__ __launch_bounds__(kWave* pdegym_policy::kWaves) void rollout1d_policy_general_kernel(pdegym_params1d P, pdegym_bufs1d Bf,
                                                                                                pdegym_rollout1d Ro, pdegym_mlp N, int B) {
  namespace pol = pdegym_policy;
  extern __shared__ __attribute__((aligned(16))) float pol_smem[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int inst = blockIdx.x * pol::kWaves + wave;
  const int od = P.sensing == PDEGYM_SENSE_FULL ? P.n : 1, xpad = pol::xpad(od);
  pol::Staged St;
  pol::Wide Wd;
  if constexpr (WIDE) Wd = pol::wide_setup(N, pol_smem, od, wave, lane);
  else St = pol::stage(N, pol_smem);      // the launch's only barrier
  const bool active = inst < B;           // wave-uniform
  if (!WIDE && !active) return;
  float* const xw = WIDE ? Wd.X + wave * Wd.ldx : pol_smem + St.end + wave * (xpad + 2 * pol::kMaxWidth);
  float* const hw = xw + xpad;
  if (!WIDE)
    for (int j = od + lane; j < xpad; j += kWave) xw[j] = 0.f;
  for (int t = 0; t < Ro.T; ++t) {
    if (active) {
      const float* orow = Ro.obs + ((size_t)t * B + inst) * od;
      for (int j = lane; j < od; j += kWave) xw[j] = orow[j];
      pol::wave_lds_sync();
      sense_noise(Ro.obs_noise, Ro.obs_seen, xw, od, B, inst, lane, t);
    }
    float a;
    if constexpr (WIDE) a = pol::eval_wide(N, Wd, od, wave, lane);
    else a = pol::lane_value(pol::eval(N, St, pol_smem, xw, hw, od, lane), 0);
    if (!active) continue;
    if (N.noise) a += N.noise[((size_t)t * B + inst) * N.noise_stride];
    if (N.clamp) a = fminf(fmaxf(a, N.lo), N.hi);
    if (lane == 0) Ro.actions[(size_t)t * B + inst] = a;
    rollout1d_general_step<EPL, PARABOLIC, NEUMANN, BURGERS>(P, Bf, Ro, B, inst, lane, t, &a);
  }
}

template <bool PARABOLIC, bool BURGERS = false>
int launch_rollout(const pdegym_params1d* prm, const pdegym_bufs1d* buf, const pdegym_rollout1d* ro, int B, void* stream) {
  if (!prm || !buf || !ro) return pdegym::fail(-1, "null params/bufs/rollout");
  if (B <= 0 || ro->T <= 0) return 0;
  const pdegym_params1d& P = *prm;
  if (P.n < 3 || P.n > PDEGYM_MAX_N1D) return pdegym::fail(-2, "rollout: n must be in [3, 2048] (register-resident rows)");
  if (P.nt < 2) return pdegym::fail(-2, "nt must be >= 2");
  if (P.sensing < PDEGYM_SENSE_FULL || P.sensing > PDEGYM_SENSE_FIRST) return pdegym::fail(-2, "bad sensing code");
  if (P.sensing != PDEGYM_SENSE_FULL && !buf->u) return pdegym::fail(-3, "rollout with scalar sensing: the state lives in bufs.u (obs slots hold the sensed values)");
  if (P.control_type != PDEGYM_CONTROL_DIRICHLET && P.control_type != PDEGYM_CONTROL_NEUMANN) return pdegym::fail(-2, "bad control_type");
  if (P.beta_f64 || P.action_kind != PDEGYM_ACTION_F32) return pdegym::fail(-2, "rollout: float32 beta and actions only");
  if (buf->history) return pdegym::fail(-2, "rollout cannot record a history buffer");
  if (P.reward_horizon != PDEGYM_HORIZON_TEMPORAL) return pdegym::fail(-2, "rollout: only the temporal reward horizon is evaluated in the rollout kernels");
  if (!buf->beta || !buf->time_index || !buf->bsum || !buf->ring || !buf->norm_now || !buf->norm_back)
    return pdegym::fail(-3, "null device buffer");
  if (!ro->obs || !ro->actions || !ro->terminated || !ro->truncated) return pdegym::fail(-3, "null rollout buffer");
  if (P.reward_kind != PDEGYM_REWARD_NONE && !ro->rewards) return pdegym::fail(-3, "null reward buffer");
  hipStream_t st = (hipStream_t)stream;
  const int nslots = P.n - (PARABOLIC ? 1 : 0);
  const int epl = (nslots + kWave - 1) / kWave;
  // the carried, register-resident form is the Dirichlet / full-state corner; everything else takes the general kernels
  const bool neumann = P.control_type == PDEGYM_CONTROL_NEUMANN;
  const bool general = neumann || P.sensing != PDEGYM_SENSE_FULL;
  if ((ro->obs_noise || ro->obs_seen) && !ro->policy) return pdegym::fail(-2, "rollout: obs_noise / obs_seen belong to the policy's input (policy is NULL)");
  if (ro->policy) {
    const pdegym_mlp& N = *ro->policy;
    const int od = P.sensing == PDEGYM_SENSE_FULL ? P.n : 1;
    if (const char* why = pdegym_policy::check(N, od, 1, true)) return pdegym::fail(-2, why);
    if (N.x_f64 || N.y_f64) return pdegym::fail(-2, "policy inside the 1D rollout kernel: float32 observations and commands");
    if (epl > 8) return pdegym::fail(-2, "policy inside the rollout kernel: rows of up to 513 nodes");
    const int lds_bytes = pdegym_policy::lds_floats(N, od) * (int)sizeof(float);
    const dim3 pgrid((B + pdegym_policy::kWaves - 1) / pdegym_policy::kWaves), pblock(kWave * pdegym_policy::kWaves);
    bool ok = true;
    const bool wide = pdegym_policy::is_wide(N);
    auto launch_pol = [&](auto kernel, signed char (&attr)[pdegym::kMaxDevices]) {
      ok = pdegym::raise_dynamic_lds_limit(reinterpret_cast<const void*>(kernel), pdegym_policy::kMaxLdsBytes, attr);
      if (ok) hipLaunchKernelGGL(kernel, pgrid, pblock, lds_bytes, st, P, *buf, *ro, N, B);
    };
    auto gop = [&](auto tag) {
      constexpr int E = decltype(tag)::value;
      static signed char attr[6][pdegym::kMaxDevices] = {};
      if (!general) {
        if (wide) launch_pol(&rollout1d_policy_kernel<E, PARABOLIC, BURGERS, true>, attr[0]);
        else launch_pol(&rollout1d_policy_kernel<E, PARABOLIC, BURGERS, false>, attr[1]);
      } else if (neumann) {
        if (wide) launch_pol(&rollout1d_policy_general_kernel<E, PARABOLIC, true, BURGERS, true>, attr[2]);
        else launch_pol(&rollout1d_policy_general_kernel<E, PARABOLIC, true, BURGERS, false>, attr[3]);
      } else {
        if (wide) launch_pol(&rollout1d_policy_general_kernel<E, PARABOLIC, false, BURGERS, true>, attr[4]);
        else launch_pol(&rollout1d_policy_general_kernel<E, PARABOLIC, false, BURGERS, false>, attr[5]);
      }
    };
    if (epl <= 1) gop(std::integral_constant<int, 1>{});
    else if (epl <= 2) gop(std::integral_constant<int, 2>{});
    else if (epl <= 3) gop(std::integral_constant<int, 3>{});
    else if (epl <= 4) gop(std::integral_constant<int, 4>{});
    else if (epl <= 5) gop(std::integral_constant<int, 5>{});
    else if (epl <= 6) gop(std::integral_constant<int, 6>{});
    else gop(std::integral_constant<int, 8>{});
    if (!ok) return pdegym::fail(-4, "cannot raise the dynamic LDS limit of rollout1d_policy_kernel");
    return pdegym::check_launch("rollout1d_policy");
  }
  const dim3 grid((B + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
  auto go = [&](auto tag) {
    constexpr int E = decltype(tag)::value;
    if (!general) hipLaunchKernelGGL((rollout1d_kernel<E, PARABOLIC, BURGERS>), grid, block, 0, st, P, *buf, *ro, B);
    else if (neumann) hipLaunchKernelGGL((rollout1d_general_kernel<E, PARABOLIC, true, BURGERS>), grid, block, 0, st, P, *buf, *ro, B);
    else hipLaunchKernelGGL((rollout1d_general_kernel<E, PARABOLIC, false, BURGERS>), grid, block, 0, st, P, *buf, *ro, B);
  };
  // the same slots-per-lane choice as launch_step: the norm reductions (hence rewards) depend on the layout
  if (epl <= 1) go(std::integral_constant<int, 1>{});
  else if (epl <= 2) go(std::integral_constant<int, 2>{});
  else if (epl <= 3) go(std::integral_constant<int, 3>{});
  else if (epl <= 4) go(std::integral_constant<int, 4>{});
  else if (epl <= 5) go(std::integral_constant<int, 5>{});
  else if (epl <= 6) go(std::integral_constant<int, 6>{});
  else if (epl <= 8) go(std::integral_constant<int, 8>{});
  else if (epl <= 12) go(std::integral_constant<int, 12>{});
  else if (epl <= 16) go(std::integral_constant<int, 16>{});
  else if (epl <= 24) go(std::integral_constant<int, 24>{});
  else go(std::integral_constant<int, 32>{});
  return pdegym::check_launch("rollout1d");
}

}  // namespace

extern "C" {

int pdegym_transport_rollout(const pdegym_params1d* prm, const pdegym_bufs1d* buf, const pdegym_rollout1d* ro, int32_t B,
                             void* stream) {
  if (prm && prm->flux == PDEGYM_FLUX_BURGERS) return launch_rollout<false, true>(prm, buf, ro, B, stream);
  return launch_rollout<false, false>(prm, buf, ro, B, stream);
}

int pdegym_parabolic_rollout(const pdegym_params1d* prm, const pdegym_bufs1d* buf, const pdegym_rollout1d* ro, int32_t B,
                             void* stream) {
  return launch_rollout<true>(prm, buf, ro, B, stream);
}

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
