// pdegym_ns256_f64.hip -- the pressure solve of NavierStokes2D on a 256 x 256 grid at the reference's own precision
// (float64: navier_stokes2D.py:186, base_env_2d.py:50), BASELINE config 5.
//
// Reference semantics restated: solve_pressure's sweeps, environments2d/navier_stokes2D.py:104-114
//   p_new = 1/4 (((W + S) + E) + N - dx dy rhs)  on interior cells, then the four Neumann copies;
// same expression tree as ns_generic_step<double> (pdegym_ns2d.hip) -> bit-identical to NumPy (tests/test_gpu_ns2d.py).
//
// p and dx dy rhs of one instance are 2 x 512 KB in float64: more than a CU holds (512 KB of registers + 160 KB of LDS), and an
// in-launch exchange between workgroups would make one workgroup wait for another.  So the K sweeps run as ceil(K / 17) PASSES
// of at most 17 sweeps, each a launch of three slabs per instance: a slab = 112 consecutive grid rows (8 waves x 14 rows, 4
// columns = 8 registers per lane and row) of which the middle ones are its OWN rows and the rest a halo of the neighbouring
// slabs; rows next to a cut go stale by one row per sweep, after 17 sweeps the own rows are still exact and are the only ones
// stored.  A pass keeps p (120 registers per lane) and 8 of its 14 dx dy rhs rows in registers, the other 6 in wave-private
// LDS (one 32-byte read per row and sweep); left / right neighbours are lanes (two v_mov_b32_dpp per double), the rows above /
// below a wave's block cross waves through a double-buffered 64 KB LDS area, one barrier per sweep, rows rotate with period
// two (UP / DOWN sweeps, see jacobi_sweep_bous) so a sweep has no register copies.  Passes ping-pong between two pressure fields.
// Redundant sweeps: 336 / 256 rows = 1.31x.  The predictor / corrector phases around the solve are ns_generic_step's
// row-wave kernels below (ns256_front_f64 / ns256_back_f64: one wave per band of 16 grid rows).
#include <hip/hip_runtime.h>

#include "pdegym.h"
#include "pdegym_common.h"
#include "pdegym_ns_common.h"
#include "pdegym_ns256_rows.h"

namespace pdegym {
namespace ns {
namespace {

constexpr int kN = 256, kCells = kN * kN;
constexpr int kNW = 8, kPR = 14, kRows = kNW * kPR;          // 112 rows per slab
constexpr int kNT = 64 * kNW;
constexpr int kRL = 6, kRR = kPR - kRL;                      // dx dy rhs rows per wave in LDS / in registers (the first kRR)
constexpr int kH = 17;                                       // sweeps per pass
constexpr int kSlabs = 3;
// slab s covers rows kLo[s] .. kLo[s] + 111 and owns rows kOwn[s] .. kOwn[s + 1] - 1; a cut row is >= kH rows away from every
// own row of the slab it bounds (0 + 112 - 17 = 95 > 88; 72 + 17 = 89, 184 - 17 = 167; 144 + 17 = 161 <= 167)
__device__ constexpr int kLo[kSlabs] = {0, 72, 144};
__device__ constexpr int kOwn[kSlabs + 1] = {0, 89, 167, 256};
static_assert(kLo[1] + kH <= kOwn[1] && kLo[1] + kRows - kH >= kOwn[2] && kLo[2] + kH <= kOwn[2] && kRows - kH >= kOwn[1] &&
              kLo[2] + kRows == kN, "every own row must be at least kH rows away from the cuts of its slab");
constexpr int kHaloBytes = 2 * 2 * kNT * 32;                 // two buffers x (top rows, bottom rows) x four doubles per thread
constexpr int kLdsBytes = kHaloBytes + kRL * kNT * 32;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget of one CU");

__device__ __forceinline__ double shr_f64(double v) {       // lane i <- lane i-1 (lane 0: 0, a domain-edge lane)
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double shl_f64(double v) {       // lane i <- lane i+1
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

struct D4 {
  double2 a, b;
};

// top / bottom halo rows of a 4-wide double patch through LDS: ht = last row of the wave above, hb = first row of the wave below
// (the first / last wave of the slab re-reads its own row: a domain wall, or a cut whose rows are allowed to go stale)
__device__ __forceinline__ void halo_tb4(const double (&top)[4], const double (&bot)[4], double (&ht)[4], double (&hb)[4], char* lds,
                                         int& xc, int tid, int w) {
  D4* base = reinterpret_cast<D4*>(lds) + (xc & 1) * (2 * kNT);
  ++xc;
  D4* eT = base;
  D4* eB = base + kNT;
  eT[tid] = D4{make_double2(top[0], top[1]), make_double2(top[2], top[3])};
  eB[tid] = D4{make_double2(bot[0], bot[1]), make_double2(bot[2], bot[3])};
  __syncthreads();
  const int up = (w > 0) ? tid - 64 : tid, dn = (w < kNW - 1) ? tid + 64 : tid;
  const D4 x = eB[up], y = eT[dn];
  ht[0] = x.a.x; ht[1] = x.a.y; ht[2] = x.b.x; ht[3] = x.b.y;
  hb[0] = y.a.x; hb[1] = y.a.y; hb[2] = y.b.x; hb[3] = y.b.y;
}

// the four Neumann copies (:110-113) on the new rows, which sit in state ST.  Selects, not branches: with control flow in the
// sweep the compiler hoists every row's lane shifts and LDS reads into the dominating block and spills ~60 registers.
template <int ST>
__device__ __forceinline__ void walls4(double (&ph)[kPR + 1][4], const EdgeFlags& E) {
  constexpr int PR = kPR;
  constexpr int n0 = bphys<PR>(0, ST), n1 = bphys<PR>(1, ST), nl = bphys<PR>(PR - 1, ST), nm = bphys<PR>(PR - 2, ST);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ph[n0][k] = E.top ? ph[n1][k] : ph[n0][k];
    ph[nl][k] = E.bot ? ph[nm][k] : ph[nl][k];
  }
#pragma unroll
  for (int a = 0; a < PR; ++a) {
    constexpr int dummy = 0;
    (void)dummy;
    double (&row)[4] = ph[bphys<PR>(a, ST)];
    row[0] = E.lef ? row[1] : row[0];
    row[3] = E.rig ? row[2] : row[3];
  }
}

// one sweep from state ST (0: UP, rows 0 -> PR-1; 1: DOWN): new row a lands in the registers of the old row it no longer needs
template <int ST>
__device__ __forceinline__ void sweep4(double (&ph)[kPR + 1][4], const double (&rq)[kRR][4], const D4* rql, const EdgeFlags& E, char* lds,
                                       int& xc, int tid, int w) {
  constexpr int PR = kPR;
  double hlast[4];
  auto q_row = [&](int a, double (&q)[4]) __attribute__((always_inline)) {
    if (a < kRR) {
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = rq[a][k];
    } else {
      const D4 v = rql[(a - kRR) * kNT];
      q[0] = v.a.x; q[1] = v.a.y; q[2] = v.b.x; q[3] = v.b.y;
    }
  };
  auto update = [&](double (&dst)[4], const double (&x)[4], const double (&sv)[4], const double (&nv)[4], const double (&q)[4])
      __attribute__((always_inline)) {
    const double xl = shr_f64(x[3]), xr = shl_f64(x[0]);
    double o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double wv = (k == 0) ? xl : x[k - 1], ev = (k == 3) ? xr : x[k + 1];
      o[k] = 0.25 * ((((wv + sv[k]) + ev) + nv[k]) - q[k]);      // navier_stokes2D.py:106-108
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = o[k];                    // dst may alias sv: every read comes first
  };
  if constexpr (ST == 0) {
    halo_tb4(ph[0], ph[PR - 1], ph[PR], hlast, lds, xc, tid, w);            // top halo -> free row PR
#pragma unroll
    for (int a = 0; a < PR; ++a) {
      double q[4];
      q_row(a, q);
      if (a == PR - 1) update(ph[a == 0 ? PR : a - 1], ph[a], ph[a == 0 ? PR : a - 1], hlast, q);
      else update(ph[a == 0 ? PR : a - 1], ph[a], ph[a == 0 ? PR : a - 1], ph[a + 1], q);
      __builtin_amdgcn_sched_barrier(0);      // row by row: left alone the scheduler hoists several rows' reads and spills
    }
    walls4<1>(ph, E);
  } else {
    halo_tb4(ph[bphys<PR>(0, 1)], ph[bphys<PR>(PR - 1, 1)], hlast, ph[PR - 1], lds, xc, tid, w);   // bottom halo -> free row PR-1
#pragma unroll
    for (int a = PR - 1; a >= 0; --a) {
      double q[4];
      q_row(a, q);
      if (a == 0) update(ph[a], ph[bphys<PR>(0, 1)], hlast, ph[a], q);
      else update(ph[a], ph[bphys<PR>(a, 1)], ph[bphys<PR>(a - 1, 1)], ph[a], q);
      __builtin_amdgcn_sched_barrier(0);
    }
    walls4<0>(ph, E);
  }
}

__global__ __launch_bounds__(kNT, 2) void ns256_slab_f64(const double* p_src, size_t src_stride, double* p_dst, size_t dst_stride,
                                                         const double* rhs_base, size_t rhs_stride, double dxdy, int nsweeps, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = blockIdx.x / kSlabs, slab = blockIdx.x - b * kSlabs;
  if (b >= B) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = 4 * lane;
  const int g0 = kLo[slab] + w * kPR;                       // grid row of this wave's first row
  const EdgeFlags E{slab == 0 && w == 0, slab == kSlabs - 1 && w == kNW - 1, lane == 0, lane == 63};
  const double* ps = p_src + (size_t)b * src_stride + (size_t)g0 * kN + c0;
  const double* rs = rhs_base + (size_t)b * rhs_stride + (size_t)g0 * kN + c0;
  D4* rql = reinterpret_cast<D4*>(smem_raw + kHaloBytes) + tid;        // row j of this thread: rql[j * kNT]
  double ph[kPR + 1][4], rq[kRR][4];
#pragma unroll
  for (int a = 0; a < kPR; ++a) {
    const double2 x = *reinterpret_cast<const double2*>(ps + a * kN), y = *reinterpret_cast<const double2*>(ps + a * kN + 2);
    const double2 r = *reinterpret_cast<const double2*>(rs + a * kN), s = *reinterpret_cast<const double2*>(rs + a * kN + 2);
    ph[a][0] = x.x; ph[a][1] = x.y; ph[a][2] = y.x; ph[a][3] = y.y;
    const double q0 = dxdy * r.x, q1 = dxdy * r.y, q2 = dxdy * s.x, q3 = dxdy * s.y;       // dx dy rhs (:108)
    if (a < kRR) {
      rq[a][0] = q0; rq[a][1] = q1; rq[a][2] = q2; rq[a][3] = q3;
    } else {
      rql[(a - kRR) * kNT] = D4{make_double2(q0, q1), make_double2(q2, q3)};
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) ph[kPR][k] = 0.0;
  int xc = 0, it = 0;
  for (; it + 2 <= nsweeps; it += 2) {
    sweep4<0>(ph, rq, rql, E, smem_raw, xc, tid, w);
    sweep4<1>(ph, rq, rql, E, smem_raw, xc, tid, w);
  }
  double* pd = p_dst + (size_t)b * dst_stride + (size_t)g0 * kN + c0;
  const int own_lo = kOwn[slab], own_hi = kOwn[slab + 1];
  if (it < nsweeps) {      // odd sweep count: one more UP sweep, the rows are stored from state 1
    sweep4<0>(ph, rq, rql, E, smem_raw, xc, tid, w);
#pragma unroll
    for (int a = 0; a < kPR; ++a) {
      const int g = g0 + a;
      if (g >= own_lo && g < own_hi) {        // wave-uniform
        const double (&row)[4] = ph[bphys<kPR>(a, 1)];
        *reinterpret_cast<double2*>(pd + a * kN) = make_double2(row[0], row[1]);
        *reinterpret_cast<double2*>(pd + a * kN + 2) = make_double2(row[2], row[3]);
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < kPR; ++a) {
      const int g = g0 + a;
      if (g >= own_lo && g < own_hi) {
        const double (&row)[4] = ph[a];
        *reinterpret_cast<double2*>(pd + a * kN) = make_double2(row[0], row[1]);
        *reinterpret_cast<double2*>(pd + a * kN + 2) = make_double2(row[2], row[3]);
      }
    }
  }
}

// ---- the phases around the solve: one wave per band of kBand grid rows, a rolled row pipeline (rows256 helpers) ---------------
// front: state rows -> predictor (:130-138) -> apply_boundary(u*, v*) (:140) -> rhs = rho/dt (d/dx u* + d/dy v*) (:101-103), ONE
//        field written (u*, v* exist only as a three-row window in registers); the two rows outside a band are evaluated again
//        by the neighbouring band (18/16 of the predictor work) instead of being exchanged.
// back:  the predictor is evaluated again from the state rows, corrector (:143-145) with the solved pressure, apply_boundary(u, v)
//        (:146), observation (:147-154), per-band partial sums of the reward's squared distance (ns_reward.py:28); the solved
//        pressure is copied home when the last pass left it in the scratch field.
using namespace rows256;
constexpr int kBand = 16, kBandsPerWg = 4, kWgPerInst = kN / (kBand * kBandsPerWg);   // 4 workgroups of 4 waves per instance

template <bool INTERLEAVED>
__global__ __launch_bounds__(64 * kBandsPerWg) void ns256_front_f64(NSConst C, NSScal<double> S, NSPtrs<double> P, int B) {
  const int b = blockIdx.x / kWgPerInst, g = blockIdx.x - b * kWgPerInst;
  if (b >= B) return;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = 4 * lane, r0 = (g * kBandsPerWg + w) * kBand;
  const double* su = INTERLEAVED ? P.state_in + (size_t)b * kCells * 2 : P.u + (size_t)b * kCells;
  const double* sv = INTERLEAVED ? nullptr : P.v + (size_t)b * kCells;
  const double* act = P.action + (size_t)b * C.action_dim;
  double* rhs = P.scratch + (size_t)b * 4 * kCells + 2 * (size_t)kCells;
  const BcSel bsel = make_bc_sel(C.bc, lane);
  const double a0 = act[0];
  double s0u[4], s0v[4], s1u[4], s1v[4], s2u[4], s2v[4];          // state rows r-1, r, r+1
  double p1u[4], p1v[4], p2u[4], p2v[4];                          // P(r-1), P(r-2)
  double f1u[4], f1v[4], f2v[4];                                  // F_u(r-2), F_v(r-2), F_v(r-3)
  load_state_row<INTERLEAVED, double>(su, sv, r0 - 2, c0, s0u, s0v);
  load_state_row<INTERLEAVED, double>(su, sv, r0 - 1, c0, s1u, s1v);
  load_state_row<INTERLEAVED, double>(su, sv, r0, c0, s2u, s2v);
#pragma unroll
  for (int k = 0; k < 4; ++k) p1u[k] = p1v[k] = p2u[k] = p2v[k] = f1u[k] = f1v[k] = f2v[k] = 0.0;
#pragma unroll 1
  for (int it = 0; it <= kBand + 2; ++it) {
    const int r = r0 - 1 + it;
    double nu_[4], nv_[4];
    load_state_row<INTERLEAVED, double>(su, sv, r + 2, c0, nu_, nv_);       // next iteration's row r+1
    double pu[4], pv[4];
    predictor_row<double>(S, r, lane, s1u, s1v, s0u, s0v, s2u, s2v, pu, pv);
    const int rr = r - 1;                                                   // boundary rule on row r-1
    double fu[4], fv[4], nbu[4], nbv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      fu[k] = p1u[k]; fv[k] = p1v[k];
      nbu[k] = rr == 0 ? pu[k] : p2u[k];
      nbv[k] = rr == 0 ? pv[k] : p2v[k];
    }
    bc_row<double>(fu, nbu, rr, c0, C.bc, 0, bsel, act, C.action_dim, a0);
    bc_row<double>(fv, nbv, rr, c0, C.bc, 1, bsel, act, C.action_dim, a0);
    if (it >= 3) {                                                          // rhs of row r-2
      const int i = r - 2;
      const double ul = lane_left(f1u[3]), ur = lane_right(f1u[0]);
      double q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double uw = (k == 0) ? ul : f1u[k - 1], ue = (k == 3) ? ur : f1u[k + 1];
        const double dudx = div_c(ue - uw, S.two_dx, S.inv_two_dx);
        const double dvdy = div_c(fv[k] - f2v[k], S.two_dy, S.inv_two_dy);
        const bool edge = (i == 0) || (i == kN - 1) || (lane == 0 && k == 0) || (lane == 63 && k == 3);
        q[k] = edge ? 0.0 : S.rho_over_dt * (dudx + dvdy);
      }
      double* dst = rhs + (size_t)i * kN + c0;
      *reinterpret_cast<double2*>(dst) = make_double2(q[0], q[1]);
      *reinterpret_cast<double2*>(dst + 2) = make_double2(q[2], q[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f2v[k] = f1v[k]; f1v[k] = fv[k]; f1u[k] = fu[k];
      p2u[k] = p1u[k]; p2v[k] = p1v[k]; p1u[k] = pu[k]; p1v[k] = pv[k];
      s0u[k] = s1u[k]; s0v[k] = s1v[k]; s1u[k] = s2u[k]; s1v[k] = s2v[k]; s2u[k] = nu_[k]; s2v[k] = nv_[k];
    }
  }
}

template <bool INTERLEAVED>
__global__ __launch_bounds__(64 * kBandsPerWg) void ns256_back_f64(NSConst C, NSScal<double> S, NSPtrs<double> P, const double* pfin_base,
                                                                  size_t pfin_stride, double* p_copy_to, int B) {
  __shared__ double red[kBandsPerWg];
  const int b = blockIdx.x / kWgPerInst, g = blockIdx.x - b * kWgPerInst;
  if (b >= B) return;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = 4 * lane, r0 = (g * kBandsPerWg + w) * kBand;
  const double* su = INTERLEAVED ? P.state_in + (size_t)b * kCells * 2 : P.u + (size_t)b * kCells;
  const double* sv = INTERLEAVED ? nullptr : P.v + (size_t)b * kCells;
  const double* act = P.action + (size_t)b * C.action_dim;
  const double* pf = pfin_base + (size_t)b * pfin_stride + c0;
  const BcSel bsel = make_bc_sel(C.bc, lane);
  const double a0 = act[0];
  const int t_new = P.time_index[b] + 1;
  const int tr = t_new < C.nt_ref ? t_new : C.nt_ref - 1;
  const double* uref = P.U_ref + (size_t)tr * kCells * 2;
  double* obs = P.obs + (size_t)b * kCells * 2;
  auto prow = [&](int row, double (&v)[4]) __attribute__((always_inline)) {
    const int rc = row < 0 ? 0 : (row > kN - 1 ? kN - 1 : row);
    const double2 x = *reinterpret_cast<const double2*>(pf + (size_t)rc * kN), y = *reinterpret_cast<const double2*>(pf + (size_t)rc * kN + 2);
    v[0] = x.x; v[1] = x.y; v[2] = y.x; v[3] = y.y;
  };
  double s0u[4], s0v[4], s1u[4], s1v[4], s2u[4], s2v[4];
  double c1u[4], c1v[4], c2u[4], c2v[4];                          // C(r-1), C(r-2)
  double ps[4], pc[4], pn[4];                                     // p rows r-1, r, r+1
  load_state_row<INTERLEAVED, double>(su, sv, r0 - 1, c0, s0u, s0v);
  load_state_row<INTERLEAVED, double>(su, sv, r0, c0, s1u, s1v);
  load_state_row<INTERLEAVED, double>(su, sv, r0 + 1, c0, s2u, s2v);
  prow(r0 - 1, ps);
  prow(r0, pc);
  prow(r0 + 1, pn);
#pragma unroll
  for (int k = 0; k < 4; ++k) c1u[k] = c1v[k] = c2u[k] = c2v[k] = 0.0;
  double acc = 0.0;
  auto finish_row = [&](int rr, const double (&cu)[4], const double (&cv)[4]) __attribute__((always_inline)) {
    double fu[4], fv[4], nbu[4], nbv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      fu[k] = c1u[k]; fv[k] = c1v[k];
      nbu[k] = rr == 0 ? cu[k] : c2u[k];
      nbv[k] = rr == 0 ? cv[k] : c2v[k];
    }
    bc_row<double>(fu, nbu, rr, c0, C.bc, 0, bsel, act, C.action_dim, a0);
    bc_row<double>(fv, nbv, rr, c0, C.bc, 1, bsel, act, C.action_dim, a0);
    const size_t o = ((size_t)rr * kN + c0) * 2;
    const double2* rrow = reinterpret_cast<const double2*>(uref + o);
    double2* orow = reinterpret_cast<double2*>(obs + o);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double2 ref = rrow[k];
      orow[k] = make_double2(fu[k], fv[k]);
      const double du = fu[k] - ref.x, dv = fv[k] - ref.y;
      acc += du * du;
      acc += dv * dv;
    }
  };
#pragma unroll 1
  for (int it = 0; it < kBand; ++it) {
    const int r = r0 + it;
    double nu_[4], nv_[4], pnn[4];
    load_state_row<INTERLEAVED, double>(su, sv, r + 2, c0, nu_, nv_);
    prow(r + 2, pnn);
    if (p_copy_to) {
      double* dst = p_copy_to + (size_t)b * kCells + (size_t)r * kN + c0;
      *reinterpret_cast<double2*>(dst) = make_double2(pc[0], pc[1]);
      *reinterpret_cast<double2*>(dst + 2) = make_double2(pc[2], pc[3]);
    }
    double cu[4], cv[4];
    predictor_row<double>(S, r, lane, s1u, s1v, s0u, s0v, s2u, s2v, cu, cv);
    {
      const double pl = lane_left(pc[3]), pr = lane_right(pc[0]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double pw = (k == 0) ? pl : pc[k - 1], pe = (k == 3) ? pr : pc[k + 1];
        const double dpdx = div_c(pe - pw, S.two_dx, S.inv_two_dx);
        const double dpdy = div_c(pn[k] - ps[k], S.two_dy, S.inv_two_dy);
        const bool edge = (r <= 0) || (r >= kN - 1) || (lane == 0 && k == 0) || (lane == 63 && k == 3);
        cu[k] = edge ? cu[k] : cu[k] - S.dt_over_rho * dpdx;
        cv[k] = edge ? cv[k] : cv[k] - S.dt_over_rho * dpdy;
      }
    }
    if (it >= 1) finish_row(r - 1, cu, cv);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c2u[k] = c1u[k]; c2v[k] = c1v[k]; c1u[k] = cu[k]; c1v[k] = cv[k];
      ps[k] = pc[k]; pc[k] = pn[k]; pn[k] = pnn[k];
      s0u[k] = s1u[k]; s0v[k] = s1v[k]; s1u[k] = s2u[k]; s1v[k] = s2v[k]; s2u[k] = nu_[k]; s2v[k] = nv_[k];
    }
  }
  // the band's last row: C(r0+15) is in c1.  It is the lower wall row only in a one-row band (never); the upper wall row (255)
  // reads C(254) = c2.  A band whose first row is the lower wall was finished inside the loop with C(1).
  {
    double dummy_u[4] = {0.0, 0.0, 0.0, 0.0}, dummy_v[4] = {0.0, 0.0, 0.0, 0.0};
    finish_row(r0 + kBand - 1, dummy_u, dummy_v);
  }
  // the first band's row 0 needs C(1): finished at it = 1 above (rr == 0 takes cu = C(1)); nothing else crosses bands
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) red[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double ssum = 0.0;
    for (int k = 0; k < kBandsPerWg; ++k) ssum += red[k];                  // fixed order: deterministic
    P.scratch[(size_t)b * 4 * kCells + g] = ssum;                          // u* quarter of the scratch: unused on this path
  }
}

__global__ void ns256_finish_f64(NSConst C, NSScal<double> S, NSPtrs<double> P, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* part = P.scratch + (size_t)b * 4 * kCells;
  double ss = 0.0;
  for (int k = 0; k < kWgPerInst; ++k) ss += part[k];                      // fixed order: deterministic
  const int t = P.time_index[b] + 1;
  const int tr = t < C.nt_ref ? t : C.nt_ref - 1;
  const double* act = P.action + (size_t)b * C.action_dim;
  double asq = 0.0;
  const double aref = P.action_ref[tr];
  for (int k = 0; k < C.action_dim; ++k) {
    const double d = act[k] - aref;
    asq += d * d;
  }
  P.reward[b] = ((-0.5 * ss) / (double)kN) / (double)kN - S.gamma_half * asq;
  P.time_index[b] = t;
  P.terminated[b] = (t >= C.nt - 1) ? 1 : 0;                               // navier_stokes2D.py:159-168
}

}  // namespace

// Pass i of the solve: `nsweeps` <= 17 sweeps from p_src into p_dst (different fields); rhs as gen_front leaves it.
static int launch_ns256_slab_f64(const double* p_src, size_t src_stride, double* p_dst, size_t dst_stride, const double* rhs, size_t rhs_stride,
                          double dxdy, int nsweeps, int B, hipStream_t st) {
  static signed char attr[pdegym::kMaxDevices] = {};
  if (nsweeps < 1 || nsweeps > kH) return pdegym::fail(-2, "a float64 256x256 pass takes 1..17 sweeps");
  if (!pdegym::raise_dynamic_lds_limit(reinterpret_cast<const void*>(&ns256_slab_f64), kLdsBytes, attr))
    return pdegym::fail(-4, "cannot raise the dynamic LDS limit of ns256_slab_f64");
  hipLaunchKernelGGL(ns256_slab_f64, dim3(kSlabs * B), dim3(kNT), kLdsBytes, st, p_src, src_stride, p_dst, dst_stride, rhs, rhs_stride, dxdy,
                     nsweeps, B);
  return 0;
}

// The whole float64 env-step: front launch -> ceil(K / 17) slab passes ping-ponging between p and scratch quarter 3 -> back launch
// (which leaves the solved pressure in p_out / p) -> finish.  With separate u, v fields the state is read from and written to them
// in place: the back kernel's bands read rows r0-1 .. r0+16 of the OLD state, so the in-place form runs the interleaved observation
// as the hand-over instead (obs is always written; u, v are filled from it by a copy launch).
__global__ __launch_bounds__(256) void ns256_split_obs_f64(const double* obs, double* u, double* v, size_t ncell2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per two cells
  if (i >= ncell2) return;
  const double2* q = reinterpret_cast<const double2*>(obs) + 2 * i;
  const double2 a = q[0], d = q[1];
  reinterpret_cast<double2*>(u)[i] = make_double2(a.x, d.x);
  reinterpret_cast<double2*>(v)[i] = make_double2(a.y, d.y);
}

int launch_ns256_step_f64(const NSConst& C, const NSScal<double>& S, const NSPtrs<double>& P, int B, hipStream_t st) {
  const size_t ncell = kCells;
  const bool inter = P.state_in != nullptr;
  const dim3 grid(kWgPerInst * B), block(64 * kBandsPerWg);
  if (inter) hipLaunchKernelGGL(ns256_front_f64<true>, grid, block, 0, st, C, S, P, B);
  else hipLaunchKernelGGL(ns256_front_f64<false>, grid, block, 0, st, C, S, P, B);
  double* bufs[2] = {P.p, P.scratch + 3 * ncell};
  const size_t strides[2] = {ncell, 4 * ncell};
  int cur = 0;
  for (int left = C.iters; left > 0; left -= kH) {
    if (int rc = launch_ns256_slab_f64(bufs[cur], strides[cur], bufs[cur ^ 1], strides[cur ^ 1], P.scratch + 2 * ncell, 4 * ncell, S.dxdy,
                                       left < kH ? left : kH, B, st))
      return rc;
    cur ^= 1;
  }
  double* home = P.p_out ? P.p_out : P.p;
  double* copy_to = (bufs[cur] == home) ? nullptr : home;
  if (inter) {
    hipLaunchKernelGGL(ns256_back_f64<true>, grid, block, 0, st, C, S, P, bufs[cur], strides[cur], copy_to, B);
  } else {
    // reads the old state from P.u / P.v, writes only the observation (a band reads rows of its neighbours' state, so u, v
    // cannot be updated in place by the same launch); u, v are split out of the observation afterwards
    hipLaunchKernelGGL(ns256_back_f64<false>, grid, block, 0, st, C, S, P, bufs[cur], strides[cur], copy_to, B);
    const size_t n2 = (size_t)B * kCells / 2;
    hipLaunchKernelGGL(ns256_split_obs_f64, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, P.obs, P.u, P.v, n2);
  }
  hipLaunchKernelGGL(ns256_finish_f64, dim3((B + 255) / 256), dim3(256), 0, st, C, S, P, B);
  return pdegym::check_launch("ns2d_slab_step_f64");
}

}  // namespace ns
}  // namespace pdegym
