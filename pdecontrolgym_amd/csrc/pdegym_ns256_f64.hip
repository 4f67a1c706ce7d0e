// pdegym_ns256_f64.hip -- NavierStokes2D on a 256 x 256 grid at the reference's own precision (float64: navier_stokes2D.py:186,
// base_env_2d.py:50), BASELINE config 5: the whole env-step as ceil(K / 25) launches of ONE kernel.
//
// Reference semantics restated (environments2d/navier_stokes2D.py): predictor :130-138, apply_boundary :68-91, rhs :101-103,
// the sweeps of solve_pressure :104-114
//   p_new = 1/4 (((W + S) + E) + N - dx dy rhs)  on interior cells, then the four Neumann copies,
// corrector :143-146, observation / reward :147-154 (ns_reward.py:28).  Same expression trees as ns_generic_step<double>
// (pdegym_ns2d.hip), IEEE division included -> bit-identical to NumPy (tests/test_gpu_ns2d.py).
//
// p and dx dy rhs of one instance are 2 x 512 KB in float64: more than a CU holds (512 KB of registers + 160 KB of LDS), and an
// in-launch exchange between workgroups would make one workgroup wait for another.  So the K sweeps run as PASSES of at most 25
// sweeps, each a launch of three slabs per instance: a slab = 120 consecutive grid rows (8 waves x 15 rows, 4 columns = 8
// registers per lane and row) of which the middle ones are its OWN rows and the rest a halo of the neighbouring slabs; rows next
// to a cut go stale by one row per sweep, after 25 sweeps the own rows (and one more on either side) are still exact and are the
// only ones stored.  A pass keeps p (128 registers per lane) and 9 of its 15 dx dy rhs rows in registers, the other 6 in
// wave-private LDS (one 32-byte read per row and sweep); left / right neighbours are lanes (two v_mov_b32_dpp per double), the
// rows above / below a wave's block cross waves through a double-buffered 64 KB LDS area, one barrier per sweep, rows rotate with
// period two (UP / DOWN sweeps, see jacobi_sweep_bous) so a sweep has no register copies.  Redundant sweeps: 360 / 256 = 1.41x.
//
// Round 4 (round 3: front -> 3 passes of 17 sweeps -> back, 11.1 MB of HBM traffic per instance against 3 MB of compulsory bytes):
// passes of 25 sweeps (K = 50: two instead of three) and the back phase inside the LAST pass -- the slab's own pressure rows are
// stored, then its waves share the slab's own rows among themselves and run predictor (again, from the state rows) -> corrector
// -> boundary rule -> observation -> reward partial sums (the solved pressure is never read back from memory by another launch
// and never copied home).  The front phase (predictor -> boundary rule -> rhs) CAN run inside the first pass too (kPassFront: every
// wave evaluates the 15 rhs rows it is about to sweep, leaves them in the rhs field and picks them up again itself) and does so
// when one launch is the whole step (K <= 25); for K > 25 it stays a launch of its own: fused it saves no bytes (the rhs field is
// written and read either way) and costs 1.6x the rows (slab overlap 1.41 x 17/15 per wave) at two waves per SIMD -- measured
// 1.52 ms per 512 env-steps fused against the separate launch's figure in docs/HISTORY.md section 4.
#include <hip/hip_runtime.h>

#include "pdegym.h"
#include "pdegym_common.h"
#include "pdegym_ns_common.h"
#include "pdegym_ns256_rows.h"

namespace pdegym {
namespace ns {
namespace {

using namespace rows256;

constexpr int kCells = kN * kN;
constexpr int kNW = 8, kPR = 15, kRows = kNW * kPR;          // 120 rows per slab
constexpr int kNT = 64 * kNW;
constexpr int kRL = 6, kRR = kPR - kRL;                      // dx dy rhs rows per wave in LDS / in registers (the first kRR)
constexpr int kH = 25;                                       // sweeps per pass
constexpr int kSlabs = 3;
// slab s covers rows kLo[s] .. kLo[s] + 119 and owns rows kOwn[s] .. kOwn[s + 1] - 1.  After kH sweeps the rows within kH of a
// cut are stale; the back phase also reads the rows next to the own range (corrector: p[r - 1], p[r + 1]), so every own row AND
// its two neighbours must be at least kH rows away from the cuts of the slab:
//   slab 0 (cut above row 119): exact rows 0 .. 94;  slab 1 (cuts at 68 and 188): 93 .. 162;  slab 2 (cut at 136): 161 .. 255
__device__ constexpr int kLo[kSlabs] = {0, 68, 136};
__device__ constexpr int kOwn[kSlabs + 1] = {0, 94, 162, 256};
static_assert(kOwn[1] <= kLo[0] + kRows - kH - 1 && kOwn[1] - 1 >= kLo[1] + kH && kOwn[2] <= kLo[1] + kRows - kH - 1 &&
                  kOwn[2] - 1 >= kLo[2] + kH && kLo[2] + kRows == kN && kLo[0] == 0,
              "every own row and its two neighbours must be at least kH rows away from the cuts of the slab");
constexpr int kHaloBytes = 2 * 2 * kNT * 32;                 // two buffers x (top rows, bottom rows) x four doubles per thread
constexpr int kLdsBytes = kHaloBytes + kRL * kNT * 32;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget of one CU");

enum : int { kPassFront = 1, kPassBack = 2 };

__device__ __forceinline__ double shr_f64(double v) { return lane_left(v); }     // lane i <- lane i-1 (lane 0: 0, a domain-edge lane)
__device__ __forceinline__ double shl_f64(double v) { return lane_right(v); }    // lane i <- lane i+1

// LDS layout (round 5): a thread's four doubles are kept as TWO 16-byte halves in two PLANES (plane a: columns 0-1, plane b: columns
// 2-3), each plane indexed by the thread id -- so every ds_read_b128 / ds_write_b128 is lane-contiguous (16-byte lane stride).  The
// round-4 array-of-struct form (32 bytes per thread) put the 16-lane groups of a ds_read_b128 on each bank twice: SQ_LDS_BANK_CONFLICT
// 30.7 M cycles per launch against 12.0 M SQ_ACTIVE_INST_LDS (profiles/r04o_summary.json; MI355X_MICROARCH.md, LDS table).
// halo area: [buffer 0/1][top, bottom][plane a, b][kNT] double2; parked rq rows: [row][plane a, b][kNT] double2
constexpr int kPlane = kNT;                                   // double2 elements per plane

// top / bottom halo rows of a 4-wide double patch through LDS: ht = last row of the wave above, hb = first row of the wave below
// (the first / last wave of the slab re-reads its own row: a domain wall, or a cut whose rows are allowed to go stale)
__device__ __forceinline__ void halo_tb4(const double (&top)[4], const double (&bot)[4], double (&ht)[4], double (&hb)[4], char* lds,
                                         int& xc, int tid, int w) {
  double2* base = reinterpret_cast<double2*>(lds) + (xc & 1) * (4 * kPlane);
  ++xc;
  double2* eT = base;                     // planes a, b of the top rows
  double2* eB = base + 2 * kPlane;        // planes a, b of the bottom rows
  eT[tid] = make_double2(top[0], top[1]);
  eT[kPlane + tid] = make_double2(top[2], top[3]);
  eB[tid] = make_double2(bot[0], bot[1]);
  eB[kPlane + tid] = make_double2(bot[2], bot[3]);
  __syncthreads();
  const int up = (w > 0) ? tid - 64 : tid, dn = (w < kNW - 1) ? tid + 64 : tid;
  const double2 xa = eB[up], xb = eB[kPlane + up], ya = eT[dn], yb = eT[kPlane + dn];
  ht[0] = xa.x; ht[1] = xa.y; ht[2] = xb.x; ht[3] = xb.y;
  hb[0] = ya.x; hb[1] = ya.y; hb[2] = yb.x; hb[3] = yb.y;
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
    double (&row)[4] = ph[bphys<PR>(a, ST)];
    row[0] = E.lef ? row[1] : row[0];
    row[3] = E.rig ? row[2] : row[3];
  }
}

// one sweep from state ST (0: UP, rows 0 -> PR-1; 1: DOWN): new row a lands in the registers of the old row it no longer needs
template <int ST>
__device__ __forceinline__ void sweep4(double (&ph)[kPR + 1][4], const double (&rq)[kRR][4], const double2* rql, const EdgeFlags& E, char* lds,
                                       int& xc, int tid, int w) {
  constexpr int PR = kPR;
  double hlast[4];
  auto q_row = [&](int a, double (&q)[4]) __attribute__((always_inline)) {
    if (a < kRR) {
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = rq[a][k];
    } else {
      const double2 va = rql[(a - kRR) * (2 * kPlane)], vb = rql[(a - kRR) * (2 * kPlane) + kPlane];
      q[0] = va.x; q[1] = va.y; q[2] = vb.x; q[3] = vb.y;
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

// ---- front phase of one wave: rhs = rho/dt (d/dx u* + d/dy v*) (:101-103) of grid rows r0 .. r0 + len - 1 ---------------------
// state rows -> predictor (:130-138) -> apply_boundary(u*, v*) (:140) -> rhs; u*, v* exist only as a three-row window in
// registers; the two rows outside the range are evaluated as well (every wave of every slab is self-sufficient: no exchange).
template <bool INTERLEAVED>
__device__ __forceinline__ void front_rows(const NSConst& C, const NSScal<double>& S, const double* su, const double* sv, const double* act,
                                           double* rhs, int r0, int len, int lane) {
  const int c0 = 4 * lane;
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
  for (int it = 0; it <= len + 2; ++it) {
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

// ---- back phase of one wave: grid rows r0 .. r0 + len - 1 (len >= 2) of the new state ---------------------------------------------
// the predictor is evaluated again from the state rows, corrector (:143-145) with the solved pressure `pf` (rows r0 - 1 ..
// r0 + len must be in place), apply_boundary(u, v) (:146), observation (:147-154); returns the wave's share of the reward's
// squared distance (ns_reward.py:28), summed over its lanes.
template <bool INTERLEAVED>
__device__ __forceinline__ double back_rows(const NSConst& C, const NSScal<double>& S, const double* su, const double* sv, const double* act,
                                            const double* pfin, const double* uref, double* obs, int r0, int len, int lane) {
  const int c0 = 4 * lane;
  const double* pf = pfin + c0;
  const BcSel bsel = make_bc_sel(C.bc, lane);
  const double a0 = act[0];
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
  for (int it = 0; it < len; ++it) {
    const int r = r0 + it;
    double nu_[4], nv_[4], pnn[4];
    load_state_row<INTERLEAVED, double>(su, sv, r + 2, c0, nu_, nv_);
    prow(r + 2, pnn);
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
  // the range's last row: C(r0 + len - 1) is in c1.  It is the lower wall row only in a one-row range (never: len >= 2); the upper
  // wall row (255) reads C(254) = c2.  A range whose first row is the lower wall was finished inside the loop with C(1).
  {
    double dummy_u[4] = {0.0, 0.0, 0.0, 0.0}, dummy_v[4] = {0.0, 0.0, 0.0, 0.0};
    finish_row(r0 + len - 1, dummy_u, dummy_v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  return acc;
}

// The front phase as a launch of its own: one wave per band of 16 grid rows, four bands per workgroup (16 waves per CU: the phase is
// bound by float64 division throughput and wants the occupancy).
constexpr int kBand = 16, kBandsPerWg = 4, kWgPerInst = kN / (kBand * kBandsPerWg);
template <bool INTERLEAVED>
__global__ __launch_bounds__(64 * kBandsPerWg) void ns256_front_f64(NSConst C, NSScal<double> S, NSPtrs<double> P, int B) {
  const int b = blockIdx.x / kWgPerInst, g = blockIdx.x - b * kWgPerInst;
  if (b >= B) return;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const double* su = INTERLEAVED ? P.state_in + (size_t)b * kCells * 2 : P.u + (size_t)b * kCells;
  const double* sv = INTERLEAVED ? nullptr : P.v + (size_t)b * kCells;
  front_rows<INTERLEAVED>(C, S, su, sv, P.action + (size_t)b * C.action_dim, P.scratch + (size_t)b * 4 * kCells + 2 * (size_t)kCells,
                          (g * kBandsPerWg + w) * kBand, kBand, lane);
}

// One pass: [front phase] -> nsweeps <= 25 sweeps from p_src into p_dst (different fields) -> [back phase].
template <bool INTERLEAVED>
__global__ __launch_bounds__(kNT, 2) void ns256_pass_f64(NSConst C, NSScal<double> S, NSPtrs<double> P, const double* p_src, size_t src_stride,
                                                         double* p_dst, size_t dst_stride, int nsweeps, int phases, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = blockIdx.x / kSlabs, slab = blockIdx.x - b * kSlabs;
  if (b >= B) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = 4 * lane;
  const int g0 = kLo[slab] + w * kPR;                       // grid row of this wave's first row
  const double* su = INTERLEAVED ? P.state_in + (size_t)b * kCells * 2 : P.u + (size_t)b * kCells;
  const double* sv = INTERLEAVED ? nullptr : P.v + (size_t)b * kCells;
  const double* act = P.action + (size_t)b * C.action_dim;
  double* rhs = P.scratch + (size_t)b * 4 * kCells + 2 * (size_t)kCells;
  if (phases & kPassFront) {
    // the 15 rhs rows this wave sweeps below; the thread that stores a value is the thread that loads it again (same lane, same
    // columns), so no barrier is needed -- only program order of one thread's accesses to one address
    front_rows<INTERLEAVED>(C, S, su, sv, act, rhs, g0, kPR, lane);
    __builtin_amdgcn_sched_barrier(0);
  }
  const EdgeFlags E{slab == 0 && w == 0, slab == kSlabs - 1 && w == kNW - 1, lane == 0, lane == 63};
  const int own_lo = kOwn[slab], own_hi = kOwn[slab + 1];
  {
    const double* ps = p_src + (size_t)b * src_stride + (size_t)g0 * kN + c0;
    const double* rs = rhs + (size_t)g0 * kN + c0;
    double2* rql = reinterpret_cast<double2*>(smem_raw + kHaloBytes) + tid;        // row j of this thread: planes rql[j * 2 kPlane], rql[j * 2 kPlane + kPlane]
    double ph[kPR + 1][4], rq[kRR][4];
#pragma unroll
    for (int a = 0; a < kPR; ++a) {
      const double2 x = *reinterpret_cast<const double2*>(ps + a * kN), y = *reinterpret_cast<const double2*>(ps + a * kN + 2);
      const double2 r = *reinterpret_cast<const double2*>(rs + a * kN), s = *reinterpret_cast<const double2*>(rs + a * kN + 2);
      ph[a][0] = x.x; ph[a][1] = x.y; ph[a][2] = y.x; ph[a][3] = y.y;
      const double q0 = S.dxdy * r.x, q1 = S.dxdy * r.y, q2 = S.dxdy * s.x, q3 = S.dxdy * s.y;       // dx dy rhs (:108)
      if (a < kRR) {
        rq[a][0] = q0; rq[a][1] = q1; rq[a][2] = q2; rq[a][3] = q3;
      } else {
        rql[(a - kRR) * (2 * kPlane)] = make_double2(q0, q1);
        rql[(a - kRR) * (2 * kPlane) + kPlane] = make_double2(q2, q3);
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
    // the back phase reads one row beyond the own range on either side: those two rows are exact here as well (see kOwn) and are
    // stored too -- the slab that owns them stores the same bits
    const int st_lo = (phases & kPassBack) ? own_lo - 1 : own_lo, st_hi = (phases & kPassBack) ? own_hi + 1 : own_hi;
    if (it < nsweeps) {      // odd sweep count: one more UP sweep, then the rows move back to their state-0 registers
      sweep4<0>(ph, rq, rql, E, smem_raw, xc, tid, w);
      __builtin_amdgcn_sched_barrier(0);
      double r0[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) r0[k] = ph[bphys<kPR>(0, 1)][k];
#pragma unroll
      for (int a = kPR - 1; a >= 1; --a) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ph[a][k] = ph[bphys<kPR>(a, 1)][k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) ph[0][k] = r0[k];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < kPR; ++a) {
      const int g = g0 + a;
      if (g >= st_lo && g < st_hi) {        // wave-uniform
        const double (&row)[4] = ph[a];
        *reinterpret_cast<double2*>(pd + a * kN) = make_double2(row[0], row[1]);
        *reinterpret_cast<double2*>(pd + a * kN + 2) = make_double2(row[2], row[3]);
      }
    }
  }
  if (!(phases & kPassBack)) return;
  // ---- back phase: the slab's own rows, shared among its eight waves (pressure rows of other waves: stored above, published by
  // the barrier; p_dst lines were never read by this CU before, so no stale copy can sit in its vector cache)
  __threadfence_block();
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  const int n_own = own_hi - own_lo, per = (n_own + kNW - 1) / kNW;
  const int r0 = own_lo + w * per;
  int len = own_hi - r0;
  len = len > per ? per : len;
  double part = 0.0;
  if (len >= 2) {          // (kOwn: every wave of every slab gets 5 .. 12 rows)
    const int t_new = P.time_index[b] + 1;
    const int tr = t_new < C.nt_ref ? t_new : C.nt_ref - 1;
    part = back_rows<INTERLEAVED>(C, S, su, sv, act, p_dst + (size_t)b * dst_stride, P.U_ref + (size_t)tr * kCells * 2,
                                  P.obs + (size_t)b * kCells * 2, r0, len, lane);
  }
  double* red = reinterpret_cast<double*>(smem_raw);        // the halo area is free now
  if (lane == 0) red[w] = part;
  __syncthreads();
  if (tid == 0) {
    double ssum = 0.0;
    for (int k = 0; k < kNW; ++k) ssum += red[k];                          // fixed order: deterministic
    P.scratch[(size_t)b * 4 * kCells + slab] = ssum;                       // u* quarter of the scratch: unused on this path
  }
}
static_assert((kOwn[1] - kOwn[0] + kNW - 1) / kNW * (kNW - 1) + 2 <= kOwn[1] - kOwn[0] &&
                  (kOwn[2] - kOwn[1] + kNW - 1) / kNW * (kNW - 1) + 2 <= kOwn[2] - kOwn[1] &&
                  (kOwn[3] - kOwn[2] + kNW - 1) / kNW * (kNW - 1) + 2 <= kOwn[3] - kOwn[2],
              "the last wave of every slab must be left with at least two own rows for the back phase");

__global__ void ns256_finish_f64(NSConst C, NSScal<double> S, NSPtrs<double> P, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* part = P.scratch + (size_t)b * 4 * kCells;
  double ss = 0.0;
  for (int k = 0; k < kSlabs; ++k) ss += part[k];                          // fixed order: deterministic
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

// With separate u, v fields the state is read from them and the interleaved observation is the hand-over: a slab's back phase
// reads state rows that another slab's back phase would overwrite, so u, v are filled from the observation by a copy launch.
__global__ __launch_bounds__(256) void ns256_split_obs_f64(const double* obs, double* u, double* v, size_t ncell2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per two cells
  if (i >= ncell2) return;
  const double2* q = reinterpret_cast<const double2*>(obs) + 2 * i;
  const double2 a = q[0], d = q[1];
  reinterpret_cast<double2*>(u)[i] = make_double2(a.x, d.x);
  reinterpret_cast<double2*>(v)[i] = make_double2(a.y, d.y);
}

}  // namespace

// The whole float64 env-step: [front launch ->] npass launches of ns256_pass_f64 (the last with the back phase; for K <= 25 one launch
// with both phases) + the per-instance finish.  The pressure travels p -> scratch quarter 3 -> p -> ... and arrives in
// its home field (p_out if given, else p) with the last pass; a pass never reads the field it writes (the slabs of an instance
// read each other's rows), so with home == p the number of passes is made even (a pass may have zero sweeps).
int launch_ns256_step_f64(const NSConst& C, const NSScal<double>& S, const NSPtrs<double>& P, int B, hipStream_t st) {
  static signed char attr_a[pdegym::kMaxDevices] = {}, attr_b[pdegym::kMaxDevices] = {};
  const size_t ncell = kCells;
  const bool inter = P.state_in != nullptr;
  if (!pdegym::raise_dynamic_lds_limit(inter ? reinterpret_cast<const void*>(&ns256_pass_f64<true>) : reinterpret_cast<const void*>(&ns256_pass_f64<false>),
                                       kLdsBytes, inter ? attr_a : attr_b))
    return pdegym::fail(-4, "cannot raise the dynamic LDS limit of ns256_pass_f64");
  double* home = P.p_out ? P.p_out : P.p;
  double* tmp = P.scratch + 3 * ncell;
  int npass = (C.iters + kH - 1) / kH;
  if (npass < 1) npass = 1;
  if (home == P.p && (npass & 1)) ++npass;
  const bool fuse_front = C.iters <= kH && home != P.p;      // one launch is the whole step
  if (!fuse_front) {
    const dim3 grid(kWgPerInst * B), block(64 * kBandsPerWg);
    if (inter) hipLaunchKernelGGL(ns256_front_f64<true>, grid, block, 0, st, C, S, P, B);
    else hipLaunchKernelGGL(ns256_front_f64<false>, grid, block, 0, st, C, S, P, B);
  }
  const double* src = P.p;
  size_t src_stride = ncell;
  int left = C.iters;
  for (int i = 0; i < npass; ++i) {
    const bool last = i == npass - 1;
    // intermediate fields alternate scratch, p, scratch, ... (p is free to be overwritten once the first pass has read it)
    double* dst = last ? home : ((i & 1) ? P.p : tmp);
    const size_t dst_stride = (dst == tmp) ? 4 * ncell : ncell;
    const int rem_passes = npass - i;
    int ns = (left + rem_passes - 1) / rem_passes;           // spread the sweeps evenly over the passes that are left
    ns = ns > kH ? kH : ns;
    const int phases = ((i == 0 && fuse_front) ? kPassFront : 0) | (last ? kPassBack : 0);
    if (inter) hipLaunchKernelGGL(ns256_pass_f64<true>, dim3(kSlabs * B), dim3(kNT), kLdsBytes, st, C, S, P, src, src_stride, dst, dst_stride, ns, phases, B);
    else hipLaunchKernelGGL(ns256_pass_f64<false>, dim3(kSlabs * B), dim3(kNT), kLdsBytes, st, C, S, P, src, src_stride, dst, dst_stride, ns, phases, B);
    left -= ns;
    src = dst;
    src_stride = dst_stride;
  }
  if (!inter) {
    const size_t n2 = (size_t)B * kCells / 2;
    hipLaunchKernelGGL(ns256_split_obs_f64, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, P.obs, P.u, P.v, n2);
  }
  hipLaunchKernelGGL(ns256_finish_f64, dim3((B + 255) / 256), dim3(256), 0, st, C, S, P, B);
  return pdegym::check_launch("ns2d_pass_step_f64");
}

}  // namespace ns
}  // namespace pdegym
