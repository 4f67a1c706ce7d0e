// pdegym_ns256.hip -- NavierStokes2D on a 256 x 256 grid (BASELINE config 5), float32: ONE launch per env-step, one
// workgroup per instance, one instance per CU.
//
// Reference semantics restated (environments2d/navier_stokes2D.py): predictor :130-138, apply_boundary :68-91 (called at
// :140 and :146), solve_pressure :94-116 (right-hand side :101-103, sweeps :104-114), corrector :143-145, observation
// :147-154, NSReward rewards/ns_reward.py:28.  Same expression trees as ns_generic_step<float> (pdegym_ns2d.hip): fields,
// pressure and observations are bit-identical to it (tests/test_gpu_ns2d.py).
//
// Why one workgroup per instance.  p and 0.25 dx dy rhs of an instance are 2 x 256 KB; a CU has 512 KB of vector registers
// and 160 KB of LDS.  Eight waves (two per SIMD, 256 registers per lane each) keep ALL of p (32 rows x 4 columns per lane =
// 128 registers) and 20-22 of their 32 right-hand-side rows in registers; the other rows sit in wave-private LDS and
// come back one ds_read_b128 per row and sweep, prefetched one two-row block ahead.  Nothing is recomputed (the earlier
// two-slab pass did 1.41x the sweeps' work on halo rows) and nothing but the state, the pressure and the observation
// crosses HBM:
//   front   rows of the state -> predictor -> boundary rule -> 0.25 dx dy rhs       (a rolled row pipeline per wave,
//           u*, v* exist only as a three-row window in registers)
//   sweeps  K Jacobi sweeps on registers: left / right neighbours are lanes (DPP), the rows above / below a wave's block
//           cross waves through a double-buffered 32 KB LDS area, one barrier per sweep (jacobi_sweep machinery of
//           pdegym_ns_common.h: period-two row rotation, two-row in-place asm blocks)
//   back    the predictor is evaluated again from the state rows (they are L2 / Infinity-Cache resident: 256 instances x
//           512 KB are in flight), corrector, boundary rule, observation, reward partial sums
// HBM traffic per env-step: read state + p, re-read state, write p + observation = 8 fields of 256 KB (the three-launch
// pipeline it replaces moved 12 and its Jacobi pass ran at 3 waves per SIMD with one workgroup-wide burst of loads).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "pdegym.h"
#include "pdegym_common.h"
#include "pdegym_ns_common.h"
#include "pdegym_ns256_rows.h"

namespace pdegym {
namespace ns {
namespace {

using namespace rows256;

constexpr int kN = 256, kCells = kN * kN;
constexpr int kWaves = 8, kPR = kN / kWaves;      // 32 grid rows per wave, 4 columns per lane
constexpr int kNT = 64 * kWaves;
#ifndef PDEGYM_NS256_LDS_ROWS
#define PDEGYM_NS256_LDS_ROWS 16
#endif
constexpr int kRL = PDEGYM_NS256_LDS_ROWS;        // right-hand-side rows per wave that live in LDS (even)
constexpr int kRR = kPR - kRL;                    // rows in registers: the first kRR of the wave's block
constexpr int kL0 = kRR;                          // first LDS row
#ifndef PDEGYM_NS256_RING
#define PDEGYM_NS256_RING 8
#endif
constexpr int kD = PDEGYM_NS256_RING;             // state rows in flight per lane in the front / back row pipelines (ring slots)
constexpr int kHaloBytes = 2 * 2 * kNT * 16;      // two buffers x (top rows, bottom rows) x one float4 per thread
constexpr int kLdsBytes = kHaloBytes + kRL * kNT * 16;
static_assert(kRL % 2 == 0 && kRL >= 2 && kRL < kPR && kRR % 2 == 0, "LDS rows come in two-row blocks");
static_assert(kLdsBytes <= 160 * 1024, "LDS budget of one CU");

__device__ __forceinline__ constexpr int rq_reg(int a) { return a; }

// ---- one Jacobi sweep (state ST: 0 = UP, 1 = DOWN; see jacobi_sweep_bous), right-hand side partly in LDS -------------
template <int ST>
__device__ __forceinline__ void sweep256(float (&ph)[kPR + 1][4], const float (&rq)[kRR][4], const float4* rql, const EdgeFlags& E,
                                         float* lds, int& xc, int tid, int ty) {
  constexpr int PR = kPR, NJL = kRL / 2;
  float hlast[4];
  float buf[2][2][4];
  auto ld = [&](int jl, int par) __attribute__((always_inline)) {
    const float4 x = rql[(2 * jl) * kNT], y = rql[(2 * jl + 1) * kNT];
    buf[par][0][0] = x.x; buf[par][0][1] = x.y; buf[par][0][2] = x.z; buf[par][0][3] = x.w;
    buf[par][1][0] = y.x; buf[par][1][1] = y.y; buf[par][1][2] = y.z; buf[par][1][3] = y.w;
  };
  if constexpr (ST == 0) {
    ld(0, 0);
    asm volatile("" ::: "memory");
    halo_tb<4, kNT, 64>(ph[bphys<PR>(0, 0)], ph[bphys<PR>(PR - 1, 0)], ph[PR], hlast, lds, xc, tid, ty);   // top halo -> free row PR
#pragma unroll
    for (int a = 0; a + 1 < PR; a += 2) {       // rows (a, a+1)
      float (&da)[4] = ph[a == 0 ? PR : a - 1];
      const bool inl = a >= kL0 && a < kL0 + kRL;
      if (inl) {
        const int jl = (a - kL0) / 2;
        if (jl + 1 < NJL) ld(jl + 1, (jl + 1) & 1);
        asm volatile("" ::: "memory");
        if (a + 2 == PR) jacobi_pair_up(da, ph[a], ph[a + 1], hlast, buf[jl & 1][0], buf[jl & 1][1]);
        else jacobi_pair_up(da, ph[a], ph[a + 1], ph[a + 2], buf[jl & 1][0], buf[jl & 1][1]);
      } else {
        if (a + 2 == PR) jacobi_pair_up(da, ph[a], ph[a + 1], hlast, rq[rq_reg(a)], rq[rq_reg(a + 1)]);
        else jacobi_pair_up(da, ph[a], ph[a + 1], ph[a + 2], rq[rq_reg(a)], rq[rq_reg(a + 1)]);
      }
    }
    jacobi_walls_state<PR, 1>(ph, E);
  } else {
    // state 1: logical row a in physical row a-1 (row 0 in PR); physical row PR-1 is free -> bottom halo
    ld(NJL - 1, (NJL - 1) & 1);        // the DOWN sweep starts with the LDS rows: requested ahead of the exchange's barrier
    asm volatile("" ::: "memory");
    halo_tb<4, kNT, 64>(ph[bphys<PR>(0, 1)], ph[bphys<PR>(PR - 1, 1)], hlast, ph[PR - 1], lds, xc, tid, ty);
#pragma unroll
    for (int a = PR - 1; a >= 1; a -= 2) {      // rows (a, a-1): new row a -> physical row a, new row a-1 -> physical row a-1
      const bool inl = (a - 1) >= kL0 && a < kL0 + kRL;
      if (inl) {
        const int jl = (a - 1 - kL0) / 2;
        if (jl >= 1) ld(jl - 1, (jl - 1) & 1);
        asm volatile("" ::: "memory");
        if (a == 1) jacobi_pair_down(ph[a], ph[a - 1], ph[bphys<PR>(0, 1)], hlast, buf[jl & 1][1], buf[jl & 1][0]);
        else jacobi_pair_down(ph[a], ph[a - 1], ph[bphys<PR>(a - 1, 1)], ph[bphys<PR>(a - 2, 1)], buf[jl & 1][1], buf[jl & 1][0]);
      } else {
        if (a == 1) jacobi_pair_down(ph[a], ph[a - 1], ph[bphys<PR>(0, 1)], hlast, rq[rq_reg(a)], rq[rq_reg(a - 1)]);
        else jacobi_pair_down(ph[a], ph[a - 1], ph[bphys<PR>(a - 1, 1)], ph[bphys<PR>(a - 2, 1)], rq[rq_reg(a)], rq[rq_reg(a - 1)]);
      }
    }
    jacobi_walls_state<PR, 0>(ph, E);
  }
}

template <bool INTERLEAVED>
__global__ __launch_bounds__(kNT, 2) void ns256_fused_step(NSConst C, NSScal<float> S, NSPtrs<float> P, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const int b = blockIdx.x;
  if (b >= B) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = 4 * lane, r0 = w * kPR;
  const EdgeFlags E{w == 0, w == kWaves - 1, lane == 0, lane == 63};
  const float* su = INTERLEAVED ? P.state_in + (size_t)b * kCells * 2 : P.u + (size_t)b * kCells;
  const float* sv = INTERLEAVED ? nullptr : P.v + (size_t)b * kCells;
  const float* act = P.action + (size_t)b * C.action_dim;
  float4* rql = reinterpret_cast<float4*>(smem_raw + kHaloBytes) + tid;   // row j of this thread: rql[j * kNT]
  const BcSel bsel = make_bc_sel(C.bc, lane);
  const float a0 = act[0];

  float rq[kRR][4];
#ifdef PDEGYM_NS256_TIMING      // developer build (tools/attic/timing_probe_ns256.py): s_memtime at the phase boundaries of every wave
  unsigned long long tm[6];
  tm[0] = __builtin_amdgcn_s_memtime();
#define NS256_STAMP(i, dep) tm[i] = __builtin_amdgcn_s_memtime() + (unsigned long long)((dep) != (dep))
#else
#define NS256_STAMP(i, dep)
#endif

  // ---- front: predictor -> apply_boundary(u*, v*) (:140) -> 0.25 dx dy rhs (:101-103, :108), a row pipeline ----
  // iteration `it` works on grid row r = r0 - 1 + it:  P(r) = predictor of row r;  F(r-1) = row r-1 after the boundary rule
  // (the lower wall row takes its inside neighbour P(1) = P(r), the upper one P(254) = P(r-2));  rq(r-2) from F_u(r-2),
  // F_v(r-3), F_v(r-1).  The two rows outside the wave's block (r0-1, r0+32) are evaluated here too instead of being
  // exchanged with the neighbouring waves (34/32 of the predictor work, no barrier).
  // State rows arrive through a ring of kD register rows with static slots (the loop is unrolled kD times): the slot that
  // held row r-1 is refilled with row r-1+kD as soon as the predictor has read it, so kD-3 rows (10 loads per lane) are in
  // flight while a row is processed -- with one row in flight the phase ran at the latency of an HBM access per row.
  // A finished rq row goes to the thread's LDS slots first (a register row cannot be picked by a run-time index); after kRL
  // rows the slots are copied into register rows by straight-line code, and the last kRL rows simply stay there.
  {
    constexpr int D = kD;
    float ru[D][4], rv[D][4];                                        // ring of state rows
    float p1u[4], p1v[4], p2u[4], p2v[4];                            // P(r-1), P(r-2)
    float f1u[4], f1v[4], f2v[4];                                    // F_u(r-2), F_v(r-2), F_v(r-3)
    // slot of row S(r+1) at iteration `it` is (it - 3) mod D; S(r) and S(r-1) sit in the two slots before it
#pragma unroll
    for (int s = 0; s < D; ++s) load_state_row<INTERLEAVED, float>(su, sv, r0 - 2 + ((s + D - 3) % D), c0, ru[s], rv[s]);
#pragma unroll
    for (int k = 0; k < 4; ++k) p1u[k] = p1v[k] = p2u[k] = p2v[k] = f1u[k] = f1v[k] = f2v[k] = 0.f;
    auto row_iter = [&](int it, auto slot_c, int slot0) __attribute__((always_inline)) {   // wave row a = it - 3 goes to LDS slot a - slot0
      constexpr int sl = decltype(slot_c)::value, sc = (sl + D - 1) % D, ss = (sl + D - 2) % D;
      const int r = r0 - 1 + it;
      float pu[4], pv[4];
      predictor_row(S, r, lane, ru[sc], rv[sc], ru[ss], rv[ss], ru[sl], rv[sl], pu, pv);
      load_state_row<INTERLEAVED, float>(su, sv, r - 1 + D, c0, ru[ss], rv[ss]);     // refill the slot row r-1 has left
      // boundary rule on row r-1
      const int rr = r - 1;
      float fu[4], fv[4], nbu[4], nbv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        fu[k] = p1u[k]; fv[k] = p1v[k];
        nbu[k] = rr == 0 ? pu[k] : p2u[k];
        nbv[k] = rr == 0 ? pv[k] : p2v[k];
      }
      bc_row(fu, nbu, rr, c0, C.bc, 0, bsel, act, C.action_dim, a0);
      bc_row(fv, nbv, rr, c0, C.bc, 1, bsel, act, C.action_dim, a0);
      // right-hand side of row r-2 = wave row a
      const int a = it - 3;
      if (a >= 0) {
        const int i = r - 2;
        const float ul = lane_left(f1u[3]), ur = lane_right(f1u[0]);
        float q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float uw = (k == 0) ? ul : f1u[k - 1], ue = (k == 3) ? ur : f1u[k + 1];
          const float dudx = div_c(ue - uw, S.two_dx, S.inv_two_dx);
          const float dvdy = div_c(fv[k] - f2v[k], S.two_dy, S.inv_two_dy);
          const float rh = S.rho_over_dt * (dudx + dvdy);
          const bool edge = (i == 0) || (i == kN - 1) || (lane == 0 && k == 0) || (lane == 63 && k == 3);
          q[k] = edge ? 0.0f : jacobi_rhs_term(S.dxdy, rh);
        }
        rql[(a - slot0) * kNT] = make_float4(q[0], q[1], q[2], q[3]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f2v[k] = f1v[k]; f1v[k] = fv[k]; f1u[k] = fu[k];
        p2u[k] = p1u[k]; p2v[k] = p1v[k]; p1u[k] = pu[k]; p1v[k] = pv[k];
      }
    };
    row_iter(0, std::integral_constant<int, (D - 3) % D>{}, 0);
    row_iter(1, std::integral_constant<int, (D - 2) % D>{}, 0);
    row_iter(2, std::integral_constant<int, (D - 1) % D>{}, 0);
    auto rows = [&](int a_lo, int a_hi, int slot0) __attribute__((always_inline)) {        // a_hi - a_lo is a multiple of D
#pragma unroll 1
      for (int it0 = a_lo + 3; it0 < a_hi + 3; it0 += D) {
        row_iter(it0 + 0, std::integral_constant<int, 0>{}, slot0);
        row_iter(it0 + 1, std::integral_constant<int, 1 % D>{}, slot0);
        row_iter(it0 + 2, std::integral_constant<int, 2 % D>{}, slot0);
        row_iter(it0 + 3, std::integral_constant<int, 3 % D>{}, slot0);
        if constexpr (D == 8) {
          row_iter(it0 + 4, std::integral_constant<int, 4 % D>{}, slot0);
          row_iter(it0 + 5, std::integral_constant<int, 5 % D>{}, slot0);
          row_iter(it0 + 6, std::integral_constant<int, 6 % D>{}, slot0);
          row_iter(it0 + 7, std::integral_constant<int, 7 % D>{}, slot0);
        }
      }
    };
    constexpr int NCH = (kRR + kRL - 1) / kRL;          // chunks of register rows
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int a_lo = c * kRL, a_hi = (c + 1) * kRL < kRR ? (c + 1) * kRL : kRR;
      rows(a_lo, a_hi, a_lo);
#pragma unroll
      for (int a = a_lo; a < a_hi; ++a) {
        const float4 x = rql[(a - a_lo) * kNT];
        rq[a][0] = x.x; rq[a][1] = x.y; rq[a][2] = x.z; rq[a][3] = x.w;
      }
    }
    rows(kRR, kPR, kRR);
  }

  NS256_STAMP(1, rq[0][0]);
  // ---- K Jacobi sweeps (:104-114): p in registers, period-two row rotation ----
  float ph[kPR + 1][4];
  {
    const float* p = P.p + (size_t)b * kCells + (size_t)r0 * kN + c0;
#pragma unroll
    for (int a = 0; a < kPR; ++a) {
      const float4 x = *reinterpret_cast<const float4*>(p + a * kN);
      ph[a][0] = x.x; ph[a][1] = x.y; ph[a][2] = x.z; ph[a][3] = x.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) ph[kPR][k] = 0.f;
  }
  NS256_STAMP(2, ph[0][0]);
  int xc = 0;
  {
    int it = 0;
    for (; it + 2 <= C.iters; it += 2) {
      sweep256<0>(ph, rq, rql, E, lds, xc, tid, w);
      sweep256<1>(ph, rq, rql, E, lds, xc, tid, w);
    }
    if (it < C.iters) {   // odd sweep count: one more UP sweep, then rotate the rows back to the identity map
      sweep256<0>(ph, rq, rql, E, lds, xc, tid, w);
      float t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = ph[kPR][k];
#pragma unroll
      for (int a = kPR; a >= 1; --a)
#pragma unroll
        for (int k = 0; k < 4; ++k) ph[a][k] = ph[a - 1][k];
#pragma unroll
      for (int k = 0; k < 4; ++k) ph[0][k] = t[k];
    }
  }
  NS256_STAMP(3, ph[0][0]);
  {
    float* pd = (P.p_out ? P.p_out : P.p) + (size_t)b * kCells + (size_t)r0 * kN + c0;
#pragma unroll
    for (int a = 0; a < kPR; ++a) *reinterpret_cast<float4*>(pd + a * kN) = make_float4(ph[a][0], ph[a][1], ph[a][2], ph[a][3]);
  }

  // ---- back: corrector (:143-145), apply_boundary(u, v) (:146), observation (:147-154), reward (ns_reward.py:28) ----
  // iteration `it` forms C(r), the corrected row r = r0 + it, from the re-evaluated predictor and p rows r-1, r, r+1, and
  // finishes row r-1 (the lower wall row takes C(1) = C(r), the upper one C(254) = C(r-2)).  Rings with static slots as in
  // the front: state rows (kD slots), pressure rows (kD slots; the wave's own rows come back from where they were just
  // stored -- same thread, same addresses, L2 hits -- so that no register row has to be picked by a run-time index; the rows
  // above / below the block come through the LDS exchange) and rows of the reference frame (4 slots).
  float acc = 0.f;
  // the lane's column offset is formed again behind an opaque copy: left to itself the compiler computes every address of this
  // phase before the front and carries it in scratch across the sweeps (whose registers are all spoken for)
  int lane_b = lane;
  asm volatile("" : "+v"(lane_b));
  const int c0b = 4 * lane_b;
  const int t_new = P.time_index[b] + 1;
  const int tr = t_new < C.nt_ref ? t_new : C.nt_ref - 1;
  {
    constexpr int D = kD;
    static_assert(kPR % kD == 0 && (kD == 4 || kD == 8), "the back loop runs kPR iterations in blocks of kD");
    float pt[4], pb[4];
    halo_tb<4, kNT, 64>(ph[0], ph[kPR - 1], pt, pb, lds, xc, tid, w);
    const float* uref = P.U_ref + (size_t)tr * kCells * 2;
    float* obs = P.obs + (size_t)b * kCells * 2;
    const float* pg = (P.p_out ? P.p_out : P.p) + (size_t)b * kCells + (size_t)r0 * kN + c0b;
    constexpr int DP = 4, DF = 2;      // pressure rows and reference rows are L2 hits: one row ahead is enough
    float ru[D][4], rv[D][4], rp[DP][4];
    float4 rf[DF][2];
    float c1u[4], c1v[4], c2u[4], c2v[4];                            // C(r-1), C(r-2)
    auto prow = [&](int a, float (&v)[4]) __attribute__((always_inline)) {                          // p row r0 + a, -1 <= a
      const int ac = a < 0 ? 0 : (a < kPR - 1 ? a : kPR - 1);
      const float4 x = *reinterpret_cast<const float4*>(pg + ac * kN);
      v[0] = a < 0 ? pt[0] : (a < kPR ? x.x : pb[0]);
      v[1] = a < 0 ? pt[1] : (a < kPR ? x.y : pb[1]);
      v[2] = a < 0 ? pt[2] : (a < kPR ? x.z : pb[2]);
      v[3] = a < 0 ? pt[3] : (a < kPR ? x.w : pb[3]);
    };
    auto urow = [&](int row, float4 (&q)[2]) __attribute__((always_inline)) {
      const int rc = row < 0 ? 0 : (row > kN - 1 ? kN - 1 : row);
      const float4* rrow = reinterpret_cast<const float4*>(uref + (rc * kN + c0b) * 2);
      q[0] = rrow[0];
      q[1] = rrow[1];
    };
    // at iteration `it`: S(r-1), S(r), S(r+1) sit in slots it, it+1, it+2 (mod D), p(r-1), p(r), p(r+1) likewise (mod DP),
    // Uref(r-1) in slot it mod DF
#pragma unroll
    for (int s = 0; s < D; ++s) load_state_row<INTERLEAVED, float>(su, sv, r0 - 1 + s, c0b, ru[s], rv[s]);
#pragma unroll
    for (int s = 0; s < DP; ++s) prow(s - 1, rp[s]);
#pragma unroll
    for (int s = 0; s < DF; ++s) urow(r0 - 1 + s, rf[s]);
#pragma unroll
    for (int k = 0; k < 4; ++k) c1u[k] = c1v[k] = c2u[k] = c2v[k] = 0.f;
    auto finish_row = [&](int rr, const float (&cu)[4], const float (&cv)[4], const float4 (&ref)[2]) __attribute__((always_inline)) {
      float fu[4], fv[4], nbu[4], nbv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        fu[k] = c1u[k]; fv[k] = c1v[k];
        nbu[k] = rr == 0 ? cu[k] : c2u[k];
        nbv[k] = rr == 0 ? cv[k] : c2v[k];
      }
      bc_row(fu, nbu, rr, c0b, C.bc, 0, bsel, act, C.action_dim, a0);
      bc_row(fv, nbv, rr, c0b, C.bc, 1, bsel, act, C.action_dim, a0);
      float4* orow = reinterpret_cast<float4*>(obs + (rr * kN + c0b) * 2);
      orow[0] = make_float4(fu[0], fv[0], fu[1], fv[1]);
      orow[1] = make_float4(fu[2], fv[2], fu[3], fv[3]);
      const float d0 = fu[0] - ref[0].x, d1 = fv[0] - ref[0].y, d2 = fu[1] - ref[0].z, d3 = fv[1] - ref[0].w;
      const float d4 = fu[2] - ref[1].x, d5 = fv[2] - ref[1].y, d6 = fu[3] - ref[1].z, d7 = fv[3] - ref[1].w;
      acc += d0 * d0;
      acc += d1 * d1;
      acc += d2 * d2;
      acc += d3 * d3;
      acc += d4 * d4;
      acc += d5 * d5;
      acc += d6 * d6;
      acc += d7 * d7;
    };
    auto row_iter = [&](int it, auto slot_c) __attribute__((always_inline)) {
      constexpr int s0 = decltype(slot_c)::value, s1 = (s0 + 1) % D, s2 = (s0 + 2) % D, sf = s0 % DF;
      constexpr int q0 = s0 % DP, q1 = (s0 + 1) % DP, q2 = (s0 + 2) % DP;
      const int r = r0 + it;
      float cu[4], cv[4];
      predictor_row(S, r, lane, ru[s1], rv[s1], ru[s0], rv[s0], ru[s2], rv[s2], cu, cv);
      {
        const float pl = lane_left(rp[q1][3]), pr = lane_right(rp[q1][0]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float pw = (k == 0) ? pl : rp[q1][k - 1], pe = (k == 3) ? pr : rp[q1][k + 1];
          const float dpdx = div_c(pe - pw, S.two_dx, S.inv_two_dx);
          const float dpdy = div_c(rp[q2][k] - rp[q0][k], S.two_dy, S.inv_two_dy);
          const bool edge = (r <= 0) || (r >= kN - 1) || (lane == 0 && k == 0) || (lane == 63 && k == 3);
          cu[k] = edge ? cu[k] : cu[k] - S.dt_over_rho * dpdx;
          cv[k] = edge ? cv[k] : cv[k] - S.dt_over_rho * dpdy;
        }
      }
      load_state_row<INTERLEAVED, float>(su, sv, r - 1 + D, c0b, ru[s0], rv[s0]);
      prow(it - 1 + DP, rp[q0]);
      if (it >= 1) finish_row(r - 1, cu, cv, rf[sf]);
      urow(r - 1 + DF, rf[sf]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        c2u[k] = c1u[k]; c2v[k] = c1v[k]; c1u[k] = cu[k]; c1v[k] = cv[k];
      }
    };
#pragma unroll 1
    for (int it0 = 0; it0 < kPR; it0 += D) {
      row_iter(it0 + 0, std::integral_constant<int, 0>{});
      row_iter(it0 + 1, std::integral_constant<int, 1 % D>{});
      row_iter(it0 + 2, std::integral_constant<int, 2 % D>{});
      row_iter(it0 + 3, std::integral_constant<int, 3 % D>{});
      if constexpr (D == 8) {
        row_iter(it0 + 4, std::integral_constant<int, 4 % D>{});
        row_iter(it0 + 5, std::integral_constant<int, 5 % D>{});
        row_iter(it0 + 6, std::integral_constant<int, 6 % D>{});
        row_iter(it0 + 7, std::integral_constant<int, 7 % D>{});
      }
    }
    // the block's last row: C(r0+31) is in c1; the wall rule of row 255 reads C(254) = c2
    finish_row(r0 + kPR - 1, c1u, c1v, rf[kPR % DF]);
  }
  NS256_STAMP(4, acc);
#ifdef PDEGYM_NS256_TIMING
  if (lane == 0) {
    unsigned int* dbg = reinterpret_cast<unsigned int*>(P.scratch + (size_t)b * 4 * kCells) + w * 8;
    for (int i = 0; i < 4; ++i) dbg[i] = (unsigned int)(tm[i + 1] - tm[i]);
    dbg[4] = (unsigned int)tm[0];
    dbg[5] = (unsigned int)tm[4];
  }
#endif
  const float ss = block_sum<float>(acc, lds);     // the halo buffers are idle now (block_sum syncs first)
  if (tid == 0) {
    float asq = 0.f;
    const float aref = P.action_ref[tr];
    for (int k = 0; k < C.action_dim; ++k) {
      const float d = act[k] - aref;
      asq += d * d;
    }
    P.reward[b] = ((-0.5f * ss) / (float)kN) / (float)kN - S.gamma_half * asq;
    P.time_index[b] = t_new;
    P.terminated[b] = (t_new >= C.nt - 1) ? 1 : 0;      // navier_stokes2D.py:159-168
  }
}

// separate-field state layout: the observation just written is copied out into u and v
__global__ __launch_bounds__(256) void ns256_split_obs(const float* obs, float* u, float* v, size_t ncell4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per four cells
  if (i >= ncell4) return;
  const float4* q = reinterpret_cast<const float4*>(obs) + 2 * i;
  const float4 a = q[0], d = q[1];
  reinterpret_cast<float4*>(u)[i] = make_float4(a.x, a.z, d.x, d.z);
  reinterpret_cast<float4*>(v)[i] = make_float4(a.y, a.w, d.y, d.w);
}

}  // namespace

int launch_ns256_fused(const NSConst& C, const NSScal<float>& S, const NSPtrs<float>& P, int B, hipStream_t st) {
  static signed char attr_i[pdegym::kMaxDevices] = {}, attr_s[pdegym::kMaxDevices] = {};
  const bool inter = P.state_in != nullptr;
  const bool ok = inter ? pdegym::raise_dynamic_lds_limit(reinterpret_cast<const void*>(&ns256_fused_step<true>), kLdsBytes, attr_i)
                        : pdegym::raise_dynamic_lds_limit(reinterpret_cast<const void*>(&ns256_fused_step<false>), kLdsBytes, attr_s);
  if (!ok) return pdegym::fail(-4, "cannot raise the dynamic LDS limit of ns256_fused_step");
  if (inter) {
    hipLaunchKernelGGL(ns256_fused_step<true>, dim3(B), dim3(kNT), kLdsBytes, st, C, S, P, B);
  } else {
    hipLaunchKernelGGL(ns256_fused_step<false>, dim3(B), dim3(kNT), kLdsBytes, st, C, S, P, B);
    const size_t n4 = (size_t)B * kCells / 4;
    hipLaunchKernelGGL(ns256_split_obs, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, P.obs, P.u, P.v, n4);
  }
  return pdegym::check_launch("ns2d_fused256_step");
}

}  // namespace ns
}  // namespace pdegym
