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
// (ns_front_kernel / ns_back_kernel in pdegym_ns2d.hip).
#include <hip/hip_runtime.h>

#include "pdegym.h"
#include "pdegym_common.h"
#include "pdegym_ns_common.h"

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

}  // namespace

int ns256_f64_passes(int iters) { return iters > 0 ? (iters + kH - 1) / kH : 0; }

// Pass i of the solve: `nsweeps` <= 17 sweeps from p_src into p_dst (different fields); rhs as gen_front leaves it.
int launch_ns256_slab_f64(const double* p_src, size_t src_stride, double* p_dst, size_t dst_stride, const double* rhs, size_t rhs_stride,
                          double dxdy, int nsweeps, int B, hipStream_t st) {
  static signed char attr[pdegym::kMaxDevices] = {};
  if (nsweeps < 1 || nsweeps > kH) return pdegym::fail(-2, "a float64 256x256 pass takes 1..17 sweeps");
  if (!pdegym::raise_dynamic_lds_limit(reinterpret_cast<const void*>(&ns256_slab_f64), kLdsBytes, attr))
    return pdegym::fail(-4, "cannot raise the dynamic LDS limit of ns256_slab_f64");
  hipLaunchKernelGGL(ns256_slab_f64, dim3(kSlabs * B), dim3(kNT), kLdsBytes, st, p_src, src_stride, p_dst, dst_stride, rhs, rhs_stride, dxdy,
                     nsweeps, B);
  return 0;
}

}  // namespace ns
}  // namespace pdegym
