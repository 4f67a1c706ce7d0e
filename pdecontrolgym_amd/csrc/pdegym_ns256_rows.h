// pdegym_ns256_rows.h -- row-wise pieces of one NavierStokes2D env-step on a 256-wide grid, one wave per grid row (lane l
// holds columns 4 l .. 4 l + 3), shared by the float32 fused step (pdegym_ns256.hip) and the float64 phases around the slab
// passes (pdegym_ns256_f64.hip).  T = float | double; the double forms keep IEEE division (div_c) -> bit parity with NumPy.
// Reference lines: predictor navier_stokes2D.py:130-138, apply_boundary :68-91, derivatives :9-22.
#pragma once
#include "pdegym_ns_common.h"

namespace pdegym {
namespace ns {
namespace rows256 {

constexpr int kN = 256;

__device__ __forceinline__ double lane_left(double v) {       // lane i <- lane i-1 (lane 0: 0, a domain-edge lane)
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double lane_right(double v) {      // lane i <- lane i+1
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
using pdegym::ns::lane_left;
using pdegym::ns::lane_right;

// (u, v) of columns 4 lane .. 4 lane + 3 of grid row `row` (clamped: rows outside the grid only feed values nobody reads)
template <bool INTERLEAVED, typename T>
__device__ __forceinline__ void load_state_row(const T* su, const T* sv, int row, int c0, T (&fu)[4], T (&fv)[4]) {
  const int r = row < 0 ? 0 : (row > kN - 1 ? kN - 1 : row);
  if constexpr (sizeof(T) == 4) {          // 16-byte loads either way
    if constexpr (INTERLEAVED) {
      const float4* q = reinterpret_cast<const float4*>(su + (r * kN + c0) * 2);
      const float4 a = q[0], d = q[1];
      fu[0] = a.x; fv[0] = a.y; fu[1] = a.z; fv[1] = a.w; fu[2] = d.x; fv[2] = d.y; fu[3] = d.z; fv[3] = d.w;
    } else {
      const float4 a = *reinterpret_cast<const float4*>(su + r * kN + c0);
      const float4 d = *reinterpret_cast<const float4*>(sv + r * kN + c0);
      fu[0] = a.x; fu[1] = a.y; fu[2] = a.z; fu[3] = a.w; fv[0] = d.x; fv[1] = d.y; fv[2] = d.z; fv[3] = d.w;
    }
  } else {
    if constexpr (INTERLEAVED) {
      const double2* q = reinterpret_cast<const double2*>(su + (r * kN + c0) * 2);
      const double2 a = q[0], b = q[1], c = q[2], d = q[3];
      fu[0] = a.x; fv[0] = a.y; fu[1] = b.x; fv[1] = b.y; fu[2] = c.x; fv[2] = c.y; fu[3] = d.x; fv[3] = d.y;
    } else {
      const double2* qu = reinterpret_cast<const double2*>(su + r * kN + c0);
      const double2* qv = reinterpret_cast<const double2*>(sv + r * kN + c0);
      const double2 a = qu[0], b = qu[1], c = qv[0], d = qv[1];
      fu[0] = a.x; fu[1] = a.y; fu[2] = b.x; fu[3] = b.y; fv[0] = c.x; fv[1] = c.y; fv[2] = d.x; fv[3] = d.y;
    }
  }
}

// predictor of grid row i (navier_stokes2D.py:130-138) from the state rows i-1 (S), i (C), i+1 (N); cells on the domain edge
// keep the state value (central_difference / laplace are zero there, :9-22)
template <typename T>
__device__ __forceinline__ void predictor_row(const NSScal<T>& S, int i, int lane, const T (&uc)[4], const T (&vc)[4],
                                              const T (&us)[4], const T (&vs)[4], const T (&un_)[4], const T (&vn_)[4],
                                              T (&uo)[4], T (&vo)[4]) {
  const T ul = lane_left(uc[3]), ur = lane_right(uc[0]), vl = lane_left(vc[3]), vr = lane_right(vc[0]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const T uw = (k == 0) ? ul : uc[k - 1], ue = (k == 3) ? ur : uc[k + 1];
    const T vw = (k == 0) ? vl : vc[k - 1], ve = (k == 3) ? vr : vc[k + 1];
    const T dudx = div_c(ue - uw, S.two_dx, S.inv_two_dx), dudy = div_c(un_[k] - us[k], S.two_dy, S.inv_two_dy);
    const T dvdx = div_c(ve - vw, S.two_dx, S.inv_two_dx), dvdy = div_c(vn_[k] - vs[k], S.two_dy, S.inv_two_dy);
    const T lapu = div_c((((uw + us[k]) - (T)4 * uc[k]) + ue) + un_[k], S.dxdy, S.inv_dxdy);
    const T lapv = div_c((((vw + vs[k]) - (T)4 * vc[k]) + ve) + vn_[k], S.dxdy, S.inv_dxdy);
    const T a = uc[k] + S.dt * (((-uc[k]) * dudx - vc[k] * dudy) + S.nu * lapu);
    const T d = vc[k] + S.dt * (((-uc[k]) * dvdx - vc[k] * dvdy) + S.nu * lapv);
    const bool edge = (i <= 0) || (i >= kN - 1) || (lane == 0 && k == 0) || (lane == 63 && k == 3);
    uo[k] = edge ? uc[k] : a;
    vo[k] = edge ? vc[k] : d;
  }
}

// apply_boundary (:76-90) restricted to grid row i, in registers.  `f` holds the row before the call (only its interior
// cells matter), `nb` the row next to it on the inside of the wall -- read only when i is the lower / upper wall row.  Passes
// in the reference's order: lower, upper (whole row), then left, right (one cell each, reading the cell the earlier pass set).
// The left / right passes run for every row of the pipelines, so they are selects on lane masks formed once per launch
// (BcSel) instead of branches on the boundary codes; the wall rows (two per instance) keep the branching form.
struct BcSel {
  bool ln[2], lw[2], rn[2], rw[2];   // per component: lane 0 takes its right neighbour (Neumann) / the wall value; lane 63 alike
  bool ld[2], rd[2];                 // wall value is 0 (Dirichlet) rather than the action (Controllable); wave-uniform
};
__device__ __forceinline__ BcSel make_bc_sel(const int (&bc)[4][2], int lane) {
  BcSel m;
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    const int cl = bc[PDEGYM_EDGE_LEFT][comp], cr = bc[PDEGYM_EDGE_RIGHT][comp];
    m.ln[comp] = lane == 0 && cl == PDEGYM_BC_NEUMANN;
    m.lw[comp] = lane == 0 && cl != PDEGYM_BC_NEUMANN;
    m.rn[comp] = lane == 63 && cr == PDEGYM_BC_NEUMANN;
    m.rw[comp] = lane == 63 && cr != PDEGYM_BC_NEUMANN;
    m.ld[comp] = cl == PDEGYM_BC_DIRICHLET;
    m.rd[comp] = cr == PDEGYM_BC_DIRICHLET;
  }
  return m;
}

template <typename T>
__device__ __forceinline__ void bc_row(T (&f)[4], const T (&nb)[4], int i, int c0, const int (&bc)[4][2], int comp,
                                       const BcSel& m, const T* act, int action_dim, T a0) {
  if (i == 0 || i == kN - 1) {        // wave-uniform, two rows per instance
    const int c = bc[i == 0 ? PDEGYM_EDGE_LOWER : PDEGYM_EDGE_UPPER][comp];
    if (c == PDEGYM_BC_NEUMANN) {
#pragma unroll
      for (int k = 0; k < 4; ++k) f[k] = nb[k];
    } else if (c == PDEGYM_BC_DIRICHLET) {
#pragma unroll
      for (int k = 0; k < 4; ++k) f[k] = (T)0;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) f[k] = action_dim == 1 ? a0 : act[c0 + k];
    }
  }
  // rows outside the grid pass through here on the pipelines' first / last iterations: keep the action index inside the array
  T ai = a0;
  if (action_dim != 1) ai = act[i < 0 ? 0 : (i > kN - 1 ? kN - 1 : i)];
  const T wl = m.ld[comp] ? (T)0 : ai, wr = m.rd[comp] ? (T)0 : ai;
  f[0] = m.ln[comp] ? f[1] : f[0];
  f[0] = m.lw[comp] ? wl : f[0];
  f[3] = m.rn[comp] ? f[2] : f[3];
  f[3] = m.rw[comp] ? wr : f[3];
}

}  // namespace rows256
}  // namespace ns
}  // namespace pdegym
