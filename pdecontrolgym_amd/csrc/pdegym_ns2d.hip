// pdegym_ns2d.hip -- gfx950 kernels for the 2D Navier-Stokes environment (collocated grid, Chorin
// projection: predictor -> Jacobi pressure Poisson -> corrector), one env-step of B instances per launch.
//
// Reference semantics restated (environments2d/navier_stokes2D.py):
//   central_difference / laplace :9-22 (interior only, zero on the boundary), predictor :130-138,
//   apply_boundary :68-91 (edges lower, upper, left, right; component u then v), solve_pressure :94-116,
//   corrector :143-146, NSReward rewards/ns_reward.py:28.
//
// Two kernel families:
//   ns_generic<T>   any grid, T = float | double.  One workgroup owns one instance for the WHOLE step
//                   (all phases + K Jacobi sweeps) so the only synchronisation is __syncthreads(); fields
//                   ping-pong through caller scratch (L2-resident).  Operation order is exactly the
//                   reference's, so the float64 build is bit-faithful to NumPy.
//   ns_tile<...>    float32 throughput path for grids up to 128x128: every thread keeps an 8x8 patch of p
//                   and rhs in VGPRs for all K sweeps and only patch halos cross threads through LDS
//                   (see docs/HISTORY.md section 4).  Same arithmetic, reassociated only where stated.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "pdegym.h"
#include "pdegym_common.h"

#include "pdegym_ns_common.h"

namespace {

using namespace pdegym::ns;

// Dispatch overrides for tests and A/B runs (pdegym_debug_set, include/pdegym.h): the product never sets them.
int g_debug[PDEGYM_DEBUG_COUNT] = {0, 0, -1, 0};
inline bool pdegym_ns_no_col() { return g_debug[PDEGYM_DEBUG_NS_NO_COL] != 0; }               // small grids take ns_generic_step
inline bool pdegym_force_generic() { return g_debug[PDEGYM_DEBUG_NS_GENERIC] != 0; }         // every grid takes ns_generic_step
inline bool pdegym_no_lds_jacobi() { return g_debug[PDEGYM_DEBUG_NS_NO_LDS_JACOBI] != 0; }   // ns_generic_step sweeps in global memory



// K Jacobi sweeps on one instance: src/dst ping-pong, result in `p`. One workgroup; caller syncs before.
template <typename T>
__device__ __forceinline__ void jacobi_sweeps(T* p, T* pB, const T* rhs, int ny, int nx, int K, T dxdy) {
  const int ncell = ny * nx;
  T* src = p;
  T* dst = pB;
  if (K & 1) {  // odd K: start from a copy in pB so the last sweep lands in p
    for (int c = threadIdx.x; c < ncell; c += blockDim.x) pB[c] = p[c];
    __syncthreads();
    src = pB;
    dst = p;
  }
  for (int it = 0; it < K; ++it) {
    for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
      const int i = c / nx, j = c - i * nx;
      if (i >= 1 && i <= ny - 2 && j >= 1 && j <= nx - 2) {
        // navier_stokes2D.py:106-108   1/4 * (W + S + E + N - dx*dy*rhs)
        const T s4 = ((src[c - 1] + src[c - nx]) + src[c + 1]) + src[c + nx];
        T val;
        if constexpr (sizeof(T) == 4) val = jacobi_update(s4, jacobi_rhs_term(dxdy, rhs[c]));
        else val = (T)0.25 * (s4 - dxdy * rhs[c]);
        dst[c] = val;
        // navier_stokes2D.py:110-113: four Neumann copies => every boundary cell ends as its nearest interior value
        const bool top = (i == 1), bot = (i == ny - 2), lef = (j == 1), rig = (j == nx - 2);
        if (top) dst[c - nx] = val;
        if (bot) dst[c + nx] = val;
        if (lef) dst[c - 1] = val;
        if (rig) dst[c + 1] = val;
        if (top && lef) dst[c - nx - 1] = val;
        if (top && rig) dst[c - nx + 1] = val;
        if (bot && lef) dst[c + nx - 1] = val;
        if (bot && rig) dst[c + nx + 1] = val;
      }
    }
    __syncthreads();
    T* t = src;
    src = dst;
    dst = t;
  }
}

// Same sweeps for grids of at most kLdsCells cells: p ping-pongs between two LDS copies and each thread keeps the
// right-hand-side term of its (up to kLdsCPT) cells in registers, so a sweep costs LDS latency + one barrier instead of
// an L2 round trip (the shipped 21x21, K = 2000 example: 2000 dependent sweeps per env-step).  Identical arithmetic.
constexpr int kLdsCells = 4096;
constexpr int kLdsCPT = 4;

template <typename T>
__device__ __forceinline__ void jacobi_sweeps_lds(T* p, const T* rhs, T* sh, int ny, int nx, int K, T dxdy) {
  const int ncell = ny * nx;
  T* src = sh;
  T* dst = sh + ncell;
  T rq[kLdsCPT];
  int flag[kLdsCPT];   // bit0 interior, bit1 top, bit2 bottom, bit3 left, bit4 right
#pragma unroll
  for (int k = 0; k < kLdsCPT; ++k) {
    const int c = threadIdx.x + k * blockDim.x;
    flag[k] = 0;
    rq[k] = 0;
    if (c < ncell) {
      const T pv = p[c];
      src[c] = pv;
      dst[c] = pv;
      const int i = c / nx, j = c - i * nx;
      if (i >= 1 && i <= ny - 2 && j >= 1 && j <= nx - 2) {
        flag[k] = 1 | ((i == 1) << 1) | ((i == ny - 2) << 2) | ((j == 1) << 3) | ((j == nx - 2) << 4);
        if constexpr (sizeof(T) == 4) rq[k] = jacobi_rhs_term(dxdy, rhs[c]);
        else rq[k] = dxdy * rhs[c];
      }
    }
  }
  __syncthreads();
  for (int it = 0; it < K; ++it) {
#pragma unroll
    for (int k = 0; k < kLdsCPT; ++k) {
      const int f = flag[k];
      if (f) {
        const int c = threadIdx.x + k * blockDim.x;
        const T s4 = ((src[c - 1] + src[c - nx]) + src[c + 1]) + src[c + nx];
        T val;
        if constexpr (sizeof(T) == 4) val = jacobi_update(s4, rq[k]);
        else val = (T)0.25 * (s4 - rq[k]);
        dst[c] = val;
        if (f != 1) {
          const bool top = f & 2, bot = f & 4, lef = f & 8, rig = f & 16;
          if (top) dst[c - nx] = val;
          if (bot) dst[c + nx] = val;
          if (lef) dst[c - 1] = val;
          if (rig) dst[c + 1] = val;
          if (top && lef) dst[c - nx - 1] = val;
          if (top && rig) dst[c - nx + 1] = val;
          if (bot && lef) dst[c + nx - 1] = val;
          if (bot && rig) dst[c + nx + 1] = val;
        }
      }
    }
    __syncthreads();
    T* t = src;
    src = dst;
    dst = t;
  }
#pragma unroll
  for (int k = 0; k < kLdsCPT; ++k) {
    const int c = threadIdx.x + k * blockDim.x;
    if (c < ncell) p[c] = src[c];
  }
}

// Grids whose two copies do not fit the LDS (up to kLds1Cells cells, e.g. float64 128x128 = 128 KB): ONE LDS copy of p; a
// sweep computes the new values of a thread's cells into registers from the old copy, then -- after a barrier -- writes
// them (and the Neumann wall copies) back.  Two barriers per sweep instead of one, still no L2 round trip.  Same arithmetic.
constexpr int kLds1CPT = 16;
constexpr int kLds1Cells = 1024 * kLds1CPT;
constexpr int kLds1MaxBytes = 160 * 1024 - 1024;

template <typename T>
__device__ __forceinline__ void jacobi_sweeps_lds1(T* p, const T* rhs, T* buf, int ny, int nx, int K, T dxdy) {
  const int ncell = ny * nx;
  T rq[kLds1CPT];
  int flag[kLds1CPT];
#pragma unroll
  for (int k = 0; k < kLds1CPT; ++k) {
    const int c = threadIdx.x + k * blockDim.x;
    flag[k] = 0;
    rq[k] = 0;
    if (c < ncell) {
      buf[c] = p[c];
      const int i = c / nx, j = c - i * nx;
      if (i >= 1 && i <= ny - 2 && j >= 1 && j <= nx - 2) {
        flag[k] = 1 | ((i == 1) << 1) | ((i == ny - 2) << 2) | ((j == 1) << 3) | ((j == nx - 2) << 4);
        if constexpr (sizeof(T) == 4) rq[k] = jacobi_rhs_term(dxdy, rhs[c]);
        else rq[k] = dxdy * rhs[c];
      }
    }
  }
  __syncthreads();
  for (int it = 0; it < K; ++it) {
    T val[kLds1CPT];
#pragma unroll
    for (int k = 0; k < kLds1CPT; ++k) {
      val[k] = 0;
      if (flag[k]) {
        const int c = threadIdx.x + k * blockDim.x;
        const T s4 = ((buf[c - 1] + buf[c - nx]) + buf[c + 1]) + buf[c + nx];
        if constexpr (sizeof(T) == 4) val[k] = jacobi_update(s4, rq[k]);
        else val[k] = (T)0.25 * (s4 - rq[k]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kLds1CPT; ++k) {
      const int f = flag[k];
      if (f) {
        const int c = threadIdx.x + k * blockDim.x;
        const T v = val[k];
        buf[c] = v;
        if (f != 1) {
          const bool top = f & 2, bot = f & 4, lef = f & 8, rig = f & 16;
          if (top) buf[c - nx] = v;
          if (bot) buf[c + nx] = v;
          if (lef) buf[c - 1] = v;
          if (rig) buf[c + 1] = v;
          if (top && lef) buf[c - nx - 1] = v;
          if (top && rig) buf[c - nx + 1] = v;
          if (bot && lef) buf[c + nx - 1] = v;
          if (bot && rig) buf[c + nx + 1] = v;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < kLds1CPT; ++k) {
    const int c = threadIdx.x + k * blockDim.x;
    if (c < ncell) p[c] = buf[c];
  }
}

template <typename T>
__device__ __forceinline__ void compute_rhs(const T* us, const T* vs, T* rhs, int ny, int nx, const NSScal<T>& S) {
  const int ncell = ny * nx;
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int i = c / nx, j = c - i * nx;
    T r = 0;
    if (i >= 1 && i <= ny - 2 && j >= 1 && j <= nx - 2) {
      // navier_stokes2D.py:101-103   rho/dt * (d/dx u* + d/dy v*)
      const T dudx = div_c(us[c + 1] - us[c - 1], S.two_dx, S.inv_two_dx);
      const T dvdy = div_c(vs[c + nx] - vs[c - nx], S.two_dy, S.inv_two_dy);
      r = S.rho_over_dt * (dudx + dvdy);
    }
    rhs[c] = r;
  }
}

// ---- the phases of one step around the pressure solve, for one instance owned by one workgroup --------------
// front: predictor -> apply_boundary -> rhs      (u*, v*, rhs left in the instance's scratch quarters 0, 1, 2)
template <typename T>
__device__ __forceinline__ void gen_front(const NSConst& C, const NSScal<T>& S, const NSPtrs<T>& P, int b) {
  const int nx = C.nx, ny = C.ny, ncell = nx * ny;
  const T* u = (P.u && !P.state_in) ? P.u + (size_t)b * ncell : nullptr;
  const T* v = (P.v && !P.state_in) ? P.v + (size_t)b * ncell : nullptr;
  T* us = P.scratch + (size_t)b * 4 * ncell;
  T* vs = us + ncell;
  T* rhs = vs + ncell;
  const T* act = P.action + (size_t)b * C.action_dim;
  // state: separate u, v fields, or the interleaved observation of the previous call
  const T* sin = P.state_in ? P.state_in + (size_t)b * ncell * 2 : nullptr;
  auto U = [&](int c) -> T { return sin ? sin[2 * (size_t)c] : u[c]; };
  auto V = [&](int c) -> T { return sin ? sin[2 * (size_t)c + 1] : v[c]; };

  // ---- predictor (navier_stokes2D.py:130-138); boundary derivatives are zero so u* = u there (:9-22) ----
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int i = c / nx, j = c - i * nx;
    const T uc = U(c), vc = V(c);
    T un = uc, vn = vc;
    if (i >= 1 && i <= ny - 2 && j >= 1 && j <= nx - 2) {
      const T uw = U(c - 1), ue = U(c + 1), usn = U(c - nx), unn = U(c + nx);
      const T vw = V(c - 1), ve = V(c + 1), vsn = V(c - nx), vnn = V(c + nx);
      const T dudx = div_c(ue - uw, S.two_dx, S.inv_two_dx), dudy = div_c(unn - usn, S.two_dy, S.inv_two_dy);
      const T dvdx = div_c(ve - vw, S.two_dx, S.inv_two_dx), dvdy = div_c(vnn - vsn, S.two_dy, S.inv_two_dy);
      const T lapu = div_c((((uw + usn) - (T)4 * uc) + ue) + unn, S.dxdy, S.inv_dxdy);
      const T lapv = div_c((((vw + vsn) - (T)4 * vc) + ve) + vnn, S.dxdy, S.inv_dxdy);
      un = uc + S.dt * (((-uc) * dudx - vc * dudy) + S.nu * lapu);
      vn = vc + S.dt * (((-uc) * dvdx - vc * dvdy) + S.nu * lapv);
    }
    us[c] = un;
    vs[c] = vn;
  }
  __syncthreads();
  // ---- apply_boundary(u*, v*, action) (:140) ----
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int i = c / nx, j = c - i * nx;
    if (i == 0 || i == ny - 1 || j == 0 || j == nx - 1) {
      const T bu = bc_value<T>(us, i, j, ny, nx, C.bc, 0, act, C.action_dim);
      const T bv = bc_value<T>(vs, i, j, ny, nx, C.bc, 1, act, C.action_dim);
      us[c] = bu;  // bc_value reads interior cells only, boundary cells are only written: no hazard
      vs[c] = bv;
    }
  }
  __syncthreads();
  // ---- rhs of the pressure Poisson problem (:101-103) ----
  compute_rhs<T>(us, vs, rhs, ny, nx, S);
}

// back: corrector (with the solved pressure pfin) -> apply_boundary -> observation, reward, flags
template <typename T>
__device__ __forceinline__ void gen_back(const NSConst& C, const NSScal<T>& S, const NSPtrs<T>& P, int b, const T* pfin, T* red) {
  const int nx = C.nx, ny = C.ny, ncell = nx * ny;
  T* u = (P.u && !P.state_in) ? P.u + (size_t)b * ncell : nullptr;
  T* v = (P.v && !P.state_in) ? P.v + (size_t)b * ncell : nullptr;
  T* pdst = (P.p_out ? P.p_out : P.p) + (size_t)b * ncell;
  T* us = P.scratch + (size_t)b * 4 * ncell;
  T* vs = us + ncell;
  const T* act = P.action + (size_t)b * C.action_dim;
  // ---- corrector (:143-145): interior; the boundary keeps u* (zero pressure derivative) until the BC pass ----
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int i = c / nx, j = c - i * nx;
    T un = us[c], vn = vs[c];
    if (i >= 1 && i <= ny - 2 && j >= 1 && j <= nx - 2) {
      const T dpdx = div_c(pfin[c + 1] - pfin[c - 1], S.two_dx, S.inv_two_dx);
      const T dpdy = div_c(pfin[c + nx] - pfin[c - nx], S.two_dy, S.inv_two_dy);
      un = un - S.dt_over_rho * dpdx;
      vn = vn - S.dt_over_rho * dpdy;
    }
    us[c] = un;  // in place: each thread reads only its own u*, v* and neighbours of p
    vs[c] = vn;
    if (pfin != pdst) pdst[c] = pfin[c];  // the split pipeline may finish in the ping-pong buffer: p is the warm start
  }
  __syncthreads();
  // ---- apply_boundary(u, v, action) (:146), observation (:147-149,:154), reward (ns_reward.py:28) ----
  int t = P.time_index[b] + 1;
  const int tr = t < C.nt_ref ? t : C.nt_ref - 1;
  const T* uref = P.U_ref + (size_t)tr * ncell * 2;
  T* obs = P.obs + (size_t)b * ncell * 2;
  T acc = 0;
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int i = c / nx, j = c - i * nx;
    T un, vn;
    if (i == 0 || i == ny - 1 || j == 0 || j == nx - 1) {
      un = bc_value<T>(us, i, j, ny, nx, C.bc, 0, act, C.action_dim);   // reads interior cells only
      vn = bc_value<T>(vs, i, j, ny, nx, C.bc, 1, act, C.action_dim);
    } else {
      un = us[c];
      vn = vs[c];
    }
    if (u) {
      u[c] = un;
      v[c] = vn;
    }
    obs[2 * (size_t)c] = un;
    obs[2 * (size_t)c + 1] = vn;
    const T du = un - uref[2 * (size_t)c], dv = vn - uref[2 * (size_t)c + 1];
    acc += du * du;
    acc += dv * dv;
  }
  const T ss = block_sum<T>(acc, red);
  if (threadIdx.x == 0) {
    T asq = 0;
    const T aref = P.action_ref[tr];
    for (int k = 0; k < C.action_dim; ++k) {
      const T d = act[k] - aref;
      asq += d * d;
    }
    // - 1/2 * ||U - Uref||^2 / nx / ny - gamma/2 * ||a - aref||^2
    P.reward[b] = (((T)-0.5 * ss) / (T)nx) / (T)ny - S.gamma_half * asq;
    P.time_index[b] = t;
    P.terminated[b] = (t >= C.nt - 1) ? 1 : 0;  // navier_stokes2D.py:159-168
  }
}

template <typename T, int LDSJ>   // 0: global-memory Jacobi, 1: two LDS copies, 2: one LDS copy
__global__ __launch_bounds__(1024) void ns_generic_step(NSConst C, NSScal<T> S, NSPtrs<T> P, int B) {
  __shared__ T red[16];
  extern __shared__ double ns_dyn_lds[];
  const int b = blockIdx.x;
  if (b >= B) return;
  const int ncell = C.nx * C.ny;
  T* p = P.p + (size_t)b * ncell;
  T* rhs = P.scratch + (size_t)b * 4 * ncell + 2 * (size_t)ncell;
  T* pB = rhs + ncell;
  gen_front<T>(C, S, P, b);
  __syncthreads();
  // pressure Poisson (:142, :94-116), result in p
  if constexpr (LDSJ == 1) {
    jacobi_sweeps_lds<T>(p, rhs, reinterpret_cast<T*>(ns_dyn_lds), C.ny, C.nx, C.iters, S.dxdy);
    __syncthreads();
  } else if constexpr (LDSJ == 2) {
    jacobi_sweeps_lds1<T>(p, rhs, reinterpret_cast<T*>(ns_dyn_lds), C.ny, C.nx, C.iters, S.dxdy);
    __syncthreads();
  } else {
    jacobi_sweeps<T>(p, pB, rhs, C.ny, C.nx, C.iters, S.dxdy);
  }
  gen_back<T>(C, S, P, b, p, red);
}

// ================================================================================================
// float32 register-tiled path: 512 threads per instance, thread (ty, tx) (16 x 32) owns the PR x PC patch at
// (ty*PR, tx*PC) in VGPRs:  (PR,PC) = (8,4) -> 128x128,  (4,2) -> 64x64.
//   * left/right patch halos: neighbouring tx are neighbouring LANES (a wave = 2 thread rows of 32 = two full
//     domain rows of patches), so they move with DPP wave shifts -- no LDS;
//   * top/bottom patch halos: one PC-wide vector per thread through a double-buffered LDS area (32 KB), one
//     barrier per exchange;
//   * p and 0.25*dx*dy*rhs stay in registers for all K Jacobi sweeps on a ROTATING row map (no register copies);
//     u*, v* wait on chip meanwhile (v* and two u* rows in registers, six u* rows in LDS);
//   * 128 VGPRs -> 4 waves per SIMD, two instances per CU.
// Same expression tree as ns_generic<float> (the fma below is exact-equivalent), so both agree bit for bit.
// ================================================================================================

template <int PR, int PC>
struct Halo {
  float t[PC], b[PC], l[PR], r[PR];
};

template <int PR, int PC>
__device__ __forceinline__ void halo_exchange(const float (&f)[PR][PC], Halo<PR, PC>& H, float* lds, int& xc, int tid, int ty) {
  halo_tb<PC>(f[0], f[PR - 1], H.t, H.b, lds, xc, tid, ty);
#pragma unroll
  for (int a = 0; a < PR; ++a) {
    H.l[a] = lane_left(f[a][PC - 1]);
    H.r[a] = lane_right(f[a][0]);
  }
}

template <int PR, int PC>
__device__ __forceinline__ void load_patch(float (&f)[PR][PC], const float* g, int n, int r0, int c0) {
  using V = typename VecOf<PC>::type;
#pragma unroll
  for (int a = 0; a < PR; ++a) unpack_row<PC>(*reinterpret_cast<const V*>(g + ((r0 + a) * n + c0)), f[a]);
}

template <int PR, int PC>
__device__ __forceinline__ void store_patch(const float (&f)[PR][PC], float* g, int n, int r0, int c0) {
  using V = typename VecOf<PC>::type;
#pragma unroll
  for (int a = 0; a < PR; ++a) *reinterpret_cast<V*>(g + ((r0 + a) * n + c0)) = pack_row<PC>(f[a]);
}

// apply_boundary on a patch: the four ordered passes (lower, upper, left, right) only touch cells of edge
// threads and only read the line next to the edge, which lives in the same patch -> no communication.
template <int PR, int PC>
__device__ __forceinline__ void apply_bc_patch(float (&f)[PR][PC], const EdgeFlags& E, const int (&bc)[4][2], int comp,
                                               const float* act, int action_dim, int r0, int c0) {
  auto aval = [&](int idx) -> float { return action_dim == 1 ? act[0] : act[idx]; };
  if (E.top) {
    const int c = bc[PDEGYM_EDGE_LOWER][comp];
#pragma unroll
    for (int b = 0; b < PC; ++b) f[0][b] = (c == PDEGYM_BC_NEUMANN) ? f[1][b] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0f : aval(c0 + b));
  }
  if (E.bot) {
    const int c = bc[PDEGYM_EDGE_UPPER][comp];
#pragma unroll
    for (int b = 0; b < PC; ++b)
      f[PR - 1][b] = (c == PDEGYM_BC_NEUMANN) ? f[PR - 2][b] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0f : aval(c0 + b));
  }
  if (E.lef) {
    const int c = bc[PDEGYM_EDGE_LEFT][comp];
#pragma unroll
    for (int a = 0; a < PR; ++a) f[a][0] = (c == PDEGYM_BC_NEUMANN) ? f[a][1] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0f : aval(r0 + a));
  }
  if (E.rig) {
    const int c = bc[PDEGYM_EDGE_RIGHT][comp];
#pragma unroll
    for (int a = 0; a < PR; ++a)
      f[a][PC - 1] = (c == PDEGYM_BC_NEUMANN) ? f[a][PC - 2] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0f : aval(r0 + a));
  }
}

// apply_boundary restricted to HR consecutive patch rows held in f (first_half: they start at patch row 0, last_half: they
// end at patch row PR-1).  Same four ordered passes as apply_bc_patch: the lower / upper pass of the block's edge row
// reads the neighbouring row of the same block (HR >= 2), the left / right passes are row-local, so running the passes
// block by block gives the values of the whole-patch version.
template <int HR, int PC, typename T>
__device__ __forceinline__ void apply_bc_rows(T (&f)[HR][PC], const EdgeFlags& E, const int (&bc)[4][2], int comp,
                                              const T* act, int action_dim, int row0, int c0, bool first_half,
                                              bool last_half) {
  static_assert(HR >= 2, "the edge row's neighbour must be in the block");
  auto aval = [&](int idx) -> T { return action_dim == 1 ? act[0] : act[idx]; };
  if (first_half && E.top) {
    const int c = bc[PDEGYM_EDGE_LOWER][comp];
#pragma unroll
    for (int b = 0; b < PC; ++b) f[0][b] = (c == PDEGYM_BC_NEUMANN) ? f[1][b] : ((c == PDEGYM_BC_DIRICHLET) ? T(0) : aval(c0 + b));
  }
  if (last_half && E.bot) {
    const int c = bc[PDEGYM_EDGE_UPPER][comp];
#pragma unroll
    for (int b = 0; b < PC; ++b)
      f[HR - 1][b] = (c == PDEGYM_BC_NEUMANN) ? f[HR - 2][b] : ((c == PDEGYM_BC_DIRICHLET) ? T(0) : aval(c0 + b));
  }
  if (E.lef) {
    const int c = bc[PDEGYM_EDGE_LEFT][comp];
#pragma unroll
    for (int a = 0; a < HR; ++a) f[a][0] = (c == PDEGYM_BC_NEUMANN) ? f[a][1] : ((c == PDEGYM_BC_DIRICHLET) ? T(0) : aval(row0 + a));
  }
  if (E.rig) {
    const int c = bc[PDEGYM_EDGE_RIGHT][comp];
#pragma unroll
    for (int a = 0; a < HR; ++a)
      f[a][PC - 1] = (c == PDEGYM_BC_NEUMANN) ? f[a][PC - 2] : ((c == PDEGYM_BC_DIRICHLET) ? T(0) : aval(row0 + a));
  }
}

template <int PR, int PC>
__device__ __forceinline__ bool on_domain_edge(const EdgeFlags& E, int a, int b) {
  return (a == 0 && E.top) || (a == PR - 1 && E.bot) || (b == 0 && E.lef) || (b == PC - 1 && E.rig);
}

// ---- Jacobi sweep on a ROTATING register file ------------------------------------------------------------
// ph holds PR+1 physical rows: the PR patch rows plus the row above (top halo).  The new value of row a is
// written into the registers of OLD row a-1 (dead once row a has been computed; new row 0 goes into the halo
// row), so a sweep needs no register copies; the logical->physical row map shifts by one per sweep and returns
// to the identity after PR+1 sweeps (the sweep loop is unrolled PR+1 times over the rotation R).
template <int PR>
__device__ constexpr int prow(int a, int r) {
  return (((a - r) % (PR + 1)) + (PR + 1)) % (PR + 1);
}

template <int PR, int PC, int R, int NT = 512, int RS = 32>
__device__ __forceinline__ void jacobi_sweep_rot(float (&ph)[PR + 1][PC], const float (&rq)[PR][PC], const EdgeFlags& E,
                                                 float* lds, int& xc, int tid, int ty) {
  float hb[PC];
  halo_tb<PC, NT, RS>(ph[prow<PR>(0, R)], ph[prow<PR>(PR - 1, R)], ph[prow<PR>(-1, R)], hb, lds, xc, tid, ty);
  // The new row a overwrites the registers of old row a-1, but the RIGHT neighbour lane still needs this lane's old
  // first column of row a (and the left one its last column) -> sample the edge columns before the row is updated.
#pragma unroll
  for (int a = 0; a < PR; ++a) {
    const int src = prow<PR>(a, R), dst = prow<PR>(a - 1, R), nxt = prow<PR>(a + 1, R);
    // ((W + S) + E) + N   (navier_stokes2D.py:106-108), stage-major over the PC independent cells of the row so that
    // consecutive instructions do not depend on each other; W / E of the edge columns ride in on the DPP adds
    float nv[PC];
    if constexpr (PC == 4) {
      // The whole row update is ONE 16-instruction block that works in place on the registers of old row a-1 (the South
      // neighbours, dead afterwards): ((W + S) + E) + N, then fma(0.25, s4, -rq) -- stage-major over the four independent
      // cells, the two DPP adds at the end of their four-instruction groups (the three plain adds in front of them are the
      // wait states of the VALU-write -> DPP-read hazard; no s_nop slots).  One block = no temporaries and nothing for the
      // scheduler to hoist: the compiler used to start all PR rows at once and spill.  Same operations, same order as the
      // generic form below.
      float (&sv)[PC] = ph[dst];
      const float (&xv)[PC] = ph[src];
      const float (&nn)[PC] = (a == PR - 1) ? hb : ph[nxt];
      asm volatile(
          "v_add_f32 %1, %4, %1\n\t"
          "v_add_f32 %2, %5, %2\n\t"
          "v_add_f32 %3, %6, %3\n\t"
          "v_add_f32_dpp %0, %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_add_f32 %0, %0, %5\n\t"
          "v_add_f32 %1, %1, %6\n\t"
          "v_add_f32 %2, %2, %7\n\t"
          "v_add_f32_dpp %3, %4, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_add_f32 %0, %0, %8\n\t"
          "v_add_f32 %1, %1, %9\n\t"
          "v_add_f32 %2, %2, %10\n\t"
          "v_add_f32 %3, %3, %11\n\t"
          "v_fma_f32 %0, %0, %16, -%12\n\t"
          "v_fma_f32 %1, %1, %16, -%13\n\t"
          "v_fma_f32 %2, %2, %16, -%14\n\t"
          "v_fma_f32 %3, %3, %16, -%15"
          : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3])
          : "v"(xv[0]), "v"(xv[1]), "v"(xv[2]), "v"(xv[3]), "v"(nn[0]), "v"(nn[1]), "v"(nn[2]), "v"(nn[3]),
            "v"(rq[a][0]), "v"(rq[a][1]), "v"(rq[a][2]), "v"(rq[a][3]), "s"(0.25f));
      continue;
    } else {
#pragma unroll
      for (int k = 0; k < PC; ++k)
        nv[k] = (k == 0) ? add_lane_left(ph[src][PC - 1], ph[dst][k]) : (ph[src][k - 1] + ph[dst][k]);
#pragma unroll
      for (int k = 0; k < PC; ++k) nv[k] = (k == PC - 1) ? add_lane_right(ph[src][0], nv[k]) : (nv[k] + ph[src][k + 1]);
    }
#pragma unroll
    for (int k = 0; k < PC; ++k) nv[k] = nv[k] + ((a == PR - 1) ? hb[k] : ph[nxt][k]);
#pragma unroll
    for (int k = 0; k < PC; ++k) nv[k] = jacobi_update(nv[k], rq[a][k]);
#pragma unroll
    for (int k = 0; k < PC; ++k) ph[dst][k] = nv[k];
  }
  // Neumann walls (:110-113) on the NEW rows (rotation R+1): every boundary cell = nearest interior value
  constexpr int n0 = prow<PR>(0, R + 1), n1 = prow<PR>(1, R + 1), nl = prow<PR>(PR - 1, R + 1), nm = prow<PR>(PR - 2, R + 1);
  if (E.top) {
#pragma unroll
    for (int k = 0; k < PC; ++k) ph[n0][k] = ph[n1][k];
  }
  if (E.bot) {
#pragma unroll
    for (int k = 0; k < PC; ++k) ph[nl][k] = ph[nm][k];
  }
  if (E.lef) {
#pragma unroll
    for (int a = 0; a < PR; ++a) ph[prow<PR>(a, R + 1)][0] = ph[prow<PR>(a, R + 1)][1];
  }
  if (E.rig) {
#pragma unroll
    for (int a = 0; a < PR; ++a) ph[prow<PR>(a, R + 1)][PC - 1] = ph[prow<PR>(a, R + 1)][PC - 2];
  }
}

// up to PR+1 sweeps, one per rotation; returns the rotation at which K was reached, or -1 after a full cycle
// RC = sweeps per cycle.  RC == PR+1 closes the rotation by itself; a shorter cycle copies the rows back to the identity
// map after RC sweeps (PR*PC moves per RC sweeps) -- used where the fully unrolled cycle is too large for the compiler
// to keep the rotating array in registers.
template <int PR, int PC, int R, int NT = 512, int RS = 32, int RC = PR + 1>
struct SweepChain {
  static __device__ __forceinline__ int run(float (&ph)[PR + 1][PC], const float (&rq)[PR][PC], const EdgeFlags& E, float* lds,
                                            int& xc, int tid, int ty, int& it, int K) {
    if (it >= K) return R;
    jacobi_sweep_rot<PR, PC, R, NT, RS>(ph, rq, E, lds, xc, tid, ty);
    ++it;
    if constexpr (R == RC - 1) {
      if constexpr (RC != PR + 1) {
        float tmp[PR][PC];
#pragma unroll
        for (int a = 0; a < PR; ++a)
#pragma unroll
          for (int k = 0; k < PC; ++k) tmp[a][k] = ph[prow<PR>(a, RC)][k];
#pragma unroll
        for (int a = 0; a < PR; ++a)
#pragma unroll
          for (int k = 0; k < PC; ++k) ph[a][k] = tmp[a][k];
      }
      return -1;
    } else {
      return SweepChain<PR, PC, R + 1, NT, RS, RC>::run(ph, rq, E, lds, xc, tid, ty, it, K);
    }
  }
};

template <int PR, int PC, int R>
__device__ __forceinline__ void unrotate(const float (&ph)[PR + 1][PC], float (&pf)[PR][PC], int rot) {
  if (rot == R) {
#pragma unroll
    for (int a = 0; a < PR; ++a)
#pragma unroll
      for (int k = 0; k < PC; ++k) pf[a][k] = ph[prow<PR>(a, R)][k];
  } else if constexpr (R < PR) {
    unrotate<PR, PC, R + 1>(ph, pf, rot);
  }
}


#ifndef PDEGYM_NS_F64_TILE_ROWS
#define PDEGYM_NS_F64_TILE_ROWS 16
#endif
#ifndef PDEGYM_NS_BACK_ROWS
#define PDEGYM_NS_BACK_ROWS 2
#endif
template <int PR, int PC, bool INTERLEAVED>
__global__ __launch_bounds__(512, 4) void ns_tile_step(NSConst C, NSScal<float> S, NSPtrs<float> P, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const int b = blockIdx.x;
  if (b >= B) return;
  constexpr int n = TileCfg<PR, PC>::N;
  constexpr int ncell = n * n;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int r0 = ty * PR, c0 = tx * PC;
  const EdgeFlags E{ty == 0, ty == 15, tx == 0, tx == 31};
  float* u = INTERLEAVED ? nullptr : P.u + (size_t)b * ncell;
  float* v = INTERLEAVED ? nullptr : P.v + (size_t)b * ncell;
  float* p = P.p + (size_t)b * ncell;
  const float* act = P.action + (size_t)b * C.action_dim;
  const float* sin = INTERLEAVED ? P.state_in + (size_t)b * ncell * 2 : nullptr;
  int xc = 0;
#ifdef PDEGYM_TIMING
  unsigned long long tm[8];
  tm[0] = __builtin_amdgcn_s_memtime();
#define PDEGYM_STAMP(i, dep) tm[i] = __builtin_amdgcn_s_memtime() + (unsigned long long)((dep) != (dep))
#else
#define PDEGYM_STAMP(i, dep)
#endif

  float rq[PR][PC];  // 0.25*dx*dy*rhs, kept for all sweeps
  // u* and v* wait on chip for the corrector: v* and two u* rows in registers (next to rq and p they fill 108 of the 128
  // registers during the sweeps), the other six u* rows in this thread's LDS slots -- nothing is parked in the caller's scratch
  // (until the end of round 3 v* and three u* rows went there and back: 176 KB of HBM traffic per 128 x 128 env-step)
  float vkeep[PR][PC];
  constexpr int kUKeep = PR - TileCfg<PR, PC>::PARK_ROWS > 0 ? PR - TileCfg<PR, PC>::PARK_ROWS : 1;
  float ukeep[kUKeep][PC];
  {
    float uf[PR][PC], vf[PR][PC];
    if constexpr (INTERLEAVED) {  // (u, v) interleaved: the previous call's observation IS the state
#pragma unroll
      for (int a = 0; a < PR; ++a) {
        const float4* row = reinterpret_cast<const float4*>(sin + ((r0 + a) * n + c0) * 2);
#pragma unroll
        for (int q = 0; q < PC / 2; ++q) {
          const float4 w = row[q];
          uf[a][2 * q] = w.x; vf[a][2 * q] = w.y; uf[a][2 * q + 1] = w.z; vf[a][2 * q + 1] = w.w;
        }
      }
    } else {
      load_patch<PR, PC>(uf, u, n, r0, c0);
      load_patch<PR, PC>(vf, v, n, r0, c0);
    }
    // ---- predictor (navier_stokes2D.py:130-138) ----
    // Only the top/bottom halo rows are kept; left/right neighbours are fetched per row with DPP right before the
    // row is overwritten in place (all lanes of a wave update the same row in lockstep, so the neighbour lane still
    // holds its OLD row) -- this keeps the phase inside the 128-VGPR budget.
    {
      float ut[PC], ub[PC], vt[PC], vb[PC];
      halo_tb<PC>(uf[0], uf[PR - 1], ut, ub, lds, xc, tid, ty);
      halo_tb<PC>(vf[0], vf[PR - 1], vt, vb, lds, xc, tid, ty);
      float pu[PC], pv[PC];  // old row a-1
#pragma unroll
      for (int k = 0; k < PC; ++k) { pu[k] = ut[k]; pv[k] = vt[k]; }
#pragma unroll
      for (int a = 0; a < PR; ++a) {
        float cu[PC], cv[PC];
#pragma unroll
        for (int k = 0; k < PC; ++k) { cu[k] = uf[a][k]; cv[k] = vf[a][k]; }
        const float ul = lane_left(cu[PC - 1]), ur = lane_right(cu[0]);
        const float vl = lane_left(cv[PC - 1]), vr = lane_right(cv[0]);
#pragma unroll
        for (int k = 0; k < PC; ++k) {
          const float uc = cu[k], vc = cv[k];
          const float uw = (k == 0) ? ul : cu[k - 1], ue = (k == PC - 1) ? ur : cu[k + 1];
          const float vw = (k == 0) ? vl : cv[k - 1], ve = (k == PC - 1) ? vr : cv[k + 1];
          const float usn = pu[k], vsn = pv[k];
          const float unn = (a == PR - 1) ? ub[k] : uf[a + 1][k];
          const float vnn = (a == PR - 1) ? vb[k] : vf[a + 1][k];
          const float dudx = div_c(ue - uw, S.two_dx, S.inv_two_dx), dudy = div_c(unn - usn, S.two_dy, S.inv_two_dy);
          const float dvdx = div_c(ve - vw, S.two_dx, S.inv_two_dx), dvdy = div_c(vnn - vsn, S.two_dy, S.inv_two_dy);
          const float lapu = div_c((((uw + usn) - 4.0f * uc) + ue) + unn, S.dxdy, S.inv_dxdy);
          const float lapv = div_c((((vw + vsn) - 4.0f * vc) + ve) + vnn, S.dxdy, S.inv_dxdy);
          const float un = uc + S.dt * (((-uc) * dudx - vc * dudy) + S.nu * lapu);
          const float vn = vc + S.dt * (((-uc) * dvdx - vc * dvdy) + S.nu * lapv);
          const bool edge = on_domain_edge<PR, PC>(E, a, k);
          uf[a][k] = edge ? uc : un;
          vf[a][k] = edge ? vc : vn;
        }
#pragma unroll
        for (int k = 0; k < PC; ++k) { pu[k] = cu[k]; pv[k] = cv[k]; }
        // one row at a time: the new row is pinned here (an empty asm that "uses" it) -- left alone the compiler sinks every row's
        // arithmetic towards its uses in the later phases while the lane shifts (convergent, cannot sink) stay put, so the shifted
        // values of ALL rows are live at once and spill
        if constexpr (PC == 4) {
          asm volatile("" : "+v"(uf[a][0]), "+v"(uf[a][1]), "+v"(uf[a][2]), "+v"(uf[a][3]), "+v"(vf[a][0]), "+v"(vf[a][1]), "+v"(vf[a][2]),
                       "+v"(vf[a][3]));
        }
      }
    }
    PDEGYM_STAMP(1, uf[0][0]);
    // ---- apply_boundary(u*, v*) (:140) ----
    apply_bc_patch<PR, PC>(uf, E, C.bc, 0, act, C.action_dim, r0, c0);
    apply_bc_patch<PR, PC>(vf, E, C.bc, 1, act, C.action_dim, r0, c0);
    {
      using V = typename VecOf<PC>::type;
      V* park = reinterpret_cast<V*>(smem_raw + TileCfg<PR, PC>::LDS_BYTES);
#pragma unroll
      for (int a = 0; a < PR; ++a) {
        if (a < TileCfg<PR, PC>::PARK_ROWS) {
          park[a * 512 + tid] = pack_row<PC>(uf[a]);      // read back by this thread only
        } else {
#pragma unroll
          for (int k = 0; k < PC; ++k) ukeep[a - TileCfg<PR, PC>::PARK_ROWS][k] = uf[a][k];
        }
      }
    }
#pragma unroll
    for (int a = 0; a < PR; ++a)
#pragma unroll
      for (int k = 0; k < PC; ++k) vkeep[a][k] = vf[a][k];
    // ---- rhs (:101-103), pre-multiplied by 0.25*dx*dy (:108) ----
    {
      float vt[PC], vb[PC], dummy_t[PC], dummy_b[PC];
      (void)dummy_t;
      (void)dummy_b;
      halo_tb<PC>(vf[0], vf[PR - 1], vt, vb, lds, xc, tid, ty);   // d/dy v* needs the rows above / below
#pragma unroll
      for (int a = 0; a < PR; ++a) {
        const float ul = lane_left(uf[a][PC - 1]), ur = lane_right(uf[a][0]);   // d/dx u* needs the lanes left / right
#pragma unroll
        for (int k = 0; k < PC; ++k) {
          const float uw = (k == 0) ? ul : uf[a][k - 1], ue = (k == PC - 1) ? ur : uf[a][k + 1];
          const float vsn = (a == 0) ? vt[k] : vf[a - 1][k], vnn = (a == PR - 1) ? vb[k] : vf[a + 1][k];
          const float dudx = div_c(ue - uw, S.two_dx, S.inv_two_dx);
          const float dvdy = div_c(vnn - vsn, S.two_dy, S.inv_two_dy);
          const float r = S.rho_over_dt * (dudx + dvdy);
          rq[a][k] = on_domain_edge<PR, PC>(E, a, k) ? 0.0f : jacobi_rhs_term(S.dxdy, r);
        }
      }
    }
  }
  PDEGYM_STAMP(2, rq[0][0]);

  // ---- K Jacobi sweeps (:104-114), p and rq in registers, rotating row map (no copies) ----
  float pf[PR][PC];
  {
    float ph[PR + 1][PC];
    {
      float tmp[PR][PC];
      load_patch<PR, PC>(tmp, p, n, r0, c0);
#pragma unroll
      for (int a = 0; a < PR; ++a)
#pragma unroll
        for (int k = 0; k < PC; ++k) ph[a][k] = tmp[a][k];
#pragma unroll
      for (int k = 0; k < PC; ++k) ph[PR][k] = 0.f;
    }
    if constexpr (PC == 4) {
      // period-two rotation (UP / DOWN sweeps, two-row in-place blocks): see jacobi_sweep_bous
      int it = 0;
      for (; it + 2 <= C.iters; it += 2) {
        jacobi_sweep_bous<PR, 0, 512, 32>(ph, rq, E, lds, xc, tid, ty);
        jacobi_sweep_bous<PR, 1, 512, 32>(ph, rq, E, lds, xc, tid, ty);
      }
      if (it < C.iters) {
        jacobi_sweep_bous<PR, 0, 512, 32>(ph, rq, E, lds, xc, tid, ty);
#pragma unroll
        for (int a = 0; a < PR; ++a)
#pragma unroll
          for (int k = 0; k < PC; ++k) pf[a][k] = ph[bphys<PR>(a, 1)][k];
      } else {
#pragma unroll
        for (int a = 0; a < PR; ++a)
#pragma unroll
          for (int k = 0; k < PC; ++k) pf[a][k] = ph[a][k];
      }
    } else {
      int it = 0, rot = 0;
      while (true) {
        const int r = SweepChain<PR, PC, 0>::run(ph, rq, E, lds, xc, tid, ty, it, C.iters);
        if (r >= 0) {
          rot = r;
          break;
        }
      }
      unrotate<PR, PC, 0>(ph, pf, rot);
    }
  }
  PDEGYM_STAMP(3, pf[0][0]);
  store_patch<PR, PC>(pf, P.p_out ? P.p_out + (size_t)b * ncell : p, n, r0, c0);

  // ---- corrector (:143-146), observation, reward ----
  float acc = 0.f;
  const int t = P.time_index[b] + 1;
  const int tr = t < C.nt_ref ? t : C.nt_ref - 1;
  {
    // Blocks of HR rows: u*, v* and the reference rows of a block are fetched, corrected, bounded, written and reduced
    // before the next block's loads are issued -- the whole-patch form (64 + 32 more live registers next to pf) spilled the
    // freshly loaded rows to scratch memory and read them back (80 bytes per lane of extra HBM traffic each way).
    constexpr int HR = PDEGYM_NS_BACK_ROWS, NBLK = PR / HR;
    static_assert(PR % HR == 0, "row blocks must tile the patch");
    // thread coordinates re-derived from an opaque copy: kept from the top of the kernel, the patch offsets and edge flags are
    // live (spilled) across the predictor and the sweeps
    int tid_c = threadIdx.x;
    asm volatile("" : "+v"(tid_c));
    const int tx_c = tid_c & 31, ty_c = tid_c >> 5;
    const int r0c = ty_c * PR, c0c = tx_c * PC;
    const EdgeFlags Ec{ty_c == 0, ty_c == 15, tx_c == 0, tx_c == 31};
    float pt[PC], pb[PC];
    halo_tb<PC>(pf[0], pf[PR - 1], pt, pb, lds, xc, tid_c, ty_c);
    const float* uref = P.U_ref + (size_t)tr * ncell * 2;
    float* obs = P.obs + (size_t)b * ncell * 2;
#pragma unroll
    for (int h = 0; h < NBLK; ++h) {
      const int a0 = h * HR;
      float uf[HR][PC], vf[HR][PC];
      using V = typename VecOf<PC>::type;
#pragma unroll
      for (int la = 0; la < HR; ++la) {     // written by this same thread above
        if (a0 + la < TileCfg<PR, PC>::PARK_ROWS)
          unpack_row<PC>(reinterpret_cast<const V*>(smem_raw + TileCfg<PR, PC>::LDS_BYTES)[(a0 + la) * 512 + tid_c], uf[la]);
        else {
#pragma unroll
          for (int k = 0; k < PC; ++k) uf[la][k] = ukeep[a0 + la - TileCfg<PR, PC>::PARK_ROWS][k];
        }
#pragma unroll
        for (int k = 0; k < PC; ++k) vf[la][k] = vkeep[a0 + la][k];
      }
#pragma unroll
      for (int la = 0; la < HR; ++la) {
        const int a = a0 + la;
        const float pl = lane_left(pf[a][PC - 1]), pr = lane_right(pf[a][0]);
#pragma unroll
        for (int k = 0; k < PC; ++k) {
          const float pw = (k == 0) ? pl : pf[a][k - 1], pe = (k == PC - 1) ? pr : pf[a][k + 1];
          const float ps = (a == 0) ? pt[k] : pf[a - 1][k], pn = (a == PR - 1) ? pb[k] : pf[a + 1][k];
          const float dpdx = div_c(pe - pw, S.two_dx, S.inv_two_dx);
          const float dpdy = div_c(pn - ps, S.two_dy, S.inv_two_dy);
          const bool edge = on_domain_edge<PR, PC>(Ec, a, k);
          uf[la][k] = edge ? uf[la][k] : uf[la][k] - S.dt_over_rho * dpdx;
          vf[la][k] = edge ? vf[la][k] : vf[la][k] - S.dt_over_rho * dpdy;
        }
      }
      apply_bc_rows<HR, PC>(uf, Ec, C.bc, 0, act, C.action_dim, r0c + a0, c0c, h == 0, h == NBLK - 1);
      apply_bc_rows<HR, PC>(vf, Ec, C.bc, 1, act, C.action_dim, r0c + a0, c0c, h == 0, h == NBLK - 1);
#pragma unroll
      for (int la = 0; la < HR; ++la) {
        if constexpr (!INTERLEAVED) {
          *reinterpret_cast<V*>(u + ((r0c + a0 + la) * n + c0c)) = pack_row<PC>(uf[la]);
          *reinterpret_cast<V*>(v + ((r0c + a0 + la) * n + c0c)) = pack_row<PC>(vf[la]);
        }
        const int o = ((r0c + a0 + la) * n + c0c) * 2;
        const float4* rrow = reinterpret_cast<const float4*>(uref + o);
        float4* orow = reinterpret_cast<float4*>(obs + o);
#pragma unroll
        for (int q = 0; q < PC / 2; ++q) {
          const float4 w = rrow[q];
          const float a0v = uf[la][2 * q], b0 = vf[la][2 * q], a1 = uf[la][2 * q + 1], b1 = vf[la][2 * q + 1];
          orow[q] = make_float4(a0v, b0, a1, b1);
          const float d0 = a0v - w.x, d1 = b0 - w.y, d2 = a1 - w.z, d3 = b1 - w.w;
          acc += d0 * d0;
          acc += d1 * d1;
          acc += d2 * d2;
          acc += d3 * d3;
        }
      }
      // pin the block's share of the reward sum here: left alone, the compiler sinks the whole sum of squares below the
      // last block and keeps (spills) every reference row until then
      asm volatile("" : "+v"(acc));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  PDEGYM_STAMP(4, acc);
  float* red = lds;   // halo buffers are idle now (block_sum syncs first)
  const float ss = block_sum<float>(acc, red);
  if (tid == 0) {
    float asq = 0.f;
    const float aref = P.action_ref[tr];
    for (int k = 0; k < C.action_dim; ++k) {
      const float d = act[k] - aref;
      asq += d * d;
    }
    P.reward[b] = ((-0.5f * ss) / (float)n) / (float)n - S.gamma_half * asq;
    P.time_index[b] = t;
    P.terminated[b] = (t >= C.nt - 1) ? 1 : 0;
#ifdef PDEGYM_TIMING
    tm[5] = __builtin_amdgcn_s_memtime();
    unsigned int* dbg = reinterpret_cast<unsigned int*>(P.scratch + (size_t)b * 4 * ncell + 2 * ncell);   // the scratch is unused here
    for (int i = 0; i < 5; ++i) dbg[i] = (unsigned int)(tm[i + 1] - tm[i]);
#endif
  }
}

// ================================================================================================
// float64 register-tiled step for 128x128 (the reference's own precision, BASELINE config 4 at float64): 1024 threads per
// instance, one WAVE per thread row: lane tx of wave ty owns the 8 x 2 patch at rows 8 ty .. 8 ty + 7, columns 2 tx, 2 tx + 1.
//   * left/right neighbours are the neighbouring lanes (two v_mov_b32_dpp per double);
//   * top/bottom halo rows (2 doubles = 16 bytes per lane) cross waves through a double-buffered LDS area, one barrier per
//     exchange;
//   * p and dx dy rhs stay in registers for all K sweeps; the rows rotate with period two (UP / DOWN sweeps, see
//     jacobi_sweep_bous) so a sweep has no register copies;
//   * every expression is the one of ns_generic_step<double> -- IEEE division included -- so the result is bit-identical to
//     NumPy (tests: goldens N3, oracle at 128x128, equality with the generic kernel).
// ================================================================================================
__device__ __forceinline__ double dpp_shr_f64(double v) {       // lane i <- lane i-1 (lane 0: 0, a domain-edge lane)
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double dpp_shl_f64(double v) {       // lane i <- lane i+1
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// top/bottom halo rows of a 2-wide double patch through LDS (waves are thread rows: RS = 64); NT threads per workgroup
template <int NT>
__device__ __forceinline__ void halo_tb_f64(const double (&top)[2], const double (&bot)[2], double (&ht)[2], double (&hb)[2], double* lds,
                                            int& xc, int tid, int ty) {
  constexpr int RS = 64;
  double2* base = reinterpret_cast<double2*>(lds) + (xc & 1) * (2 * NT);
  ++xc;
  double2* eT = base;
  double2* eB = base + NT;
  eT[tid] = make_double2(top[0], top[1]);
  eB[tid] = make_double2(bot[0], bot[1]);
  __syncthreads();
  const int up = (ty > 0) ? tid - RS : tid, dn = (ty < NT / RS - 1) ? tid + RS : tid;
  const double2 a = eB[up], b = eT[dn];
  ht[0] = a.x; ht[1] = a.y;
  hb[0] = b.x; hb[1] = b.y;
}

template <int PR>
__device__ __forceinline__ void apply_bc_patch_f64(double (&f)[PR][2], const EdgeFlags& E, const int (&bc)[4][2], int comp,
                                                   const double* act, int action_dim, int r0, int c0) {
  auto aval = [&](int idx) -> double { return action_dim == 1 ? act[0] : act[idx]; };
  if (E.top) {
    const int c = bc[PDEGYM_EDGE_LOWER][comp];
#pragma unroll
    for (int b = 0; b < 2; ++b) f[0][b] = (c == PDEGYM_BC_NEUMANN) ? f[1][b] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0 : aval(c0 + b));
  }
  if (E.bot) {
    const int c = bc[PDEGYM_EDGE_UPPER][comp];
#pragma unroll
    for (int b = 0; b < 2; ++b) f[PR - 1][b] = (c == PDEGYM_BC_NEUMANN) ? f[PR - 2][b] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0 : aval(c0 + b));
  }
  if (E.lef) {
    const int c = bc[PDEGYM_EDGE_LEFT][comp];
#pragma unroll
    for (int a = 0; a < PR; ++a) f[a][0] = (c == PDEGYM_BC_NEUMANN) ? f[a][1] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0 : aval(r0 + a));
  }
  if (E.rig) {
    const int c = bc[PDEGYM_EDGE_RIGHT][comp];
#pragma unroll
    for (int a = 0; a < PR; ++a) f[a][1] = (c == PDEGYM_BC_NEUMANN) ? f[a][0] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0 : aval(r0 + a));
  }
}

// Neumann pressure walls on the new rows in state ST (see bphys)
template <int PR, int ST>
__device__ __forceinline__ void jacobi_walls_f64(double (&ph)[PR + 1][2], const EdgeFlags& E) {
  constexpr int n0 = bphys<PR>(0, ST), n1 = bphys<PR>(1, ST), nl = bphys<PR>(PR - 1, ST), nm = bphys<PR>(PR - 2, ST);
  if (E.top) { ph[n0][0] = ph[n1][0]; ph[n0][1] = ph[n1][1]; }
  if (E.bot) { ph[nl][0] = ph[nm][0]; ph[nl][1] = ph[nm][1]; }
  if (E.lef) {
#pragma unroll
    for (int a = 0; a < PR; ++a) ph[bphys<PR>(a, ST)][0] = ph[bphys<PR>(a, ST)][1];
  }
  if (E.rig) {
#pragma unroll
    for (int a = 0; a < PR; ++a) ph[bphys<PR>(a, ST)][1] = ph[bphys<PR>(a, ST)][0];
  }
}

// one float64 Jacobi sweep from state ST (0: UP, rows 0 -> PR-1, 1: DOWN); 1/4 (((W + S) + E) + N - dx dy rhs) (:106-108)
template <int PR, int ST>
__device__ __forceinline__ void jacobi_sweep_f64(double (&ph)[PR + 1][2], const double (&rq)[PR][2], const EdgeFlags& E, double* lds,
                                                 int& xc, int tid, int ty) {
  double hlast[2];
  auto update = [&](double (&dst)[2], const double (&x)[2], const double (&sv)[2], const double (&nv)[2], const double (&q)[2]) {
    const double xl = dpp_shr_f64(x[1]), xr = dpp_shl_f64(x[0]);
    const double s0 = ((xl + sv[0]) + x[1]) + nv[0];
    const double s1 = ((x[0] + sv[1]) + xr) + nv[1];
    dst[0] = 0.25 * (s0 - q[0]);
    dst[1] = 0.25 * (s1 - q[1]);
  };
  if constexpr (ST == 0) {
    halo_tb_f64<64 * (128 / PR)>(ph[0], ph[PR - 1], ph[PR], hlast, lds, xc, tid, ty);            // top halo -> free row PR
#pragma unroll
    for (int a = 0; a < PR; ++a) {
      // new row a overwrites old row a-1 (its South neighbour, dead afterwards); dst may alias sv: all reads come first
      double out[2];
      if (a == PR - 1) update(out, ph[a], ph[a == 0 ? PR : a - 1], hlast, rq[a]);
      else update(out, ph[a], ph[a == 0 ? PR : a - 1], ph[a + 1], rq[a]);
      ph[a == 0 ? PR : a - 1][0] = out[0];
      ph[a == 0 ? PR : a - 1][1] = out[1];
    }
    jacobi_walls_f64<PR, 1>(ph, E);
  } else {
    halo_tb_f64<64 * (128 / PR)>(ph[bphys<PR>(0, 1)], ph[bphys<PR>(PR - 1, 1)], hlast, ph[PR - 1], lds, xc, tid, ty);   // bottom halo -> free row PR-1
#pragma unroll
    for (int a = PR - 1; a >= 0; --a) {
      double out[2];
      if (a == 0) update(out, ph[bphys<PR>(0, 1)], hlast, ph[a], rq[a]);
      else update(out, ph[bphys<PR>(a, 1)], ph[bphys<PR>(a - 1, 1)], ph[a], rq[a]);
      ph[a][0] = out[0];
      ph[a][1] = out[1];
    }
    jacobi_walls_f64<PR, 0>(ph, E);
  }
}

// Two shapes.  PR = 8: 16 waves of 8 rows, 128 registers per lane, five u* rows wait in LDS, the others and v* in the caller's
// scratch (round 2).  PR = 16 (round 3): 8 waves of 16 rows, 256 registers per lane -- float64 issues at a quarter of the
// float32 rate, two waves per SIMD keep the pipe as busy as four -- and nothing is parked in memory: ALL u* rows wait in LDS
// (128 KB next to 32 KB of halo buffers), ALL v* rows in registers: HBM traffic 753 -> ~430 MB per 512 env-steps.
template <int PR>
struct F64Tile {
  static constexpr int NW = 128 / PR, NT = 64 * NW;
  static constexpr int HALO_BYTES = 2 * 2 * NT * 16;            // two buffers x (top, bottom) x one double2 per thread
  static constexpr int PARK_ROWS = PR == 16 ? 16 : 5;           // u* rows in LDS (16 bytes per thread and row)
  static constexpr bool V_IN_REGS = PR == 16;
  static constexpr int LDS_BYTES = HALO_BYTES + PARK_ROWS * NT * 16;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget of one CU");
};
template <bool INTERLEAVED, int PR>
__global__ __launch_bounds__(64 * (128 / PR), PR == 16 ? 2 : 4) void ns_tile_step_f64(NSConst C, NSScal<double> S, NSPtrs<double> P, int B) {
  using Cfg = F64Tile<PR>;
  constexpr int kF64HaloBytes = Cfg::HALO_BYTES, kF64ParkRows = Cfg::PARK_ROWS, NT = Cfg::NT;
  constexpr int n = 128, ncell = n * n;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* lds = reinterpret_cast<double*>(smem_raw);
  const int b = blockIdx.x;
  if (b >= B) return;
  const int tid = threadIdx.x, tx = tid & 63;
  const int ty = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = ty * PR, c0 = tx * 2;
  const EdgeFlags E{ty == 0, ty == Cfg::NW - 1, tx == 0, tx == 63};
  double* u = INTERLEAVED ? nullptr : P.u + (size_t)b * ncell;
  double* v = INTERLEAVED ? nullptr : P.v + (size_t)b * ncell;
  const double* p = P.p + (size_t)b * ncell;
  double* pout = (P.p_out ? P.p_out : P.p) + (size_t)b * ncell;
  double* us = P.scratch + (size_t)b * 4 * ncell;
  double* vs = us + ncell;
  const double* act = P.action + (size_t)b * C.action_dim;
  const double* sin = INTERLEAVED ? P.state_in + (size_t)b * ncell * 2 : nullptr;
  int xc = 0;
  auto edge_cell = [&](int a, int k) { return (a == 0 && E.top) || (a == PR - 1 && E.bot) || (k == 0 && E.lef) || (k == 1 && E.rig); };

  double rq[PR][2];     // dx dy rhs, kept for all sweeps
  double keep_v[Cfg::V_IN_REGS ? PR : 1][2];    // PR = 16: v* waits here for the corrector phase
  (void)keep_v;
  {
    auto load_row = [&](int grow, double (&ru)[2], double (&rv)[2]) {
      if constexpr (INTERLEAVED) {
        const double2* row = reinterpret_cast<const double2*>(sin + ((size_t)grow * n + c0) * 2);
        const double2 w0 = row[0], w1 = row[1];
        ru[0] = w0.x; rv[0] = w0.y; ru[1] = w1.x; rv[1] = w1.y;
      } else {
        const double2 wu = *reinterpret_cast<const double2*>(u + grow * n + c0);
        const double2 wv = *reinterpret_cast<const double2*>(v + grow * n + c0);
        ru[0] = wu.x; ru[1] = wu.y; rv[0] = wv.x; rv[1] = wv.y;
      }
    };
    // ---- predictor (:130-138), apply_boundary(u*, v*) (:140), rhs (:101-103, :108) as a ROW PIPELINE ----
    // Iteration a: load the old row a+1, predict row a, finalise row a-1 (boundary rule), store it (u* partly in LDS, v* in the
    // caller's scratch) and form dx dy rhs of row a-2, whose stencil reaches the finalised rows a-3 ... a-1.  A row's values
    // are dead two iterations after they are made, so next to rq only a five-row window is live -- with the whole patch of
    // u, v, u*, v* in flight around the division-heavy stencil the compiler spilled ~190 bytes per lane to scratch memory
    // (HBM traffic both ways).  The old rows just above / below the patch come straight from the state in global memory (the
    // neighbouring thread's own rows: L2 hits); domain-edge threads read a valid row whose values are never used.
    // The lower / upper rule of apply_boundary reads the RAW prediction of row 1 / row PR-2 (those passes run before the
    // left / right passes, navier_stokes2D.py:76-90), the left / right rule is row-local: finalise() applies them in that order.
    {
      double old_p[2][2], old_c[2][2], old_n[2][2];       // [field u / v][column]: old rows a-1, a, a+1
      double raw_u[PR][2], raw_v[PR][2], fin_u[PR][2], fin_v[PR][2];
      double2* park = reinterpret_cast<double2*>(smem_raw + kF64HaloBytes);
      auto aval = [&](int idx) -> double { return C.action_dim == 1 ? act[0] : act[idx]; };
      auto finalise = [&](int i) {
#pragma unroll
        for (int comp = 0; comp < 2; ++comp) {
          const double (&raw)[PR][2] = comp == 0 ? raw_u : raw_v;
          double (&fin)[PR][2] = comp == 0 ? fin_u : fin_v;
          double t0 = raw[i][0], t1 = raw[i][1];
          if (i == 0 && E.top) {
            const int c = C.bc[PDEGYM_EDGE_LOWER][comp];
            t0 = (c == PDEGYM_BC_NEUMANN) ? raw[1][0] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0 : aval(c0));
            t1 = (c == PDEGYM_BC_NEUMANN) ? raw[1][1] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0 : aval(c0 + 1));
          }
          if (i == PR - 1 && E.bot) {
            const int c = C.bc[PDEGYM_EDGE_UPPER][comp];
            t0 = (c == PDEGYM_BC_NEUMANN) ? raw[PR - 2][0] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0 : aval(c0));
            t1 = (c == PDEGYM_BC_NEUMANN) ? raw[PR - 2][1] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0 : aval(c0 + 1));
          }
          if (E.lef) {
            const int c = C.bc[PDEGYM_EDGE_LEFT][comp];
            t0 = (c == PDEGYM_BC_NEUMANN) ? t1 : ((c == PDEGYM_BC_DIRICHLET) ? 0.0 : aval(r0 + i));
          }
          if (E.rig) {
            const int c = C.bc[PDEGYM_EDGE_RIGHT][comp];
            t1 = (c == PDEGYM_BC_NEUMANN) ? t0 : ((c == PDEGYM_BC_DIRICHLET) ? 0.0 : aval(r0 + i));
          }
          fin[i][0] = t0;
          fin[i][1] = t1;
        }
        if (i < kF64ParkRows) park[i * NT + tid] = make_double2(fin_u[i][0], fin_u[i][1]);
        else *reinterpret_cast<double2*>(us + (r0 + i) * n + c0) = make_double2(fin_u[i][0], fin_u[i][1]);
        if constexpr (Cfg::V_IN_REGS) {
          keep_v[i][0] = fin_v[i][0];
          keep_v[i][1] = fin_v[i][1];
        } else {
          *reinterpret_cast<double2*>(vs + (r0 + i) * n + c0) = make_double2(fin_v[i][0], fin_v[i][1]);
        }
      };
      auto rhs_row = [&](int i, const double (&vbelow)[2], const double (&vabove)[2]) {
        const double ul = dpp_shr_f64(fin_u[i][1]), ur = dpp_shl_f64(fin_u[i][0]);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const double uw = (k == 0) ? ul : fin_u[i][0], ue = (k == 1) ? ur : fin_u[i][1];
          const double dudx = div_c(ue - uw, S.two_dx, S.inv_two_dx);
          const double dvdy = div_c(vabove[k] - vbelow[k], S.two_dy, S.inv_two_dy);
          const double r = S.rho_over_dt * (dudx + dvdy);
          rq[i][k] = edge_cell(i, k) ? 0.0 : S.dxdy * r;
        }
      };
      load_row(E.top ? r0 : r0 - 1, old_p[0], old_p[1]);
      load_row(r0, old_c[0], old_c[1]);
#pragma unroll
      for (int a = 0; a < PR; ++a) {
        load_row((a == PR - 1 && E.bot) ? r0 + PR - 1 : r0 + a + 1, old_n[0], old_n[1]);
        {
          const double (&cu)[2] = old_c[0], (&cv)[2] = old_c[1];
          const double ul = dpp_shr_f64(cu[1]), ur = dpp_shl_f64(cu[0]), vl = dpp_shr_f64(cv[1]), vr = dpp_shl_f64(cv[0]);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const double uc = cu[k], vc = cv[k];
            const double uw = (k == 0) ? ul : cu[0], ue = (k == 1) ? ur : cu[1];
            const double vw = (k == 0) ? vl : cv[0], ve = (k == 1) ? vr : cv[1];
            const double usn = old_p[0][k], vsn = old_p[1][k], unn = old_n[0][k], vnn = old_n[1][k];
            const double dudx = div_c(ue - uw, S.two_dx, S.inv_two_dx), dudy = div_c(unn - usn, S.two_dy, S.inv_two_dy);
            const double dvdx = div_c(ve - vw, S.two_dx, S.inv_two_dx), dvdy = div_c(vnn - vsn, S.two_dy, S.inv_two_dy);
            const double lapu = div_c((((uw + usn) - 4.0 * uc) + ue) + unn, S.dxdy, S.inv_dxdy);
            const double lapv = div_c((((vw + vsn) - 4.0 * vc) + ve) + vnn, S.dxdy, S.inv_dxdy);
            const double un = uc + S.dt * (((-uc) * dudx - vc * dudy) + S.nu * lapu);
            const double vn = vc + S.dt * (((-uc) * dvdx - vc * dvdy) + S.nu * lapv);
            const bool edge = edge_cell(a, k);
            raw_u[a][k] = edge ? uc : un;
            raw_v[a][k] = edge ? vc : vn;
          }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int k = 0; k < 2; ++k) { old_p[f][k] = old_c[f][k]; old_c[f][k] = old_n[f][k]; }
        if (a >= 1) finalise(a - 1);
        if (a >= 3) rhs_row(a - 2, fin_v[a - 3], fin_v[a - 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      finalise(PR - 1);
      rhs_row(PR - 2, fin_v[PR - 3], fin_v[PR - 1]);
      double vt[2], vb[2];
      halo_tb_f64<NT>(fin_v[0], fin_v[PR - 1], vt, vb, lds, xc, tid, ty);
      rhs_row(0, vt, fin_v[1]);
      rhs_row(PR - 1, fin_v[PR - 2], vb);
    }
  }

  // ---- K Jacobi sweeps (:104-114) ----
  double pf[PR][2];
  {
    double ph[PR + 1][2];
#pragma unroll
    for (int a = 0; a < PR; ++a) {
      const double2 w = *reinterpret_cast<const double2*>(p + (r0 + a) * n + c0);
      ph[a][0] = w.x; ph[a][1] = w.y;
    }
    ph[PR][0] = ph[PR][1] = 0.0;
    int it = 0;
    for (; it + 2 <= C.iters; it += 2) {
      jacobi_sweep_f64<PR, 0>(ph, rq, E, lds, xc, tid, ty);
      jacobi_sweep_f64<PR, 1>(ph, rq, E, lds, xc, tid, ty);
    }
    if (it < C.iters) {
      jacobi_sweep_f64<PR, 0>(ph, rq, E, lds, xc, tid, ty);
#pragma unroll
      for (int a = 0; a < PR; ++a) { pf[a][0] = ph[bphys<PR>(a, 1)][0]; pf[a][1] = ph[bphys<PR>(a, 1)][1]; }
    } else {
#pragma unroll
      for (int a = 0; a < PR; ++a) { pf[a][0] = ph[a][0]; pf[a][1] = ph[a][1]; }
    }
  }
#pragma unroll
  for (int a = 0; a < PR; ++a) *reinterpret_cast<double2*>(pout + (r0 + a) * n + c0) = make_double2(pf[a][0], pf[a][1]);

  // ---- corrector (:143-146), observation, reward ----
  double acc = 0.0;
  const int t = P.time_index[b] + 1;
  const int tr = t < C.nt_ref ? t : C.nt_ref - 1;
  {
    // blocks of two rows, each finished (corrector, boundary, stores, its share of the reward sum) before the next one's
    // loads are issued: see ns_tile_step -- the whole-patch form spilled 268 bytes per lane
    constexpr int HR = 2, NBLK = PR / HR;
    double pt[2], pb[2];
    halo_tb_f64<NT>(pf[0], pf[PR - 1], pt, pb, lds, xc, tid, ty);
    const double* uref = P.U_ref + (size_t)tr * ncell * 2;
    double* obs = P.obs + (size_t)b * ncell * 2;
#pragma unroll
    for (int h = 0; h < NBLK; ++h) {
      const int a0 = h * HR;
      double uf[HR][2], vf[HR][2];
#pragma unroll
      for (int la = 0; la < HR; ++la) {
        const double2 wu = (a0 + la < kF64ParkRows)
                               ? reinterpret_cast<const double2*>(smem_raw + kF64HaloBytes)[(a0 + la) * NT + tid]
                               : *reinterpret_cast<const double2*>(us + (r0 + a0 + la) * n + c0);   // written by this same thread above
        uf[la][0] = wu.x; uf[la][1] = wu.y;
        if constexpr (Cfg::V_IN_REGS) {
          vf[la][0] = keep_v[a0 + la][0];
          vf[la][1] = keep_v[a0 + la][1];
        } else {
          const double2 wv = *reinterpret_cast<const double2*>(vs + (r0 + a0 + la) * n + c0);
          vf[la][0] = wv.x; vf[la][1] = wv.y;
        }
      }
#pragma unroll
      for (int la = 0; la < HR; ++la) {
        const int a = a0 + la;
        const double pl = dpp_shr_f64(pf[a][1]), pr = dpp_shl_f64(pf[a][0]);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const double pw = (k == 0) ? pl : pf[a][0], pe = (k == 1) ? pr : pf[a][1];
          const double ps = (a == 0) ? pt[k] : pf[a - 1][k], pn = (a == PR - 1) ? pb[k] : pf[a + 1][k];
          const double dpdx = div_c(pe - pw, S.two_dx, S.inv_two_dx);
          const double dpdy = div_c(pn - ps, S.two_dy, S.inv_two_dy);
          const bool edge = edge_cell(a, k);
          uf[la][k] = edge ? uf[la][k] : uf[la][k] - S.dt_over_rho * dpdx;
          vf[la][k] = edge ? vf[la][k] : vf[la][k] - S.dt_over_rho * dpdy;
        }
      }
      apply_bc_rows<HR, 2, double>(uf, E, C.bc, 0, act, C.action_dim, r0 + a0, c0, h == 0, h == NBLK - 1);
      apply_bc_rows<HR, 2, double>(vf, E, C.bc, 1, act, C.action_dim, r0 + a0, c0, h == 0, h == NBLK - 1);
#pragma unroll
      for (int la = 0; la < HR; ++la) {
        const int a = a0 + la;
        if constexpr (!INTERLEAVED) {
          *reinterpret_cast<double2*>(u + (r0 + a) * n + c0) = make_double2(uf[la][0], uf[la][1]);
          *reinterpret_cast<double2*>(v + (r0 + a) * n + c0) = make_double2(vf[la][0], vf[la][1]);
        }
        const size_t o = ((size_t)(r0 + a) * n + c0) * 2;
        const double2 w0 = *reinterpret_cast<const double2*>(uref + o), w1 = *reinterpret_cast<const double2*>(uref + o + 2);
        *reinterpret_cast<double2*>(obs + o) = make_double2(uf[la][0], vf[la][0]);
        *reinterpret_cast<double2*>(obs + o + 2) = make_double2(uf[la][1], vf[la][1]);
        // accumulation order of gen_back: cells in index order, du^2 then dv^2 (the reduction order across lanes differs: rtol 1e-12)
        const double d0 = uf[la][0] - w0.x, d1 = vf[la][0] - w0.y, d2 = uf[la][1] - w1.x, d3 = vf[la][1] - w1.y;
        acc += d0 * d0;
        acc += d1 * d1;
        acc += d2 * d2;
        acc += d3 * d3;
      }
      asm volatile("" : "+v"(acc));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const double ss = block_sum<double>(acc, lds);
  if (tid == 0) {
    double asq = 0.0;
    const double aref = P.action_ref[tr];
    for (int k = 0; k < C.action_dim; ++k) {
      const double d = act[k] - aref;
      asq += d * d;
    }
    P.reward[b] = ((-0.5 * ss) / (double)n) / (double)n - S.gamma_half * asq;
    P.time_index[b] = t;
    P.terminated[b] = (t >= C.nt - 1) ? 1 : 0;
  }
}

// ================================================================================================
// Small grids (the reference's shipped example: 21 x 21, K = 2000 sweeps, float64 -- examples/NavierStokes/NS2Dppo.py):
// one LANE per grid column, the column's ny cells in registers, floor(64 / nx) instances side by side in one wave.
// East / West neighbours are the adjacent lanes (DPP wave shifts), North / South neighbours are registers of the same
// lane: no LDS, no barriers -- a sweep is ~16 instructions per row for up to three instances, against one workgroup barrier
// per sweep and instance in ns_generic_step (whose 2000 barriers are what its 0.7 ms per env-step consist of).
// Lanes of neighbouring instances never feed each other: only interior columns 1 .. nx-2 are updated by stencils, and their
// neighbours are columns 0 .. nx-1 of the same instance.  Pressure walls: the reference's four Neumann copies make every
// wall cell equal its nearest interior cell after each sweep, so from the second sweep on a stencil next to a wall reads
// the cell's own old value instead of the wall (first sweep: the walls as given); the walls are written out once at the end.
// Same expression trees as ns_generic_step<T> (IEEE division for double) -> bit-identical fields (tested against the goldens
// of the reference and the generic kernel); the reward is summed per column then over the instance's lanes in order.
// ================================================================================================
__device__ __forceinline__ float lane_from_left(float v) { return lane_left(v); }
__device__ __forceinline__ float lane_from_right(float v) { return lane_right(v); }
__device__ __forceinline__ double lane_from_left(double v) { return dpp_shr_f64(v); }
__device__ __forceinline__ double lane_from_right(double v) { return dpp_shl_f64(v); }
// A lane shift must execute with every lane of the wave active: when its only use is a per-lane select (edge lanes keep
// their value, the lanes next to a wall substitute their own), the compiler may fold the shift into the selecting lanes'
// branch, and a DPP read from a lane that is masked off returns 0.  The empty asm pins the shift where it is written.
template <typename T>
__device__ __forceinline__ T pinned_from_left(T v) {
  T r = lane_from_left(v);
  asm volatile("" : "+v"(r));
  return r;
}
template <typename T>
__device__ __forceinline__ T pinned_from_right(T v) {
  T r = lane_from_right(v);
  asm volatile("" : "+v"(r));
  return r;
}

// Fused auto-reset, run right after the step kernels of the same call (same stream, no host round trip): instances whose
// step ended terminated keep their last observation in final_obs and restart from a pool row.
template <typename T>
struct NSAutoReset {
  const T* u0;
  const T* v0;
  const T* p0;
  T* final_obs;
  int* reset_count;
  int pool_rows;
};

template <typename T, int NY>     // NY = C.ny exactly: every row index below is a compile-time constant
__device__ __forceinline__ void ns_col_body(const NSConst& C, const NSScal<T>& S, const NSPtrs<T>& P, int B, T* red) {
  static_assert(NY >= 3, "a grid has at least one interior row");
  const int nx = C.nx, ncell = nx * NY;
  const int lane = threadIdx.x;
  const int G = 64 / nx;
  const int g = lane / nx;
  const int live_g = g < G ? g : G - 1;
  const int j = g < G ? lane - g * nx : nx - 1;                 // idle lanes shadow a valid cell and never store
  const int b_raw = blockIdx.x * G + live_g;
  const bool live = g < G && b_raw < B;
  const int b = b_raw < B ? b_raw : B - 1;
  const bool lef = j == 0, rig = j == nx - 1, icol = !lef && !rig;
  const T* act = P.action + (size_t)b * C.action_dim;
  const T* sin = P.state_in ? P.state_in + (size_t)b * ncell * 2 : nullptr;
  T* ug = P.u ? P.u + (size_t)b * ncell : nullptr;
  T* vg = P.v ? P.v + (size_t)b * ncell : nullptr;
  T* us = P.scratch + (size_t)b * 4 * ncell;
  T* vs = us + ncell;
  // one command per instance or one per edge node: an unconditional load from a selected index (a conditional LOAD keeps a branch
  // per use inside the unrolled row loops, and with the branches the register allocator spilled half the kernel)
  const int per_node = C.action_dim == 1 ? 0 : 1;
  auto aval = [&](int idx) -> T { return act[idx * per_node]; };

  // apply_boundary (navier_stokes2D.py:76-90) on a column-per-lane field: lower / upper rule on every lane's end cells, then
  // the left / right rule on the edge lanes, which for a Neumann edge copies the (already ruled) neighbouring column
  auto apply_bc = [&](T (&f)[NY], int comp) {
    const int cl = C.bc[PDEGYM_EDGE_LOWER][comp], cu = C.bc[PDEGYM_EDGE_UPPER][comp];
    f[0] = (cl == PDEGYM_BC_NEUMANN) ? f[1] : ((cl == PDEGYM_BC_DIRICHLET) ? (T)0 : aval(j));
    f[NY - 1] = (cu == PDEGYM_BC_NEUMANN) ? f[NY - 2] : ((cu == PDEGYM_BC_DIRICHLET) ? (T)0 : aval(j));
    const int cL = C.bc[PDEGYM_EDGE_LEFT][comp], cR = C.bc[PDEGYM_EDGE_RIGHT][comp];
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const T from_r = pinned_from_right(f[i]), from_l = pinned_from_left(f[i]);
      const T vl = (cL == PDEGYM_BC_NEUMANN) ? from_r : ((cL == PDEGYM_BC_DIRICHLET) ? (T)0 : aval(i));
      const T vr = (cR == PDEGYM_BC_NEUMANN) ? from_l : ((cR == PDEGYM_BC_DIRICHLET) ? (T)0 : aval(i));
      f[i] = lef ? vl : (rig ? vr : f[i]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  T u[NY], v[NY];
  {
    // either state layout through one pair of base pointers and a stride (no branch per row)
    const T* bu = sin ? sin : ug;
    const T* bv = sin ? sin + 1 : vg;
    const int st = sin ? 2 : 1;
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int c = (i * nx + j) * st;
      u[i] = bu[c];
      v[i] = bv[c];
    }
  }
  // ---- predictor (navier_stokes2D.py:130-138), in place with the old row below carried along ----
  {
    T pu = u[0], pv = v[0];
#pragma unroll
    for (int i = 1; i < NY - 1; ++i) {
      const T uc = u[i], vc = v[i];
      const T uw = pinned_from_left(uc), ue = pinned_from_right(uc), vw = pinned_from_left(vc), ve = pinned_from_right(vc);
      const T usn = pu, vsn = pv, unn = u[i + 1], vnn = v[i + 1];
      const T dudx = div_c(ue - uw, S.two_dx, S.inv_two_dx), dudy = div_c(unn - usn, S.two_dy, S.inv_two_dy);
      const T dvdx = div_c(ve - vw, S.two_dx, S.inv_two_dx), dvdy = div_c(vnn - vsn, S.two_dy, S.inv_two_dy);
      const T lapu = div_c((((uw + usn) - (T)4 * uc) + ue) + unn, S.dxdy, S.inv_dxdy);
      const T lapv = div_c((((vw + vsn) - (T)4 * vc) + ve) + vnn, S.dxdy, S.inv_dxdy);
      const T un = uc + S.dt * (((-uc) * dudx - vc * dudy) + S.nu * lapu);
      const T vn = vc + S.dt * (((-uc) * dvdx - vc * dvdy) + S.nu * lapv);
      u[i] = icol ? un : uc;
      v[i] = icol ? vn : vc;
      pu = uc;
      pv = vc;
      __builtin_amdgcn_sched_barrier(0);      // one row at a time: interleaved rows multiply the live temporaries
    }
  }
  apply_bc(u, 0);
  apply_bc(v, 1);
  // ---- rhs (:101-103) times dx dy (:108); u*, v* wait in the caller's scratch during the sweeps ----
  T rq[NY];
#pragma unroll
  for (int i = 0; i < NY; ++i) {
    rq[i] = (T)0;
    if (live) { us[i * nx + j] = u[i]; vs[i * nx + j] = v[i]; }
    if (i >= 1 && i < NY - 1) {
      const T ue = pinned_from_right(u[i]), uw = pinned_from_left(u[i]);
      const T dudx = div_c(ue - uw, S.two_dx, S.inv_two_dx);
      const T dvdy = div_c(v[i + 1] - v[i - 1], S.two_dy, S.inv_two_dy);
      const T r = S.rho_over_dt * (dudx + dvdy);
      if constexpr (sizeof(T) == 4) rq[i] = icol ? jacobi_rhs_term(S.dxdy, r) : 0.f;
      else rq[i] = icol ? S.dxdy * r : (T)0;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- K Jacobi sweeps (:104-114) ----
  T p[NY];
  {
    const T* pg = P.p + (size_t)b * ncell;
#pragma unroll
    for (int i = 0; i < NY; ++i) p[i] = pg[i * nx + j];
  }
  // FIRST: the walls as given (edge lanes and rows 0, NY-1 hold them); afterwards a stencil next to a wall reads the cell's
  // own old value (= what the Neumann copies of the previous sweep left in the wall).  Edge lanes run the same arithmetic on
  // whatever their neighbours hold: nothing reads them until the walls are written out after the last sweep.
  const bool next_to_left = j == 1, next_to_right = j == nx - 2;
  auto sweep = [&](auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    T below = p[0];                      // old value of the row below
#pragma unroll
    for (int i = 1; i < NY - 1; ++i) {
      const T cur = p[i];
      const T wl = pinned_from_left(cur), er = pinned_from_right(cur);
      const T w = (!FIRST && next_to_left) ? cur : wl;
      const T e = (!FIRST && next_to_right) ? cur : er;
      const T sv = (!FIRST && i == 1) ? cur : below;
      const T nv = (!FIRST && i == NY - 2) ? cur : p[i + 1];
      const T s4 = ((w + sv) + e) + nv;
      T val;
      if constexpr (sizeof(T) == 4) val = jacobi_update(s4, rq[i]);
      else val = (T)0.25 * (s4 - rq[i]);
      p[i] = FIRST ? (icol ? val : cur) : val;
      below = cur;
    }
  };
  // every later sweep reads one array and writes the other (p -> q, q -> p): the old row below is still in place when row i
  // needs it, so nothing is carried along (one 64-bit move per row and sweep less; p, q and rq are 126 float64 registers)
  auto sweep_into = [&](const T (&src)[NY], T (&dst)[NY]) {
#pragma unroll
    for (int i = 1; i < NY - 1; ++i) {
      const T cur = src[i];
      const T wl = pinned_from_left(cur), er = pinned_from_right(cur);
      const T w = next_to_left ? cur : wl;
      const T e = next_to_right ? cur : er;
      const T sv = (i == 1) ? cur : src[i - 1];
      const T nv = (i == NY - 2) ? cur : src[i + 1];
      const T s4 = ((w + sv) + e) + nv;
      if constexpr (sizeof(T) == 4) dst[i] = jacobi_update(s4, rq[i]);
      else dst[i] = (T)0.25 * (s4 - rq[i]);
    }
  };
  if (C.iters > 0) {
    sweep(std::true_type{});
    T q[NY];
#pragma unroll
    for (int i = 0; i < NY; ++i) q[i] = p[i];
    int it = 1;
    for (; it + 2 <= C.iters; it += 2) {
      sweep_into(p, q);
      sweep_into(q, p);
    }
    if (it < C.iters) {
      sweep_into(p, q);
#pragma unroll
      for (int i = 1; i < NY - 1; ++i) p[i] = q[i];
    }
    // the four Neumann copies of the last sweep (:110-113): every wall cell = its nearest interior cell
    p[0] = p[1];
    p[NY - 1] = p[NY - 2];
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const T from_r = pinned_from_right(p[i]), from_l = pinned_from_left(p[i]);
      p[i] = lef ? from_r : (rig ? from_l : p[i]);
    }
  }
  {
    T* pd = (P.p_out ? P.p_out : P.p) + (size_t)b * ncell;
#pragma unroll
    for (int i = 0; i < NY; ++i)
      if (live) pd[i * nx + j] = p[i];
  }
  // ---- corrector (:143-146), observation, reward ----
#pragma unroll
  for (int i = 0; i < NY; ++i) {
    u[i] = us[i * nx + j];      // written by this same lane above (idle lanes: never written, never used)
    v[i] = vs[i * nx + j];
  }
#pragma unroll
  for (int i = 1; i < NY - 1; ++i) {
    const T pe = pinned_from_right(p[i]), pw = pinned_from_left(p[i]);
    const T dpdx = div_c(pe - pw, S.two_dx, S.inv_two_dx);
    const T dpdy = div_c(p[i + 1] - p[i - 1], S.two_dy, S.inv_two_dy);
    u[i] = icol ? u[i] - S.dt_over_rho * dpdx : u[i];
    v[i] = icol ? v[i] - S.dt_over_rho * dpdy : v[i];
    __builtin_amdgcn_sched_barrier(0);
  }
  apply_bc(u, 0);
  apply_bc(v, 1);
  const int t = P.time_index[b] + 1;
  const int tr = t < C.nt_ref ? t : C.nt_ref - 1;
  const T* uref = P.U_ref + (size_t)tr * ncell * 2;
  T* obs = P.obs + (size_t)b * ncell * 2;
  T acc = 0;
#pragma unroll
  for (int i = 0; i < NY; ++i) {
    const int c = i * nx + j;
    if (live) {
      if (ug) { ug[c] = u[i]; vg[c] = v[i]; }
      obs[2 * c] = u[i];
      obs[2 * c + 1] = v[i];
    }
    const T du = u[i] - uref[2 * c], dv = v[i] - uref[2 * c + 1];
    acc += du * du;
    acc += dv * dv;
    __builtin_amdgcn_sched_barrier(0);
  }
  red[lane] = acc;
  __syncthreads();
  if (live && j == 0) {
    T ss = 0;
    for (int k = 0; k < nx; ++k) ss += red[lane + k];
    T asq = 0;
    const T aref = P.action_ref[tr];
    for (int k = 0; k < C.action_dim; ++k) {
      const T d = act[k] - aref;
      asq += d * d;
    }
    P.reward[b] = (((T)-0.5 * ss) / (T)nx) / (T)NY - S.gamma_half * asq;
    P.time_index[b] = t;
    P.terminated[b] = (t >= C.nt - 1) ? 1 : 0;
  }
}

template <typename T, int NY>
__global__ __launch_bounds__(64, (sizeof(T) == 8 || NY > 21 ? 2 : 3)) void ns_col_step(NSConst C, NSScal<T> S, NSPtrs<T> P, int B) {
  __shared__ T red[64];
  ns_col_body<T, NY>(C, S, P, B, red);
}

// The same kernel built for ONE wave per SIMD (512 registers per lane, arch + accumulation): the float64 instantiations then keep
// everything on chip -- 372 registers and no scratch memory at 21 rows, where the two-wave build spills 222 -- and a batch that does
// not fill the chip twice over is quicker on it (21 x 21, K = 2000: B = 1024 1.11 -> 0.88 ms, B = 3072 1.15 -> 0.95 ms); with more
// than one wave per SIMD to run the two-wave build wins (B = 8192 2.61 vs 2.68 ms, B = 32768 8.75 vs 9.71 ms): launch_ns_col picks.
template <typename T, int NY>
__global__ __launch_bounds__(64, 1) void ns_col_step_w1(NSConst C, NSScal<T> S, NSPtrs<T> P, int B) {
  __shared__ T red[64];
  ns_col_body<T, NY>(C, S, P, B, red);
}

// T env-steps in ONE launch (pdegym_ns2d_rollout_*, small grids): iteration t is ns_col_step's body with the state read from
// observation slot t and written to slot t + 1, the command / reward / flag taken from / written to row t of the rollout arrays,
// followed by the fused auto-reset of ns_step (final_obs, pool row -> slot t + 1 and p, time index, restart counter) -- every
// value equals what T pdegym_ns2d_step calls produce, bit for bit.  A wave re-reads only what its own lanes stored (its up to
// three instances: state slots, p, time index, flag), so iterations are separated by a workgroup-scope release / acquire pair.
template <typename T>
struct NSRollout {
  int T_steps;
  T* obs;
  const T* actions;
  T* rewards;
  uint8_t* terminated;
};

// (called, not inlined, from the rollout loop: with the step body inlined into a loop clang 22 / ROCm 7.2 crashes in instcombine for
// one instantiation or another -- which one moves with every change of the body.  The call costs the callee-saved registers a
// round trip through scratch memory per env-step: tools/attic/bench_ns_rollout.py)
template <typename T, int NY>
__device__ __attribute__((noinline)) void ns_col_body_call(const NSConst& C, const NSScal<T>& S, const NSPtrs<T>& P, int B, T* red) {
  ns_col_body<T, NY>(C, S, P, B, red);
}

template <typename T, int NY>
__global__ __launch_bounds__(64, (sizeof(T) == 8 || NY > 21 ? 2 : 3)) void ns_col_rollout(NSConst C, NSScal<T> S, NSPtrs<T> P, NSRollout<T> Ro,
                                                                                         NSAutoReset<T> R, int B) {
  __shared__ T red[64];
  const int nx = C.nx, ncell = nx * NY;
  const int lane = threadIdx.x;
  const int G = 64 / nx;
  const int g = lane / nx;
  const int j = lane - g * nx;
  const int b = blockIdx.x * G + g;
  const bool live = g < G && b < B;
  const size_t slot = (size_t)B * ncell * 2;
  for (int t = 0; t < Ro.T_steps; ++t) {
    const NSPtrs<T> Q{nullptr, nullptr, P.p, P.scratch, Ro.actions + (size_t)t * B * C.action_dim, P.time_index, P.U_ref, P.action_ref,
                      Ro.obs + (size_t)(t + 1) * slot, Ro.rewards + (size_t)t * B, Ro.terminated + (size_t)t * B,
                      Ro.obs + (size_t)t * slot, nullptr};
    ns_col_body_call<T, NY>(C, S, Q, B, red);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (R.u0) {                                       // fused auto-reset, as ns_auto_reset_kernel / _finish after a step call
      const bool done = live && Q.terminated[b] != 0;
      if (done) {
        const int rows = R.pool_rows > 0 ? R.pool_rows : B;
        const long long k = R.reset_count ? (long long)R.reset_count[b] : 0;
        const size_t src = (size_t)(((long long)b + k * (long long)B) % rows) * ncell, off = (size_t)b * ncell;
#pragma unroll
        for (int i = 0; i < NY; ++i) {
          const size_t c = off + (size_t)i * nx + j;
          if (R.final_obs) {
            R.final_obs[2 * c] = Q.obs[2 * c];
            R.final_obs[2 * c + 1] = Q.obs[2 * c + 1];
          }
          Q.p[c] = R.p0[src + (size_t)i * nx + j];
          Q.obs[2 * c] = R.u0[src + (size_t)i * nx + j];
          Q.obs[2 * c + 1] = R.v0[src + (size_t)i * nx + j];
        }
      }
      __builtin_amdgcn_wave_barrier();                // every lane of the group has read reset_count before lane j == 0 bumps it
      if (done && j == 0) {
        P.time_index[b] = 0;
        if (R.reset_count) R.reset_count[b] += 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
}

// The column kernel is instantiated for the grid height of the reference's shipped example (21 rows) and a few neighbours
// (8, 11, 16, 26, 31, 32); any width up to 64.  One wave works through all K sweeps of its (up to three) instances alone: in
// float64 a lone instance finishes sooner on the workgroup-per-instance kernel (seven waves per instance; 0.77 vs 0.98 ms per
// env-step at 21 x 21, K = 2000), so batches below a minimum (PDEGYM_DEBUG_NS_COL_MIN_BATCH overrides it; default: 400 for float64 -- round 4, tools/attic/probe_ns_col_min_batch.py: column kernel 0.87 ms flat up to 3072 instances, workgroup kernel 0.80 ms at 256, 0.96 at 512, 1.30 at 768 -- and 1 for float32, where
// the two kernels are equal at B = 1) stay there.
template <typename T>
bool launch_ns_col(const NSConst& C, const NSScal<T>& S, const NSPtrs<T>& P, int B, hipStream_t st) {
  if (C.nx < 3 || C.nx > 64) return false;
  const int dbg = g_debug[PDEGYM_DEBUG_NS_COL_MIN_BATCH];
  // float64 crossover: 400 instances on the 1024 SIMDs of an unpartitioned MI355X (three instances per wave at 21 columns), scaled
  // with the SIMD count of the device the call runs on
  const int simds = pdegym::simd_count();
  const int min_batch = dbg >= 0 ? dbg : (sizeof(T) == 8 ? (int)((400LL * simds + 512) / 1024) : 1);
  if (B < min_batch) return false;
  const int G = 64 / C.nx;
  const dim3 grid((B + G - 1) / G), block(64);
  if constexpr (sizeof(T) == 8) {
    if ((int)grid.x <= simds) {  // at most one wave per SIMD anyway: the build without spills (ns_col_step_w1)
      switch (C.ny) {
        case 16: hipLaunchKernelGGL((ns_col_step_w1<T, 16>), grid, block, 0, st, C, S, P, B); return true;
        case 21: hipLaunchKernelGGL((ns_col_step_w1<T, 21>), grid, block, 0, st, C, S, P, B); return true;
        case 26: hipLaunchKernelGGL((ns_col_step_w1<T, 26>), grid, block, 0, st, C, S, P, B); return true;
        default: break;          // 8 / 11 rows do not spill either way; 31 / 32 rows spill either way
      }
    }
  }
  switch (C.ny) {
    case 8: hipLaunchKernelGGL((ns_col_step<T, 8>), grid, block, 0, st, C, S, P, B); return true;
    case 11: hipLaunchKernelGGL((ns_col_step<T, 11>), grid, block, 0, st, C, S, P, B); return true;
    case 16: hipLaunchKernelGGL((ns_col_step<T, 16>), grid, block, 0, st, C, S, P, B); return true;
    case 21: hipLaunchKernelGGL((ns_col_step<T, 21>), grid, block, 0, st, C, S, P, B); return true;
    case 26: hipLaunchKernelGGL((ns_col_step<T, 26>), grid, block, 0, st, C, S, P, B); return true;
    case 31: hipLaunchKernelGGL((ns_col_step<T, 31>), grid, block, 0, st, C, S, P, B); return true;
    case 32: hipLaunchKernelGGL((ns_col_step<T, 32>), grid, block, 0, st, C, S, P, B); return true;
    default: return false;
  }
}

template <typename T>
bool launch_ns_col_rollout(const NSConst& C, const NSScal<T>& S, const NSPtrs<T>& P, const NSRollout<T>& Ro, const NSAutoReset<T>& R, int B,
                           hipStream_t st) {
  if (C.nx < 3 || C.nx > 64) return false;
  const int G = 64 / C.nx;
  const dim3 grid((B + G - 1) / G), block(64);
  switch (C.ny) {
    case 8: hipLaunchKernelGGL((ns_col_rollout<T, 8>), grid, block, 0, st, C, S, P, Ro, R, B); return true;
    case 11: hipLaunchKernelGGL((ns_col_rollout<T, 11>), grid, block, 0, st, C, S, P, Ro, R, B); return true;
    case 16: hipLaunchKernelGGL((ns_col_rollout<T, 16>), grid, block, 0, st, C, S, P, Ro, R, B); return true;
    case 21: hipLaunchKernelGGL((ns_col_rollout<T, 21>), grid, block, 0, st, C, S, P, Ro, R, B); return true;
    case 26: hipLaunchKernelGGL((ns_col_rollout<T, 26>), grid, block, 0, st, C, S, P, Ro, R, B); return true;
    case 31: hipLaunchKernelGGL((ns_col_rollout<T, 31>), grid, block, 0, st, C, S, P, Ro, R, B); return true;
    case 32: hipLaunchKernelGGL((ns_col_rollout<T, 32>), grid, block, 0, st, C, S, P, Ro, R, B); return true;
    default: return false;
  }
}

template <typename T, int LDSJ>
__global__ __launch_bounds__(1024) void ns_generic_pressure(NSConst C, NSScal<T> S, const T* ug, const T* vg, const T* p_in,
                                                             T* p_out, T* scratch, int B) {
  extern __shared__ double ns_dyn_lds[];
  const int b = blockIdx.x;
  if (b >= B) return;
  const int nx = C.nx, ny = C.ny, ncell = nx * ny;
  const T* u = ug + (size_t)b * ncell;
  const T* v = vg + (size_t)b * ncell;
  const T* pi = p_in + (size_t)b * ncell;
  T* po = p_out + (size_t)b * ncell;
  T* rhs = scratch + (size_t)b * 2 * ncell;
  T* pB = rhs + ncell;
  compute_rhs<T>(u, v, rhs, ny, nx, S);
  if (po != pi) {
    for (int c = threadIdx.x; c < ncell; c += blockDim.x) po[c] = pi[c];
  }
  __syncthreads();
  if constexpr (LDSJ == 1) jacobi_sweeps_lds<T>(po, rhs, reinterpret_cast<T*>(ns_dyn_lds), ny, nx, C.iters, S.dxdy);
  else if constexpr (LDSJ == 2) jacobi_sweeps_lds1<T>(po, rhs, reinterpret_cast<T*>(ns_dyn_lds), ny, nx, C.iters, S.dxdy);
  else jacobi_sweeps<T>(po, pB, rhs, ny, nx, C.iters, S.dxdy);
}

template <typename T>
__global__ void ns_reset_kernel(NSConst C, NSPtrs<T> P, const T* u0, const T* v0, const T* p0, const uint8_t* mask, int B) {
  const int b = blockIdx.y;
  if (b >= B) return;
  if (mask && !mask[b]) return;
  const int ncell = C.nx * C.ny;
  const size_t off = (size_t)b * ncell;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += gridDim.x * blockDim.x) {
    const T a = u0[off + c], bb = v0[off + c];
    if (P.u) {
      P.u[off + c] = a;
      P.v[off + c] = bb;
    }
    P.p[off + c] = p0[off + c];
    P.obs[2 * (off + c)] = a;
    P.obs[2 * (off + c) + 1] = bb;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    P.time_index[b] = 0;
    P.terminated[b] = 0;
  }
}

template <typename T>
__global__ void ns_auto_reset_kernel(NSConst C, NSPtrs<T> P, NSAutoReset<T> R, int B) {
  const int b = blockIdx.y;
  if (b >= B || !P.terminated[b]) return;
  const int ncell = C.nx * C.ny;
  const int rows = R.pool_rows > 0 ? R.pool_rows : B;
  const long long k = R.reset_count ? (long long)R.reset_count[b] : 0;
  const size_t src = (size_t)(((long long)b + k * (long long)B) % rows) * ncell, off = (size_t)b * ncell;
  T* pnew = P.p_out ? P.p_out : P.p;      // where this call left the pressure (the caller swaps p and p_out afterwards)
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += gridDim.x * blockDim.x) {
    if (R.final_obs) {
      R.final_obs[2 * (off + c)] = P.obs[2 * (off + c)];
      R.final_obs[2 * (off + c) + 1] = P.obs[2 * (off + c) + 1];
    }
    const T a = R.u0[src + c], bb = R.v0[src + c];
    if (P.u && !P.state_in) {
      P.u[off + c] = a;
      P.v[off + c] = bb;
    }
    pnew[off + c] = R.p0[src + c];
    P.obs[2 * (off + c)] = a;
    P.obs[2 * (off + c) + 1] = bb;
  }
}

// time_index / reset_count are updated by a second tiny launch so that every block of the copy kernel sees the old values
template <typename T>
__global__ void ns_auto_reset_finish(NSPtrs<T> P, NSAutoReset<T> R, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B || !P.terminated[b]) return;
  P.time_index[b] = 0;
  if (R.reset_count) R.reset_count[b] += 1;
}

template <typename T>
int fill(const pdegym_params_ns2d* prm, NSConst& C, NSScal<T>& S) {
  if (!prm) return pdegym::fail(-1, "null params");
  if (prm->nx < 3 || prm->ny < 3) return pdegym::fail(-2, "grid must be at least 3x3");
  if (prm->iters < 0) return pdegym::fail(-2, "iters must be >= 0");
  if (prm->action_dim != 1 && (prm->action_dim != prm->nx || prm->nx != prm->ny))
    return pdegym::fail(-2, "action_dim must be 1 or the edge length of a square grid");
  C.nx = prm->nx;
  C.ny = prm->ny;
  C.nt = prm->nt;
  C.iters = prm->iters;
  C.action_dim = prm->action_dim;
  for (int e = 0; e < 4; ++e)
    for (int k = 0; k < 2; ++k) {
      if (prm->bc[e][k] < 0 || prm->bc[e][k] > 2) return pdegym::fail(-2, "bad boundary condition code");
      C.bc[e][k] = prm->bc[e][k];
    }
  // Python evaluates these scalar sub-expressions in double before they meet an array (navier_stokes2D.py:12,14,21,
  // 103,108,144): 2*step, dx*dy, rho/dt, dt/rho
  S.dt = (T)prm->dt;
  S.two_dx = (T)(2 * prm->dx);
  S.two_dy = (T)(2 * prm->dy);
  S.dxdy = (T)(prm->dx * prm->dy);
  S.nu = (T)prm->viscosity;
  S.rho_over_dt = (T)(prm->density / prm->dt);
  S.dt_over_rho = (T)(prm->dt / prm->density);
  S.gamma_half = (T)(prm->gamma / 2);
  S.inv_two_dx = (T)(1.0 / (2 * prm->dx));
  S.inv_two_dy = (T)(1.0 / (2 * prm->dy));
  S.inv_dxdy = (T)(1.0 / (prm->dx * prm->dy));
  return 0;
}

// LDS-resident Jacobi: every thread owns at most kLdsCPT cells
// LDS-resident Jacobi: every thread owns at most kLdsCPT cells.  A lone wave issues an instruction only every ~6.5 cycles,
// so when the batch leaves CUs idle anyway (B <= 512: at most two workgroups per CU) one thread per cell (up to 512) cuts the
// per-sweep latency (21x21, K = 2000, B = 1: 1.06 -> 0.74 ms per env-step); big batches keep the 4-cells-per-thread shape,
// which has the better throughput (fewer barrier participants per instance).
inline int lds_block_threads(int ncell, int B) {
  if (B <= 512 && ncell <= 2048) {
    int t = (ncell + 63) / 64 * 64;
    t = t < 128 ? 128 : (t > 512 ? 512 : t);
    if (t * kLdsCPT >= ncell) return t;
  }
  if (ncell <= 1024) return 256;
  if (ncell <= 2048) return 512;
  return 1024;
}

// 1: two LDS copies (<= kLdsCells cells); 2: one LDS copy (larger grids that still fit the 160 KB of a CU); 0: global memory
template <typename T>
inline int lds_jacobi_mode(int ncell) {
  if (pdegym_no_lds_jacobi()) return 0;
  if (ncell <= kLdsCells) return 1;
  if (ncell <= kLds1Cells && (size_t)ncell * sizeof(T) <= (size_t)kLds1MaxBytes) return 2;
  return 0;
}

inline int block_threads(int ncell) {
  if (ncell >= 4096) return 1024;
  if (ncell >= 1024) return 512;
  return 256;
}

template <typename T>
int ns_step_launch(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, int B, void* stream);

template <typename T>
int ns_step(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, int B, void* stream) {
  if (int rc = ns_step_launch<T>(prm, buf, B, stream)) return rc;
  if (B <= 0 || !buf->reset_u0) return 0;
  if (!buf->reset_v0 || !buf->reset_p0) return pdegym::fail(-3, "reset_u0 needs reset_v0 and reset_p0");
  NSConst C;
  NSScal<T> S;
  if (int rc = fill<T>(prm, C, S)) return rc;
  NSPtrs<T> P{(T*)buf->u, (T*)buf->v, (T*)buf->p, (T*)buf->scratch, (const T*)buf->action, buf->time_index,
              (const T*)buf->U_ref, (const T*)buf->action_ref, (T*)buf->obs, (T*)buf->reward, buf->terminated,
              (const T*)buf->state_in, (T*)buf->p_out};
  NSAutoReset<T> R{(const T*)buf->reset_u0, (const T*)buf->reset_v0, (const T*)buf->reset_p0, (T*)buf->final_obs, buf->reset_count,
                   buf->reset_pool_rows};
  const int ncell = C.nx * C.ny;
  const int gx = (ncell + 255) / 256 > 64 ? 64 : (ncell + 255) / 256;
  hipLaunchKernelGGL(ns_auto_reset_kernel<T>, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, C, P, R, B);
  hipLaunchKernelGGL(ns_auto_reset_finish<T>, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, R, B);
  return pdegym::check_launch("ns2d_auto_reset");
}

template <typename T>
int ns_step_launch(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, int B, void* stream) {
  NSConst C;
  NSScal<T> S;
  if (int rc = fill<T>(prm, C, S)) return rc;
  if (!buf) return pdegym::fail(-1, "null bufs");
  if (B <= 0) return 0;
  if (!buf->p || !buf->scratch || !buf->action || !buf->time_index || !buf->U_ref || !buf->action_ref || !buf->obs ||
      !buf->reward || !buf->terminated)
    return pdegym::fail(-3, "null device buffer");
  if (!buf->state_in && (!buf->u || !buf->v)) return pdegym::fail(-3, "either state_in or both u and v must be given");
  if ((buf->u == nullptr) != (buf->v == nullptr)) return pdegym::fail(-3, "u and v must be given together");
  if (buf->state_in == buf->obs) return pdegym::fail(-3, "state_in must not alias obs (double-buffer the observations)");
  if (buf->nt_ref < 1) return pdegym::fail(-2, "nt_ref must be >= 1");
  C.nt_ref = buf->nt_ref;
  NSPtrs<T> P{(T*)buf->u, (T*)buf->v, (T*)buf->p, (T*)buf->scratch, (const T*)buf->action, buf->time_index,
              (const T*)buf->U_ref, (const T*)buf->action_ref, (T*)buf->obs, (T*)buf->reward, buf->terminated,
              (const T*)buf->state_in, (T*)buf->p_out};
  if (buf->p_out && buf->p_out == buf->p) return pdegym::fail(-3, "p_out must not alias p");
  // small grids: one lane per column, up to three instances per wave, no barriers (ns_col_step)
  if (!pdegym_force_generic() && !pdegym_ns_no_col() && launch_ns_col<T>(C, S, P, B, (hipStream_t)stream))
    return pdegym::check_launch("ns2d_col_step");
  if constexpr (sizeof(T) == 4) {
    // register-tiled float32 path for the square grids it is instantiated for (BASELINE config 4 is 128x128)
    if (!pdegym_force_generic() && C.nx == C.ny && (C.nx == 128 || C.nx == 64)) {
      constexpr int lds128 = TileCfg<8, 4>::LDS_BYTES + TileCfg<8, 4>::PARK_BYTES, lds64 = TileCfg<4, 2>::LDS_BYTES;
      if (C.nx == 128) {   // 80 KB of dynamic LDS: above the 64 KB a kernel gets without asking
        static signed char attr_i[pdegym::kMaxDevices] = {}, attr_s[pdegym::kMaxDevices] = {};
        const bool ok = buf->state_in
            ? pdegym::raise_dynamic_lds_limit(reinterpret_cast<const void*>(&ns_tile_step<8, 4, true>), lds128, attr_i)
            : pdegym::raise_dynamic_lds_limit(reinterpret_cast<const void*>(&ns_tile_step<8, 4, false>), lds128, attr_s);
        if (!ok) return pdegym::fail(-4, "cannot raise the dynamic LDS limit of ns_tile_step");
      }
      // state_in given -> the velocity state is the previous observation and u, v are not written (if the caller
      // also passed u, v they are simply left untouched)
      const bool inter = buf->state_in != nullptr;
      if (C.nx == 128 && inter)
        hipLaunchKernelGGL((ns_tile_step<8, 4, true>), dim3(B), dim3(512), lds128, (hipStream_t)stream, C, S, P, B);
      else if (C.nx == 128)
        hipLaunchKernelGGL((ns_tile_step<8, 4, false>), dim3(B), dim3(512), lds128, (hipStream_t)stream, C, S, P, B);
      else if (inter)
        hipLaunchKernelGGL((ns_tile_step<4, 2, true>), dim3(B), dim3(512), lds64, (hipStream_t)stream, C, S, P, B);
      else
        hipLaunchKernelGGL((ns_tile_step<4, 2, false>), dim3(B), dim3(512), lds64, (hipStream_t)stream, C, S, P, B);
      return pdegym::check_launch("ns2d_tile_step");
    }
  }
  if constexpr (sizeof(T) == 8) {
    // register-tiled float64 path for 128x128 (BASELINE config 4 at the reference's own precision)
    if (!pdegym_force_generic() && C.nx == 128 && C.ny == 128) {
      constexpr int kPRf64 = PDEGYM_NS_F64_TILE_ROWS;
      constexpr int lds_bytes = F64Tile<kPRf64>::LDS_BYTES, nt = F64Tile<kPRf64>::NT;
      static signed char attr_a[pdegym::kMaxDevices] = {}, attr_b[pdegym::kMaxDevices] = {};
      if (buf->state_in) {
        if (!pdegym::raise_dynamic_lds_limit(reinterpret_cast<const void*>(&ns_tile_step_f64<true, kPRf64>), lds_bytes, attr_a))
          return pdegym::fail(-4, "cannot raise the dynamic LDS limit");
        hipLaunchKernelGGL((ns_tile_step_f64<true, kPRf64>), dim3(B), dim3(nt), lds_bytes, (hipStream_t)stream, C, S, P, B);
      } else {
        if (!pdegym::raise_dynamic_lds_limit(reinterpret_cast<const void*>(&ns_tile_step_f64<false, kPRf64>), lds_bytes, attr_b))
          return pdegym::fail(-4, "cannot raise the dynamic LDS limit");
        hipLaunchKernelGGL((ns_tile_step_f64<false, kPRf64>), dim3(B), dim3(nt), lds_bytes, (hipStream_t)stream, C, S, P, B);
      }
      return pdegym::check_launch("ns2d_tile_step_f64");
    }
  }
  if constexpr (sizeof(T) == 8) {
    // 256x256 float64 (BASELINE config 5 at the reference's precision): pdegym_ns256_f64.hip
    if (!pdegym_force_generic() && C.nx == 256 && C.ny == 256) return launch_ns256_step_f64(C, S, P, B, (hipStream_t)stream);
  }
  if constexpr (sizeof(T) == 4) {
    // 256x256 float32 (BASELINE config 5): the whole env-step in one launch, one workgroup per instance (pdegym_ns256.hip)
    if (!pdegym_force_generic() && C.nx == 256 && C.ny == 256) return launch_ns256_fused(C, S, P, B, (hipStream_t)stream);
  }
  const int ncell = C.nx * C.ny;
  const int mode = lds_jacobi_mode<T>(ncell);
  if (mode == 1) {
    hipLaunchKernelGGL((ns_generic_step<T, 1>), dim3(B), dim3(lds_block_threads(ncell, B)), 2 * (size_t)ncell * sizeof(T),
                       (hipStream_t)stream, C, S, P, B);
  } else if (mode == 2) {
    static signed char attr_done[pdegym::kMaxDevices] = {};
    if (!pdegym::raise_dynamic_lds_limit(reinterpret_cast<const void*>(&ns_generic_step<T, 2>), kLds1MaxBytes, attr_done))
      return pdegym::fail(-4, "cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL((ns_generic_step<T, 2>), dim3(B), dim3(1024), (size_t)ncell * sizeof(T), (hipStream_t)stream, C, S, P, B);
  } else {
    hipLaunchKernelGGL((ns_generic_step<T, 0>), dim3(B), dim3(block_threads(ncell)), 0, (hipStream_t)stream, C, S, P, B);
  }
  return pdegym::check_launch("ns2d_step");
}

template <typename T>
int ns_pressure(const pdegym_params_ns2d* prm, const void* u, const void* v, const void* p_in, void* p_out, void* scratch,
                int B, void* stream) {
  NSConst C;
  NSScal<T> S;
  if (int rc = fill<T>(prm, C, S)) return rc;
  if (B <= 0) return 0;
  if (!u || !v || !p_in || !p_out || !scratch) return pdegym::fail(-3, "null device buffer");
  C.nt_ref = 1;
  const int ncell = C.nx * C.ny;
  const int mode = lds_jacobi_mode<T>(ncell);
  if (mode == 1) {
    hipLaunchKernelGGL((ns_generic_pressure<T, 1>), dim3(B), dim3(lds_block_threads(ncell, B)), 2 * (size_t)ncell * sizeof(T),
                       (hipStream_t)stream, C, S, (const T*)u, (const T*)v, (const T*)p_in, (T*)p_out, (T*)scratch, B);
  } else if (mode == 2) {
    static signed char attr_done[pdegym::kMaxDevices] = {};
    if (!pdegym::raise_dynamic_lds_limit(reinterpret_cast<const void*>(&ns_generic_pressure<T, 2>), kLds1MaxBytes, attr_done))
      return pdegym::fail(-4, "cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL((ns_generic_pressure<T, 2>), dim3(B), dim3(1024), (size_t)ncell * sizeof(T), (hipStream_t)stream, C, S,
                       (const T*)u, (const T*)v, (const T*)p_in, (T*)p_out, (T*)scratch, B);
  } else {
    hipLaunchKernelGGL((ns_generic_pressure<T, 0>), dim3(B), dim3(block_threads(ncell)), 0, (hipStream_t)stream, C, S,
                       (const T*)u, (const T*)v, (const T*)p_in, (T*)p_out, (T*)scratch, B);
  }
  return pdegym::check_launch("ns2d_solve_pressure");
}

template <typename T>
int ns_reset(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, const void* u0, const void* v0, const void* p0,
             const uint8_t* mask, int B, void* stream) {
  NSConst C;
  NSScal<T> S;
  if (int rc = fill<T>(prm, C, S)) return rc;
  if (!buf || !u0 || !v0 || !p0) return pdegym::fail(-1, "null bufs/initial fields");
  if (B <= 0) return 0;
  C.nt_ref = 1;
  NSPtrs<T> P{(T*)buf->u, (T*)buf->v, (T*)buf->p, (T*)buf->scratch, (const T*)buf->action, buf->time_index,
              (const T*)buf->U_ref, (const T*)buf->action_ref, (T*)buf->obs, (T*)buf->reward, buf->terminated, nullptr, nullptr};
  const int ncell = C.nx * C.ny;
  const int gx = (ncell + 255) / 256 > 64 ? 64 : (ncell + 255) / 256;
  hipLaunchKernelGGL(ns_reset_kernel<T>, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, C, P, (const T*)u0, (const T*)v0,
                     (const T*)p0, mask, B);
  return pdegym::check_launch("ns2d_reset");
}

template <typename T>
int ns_rollout(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, const pdegym_rollout_ns2d* ro, int B, void* stream) {
  NSConst C;
  NSScal<T> S;
  if (int rc = fill<T>(prm, C, S)) return rc;
  if (!buf || !ro) return pdegym::fail(-1, "null bufs / rollout descriptor");
  if (B <= 0 || ro->T <= 0) return 0;
  if (!buf->p || !buf->scratch || !buf->time_index || !buf->U_ref || !buf->action_ref) return pdegym::fail(-3, "null device buffer");
  if (!ro->obs || !ro->actions || !ro->rewards || !ro->terminated) return pdegym::fail(-3, "null rollout buffer");
  if (buf->nt_ref < 1) return pdegym::fail(-2, "nt_ref must be >= 1");
  if (buf->reset_u0 && (!buf->reset_v0 || !buf->reset_p0)) return pdegym::fail(-3, "reset_u0 needs reset_v0 and reset_p0");
  C.nt_ref = buf->nt_ref;
  NSPtrs<T> P{nullptr, nullptr, (T*)buf->p, (T*)buf->scratch, nullptr, buf->time_index, (const T*)buf->U_ref, (const T*)buf->action_ref,
              nullptr, nullptr, nullptr, nullptr, nullptr};
  NSRollout<T> Ro{ro->T, (T*)ro->obs, (const T*)ro->actions, (T*)ro->rewards, ro->terminated};
  NSAutoReset<T> R{(const T*)buf->reset_u0, (const T*)buf->reset_v0, (const T*)buf->reset_p0, (T*)buf->final_obs, buf->reset_count,
                   buf->reset_pool_rows};
  if (!launch_ns_col_rollout<T>(C, S, P, Ro, R, B, (hipStream_t)stream))
    return pdegym::fail(-2, "ns2d rollout: grids of 8, 11, 16, 21, 26, 31 or 32 rows and at most 64 columns (the column-per-lane kernel)");
  return pdegym::check_launch("ns2d_col_rollout");
}

}  // namespace

extern "C" {

int pdegym_ns2d_rollout_f32(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, const pdegym_rollout_ns2d* ro, int32_t B, void* stream) {
  return ns_rollout<float>(prm, buf, ro, B, stream);
}
int pdegym_ns2d_rollout_f64(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, const pdegym_rollout_ns2d* ro, int32_t B, void* stream) {
  return ns_rollout<double>(prm, buf, ro, B, stream);
}

int32_t pdegym_debug_set(int32_t key, int32_t value) {
  if (key < 0 || key >= PDEGYM_DEBUG_COUNT) return pdegym::fail(-2, "unknown debug key");
  const int32_t old = g_debug[key];
  g_debug[key] = value;
  return old;
}

int pdegym_ns2d_step_f32(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, int32_t B, void* stream) {
  return ns_step<float>(prm, buf, B, stream);
}
int pdegym_ns2d_step_f64(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, int32_t B, void* stream) {
  return ns_step<double>(prm, buf, B, stream);
}
int pdegym_ns2d_solve_pressure_f32(const pdegym_params_ns2d* prm, const void* u, const void* v, const void* p_in,
                                   void* p_out, void* scratch, int32_t B, void* stream) {
  return ns_pressure<float>(prm, u, v, p_in, p_out, scratch, B, stream);
}
int pdegym_ns2d_solve_pressure_f64(const pdegym_params_ns2d* prm, const void* u, const void* v, const void* p_in,
                                   void* p_out, void* scratch, int32_t B, void* stream) {
  return ns_pressure<double>(prm, u, v, p_in, p_out, scratch, B, stream);
}
int pdegym_ns2d_reset_masked_f32(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, const void* u0,
                                 const void* v0, const void* p0, const uint8_t* mask, int32_t B, void* stream) {
  return ns_reset<float>(prm, buf, u0, v0, p0, mask, B, stream);
}
int pdegym_ns2d_reset_masked_f64(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, const void* u0,
                                 const void* v0, const void* p0, const uint8_t* mask, int32_t B, void* stream) {
  return ns_reset<double>(prm, buf, u0, v0, p0, mask, B, stream);
}

}  // extern "C"
