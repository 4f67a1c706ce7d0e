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
//                   (see DESIGN.md).  Same arithmetic, reassociated only where stated.
#include <hip/hip_runtime.h>

#include "pdegym.h"
#include "pdegym_common.h"

namespace {

struct NSConst {
  int nx, ny, nt, iters, action_dim, nt_ref;
  int bc[4][2];
};

template <typename T>
struct NSScal {
  T dt, two_dx, two_dy, dxdy, nu, rho_over_dt, dt_over_rho, gamma_half, inv_unused;
};

template <typename T>
struct NSPtrs {
  T* u;
  T* v;
  T* p;
  T* scratch;
  const T* action;
  int* time_index;
  const T* U_ref;
  const T* action_ref;
  T* obs;
  T* reward;
  uint8_t* terminated;
};

// Value of boundary cell (i,j) after apply_boundary's four ordered passes (navier_stokes2D.py:76-90), as a
// closed form of the INTERIOR values of f: left/right passes run last and cover whole columns, so corners are
// decided by the left/right condition, reading the neighbour that the lower/upper pass has already set.
template <typename T>
__device__ __forceinline__ T bc_value(const T* __restrict__ f, int i, int j, int ny, int nx, const int (&bc)[4][2],
                                      int comp, const T* __restrict__ act, int action_dim) {
  auto aval = [&](int idx) -> T { return action_dim == 1 ? act[0] : act[idx]; };
  auto row_rule = [&](int ii, int jj) -> T {
    const int c = (ii == 0) ? bc[PDEGYM_EDGE_LOWER][comp] : bc[PDEGYM_EDGE_UPPER][comp];
    if (c == PDEGYM_BC_NEUMANN) return f[(size_t)((ii == 0) ? 1 : ny - 2) * nx + jj];
    if (c == PDEGYM_BC_DIRICHLET) return (T)0;
    return aval(jj);
  };
  if (j == 0 || j == nx - 1) {
    const int c = (j == 0) ? bc[PDEGYM_EDGE_LEFT][comp] : bc[PDEGYM_EDGE_RIGHT][comp];
    if (c == PDEGYM_BC_DIRICHLET) return (T)0;
    if (c == PDEGYM_BC_CONTROLLABLE) return aval(i);
    const int jj = (j == 0) ? 1 : nx - 2;
    if (i == 0 || i == ny - 1) return row_rule(i, jj);
    return f[(size_t)i * nx + jj];
  }
  return row_rule(i, j);
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red /* >= 16 entries of LDS */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  T s = 0;
  for (int k = 0; k < nw; ++k) s += red[k];
  return s;
}

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
        const T val = (T)0.25 * (s4 - dxdy * rhs[c]);
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

template <typename T>
__device__ __forceinline__ void compute_rhs(const T* us, const T* vs, T* rhs, int ny, int nx, const NSScal<T>& S) {
  const int ncell = ny * nx;
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int i = c / nx, j = c - i * nx;
    T r = 0;
    if (i >= 1 && i <= ny - 2 && j >= 1 && j <= nx - 2) {
      // navier_stokes2D.py:101-103   rho/dt * (d/dx u* + d/dy v*)
      const T dudx = (us[c + 1] - us[c - 1]) / S.two_dx;
      const T dvdy = (vs[c + nx] - vs[c - nx]) / S.two_dy;
      r = S.rho_over_dt * (dudx + dvdy);
    }
    rhs[c] = r;
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void ns_generic_step(NSConst C, NSScal<T> S, NSPtrs<T> P, int B) {
  __shared__ T red[16];
  const int b = blockIdx.x;
  if (b >= B) return;
  const int nx = C.nx, ny = C.ny, ncell = nx * ny;
  T* u = P.u + (size_t)b * ncell;
  T* v = P.v + (size_t)b * ncell;
  T* p = P.p + (size_t)b * ncell;
  T* us = P.scratch + (size_t)b * 4 * ncell;
  T* vs = us + ncell;
  T* rhs = vs + ncell;
  T* pB = rhs + ncell;
  const T* act = P.action + (size_t)b * C.action_dim;

  // ---- predictor (navier_stokes2D.py:130-138); boundary derivatives are zero so u* = u there (:9-22) ----
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int i = c / nx, j = c - i * nx;
    const T uc = u[c], vc = v[c];
    T un = uc, vn = vc;
    if (i >= 1 && i <= ny - 2 && j >= 1 && j <= nx - 2) {
      const T uw = u[c - 1], ue = u[c + 1], usn = u[c - nx], unn = u[c + nx];
      const T vw = v[c - 1], ve = v[c + 1], vsn = v[c - nx], vnn = v[c + nx];
      const T dudx = (ue - uw) / S.two_dx, dudy = (unn - usn) / S.two_dy;
      const T dvdx = (ve - vw) / S.two_dx, dvdy = (vnn - vsn) / S.two_dy;
      const T lapu = ((((uw + usn) - (T)4 * uc) + ue) + unn) / S.dxdy;
      const T lapv = ((((vw + vsn) - (T)4 * vc) + ve) + vnn) / S.dxdy;
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
  // ---- pressure Poisson (:142, :94-116) ----
  compute_rhs<T>(us, vs, rhs, ny, nx, S);
  __syncthreads();
  jacobi_sweeps<T>(p, pB, rhs, ny, nx, C.iters, S.dxdy);
  // ---- corrector (:143-145): interior; the boundary keeps u* (zero pressure derivative) until the BC pass ----
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int i = c / nx, j = c - i * nx;
    T un = us[c], vn = vs[c];
    if (i >= 1 && i <= ny - 2 && j >= 1 && j <= nx - 2) {
      const T dpdx = (p[c + 1] - p[c - 1]) / S.two_dx;
      const T dpdy = (p[c + nx] - p[c - nx]) / S.two_dy;
      un = un - S.dt_over_rho * dpdx;
      vn = vn - S.dt_over_rho * dpdy;
    }
    u[c] = un;
    v[c] = vn;
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
      un = bc_value<T>(u, i, j, ny, nx, C.bc, 0, act, C.action_dim);
      vn = bc_value<T>(v, i, j, ny, nx, C.bc, 1, act, C.action_dim);
      u[c] = un;
      v[c] = vn;
    } else {
      un = u[c];
      vn = v[c];
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

template <typename T>
__global__ __launch_bounds__(1024) void ns_generic_pressure(NSConst C, NSScal<T> S, const T* ug, const T* vg, const T* p_in,
                                                             T* p_out, T* scratch, int B) {
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
  jacobi_sweeps<T>(po, pB, rhs, ny, nx, C.iters, S.dxdy);
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
    P.u[off + c] = a;
    P.v[off + c] = bb;
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
int fill(const pdegym_params_ns2d* prm, NSConst& C, NSScal<T>& S) {
  if (!prm) return pdegym::fail(-1, "null params");
  if (prm->nx < 4 || prm->ny < 4) return pdegym::fail(-2, "grid must be at least 4x4");
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
  S.inv_unused = 0;
  return 0;
}

inline int block_threads(int ncell) {
  if (ncell >= 4096) return 1024;
  if (ncell >= 1024) return 512;
  return 256;
}

template <typename T>
int ns_step(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, int B, void* stream) {
  NSConst C;
  NSScal<T> S;
  if (int rc = fill<T>(prm, C, S)) return rc;
  if (!buf) return pdegym::fail(-1, "null bufs");
  if (B <= 0) return 0;
  if (!buf->u || !buf->v || !buf->p || !buf->scratch || !buf->action || !buf->time_index || !buf->U_ref ||
      !buf->action_ref || !buf->obs || !buf->reward || !buf->terminated)
    return pdegym::fail(-3, "null device buffer");
  if (buf->nt_ref < 1) return pdegym::fail(-2, "nt_ref must be >= 1");
  C.nt_ref = buf->nt_ref;
  NSPtrs<T> P{(T*)buf->u, (T*)buf->v, (T*)buf->p, (T*)buf->scratch, (const T*)buf->action, buf->time_index,
              (const T*)buf->U_ref, (const T*)buf->action_ref, (T*)buf->obs, (T*)buf->reward, buf->terminated};
  hipLaunchKernelGGL(ns_generic_step<T>, dim3(B), dim3(block_threads(C.nx * C.ny)), 0, (hipStream_t)stream, C, S, P, B);
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
  hipLaunchKernelGGL(ns_generic_pressure<T>, dim3(B), dim3(block_threads(C.nx * C.ny)), 0, (hipStream_t)stream, C, S,
                     (const T*)u, (const T*)v, (const T*)p_in, (T*)p_out, (T*)scratch, B);
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
              (const T*)buf->U_ref, (const T*)buf->action_ref, (T*)buf->obs, (T*)buf->reward, buf->terminated};
  const int ncell = C.nx * C.ny;
  const int gx = (ncell + 255) / 256 > 64 ? 64 : (ncell + 255) / 256;
  hipLaunchKernelGGL(ns_reset_kernel<T>, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, C, P, (const T*)u0, (const T*)v0,
                     (const T*)p0, mask, B);
  return pdegym::check_launch("ns2d_reset");
}

}  // namespace

extern "C" {

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
