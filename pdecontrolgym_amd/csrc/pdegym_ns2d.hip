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

#include <cstdlib>

#include "pdegym.h"
#include "pdegym_common.h"

namespace {

struct NSConst {
  int nx, ny, nt, iters, action_dim, nt_ref;
  int bc[4][2];
};

template <typename T>
struct NSScal {
  T dt, two_dx, two_dy, dxdy, nu, rho_over_dt, dt_over_rho, gamma_half;
  T inv_two_dx, inv_two_dy, inv_dxdy;  // float32 throughput mode multiplies by reciprocals
};

// Division by a grid constant.  double: true IEEE division (bit parity with NumPy).  float: multiply by the
// reciprocal (rounded once on the host) -- inside the stated float32 tolerance, ~10x fewer instructions.
__device__ __forceinline__ double div_c(double a, double c, double /*inv_c*/) { return a / c; }
__device__ __forceinline__ float div_c(float a, float /*c*/, float inv_c) { return a * inv_c; }

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
      const T dudx = div_c(us[c + 1] - us[c - 1], S.two_dx, S.inv_two_dx);
      const T dvdy = div_c(vs[c + nx] - vs[c - nx], S.two_dy, S.inv_two_dy);
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
  // ---- pressure Poisson (:142, :94-116) ----
  compute_rhs<T>(us, vs, rhs, ny, nx, S);
  __syncthreads();
  jacobi_sweeps<T>(p, pB, rhs, ny, nx, C.iters, S.dxdy);
  // ---- corrector (:143-145): interior; the boundary keeps u* (zero pressure derivative) until the BC pass ----
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int i = c / nx, j = c - i * nx;
    T un = us[c], vn = vs[c];
    if (i >= 1 && i <= ny - 2 && j >= 1 && j <= nx - 2) {
      const T dpdx = div_c(p[c + 1] - p[c - 1], S.two_dx, S.inv_two_dx);
      const T dpdy = div_c(p[c + nx] - p[c - nx], S.two_dy, S.inv_two_dy);
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


// ================================================================================================
// float32 register-tiled path: grid side n = 16*PS (PS = 8 -> 128x128, PS = 4 -> 64x64).
// 256 threads per instance, thread (ty,tx) owns the PSxPS patch at (ty*PS, tx*PS) in VGPRs.  Only patch
// edges cross threads, through a double-buffered LDS halo area (one barrier per exchange).  p and
// dx*dy*rhs stay in registers for all K Jacobi sweeps; u*, v* are parked in caller scratch meanwhile.
// Arithmetic is the same expression tree as ns_generic<float>, so both paths agree bit for bit.
// ================================================================================================
template <int PS>
struct TileCfg {
  static constexpr int NP = PS / 4;                 // float4 planes per patch edge
  static constexpr int BUF = 4 * NP * 256;          // float4 per halo buffer
  static constexpr int LDS_BYTES = 2 * BUF * 16;    // two buffers
};

struct EdgeFlags {
  bool top, bot, lef, rig;
};

template <int PS>
struct Halo {
  float t[PS], b[PS], l[PS], r[PS];
};

template <int PS>
__device__ __forceinline__ void halo_exchange(const float (&f)[PS][PS], Halo<PS>& H, float4* lds, int& xc, int tid, int ty,
                                              int tx) {
  constexpr int NP = TileCfg<PS>::NP;
  float4* base = lds + (xc & 1) * TileCfg<PS>::BUF;
  ++xc;
  float4* eT = base;
  float4* eB = base + NP * 256;
  float4* eL = base + 2 * NP * 256;
  float4* eR = base + 3 * NP * 256;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    eT[q * 256 + tid] = make_float4(f[0][4 * q], f[0][4 * q + 1], f[0][4 * q + 2], f[0][4 * q + 3]);
    eB[q * 256 + tid] = make_float4(f[PS - 1][4 * q], f[PS - 1][4 * q + 1], f[PS - 1][4 * q + 2], f[PS - 1][4 * q + 3]);
    eL[q * 256 + tid] = make_float4(f[4 * q][0], f[4 * q + 1][0], f[4 * q + 2][0], f[4 * q + 3][0]);
    eR[q * 256 + tid] = make_float4(f[4 * q][PS - 1], f[4 * q + 1][PS - 1], f[4 * q + 2][PS - 1], f[4 * q + 3][PS - 1]);
  }
  __syncthreads();
  // Threads on the domain edge have no neighbour on that side: they re-read their own edge (a valid address) and
  // never use the value (edge cells are overwritten by the wall / boundary rules), so every read is unconditional.
  const int up = (ty > 0) ? tid - 16 : tid, dn = (ty < 15) ? tid + 16 : tid;
  const int lf = (tx > 0) ? tid - 1 : tid, rt = (tx < 15) ? tid + 1 : tid;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const float4 a = eB[q * 256 + up];   // row above = bottom edge of (ty-1, tx)
    const float4 b = eT[q * 256 + dn];   // row below = top edge of (ty+1, tx)
    const float4 c = eR[q * 256 + lf];   // column to the left = right edge of (ty, tx-1)
    const float4 d = eL[q * 256 + rt];   // column to the right = left edge of (ty, tx+1)
    H.t[4 * q] = a.x; H.t[4 * q + 1] = a.y; H.t[4 * q + 2] = a.z; H.t[4 * q + 3] = a.w;
    H.b[4 * q] = b.x; H.b[4 * q + 1] = b.y; H.b[4 * q + 2] = b.z; H.b[4 * q + 3] = b.w;
    H.l[4 * q] = c.x; H.l[4 * q + 1] = c.y; H.l[4 * q + 2] = c.z; H.l[4 * q + 3] = c.w;
    H.r[4 * q] = d.x; H.r[4 * q + 1] = d.y; H.r[4 * q + 2] = d.z; H.r[4 * q + 3] = d.w;
  }
}

// ---- Jacobi sweep on a ROTATING register file ------------------------------------------------------------
// ph holds PS+1 physical rows: the PS patch rows plus the row above (top halo).  The new value of row a is
// written into the registers of OLD row a-1 (dead once row a has been computed; new row 0 goes into the halo
// row), so a sweep needs no register copies at all; the logical->physical row map shifts by one per sweep and
// returns to the identity after PS+1 sweeps (the sweep loop is unrolled PS+1 times over the rotation R).
template <int PS>
__device__ constexpr int prow(int a, int r) {
  return (((a - r) % (PS + 1)) + (PS + 1)) % (PS + 1);
}

template <int PS, int R>
__device__ __forceinline__ void jacobi_sweep_rot(float (&ph)[PS + 1][PS], const float (&rr)[PS][PS], const EdgeFlags& E,
                                                 float4* lds, int& xc, int tid, int ty, int tx) {
  constexpr int NP = TileCfg<PS>::NP;
  float4* base = lds + (xc & 1) * TileCfg<PS>::BUF;
  ++xc;
  float4* eT = base;
  float4* eB = base + NP * 256;
  float4* eL = base + 2 * NP * 256;
  float4* eR = base + 3 * NP * 256;
  constexpr int r0 = prow<PS>(0, R), rl = prow<PS>(PS - 1, R);
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    eT[q * 256 + tid] = make_float4(ph[r0][4 * q], ph[r0][4 * q + 1], ph[r0][4 * q + 2], ph[r0][4 * q + 3]);
    eB[q * 256 + tid] = make_float4(ph[rl][4 * q], ph[rl][4 * q + 1], ph[rl][4 * q + 2], ph[rl][4 * q + 3]);
    eL[q * 256 + tid] = make_float4(ph[prow<PS>(4 * q, R)][0], ph[prow<PS>(4 * q + 1, R)][0], ph[prow<PS>(4 * q + 2, R)][0],
                                    ph[prow<PS>(4 * q + 3, R)][0]);
    eR[q * 256 + tid] = make_float4(ph[prow<PS>(4 * q, R)][PS - 1], ph[prow<PS>(4 * q + 1, R)][PS - 1],
                                    ph[prow<PS>(4 * q + 2, R)][PS - 1], ph[prow<PS>(4 * q + 3, R)][PS - 1]);
  }
  __syncthreads();
  const int up = (ty > 0) ? tid - 16 : tid, dn = (ty < 15) ? tid + 16 : tid;
  const int lf = (tx > 0) ? tid - 1 : tid, rt = (tx < 15) ? tid + 1 : tid;
  float hb[PS], hl[PS], hr[PS];
  constexpr int rt_row = prow<PS>(-1, R);  // top halo lands in the free physical row
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const float4 a = eB[q * 256 + up];
    const float4 b = eT[q * 256 + dn];
    const float4 c = eR[q * 256 + lf];
    const float4 d = eL[q * 256 + rt];
    ph[rt_row][4 * q] = a.x; ph[rt_row][4 * q + 1] = a.y; ph[rt_row][4 * q + 2] = a.z; ph[rt_row][4 * q + 3] = a.w;
    hb[4 * q] = b.x; hb[4 * q + 1] = b.y; hb[4 * q + 2] = b.z; hb[4 * q + 3] = b.w;
    hl[4 * q] = c.x; hl[4 * q + 1] = c.y; hl[4 * q + 2] = c.z; hl[4 * q + 3] = c.w;
    hr[4 * q] = d.x; hr[4 * q + 1] = d.y; hr[4 * q + 2] = d.z; hr[4 * q + 3] = d.w;
  }
#pragma unroll
  for (int a = 0; a < PS; ++a) {
    constexpr int dummy = 0;
    (void)dummy;
    const int src = prow<PS>(a, R), dst = prow<PS>(a - 1, R), nxt = prow<PS>(a + 1, R);
#pragma unroll
    for (int k = 0; k < PS; ++k) {
      const float w = (k == 0) ? hl[a] : ph[src][k - 1], e = (k == PS - 1) ? hr[a] : ph[src][k + 1];
      const float nn = (a == PS - 1) ? hb[k] : ph[nxt][k];
      const float s4 = ((w + ph[dst][k]) + e) + nn;          // ((W + S) + E) + N   (navier_stokes2D.py:106-108)
      ph[dst][k] = 0.25f * (s4 - rr[a][k]);
    }
  }
  // Neumann walls (:110-113) on the NEW rows (rotation R+1): every boundary cell = nearest interior value
  constexpr int n0 = prow<PS>(0, R + 1), n1 = prow<PS>(1, R + 1), nl = prow<PS>(PS - 1, R + 1), nm = prow<PS>(PS - 2, R + 1);
  if (E.top) {
#pragma unroll
    for (int k = 0; k < PS; ++k) ph[n0][k] = ph[n1][k];
  }
  if (E.bot) {
#pragma unroll
    for (int k = 0; k < PS; ++k) ph[nl][k] = ph[nm][k];
  }
  if (E.lef) {
#pragma unroll
    for (int a = 0; a < PS; ++a) ph[prow<PS>(a, R + 1)][0] = ph[prow<PS>(a, R + 1)][1];
  }
  if (E.rig) {
#pragma unroll
    for (int a = 0; a < PS; ++a) ph[prow<PS>(a, R + 1)][PS - 1] = ph[prow<PS>(a, R + 1)][PS - 2];
  }
}

// K sweeps, unrolled over the PS+1 rotations; returns with the patch back in logical order in pf.
template <int PS, int R>
struct SweepChain {
  static __device__ __forceinline__ int run(float (&ph)[PS + 1][PS], const float (&rr)[PS][PS], const EdgeFlags& E, float4* lds,
                                            int& xc, int tid, int ty, int tx, int& it, int K) {
    if (it >= K) return R;
    jacobi_sweep_rot<PS, R>(ph, rr, E, lds, xc, tid, ty, tx);
    ++it;
    if constexpr (R == PS) return -1;  // full cycle done: identity map again
    else return SweepChain<PS, R + 1>::run(ph, rr, E, lds, xc, tid, ty, tx, it, K);
  }
};

template <int PS, int R>
__device__ __forceinline__ void unrotate(const float (&ph)[PS + 1][PS], float (&pf)[PS][PS], int rot) {
  if (rot == R) {
#pragma unroll
    for (int a = 0; a < PS; ++a)
#pragma unroll
      for (int k = 0; k < PS; ++k) pf[a][k] = ph[prow<PS>(a, R)][k];
  } else if constexpr (R < PS) {
    unrotate<PS, R + 1>(ph, pf, rot);
  }
}

template <int PS>
__device__ __forceinline__ void load_patch(float (&f)[PS][PS], const float* g, int n, int r0, int c0) {
#pragma unroll
  for (int a = 0; a < PS; ++a) {
    const float4* row = reinterpret_cast<const float4*>(g + (size_t)(r0 + a) * n + c0);
#pragma unroll
    for (int q = 0; q < PS / 4; ++q) {
      const float4 w = row[q];
      f[a][4 * q] = w.x; f[a][4 * q + 1] = w.y; f[a][4 * q + 2] = w.z; f[a][4 * q + 3] = w.w;
    }
  }
}

template <int PS>
__device__ __forceinline__ void store_patch(const float (&f)[PS][PS], float* g, int n, int r0, int c0) {
#pragma unroll
  for (int a = 0; a < PS; ++a) {
    float4* row = reinterpret_cast<float4*>(g + (size_t)(r0 + a) * n + c0);
#pragma unroll
    for (int q = 0; q < PS / 4; ++q) row[q] = make_float4(f[a][4 * q], f[a][4 * q + 1], f[a][4 * q + 2], f[a][4 * q + 3]);
  }
}

// apply_boundary on a patch: the four ordered passes (lower, upper, left, right) only touch cells of edge
// threads and only read the line next to the edge, which lives in the same patch -> no communication.
template <int PS>
__device__ __forceinline__ void apply_bc_patch(float (&f)[PS][PS], const EdgeFlags& E, const int (&bc)[4][2], int comp,
                                               const float* act, int action_dim, int r0, int c0) {
  auto aval = [&](int idx) -> float { return action_dim == 1 ? act[0] : act[idx]; };
  if (E.top) {
    const int c = bc[PDEGYM_EDGE_LOWER][comp];
#pragma unroll
    for (int b = 0; b < PS; ++b) f[0][b] = (c == PDEGYM_BC_NEUMANN) ? f[1][b] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0f : aval(c0 + b));
  }
  if (E.bot) {
    const int c = bc[PDEGYM_EDGE_UPPER][comp];
#pragma unroll
    for (int b = 0; b < PS; ++b)
      f[PS - 1][b] = (c == PDEGYM_BC_NEUMANN) ? f[PS - 2][b] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0f : aval(c0 + b));
  }
  if (E.lef) {
    const int c = bc[PDEGYM_EDGE_LEFT][comp];
#pragma unroll
    for (int a = 0; a < PS; ++a) f[a][0] = (c == PDEGYM_BC_NEUMANN) ? f[a][1] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0f : aval(r0 + a));
  }
  if (E.rig) {
    const int c = bc[PDEGYM_EDGE_RIGHT][comp];
#pragma unroll
    for (int a = 0; a < PS; ++a)
      f[a][PS - 1] = (c == PDEGYM_BC_NEUMANN) ? f[a][PS - 2] : ((c == PDEGYM_BC_DIRICHLET) ? 0.0f : aval(r0 + a));
  }
}

template <int PS>
__device__ __forceinline__ bool on_domain_edge(const EdgeFlags& E, int a, int b) {
  return (a == 0 && E.top) || (a == PS - 1 && E.bot) || (b == 0 && E.lef) || (b == PS - 1 && E.rig);
}

template <int PS>
__global__ __launch_bounds__(256, 2) void ns_tile_step(NSConst C, NSScal<float> S, NSPtrs<float> P, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float4* lds = reinterpret_cast<float4*>(smem_raw);
  const int b = blockIdx.x;
  if (b >= B) return;
  constexpr int n = 16 * PS;
  constexpr int ncell = n * n;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int r0 = ty * PS, c0 = tx * PS;
  const EdgeFlags E{ty == 0, ty == 15, tx == 0, tx == 15};
  float* u = P.u + (size_t)b * ncell;
  float* v = P.v + (size_t)b * ncell;
  float* p = P.p + (size_t)b * ncell;
  float* us = P.scratch + (size_t)b * 4 * ncell;
  float* vs = us + ncell;
  const float* act = P.action + (size_t)b * C.action_dim;
  int xc = 0;
#ifdef PDEGYM_TIMING
  unsigned long long tm[8];
  tm[0] = __builtin_amdgcn_s_memtime();
#define PDEGYM_STAMP(i, dep) tm[i] = __builtin_amdgcn_s_memtime() + (unsigned long long)((dep) != (dep))
#else
#define PDEGYM_STAMP(i, dep)
#endif

  float rr[PS][PS];  // dx*dy*rhs, kept for all sweeps
  {
    float uf[PS][PS], vf[PS][PS];
    load_patch<PS>(uf, u, n, r0, c0);
    load_patch<PS>(vf, v, n, r0, c0);
    // ---- predictor (navier_stokes2D.py:130-138) ----
    {
      Halo<PS> HU, HV;
      halo_exchange<PS>(uf, HU, lds, xc, tid, ty, tx);
      halo_exchange<PS>(vf, HV, lds, xc, tid, ty, tx);
      float pu[PS], pv[PS];  // old row a-1
#pragma unroll
      for (int k = 0; k < PS; ++k) { pu[k] = HU.t[k]; pv[k] = HV.t[k]; }
#pragma unroll
      for (int a = 0; a < PS; ++a) {
        float cu[PS], cv[PS];
#pragma unroll
        for (int k = 0; k < PS; ++k) { cu[k] = uf[a][k]; cv[k] = vf[a][k]; }
#pragma unroll
        for (int k = 0; k < PS; ++k) {
          const float uc = cu[k], vc = cv[k];
          const float uw = (k == 0) ? HU.l[a] : cu[k - 1], ue = (k == PS - 1) ? HU.r[a] : cu[k + 1];
          const float vw = (k == 0) ? HV.l[a] : cv[k - 1], ve = (k == PS - 1) ? HV.r[a] : cv[k + 1];
          const float usn = pu[k], vsn = pv[k];
          const float unn = (a == PS - 1) ? HU.b[k] : uf[a + 1][k];
          const float vnn = (a == PS - 1) ? HV.b[k] : vf[a + 1][k];
          const float dudx = div_c(ue - uw, S.two_dx, S.inv_two_dx), dudy = div_c(unn - usn, S.two_dy, S.inv_two_dy);
          const float dvdx = div_c(ve - vw, S.two_dx, S.inv_two_dx), dvdy = div_c(vnn - vsn, S.two_dy, S.inv_two_dy);
          const float lapu = div_c((((uw + usn) - 4.0f * uc) + ue) + unn, S.dxdy, S.inv_dxdy);
          const float lapv = div_c((((vw + vsn) - 4.0f * vc) + ve) + vnn, S.dxdy, S.inv_dxdy);
          const float un = uc + S.dt * (((-uc) * dudx - vc * dudy) + S.nu * lapu);
          const float vn = vc + S.dt * (((-uc) * dvdx - vc * dvdy) + S.nu * lapv);
          const bool edge = on_domain_edge<PS>(E, a, k);
          uf[a][k] = edge ? uc : un;
          vf[a][k] = edge ? vc : vn;
        }
#pragma unroll
        for (int k = 0; k < PS; ++k) { pu[k] = cu[k]; pv[k] = cv[k]; }
      }
    }
    PDEGYM_STAMP(1, uf[0][0]);
    // ---- apply_boundary(u*, v*) (:140) ----
    apply_bc_patch<PS>(uf, E, C.bc, 0, act, C.action_dim, r0, c0);
    apply_bc_patch<PS>(vf, E, C.bc, 1, act, C.action_dim, r0, c0);
    store_patch<PS>(uf, us, n, r0, c0);
    store_patch<PS>(vf, vs, n, r0, c0);
    // ---- rhs (:101-103), pre-multiplied by dx*dy (:108) ----
    {
      Halo<PS> HU, HV;
      halo_exchange<PS>(uf, HU, lds, xc, tid, ty, tx);
      halo_exchange<PS>(vf, HV, lds, xc, tid, ty, tx);
#pragma unroll
      for (int a = 0; a < PS; ++a)
#pragma unroll
        for (int k = 0; k < PS; ++k) {
          const float uw = (k == 0) ? HU.l[a] : uf[a][k - 1], ue = (k == PS - 1) ? HU.r[a] : uf[a][k + 1];
          const float vsn = (a == 0) ? HV.t[k] : vf[a - 1][k], vnn = (a == PS - 1) ? HV.b[k] : vf[a + 1][k];
          const float dudx = div_c(ue - uw, S.two_dx, S.inv_two_dx);
          const float dvdy = div_c(vnn - vsn, S.two_dy, S.inv_two_dy);
          const float r = S.rho_over_dt * (dudx + dvdy);
          rr[a][k] = on_domain_edge<PS>(E, a, k) ? 0.0f : S.dxdy * r;
        }
    }
  }

  PDEGYM_STAMP(2, rr[0][0]);
  // ---- K Jacobi sweeps (:104-114), p and rr in registers, rotating row map (no copies) ----
  float pf[PS][PS];
  {
    float ph[PS + 1][PS];
    {
      float tmp[PS][PS];
      load_patch<PS>(tmp, p, n, r0, c0);
#pragma unroll
      for (int a = 0; a < PS; ++a)
#pragma unroll
        for (int k = 0; k < PS; ++k) ph[a][k] = tmp[a][k];
#pragma unroll
      for (int k = 0; k < PS; ++k) ph[PS][k] = 0.f;
    }
    int it = 0, rot = 0;
    while (true) {
      const int r = SweepChain<PS, 0>::run(ph, rr, E, lds, xc, tid, ty, tx, it, C.iters);
      if (r >= 0) {
        rot = r;
        break;
      }
    }
    unrotate<PS, 0>(ph, pf, rot);
  }
  PDEGYM_STAMP(3, pf[0][0]);
  store_patch<PS>(pf, p, n, r0, c0);

  // ---- corrector (:143-146), observation, reward ----
  float acc = 0.f;
  const int t = P.time_index[b] + 1;
  const int tr = t < C.nt_ref ? t : C.nt_ref - 1;
  {
    Halo<PS> H;
    halo_exchange<PS>(pf, H, lds, xc, tid, ty, tx);
    float uf[PS][PS], vf[PS][PS];
    load_patch<PS>(uf, us, n, r0, c0);   // written by this same thread above
    load_patch<PS>(vf, vs, n, r0, c0);
#pragma unroll
    for (int a = 0; a < PS; ++a)
#pragma unroll
      for (int k = 0; k < PS; ++k) {
        const float pw = (k == 0) ? H.l[a] : pf[a][k - 1], pe = (k == PS - 1) ? H.r[a] : pf[a][k + 1];
        const float ps = (a == 0) ? H.t[k] : pf[a - 1][k], pn = (a == PS - 1) ? H.b[k] : pf[a + 1][k];
        const float dpdx = div_c(pe - pw, S.two_dx, S.inv_two_dx);
        const float dpdy = div_c(pn - ps, S.two_dy, S.inv_two_dy);
        const bool edge = on_domain_edge<PS>(E, a, k);
        uf[a][k] = edge ? uf[a][k] : uf[a][k] - S.dt_over_rho * dpdx;
        vf[a][k] = edge ? vf[a][k] : vf[a][k] - S.dt_over_rho * dpdy;
      }
    apply_bc_patch<PS>(uf, E, C.bc, 0, act, C.action_dim, r0, c0);
    apply_bc_patch<PS>(vf, E, C.bc, 1, act, C.action_dim, r0, c0);
    store_patch<PS>(uf, u, n, r0, c0);
    store_patch<PS>(vf, v, n, r0, c0);
    const float* uref = P.U_ref + (size_t)tr * ncell * 2;
    float* obs = P.obs + (size_t)b * ncell * 2;
#pragma unroll
    for (int a = 0; a < PS; ++a) {
      const size_t o = ((size_t)(r0 + a) * n + c0) * 2;
      const float4* rrow = reinterpret_cast<const float4*>(uref + o);
      float4* orow = reinterpret_cast<float4*>(obs + o);
#pragma unroll
      for (int q = 0; q < PS / 2; ++q) {
        const float4 w = rrow[q];
        const float a0 = uf[a][2 * q], b0 = vf[a][2 * q], a1 = uf[a][2 * q + 1], b1 = vf[a][2 * q + 1];
        orow[q] = make_float4(a0, b0, a1, b1);
        const float d0 = a0 - w.x, d1 = b0 - w.y, d2 = a1 - w.z, d3 = b1 - w.w;
        acc += d0 * d0;
        acc += d1 * d1;
        acc += d2 * d2;
        acc += d3 * d3;
      }
    }
  }
  PDEGYM_STAMP(4, acc);
  float* red = reinterpret_cast<float*>(lds);   // halo buffers are idle now (block_sum syncs first)
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
    unsigned int* dbg = reinterpret_cast<unsigned int*>(us + 2 * ncell);   // rhs quarter of the scratch is unused here
    for (int i = 0; i < 5; ++i) dbg[i] = (unsigned int)(tm[i + 1] - tm[i]);
#endif
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
  S.inv_two_dx = (T)(1.0 / (2 * prm->dx));
  S.inv_two_dy = (T)(1.0 / (2 * prm->dy));
  S.inv_dxdy = (T)(1.0 / (prm->dx * prm->dy));
  return 0;
}

// PDEGYM_NS_GENERIC=1 in the environment routes float32 steps through ns_generic (A/B testing of the tiled path)
inline bool pdegym_force_generic() {
  const char* e = getenv("PDEGYM_NS_GENERIC");
  return e && e[0] == '1';
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
  if constexpr (sizeof(T) == 4) {
    // register-tiled float32 path for the square grids it is instantiated for (BASELINE config 4 is 128x128)
    if (!pdegym_force_generic() && C.nx == C.ny && (C.nx == 128 || C.nx == 64)) {
      if (C.nx == 128)
        hipLaunchKernelGGL(ns_tile_step<8>, dim3(B), dim3(256), TileCfg<8>::LDS_BYTES, (hipStream_t)stream, C, S, P, B);
      else
        hipLaunchKernelGGL(ns_tile_step<4>, dim3(B), dim3(256), TileCfg<4>::LDS_BYTES, (hipStream_t)stream, C, S, P, B);
      return pdegym::check_launch("ns2d_tile_step");
    }
  }
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
