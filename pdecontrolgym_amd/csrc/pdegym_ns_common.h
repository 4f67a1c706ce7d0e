// pdegym_ns_common.h -- types and device helpers shared by the Navier-Stokes translation units (pdegym_ns2d.hip: generic,
// tile and column kernels and the C-ABI entry points; pdegym_ns256.hip: the fused 256 x 256 float32 step).
// Reference semantics restated: environments2d/navier_stokes2D.py (see pdegym_ns2d.hip for the line map).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pdegym.h"

#ifndef PDEGYM_NS_DPP_NOP
#define PDEGYM_NS_DPP_NOP 1
#endif
#if PDEGYM_NS_DPP_NOP
#define PDEGYM_DPP_NOP "s_nop 0\n\t"
#else
#define PDEGYM_DPP_NOP
#endif

namespace pdegym {
namespace ns {

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

// Jacobi update 1/4*(s4 - dx*dy*rhs) (navier_stokes2D.py:106-108).  float: the per-sweep constant q = dx*dy*rhs is
// pre-scaled by 0.25 (exact) and the update is one fma: RN(0.25*s4 - 0.25*q) == 0.25*RN(s4 - q) because scaling by
// a power of two commutes with rounding (outside the subnormal range) -- bit-identical, one instruction fewer.
__device__ __forceinline__ float jacobi_rhs_term(float dxdy, float rhs) { return 0.25f * (dxdy * rhs); }
__device__ __forceinline__ float jacobi_update(float s4, float rq) { return __builtin_fmaf(0.25f, s4, -rq); }

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
  const T* state_in;  // optional [B, ny, nx, 2]: (u, v) of the previous call's observation; then u, v may be NULL
  T* p_out;           // optional [B, ny, nx]: the solved pressure goes here instead of back into p
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

template <int N>
struct VecOf;
template <>
struct VecOf<4> { using type = float4; };
template <>
struct VecOf<2> { using type = float2; };

template <int PC>
__device__ __forceinline__ typename VecOf<PC>::type pack_row(const float (&r)[PC]) {
  if constexpr (PC == 4) return make_float4(r[0], r[1], r[2], r[3]);
  else return make_float2(r[0], r[1]);
}
template <int PC>
__device__ __forceinline__ void unpack_row(const typename VecOf<PC>::type& v, float (&r)[PC]) {
  r[0] = v.x;
  r[1] = v.y;
  if constexpr (PC == 4) {
    r[2] = v.z;
    r[3] = v.w;
  }
}

#ifndef PDEGYM_NS_PARK_ROWS
#define PDEGYM_NS_PARK_ROWS 6
#endif
template <int PR, int PC>
struct TileCfg {
  static constexpr int NT = 512;
  static constexpr int N = 32 * PC;                      // grid side (== 16 * PR)
  static constexpr int BUF = 2 * NT;                     // vectors per halo buffer (top edges, bottom edges)
  static constexpr int LDS_BYTES = 2 * BUF * PC * 4;     // two buffers
  // rows of u* that wait in LDS (one PC-wide vector per thread and row) while the pressure solve runs -- the others and v* stay
  // in registers: 128x128 -> 6 of 8 rows = 48 KB, so that two workgroups (2 x 80 KB) share a CU's 160 KB exactly
  // (5 rows: B = 4096 528 instead of 521 us; 4 rows: 533)
  static constexpr int PARK_ROWS = (PR == 8 && PC == 4) ? PDEGYM_NS_PARK_ROWS : 0;
  static constexpr int PARK_BYTES = PARK_ROWS * NT * PC * 4;
  static_assert(16 * PR == 32 * PC, "square grids only");
};

struct EdgeFlags {
  bool top, bot, lef, rig;
};

// lane i <- lane i-1 / lane i+1 (DPP wave_shr:1 / wave_shl:1); the lane without a source gets 0 (never used:
// it is a domain-edge thread)
__device__ __forceinline__ float lane_left(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_right(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// y + (value of x in the lane to the left / right): the DPP shift rides on the add itself (hipcc keeps a separate
// v_mov_b32_dpp otherwise).  s_nop 1 covers the VALU-write -> DPP-read hazard for operands hipcc cannot see into.
__device__ __forceinline__ float add_lane_left(float x, float y) {
  float r;
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ float add_lane_right(float x, float y) {
  float r;
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// top/bottom halo rows through LDS: ht = bottom row of the thread above, hb = top row of the thread below.
// Domain-edge threads re-read their own row (valid address, value never used): every access is unconditional.
template <int PC, int NT = 512, int RS = 32>
__device__ __forceinline__ void halo_tb(const float (&top)[PC], const float (&bot)[PC], float (&ht)[PC], float (&hb)[PC],
                                        float* lds, int& xc, int tid, int ty) {
  using V = typename VecOf<PC>::type;
  V* base = reinterpret_cast<V*>(lds) + (xc & 1) * (2 * NT);
  ++xc;
  V* eT = base;
  V* eB = base + NT;
  eT[tid] = pack_row<PC>(top);
  eB[tid] = pack_row<PC>(bot);
  __syncthreads();
  const int up = (ty > 0) ? tid - RS : tid, dn = (ty < NT / RS - 1) ? tid + RS : tid;
  const V a = eB[up];
  const V b = eT[dn];
  unpack_row<PC>(a, ht);
  unpack_row<PC>(b, hb);
}

// ---- boustrophedon sweeps: a rotation with period two ---------------------------------------------------------------
// ph has PR+1 physical rows.  State 0: logical patch row a sits in physical row a, physical row PR is free.  An UP sweep
// (rows 0 -> PR-1) writes new row a into the registers of old row a-1 (the top halo is loaded into the free row), leaving
// logical row a in physical row a-1 (row 0 in PR) and physical row PR-1 free (state 1).  A DOWN sweep (rows PR-1 -> 0) writes
// new row a into the registers of old row a+1 (the bottom halo is loaded into the free row) and restores state 0.  Two sweep
// bodies instead of PR+1, no register copies, and the halo row a sweep needs FIRST is the row its neighbour produced first
// in the previous sweep.  Arithmetic and its order are those of jacobi_sweep_rot / ns_generic_step<float>.
template <int PR>
__device__ constexpr int bphys(int a, int state) {          // physical row of logical row a (a = -1: top halo in an UP sweep)
  return state == 0 ? a : (a == 0 ? PR : a - 1);
}

// ((W + S) + E) + N -> fma(0.25, ., -rq), in place on the registers of the South row (UP) ...
__device__ __forceinline__ void jacobi_row_into_south(float (&sv)[4], const float (&xv)[4], const float (&nn)[4], const float (&rq)[4]) {
  asm volatile(
      "v_add_f32 %1, %4, %1\n\t"
      "v_add_f32 %2, %5, %2\n\t"
      "v_add_f32 %3, %6, %3\n\t"
      PDEGYM_DPP_NOP
      "v_add_f32_dpp %0, %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32 %0, %0, %5\n\t"
      "v_add_f32 %1, %1, %6\n\t"
      "v_add_f32 %2, %2, %7\n\t"
      PDEGYM_DPP_NOP
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
      : "v"(xv[0]), "v"(xv[1]), "v"(xv[2]), "v"(xv[3]), "v"(nn[0]), "v"(nn[1]), "v"(nn[2]), "v"(nn[3]), "v"(rq[0]), "v"(rq[1]),
        "v"(rq[2]), "v"(rq[3]), "s"(0.25f));
}
// ... and on the registers of the North row (DOWN): the partial sums (W + S) + E need four temporaries
__device__ __forceinline__ void jacobi_row_into_north(float (&nv)[4], const float (&xv)[4], const float (&ss)[4], const float (&rq)[4]) {
  float t0, t1, t2, t3;
  asm volatile(
      "v_add_f32 %5, %8, %13\n\t"
      "v_add_f32 %6, %9, %14\n\t"
      "v_add_f32 %7, %10, %15\n\t"
      PDEGYM_DPP_NOP
      "v_add_f32_dpp %4, %11, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32 %4, %4, %9\n\t"
      "v_add_f32 %5, %5, %10\n\t"
      "v_add_f32 %6, %6, %11\n\t"
      PDEGYM_DPP_NOP
      "v_add_f32_dpp %7, %8, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32 %0, %4, %0\n\t"
      "v_add_f32 %1, %5, %1\n\t"
      "v_add_f32 %2, %6, %2\n\t"
      "v_add_f32 %3, %7, %3\n\t"
      "v_fma_f32 %0, %0, %20, -%16\n\t"
      "v_fma_f32 %1, %1, %20, -%17\n\t"
      "v_fma_f32 %2, %2, %20, -%18\n\t"
      "v_fma_f32 %3, %3, %20, -%19"
      : "+v"(nv[0]), "+v"(nv[1]), "+v"(nv[2]), "+v"(nv[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(xv[0]), "v"(xv[1]), "v"(xv[2]), "v"(xv[3]), "v"(ss[0]), "v"(ss[1]), "v"(ss[2]), "v"(ss[3]), "v"(rq[0]), "v"(rq[1]),
        "v"(rq[2]), "v"(rq[3]), "s"(0.25f));
}

// An s_nop 0 goes ahead of every DPP add: without it a DPP operand costs the SIMD ~15 cycles in these blocks -- the two waves of a
// SIMD stop overlapping, any density of DPP from 2 in 16 up runs at ~4.2 cycles per instruction instead of 2.3-2.5 -- with it
// ~6 (tools/attic/ubench_dpp5.hip: pair block 261 -> 207 cycles per SIMD at two waves, 451 -> 386 at four; 166 / 294 without any
// lane crossing).
// Two rows per block (8 independent dependency chains instead of 4: a wave issues only about every 8 cycles along ONE
// chain of 4, so three or four waves per SIMD cannot fill it with single-row blocks -- tools/attic/ubench_dpp.hip).
// UP, rows a (A) and a+1 (B):  da = old row a-1 (South of A, becomes new row a), db = old row a (centre of A, South of B,
// becomes new row a+1), xb = old row a+1 (North of A, centre of B), nb = old row a+2 / bottom halo (North of B).
// A accumulates in place; B keeps its partial sums in four temporaries until A has read db for the last time (A8).
__device__ __forceinline__ void jacobi_pair_up(float (&da)[4], float (&db)[4], const float (&xb)[4], const float (&nb)[4],
                                               const float (&rqa)[4], const float (&rqb)[4]) {
  float t0, t1, t2, t3;
  asm volatile(
      "v_add_f32 %1, %4, %1\n\t"                  // A: W + S
      "v_add_f32 %9, %12, %5\n\t"                 // B
      "v_add_f32 %2, %5, %2\n\t"
      "v_add_f32 %10, %13, %6\n\t"
      "v_add_f32 %3, %6, %3\n\t"
      "v_add_f32 %11, %14, %7\n\t"
      PDEGYM_DPP_NOP
      "v_add_f32_dpp %0, %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      PDEGYM_DPP_NOP
      "v_add_f32_dpp %8, %15, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32 %0, %0, %5\n\t"                  // + E
      "v_add_f32 %8, %8, %13\n\t"
      "v_add_f32 %1, %1, %6\n\t"
      "v_add_f32 %9, %9, %14\n\t"
      "v_add_f32 %2, %2, %7\n\t"
      "v_add_f32 %10, %10, %15\n\t"
      PDEGYM_DPP_NOP
      "v_add_f32_dpp %3, %4, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      PDEGYM_DPP_NOP
      "v_add_f32_dpp %11, %12, %11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32 %0, %0, %12\n\t"                 // + N
      "v_add_f32 %8, %8, %16\n\t"
      "v_add_f32 %1, %1, %13\n\t"
      "v_add_f32 %9, %9, %17\n\t"
      "v_add_f32 %2, %2, %14\n\t"
      "v_add_f32 %10, %10, %18\n\t"
      "v_add_f32 %3, %3, %15\n\t"
      "v_add_f32 %11, %11, %19\n\t"
      "v_fma_f32 %0, %0, %28, -%20\n\t"           // 0.25 * s4 - rq ; B lands in db (A no longer reads it)
      "v_fma_f32 %4, %8, %28, -%24\n\t"
      "v_fma_f32 %1, %1, %28, -%21\n\t"
      "v_fma_f32 %5, %9, %28, -%25\n\t"
      "v_fma_f32 %2, %2, %28, -%22\n\t"
      "v_fma_f32 %6, %10, %28, -%26\n\t"
      "v_fma_f32 %3, %3, %28, -%23\n\t"
      "v_fma_f32 %7, %11, %28, -%27"
      : "+v"(da[0]), "+v"(da[1]), "+v"(da[2]), "+v"(da[3]), "+v"(db[0]), "+v"(db[1]), "+v"(db[2]), "+v"(db[3]), "=&v"(t0), "=&v"(t1),
        "=&v"(t2), "=&v"(t3)
      : "v"(xb[0]), "v"(xb[1]), "v"(xb[2]), "v"(xb[3]), "v"(nb[0]), "v"(nb[1]), "v"(nb[2]), "v"(nb[3]), "v"(rqa[0]), "v"(rqa[1]),
        "v"(rqa[2]), "v"(rqa[3]), "v"(rqb[0]), "v"(rqb[1]), "v"(rqb[2]), "v"(rqb[3]), "s"(0.25f));
}

// DOWN, rows a (A) and a-1 (B):  da = old row a+1 / bottom halo (North of A, becomes new row a), db = old row a (centre of A,
// North of B, becomes new row a-1), xb = old row a-1 (South of A, centre of B), sb = old row a-2 / top halo (South of B).
// North is the LAST addend of ((W + S) + E) + N, so both rows form (W + S) + E in temporaries first.
__device__ __forceinline__ void jacobi_pair_down(float (&da)[4], float (&db)[4], const float (&xb)[4], const float (&sb)[4],
                                                 const float (&rqa)[4], const float (&rqb)[4]) {
  float a0, a1, a2, a3, b0, b1, b2, b3;
  asm volatile(
      "v_add_f32 %9, %4, %17\n\t"                 // A: W + S   (W = db[k-1], S = xb[k])
      "v_add_f32 %13, %16, %21\n\t"               // B:          (W = xb[k-1], S = sb[k])
      "v_add_f32 %10, %5, %18\n\t"
      "v_add_f32 %14, %17, %22\n\t"
      "v_add_f32 %11, %6, %19\n\t"
      "v_add_f32 %15, %18, %23\n\t"
      PDEGYM_DPP_NOP
      "v_add_f32_dpp %8, %7, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      PDEGYM_DPP_NOP
      "v_add_f32_dpp %12, %19, %20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32 %8, %8, %5\n\t"                  // + E
      "v_add_f32 %12, %12, %17\n\t"
      "v_add_f32 %9, %9, %6\n\t"
      "v_add_f32 %13, %13, %18\n\t"
      "v_add_f32 %10, %10, %7\n\t"
      "v_add_f32 %14, %14, %19\n\t"
      PDEGYM_DPP_NOP
      "v_add_f32_dpp %11, %4, %11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      PDEGYM_DPP_NOP
      "v_add_f32_dpp %15, %16, %15 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32 %0, %8, %0\n\t"                  // + N, in place on the North rows (A has read db for the last time)
      "v_add_f32 %4, %12, %4\n\t"
      "v_add_f32 %1, %9, %1\n\t"
      "v_add_f32 %5, %13, %5\n\t"
      "v_add_f32 %2, %10, %2\n\t"
      "v_add_f32 %6, %14, %6\n\t"
      "v_add_f32 %3, %11, %3\n\t"
      "v_add_f32 %7, %15, %7"
      : "+v"(da[0]), "+v"(da[1]), "+v"(da[2]), "+v"(da[3]), "+v"(db[0]), "+v"(db[1]), "+v"(db[2]), "+v"(db[3]), "=&v"(a0), "=&v"(a1),
        "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
      : "v"(xb[0]), "v"(xb[1]), "v"(xb[2]), "v"(xb[3]), "v"(sb[0]), "v"(sb[1]), "v"(sb[2]), "v"(sb[3]));
  asm volatile(
      "v_fma_f32 %0, %0, %16, -%8\n\t"
      "v_fma_f32 %4, %4, %16, -%12\n\t"
      "v_fma_f32 %1, %1, %16, -%9\n\t"
      "v_fma_f32 %5, %5, %16, -%13\n\t"
      "v_fma_f32 %2, %2, %16, -%10\n\t"
      "v_fma_f32 %6, %6, %16, -%14\n\t"
      "v_fma_f32 %3, %3, %16, -%11\n\t"
      "v_fma_f32 %7, %7, %16, -%15"
      : "+v"(da[0]), "+v"(da[1]), "+v"(da[2]), "+v"(da[3]), "+v"(db[0]), "+v"(db[1]), "+v"(db[2]), "+v"(db[3])
      : "v"(rqa[0]), "v"(rqa[1]), "v"(rqa[2]), "v"(rqa[3]), "v"(rqb[0]), "v"(rqb[1]), "v"(rqb[2]), "v"(rqb[3]), "s"(0.25f));
}

// Neumann walls (:110-113) on the new rows, which sit in state `st`
template <int PR, int ST>
__device__ __forceinline__ void jacobi_walls_state(float (&ph)[PR + 1][4], const EdgeFlags& E) {
  constexpr int n0 = bphys<PR>(0, ST), n1 = bphys<PR>(1, ST), nl = bphys<PR>(PR - 1, ST), nm = bphys<PR>(PR - 2, ST);
  if (E.top) {
#pragma unroll
    for (int k = 0; k < 4; ++k) ph[n0][k] = ph[n1][k];
  }
  if (E.bot) {
#pragma unroll
    for (int k = 0; k < 4; ++k) ph[nl][k] = ph[nm][k];
  }
  if (E.lef) {
#pragma unroll
    for (int a = 0; a < PR; ++a) ph[bphys<PR>(a, ST)][0] = ph[bphys<PR>(a, ST)][1];
  }
  if (E.rig) {
#pragma unroll
    for (int a = 0; a < PR; ++a) ph[bphys<PR>(a, ST)][3] = ph[bphys<PR>(a, ST)][2];
  }
}

// One sweep starting from state ST (0: UP, 1: DOWN).  Halo rows cross thread rows through the double-buffered LDS area of
// halo_tb (one barrier); the halo used LAST lands in four temporaries.
template <int PR, int ST, int NT, int RS>
__device__ __forceinline__ void jacobi_sweep_bous(float (&ph)[PR + 1][4], const float (&rq)[PR][4], const EdgeFlags& E, float* lds,
                                                  int& xc, int tid, int ty) {
  float hlast[4];
  if constexpr (ST == 0) {
    halo_tb<4, NT, RS>(ph[bphys<PR>(0, 0)], ph[bphys<PR>(PR - 1, 0)], ph[PR], hlast, lds, xc, tid, ty);   // top halo -> free row PR
#pragma unroll
    for (int a = 0; a + 1 < PR; a += 2) {       // rows (a, a+1)
      float (&da)[4] = ph[a == 0 ? PR : a - 1];
      if (a + 2 == PR) jacobi_pair_up(da, ph[a], ph[a + 1], hlast, rq[a], rq[a + 1]);
      else jacobi_pair_up(da, ph[a], ph[a + 1], ph[a + 2], rq[a], rq[a + 1]);
    }
    if constexpr (PR % 2 == 1) jacobi_row_into_south(ph[PR - 2], ph[PR - 1], hlast, rq[PR - 1]);
    jacobi_walls_state<PR, 1>(ph, E);
  } else {
    // state 1: logical row a in physical row a-1 (row 0 in PR); physical row PR-1 is free -> bottom halo
    halo_tb<4, NT, RS>(ph[bphys<PR>(0, 1)], ph[bphys<PR>(PR - 1, 1)], hlast, ph[PR - 1], lds, xc, tid, ty);
#pragma unroll
    for (int a = PR - 1; a >= 1; a -= 2) {      // rows (a, a-1): new row a -> physical row a, new row a-1 -> physical row a-1
      // physical rows: old logical a+1 (or bottom halo) = a, old a = a-1, old a-1 = bphys(a-1), old a-2 = bphys(a-2) / top halo
      if (a == 1) jacobi_pair_down(ph[a], ph[a - 1], ph[bphys<PR>(0, 1)], hlast, rq[a], rq[a - 1]);
      else jacobi_pair_down(ph[a], ph[a - 1], ph[bphys<PR>(a - 1, 1)], ph[bphys<PR>(a - 2, 1)], rq[a], rq[a - 1]);
    }
    if constexpr (PR % 2 == 1) jacobi_row_into_north(ph[0], ph[bphys<PR>(0, 1)], hlast, rq[0]);
    jacobi_walls_state<PR, 0>(ph, E);
  }
}

// pdegym_ns256.hip: the whole 256 x 256 float32 env-step in one launch (state_in or separate u, v; any sweep count)
int launch_ns256_fused(const NSConst& C, const NSScal<float>& S, const NSPtrs<float>& P, int B, hipStream_t st);

// pdegym_ns256_f64.hip: the 256 x 256 float64 env-step (front launch, slab passes of the pressure solve, back launch, finish)
int launch_ns256_step_f64(const NSConst& C, const NSScal<double>& S, const NSPtrs<double>& P, int B, hipStream_t st);

}  // namespace ns
}  // namespace pdegym
