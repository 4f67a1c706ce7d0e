// Micro-benchmark: WHY does a lane-crossing operand cost ~15 cycles inside the Jacobi row group (tools/ubench_dpp.hip) when a
// stream of nothing but DPP instructions runs at ~5?  Same 16-instruction group, variants of where the two DPP reads sit and
// what they read.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_dpp5 tools/ubench_dpp5.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define DPPR "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define DPPL "wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define OPS : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]) \
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "s"(0.25f)

template <int MODE>
__device__ __forceinline__ void group(float (&s)[4], const float (&x)[4], const float (&n)[4], const float (&q)[4]) {
  if constexpr (MODE == 0) {          // no lane crossing
    asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\tv_add_f32 %0, %7, %0\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %4, %3\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15" OPS);
  } else if constexpr (MODE == 1) {   // as the kernels: DPP adds 4th and 8th
    asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\tv_add_f32_dpp %0, %7, %0 " DPPR "\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32_dpp %3, %4, %3 " DPPL "\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15" OPS);
  } else if constexpr (MODE == 2) {   // the DPP operands are registers nobody ever writes (q): is it the read-after-write distance?
    asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\tv_add_f32_dpp %0, %15, %0 " DPPR "\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32_dpp %3, %12, %3 " DPPL "\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15" OPS);
  } else if constexpr (MODE == 3) {   // both DPP adds first, back to back
    asm volatile("v_add_f32_dpp %0, %7, %0 " DPPR "\n\tv_add_f32_dpp %3, %4, %3 " DPPL "\n\t"
                 "v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15" OPS);
  } else if constexpr (MODE == 4) {   // 16 DPP adds: is a DPP-only group cheap (the pure-stream number)?
    asm volatile("v_add_f32_dpp %1, %4, %1 " DPPR "\n\tv_add_f32_dpp %2, %5, %2 " DPPR "\n\tv_add_f32_dpp %3, %6, %3 " DPPR "\n\tv_add_f32_dpp %0, %7, %0 " DPPR "\n\t"
                 "v_add_f32_dpp %0, %5, %0 " DPPR "\n\tv_add_f32_dpp %1, %6, %1 " DPPR "\n\tv_add_f32_dpp %2, %7, %2 " DPPR "\n\tv_add_f32_dpp %3, %4, %3 " DPPL "\n\t"
                 "v_add_f32_dpp %0, %8, %0 " DPPR "\n\tv_add_f32_dpp %1, %9, %1 " DPPR "\n\tv_add_f32_dpp %2, %10, %2 " DPPR "\n\tv_add_f32_dpp %3, %11, %3 " DPPR "\n\t"
                 "v_add_f32_dpp %0, %12, %0 " DPPR "\n\tv_add_f32_dpp %1, %13, %1 " DPPR "\n\tv_add_f32_dpp %2, %14, %2 " DPPR "\n\tv_add_f32_dpp %3, %15, %3 " DPPR OPS);
  } else if constexpr (MODE == 5) {   // quad_perm identity (no data actually crosses) in the kernels' positions
    asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\tv_add_f32_dpp %0, %7, %0 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32_dpp %3, %4, %3 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15" OPS);
  } else if constexpr (MODE == 6) {   // the fma as VOP2 fmac-free form is VOP3 (8 bytes); here: is it the VOP3 + DPP mix? use v_mul instead of fma
    asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\tv_add_f32_dpp %0, %7, %0 " DPPR "\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32_dpp %3, %4, %3 " DPPL "\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_sub_f32 %0, %0, %12\n\tv_sub_f32 %1, %1, %13\n\tv_sub_f32 %2, %2, %14\n\tv_sub_f32 %3, %3, %15" OPS);
  } else if constexpr (MODE == 8) {   // every instruction 8 bytes (VOP3 adds), DPP adds at 8-byte aligned offsets
    asm volatile("v_add_f32_e64 %1, %4, %1\n\tv_add_f32_e64 %2, %5, %2\n\tv_add_f32_e64 %3, %6, %3\n\tv_add_f32_dpp %0, %7, %0 " DPPR "\n\t"
                 "v_add_f32_e64 %0, %0, %5\n\tv_add_f32_e64 %1, %1, %6\n\tv_add_f32_e64 %2, %2, %7\n\tv_add_f32_dpp %3, %4, %3 " DPPL "\n\t"
                 "v_add_f32_e64 %0, %0, %8\n\tv_add_f32_e64 %1, %1, %9\n\tv_add_f32_e64 %2, %2, %10\n\tv_add_f32_e64 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15" OPS);
  } else if constexpr (MODE == 9) {   // every instruction 8 bytes, no DPP
    asm volatile("v_add_f32_e64 %1, %4, %1\n\tv_add_f32_e64 %2, %5, %2\n\tv_add_f32_e64 %3, %6, %3\n\tv_add_f32_e64 %0, %7, %0\n\t"
                 "v_add_f32_e64 %0, %0, %5\n\tv_add_f32_e64 %1, %1, %6\n\tv_add_f32_e64 %2, %2, %7\n\tv_add_f32_e64 %3, %4, %3\n\t"
                 "v_add_f32_e64 %0, %0, %8\n\tv_add_f32_e64 %1, %1, %9\n\tv_add_f32_e64 %2, %2, %10\n\tv_add_f32_e64 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15" OPS);
  } else if constexpr (MODE == 10) {  // s_nop 0 after each DPP add (one extra issue slot, 4 bytes: realigns the stream)
    asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\ts_nop 0\n\tv_add_f32_dpp %0, %7, %0 " DPPR "\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\ts_nop 0\n\tv_add_f32_dpp %3, %4, %3 " DPPL "\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15" OPS);
  } else if constexpr (MODE >= 20 && MODE < 80) {
    constexpr int NB = (MODE < 40) ? MODE - 20 : (MODE >= 60 ? MODE - 60 : -1);
    constexpr int NA = (MODE >= 40 && MODE < 60) ? MODE - 40 : (MODE >= 60 ? MODE - 60 : -1);
    asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\t"
                 ".if %17 >= 0\n\ts_nop %17\n\t.endif\n\t"
                 "v_add_f32_dpp %0, %7, %0 " DPPR "\n\t"
                 ".if %18 >= 0\n\ts_nop %18\n\t.endif\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\t"
                 ".if %17 >= 0\n\ts_nop %17\n\t.endif\n\t"
                 "v_add_f32_dpp %3, %4, %3 " DPPL "\n\t"
                 ".if %18 >= 0\n\ts_nop %18\n\t.endif\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15"
                 : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3])
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "s"(0.25f),
                   "i"(NB), "i"(NA));
  } else if constexpr (MODE == 7) {   // plain group with v_sub instead of fma (baseline for 6)
    asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\tv_add_f32 %0, %7, %0\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %4, %3\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_sub_f32 %0, %0, %12\n\tv_sub_f32 %1, %1, %13\n\tv_sub_f32 %2, %2, %14\n\tv_sub_f32 %3, %3, %15" OPS);
  }
}

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(float* out, int iters) {
  constexpr int PR = 15;
  if constexpr (MODE >= 100) {         // MODE 100 + k: every k-th group carries the two DPP adds, the others none
    constexpr int K = MODE - 100;
    float ph[PR + 1][4], rq[PR][4];
#pragma unroll
    for (int r = 0; r <= PR; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) { ph[r][c] = threadIdx.x * 1e-3f + r + c; if (r < PR) rq[r][c] = 0.01f * (r + c); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int a = 0; a < PR; ++a) {
        if (a % K == 0) group<1>(ph[a == 0 ? PR : a - 1], ph[a], ph[a + 1 == PR ? PR : a + 1], rq[a]);
        else group<0>(ph[a == 0 ? PR : a - 1], ph[a], ph[a + 1 == PR ? PR : a + 1], rq[a]);
      }
#pragma unroll
      for (int a = PR - 1; a >= 0; --a) {
        if (a % K == 0) group<1>(ph[a], ph[a == 0 ? PR : a - 1], ph[a == 0 ? PR - 1 : (a == 1 ? PR : a - 2)], rq[a]);
        else group<0>(ph[a], ph[a == 0 ? PR : a - 1], ph[a == 0 ? PR - 1 : (a == 1 ? PR : a - 2)], rq[a]);
      }
    }
    float sum = 0;
#pragma unroll
    for (int r = 0; r <= PR; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) sum += ph[r][c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    return;
  }
  if constexpr (MODE == 50) {          // odd waves run the DPP groups, even waves the plain ones
    float ph[PR + 1][4], rq[PR][4];
#pragma unroll
    for (int r = 0; r <= PR; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) { ph[r][c] = threadIdx.x * 1e-3f + r + c; if (r < PR) rq[r][c] = 0.01f * (r + c); }
    if ((threadIdx.x >> 6) & 1) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < PR; ++a) group<1>(ph[a == 0 ? PR : a - 1], ph[a], ph[a + 1 == PR ? PR : a + 1], rq[a]);
#pragma unroll
        for (int a = PR - 1; a >= 0; --a) group<1>(ph[a], ph[a == 0 ? PR : a - 1], ph[a == 0 ? PR - 1 : (a == 1 ? PR : a - 2)], rq[a]);
      }
    } else {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < PR; ++a) group<0>(ph[a == 0 ? PR : a - 1], ph[a], ph[a + 1 == PR ? PR : a + 1], rq[a]);
#pragma unroll
        for (int a = PR - 1; a >= 0; --a) group<0>(ph[a], ph[a == 0 ? PR : a - 1], ph[a == 0 ? PR - 1 : (a == 1 ? PR : a - 2)], rq[a]);
      }
    }
    float sum = 0;
#pragma unroll
    for (int r = 0; r <= PR; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) sum += ph[r][c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    return;
  }
  float ph[PR + 1][4], rq[PR][4];
#pragma unroll
  for (int r = 0; r <= PR; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) { ph[r][c] = threadIdx.x * 1e-3f + r + c; if (r < PR) rq[r][c] = 0.01f * (r + c); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int a = 0; a < PR; ++a) group<MODE>(ph[a == 0 ? PR : a - 1], ph[a], ph[a + 1 == PR ? PR : a + 1], rq[a]);
#pragma unroll
    for (int a = PR - 1; a >= 0; --a) group<MODE>(ph[a], ph[a == 0 ? PR : a - 1], ph[a == 0 ? PR - 1 : (a == 1 ? PR : a - 2)], rq[a]);
  }
  float sum = 0;
#pragma unroll
  for (int r = 0; r <= PR; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) sum += ph[r][c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int V>
__device__ __forceinline__ void pair_up(float (&da)[4], float (&db)[4], const float (&xb)[4], const float (&nb)[4],
                                        const float (&rqa)[4], const float (&rqb)[4]) {
  float t0, t1, t2, t3;
#define D1 " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define D2 " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define NOPA(v) ".if %29 == " #v "\n\ts_nop 0\n\t.endif\n\t"
#define NOPB ".if %29 == 4\n\ts_nop 0\n\t.endif\n\t.if %29 == 6\n\ts_nop 0\n\t.endif\n\t"
#define NOPC ".if %29 == 4\n\ts_nop 0\n\t.endif\n\t.if %29 == 5\n\ts_nop 1\n\t.endif\n\t.if %29 == 7\n\ts_nop 0\n\ts_nop 0\n\t.endif\n\t"
  asm volatile(
      "v_add_f32 %1, %4, %1\n\t"
      "v_add_f32 %9, %12, %5\n\t"
      "v_add_f32 %2, %5, %2\n\t"
      "v_add_f32 %10, %13, %6\n\t"
      "v_add_f32 %3, %6, %3\n\t"
      "v_add_f32 %11, %14, %7\n\t"
      ".if %29 == 3\n\tv_add_f32 %0, %7, %0\n\tv_add_f32 %8, %15, %4\n\t.else\n\t"
      NOPB "v_add_f32_dpp %0, %7, %0" D1 "\n\t" NOPA(1) NOPC
      NOPB "v_add_f32_dpp %8, %15, %4" D1 "\n\t" NOPA(1) NOPA(2) NOPC
      ".endif\n\t"
      "v_add_f32 %0, %0, %5\n\t"
      "v_add_f32 %8, %8, %13\n\t"
      "v_add_f32 %1, %1, %6\n\t"
      "v_add_f32 %9, %9, %14\n\t"
      "v_add_f32 %2, %2, %7\n\t"
      "v_add_f32 %10, %10, %15\n\t"
      ".if %29 == 3\n\tv_add_f32 %3, %4, %3\n\tv_add_f32 %11, %12, %11\n\t.else\n\t"
      NOPB "v_add_f32_dpp %3, %4, %3" D2 "\n\t" NOPA(1) NOPC
      NOPB "v_add_f32_dpp %11, %12, %11" D2 "\n\t" NOPA(1) NOPA(2) NOPC
      ".endif\n\t"
      "v_add_f32 %0, %0, %12\n\t"
      "v_add_f32 %8, %8, %16\n\t"
      "v_add_f32 %1, %1, %13\n\t"
      "v_add_f32 %9, %9, %17\n\t"
      "v_add_f32 %2, %2, %14\n\t"
      "v_add_f32 %10, %10, %18\n\t"
      "v_add_f32 %3, %3, %15\n\t"
      "v_add_f32 %11, %11, %19\n\t"
      "v_fma_f32 %0, %0, %28, -%20\n\t"
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
        "v"(rqa[2]), "v"(rqa[3]), "v"(rqb[0]), "v"(rqb[1]), "v"(rqb[2]), "v"(rqb[3]), "s"(0.25f), "i"(V));
}

template <int V, int NT>
__global__ __launch_bounds__(NT) void kpair(float* out, int iters) {
  constexpr int PR = 16;
  float ph[PR + 2][4], rq[PR][4];
#pragma unroll
  for (int r = 0; r <= PR + 1; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) { ph[r][c] = threadIdx.x * 1e-3f + r + c; if (r < PR) rq[r][c] = 0.01f * (r + c); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)
#pragma unroll
      for (int a = 0; a < PR; a += 2) pair_up<V>(ph[a == 0 ? PR : a - 1], ph[a], ph[a + 1], ph[a + 2 == PR ? PR + 1 : a + 2], rq[a], rq[a + 1]);
  }
  float sum = 0;
#pragma unroll
  for (int r = 0; r <= PR + 1; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) sum += ph[r][c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int V, int NT>
void runpair(const char* name) {
  float* out;
  (void)hipMalloc(&out, (size_t)256 * NT * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((kpair<V, NT>), dim3(256), dim3(NT), 0, 0, out, 50);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((kpair<V, NT>), dim3(256), dim3(NT), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9 / iters;     // 16 pair blocks of 32 arithmetic instructions
  printf("%-44s waves/SIMD=%d  %.1f cycles per pair block and SIMD (32 arithmetic instructions per wave)\n", name, NT / 256, cyc / 16.0);
  (void)hipFree(out);
}

template <int MODE, int NT, int PR>
__global__ __launch_bounds__(NT) void kpr(float* out, int iters) {
  float ph[PR + 1][4], rq[PR][4];
#pragma unroll
  for (int r = 0; r <= PR; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) { ph[r][c] = threadIdx.x * 1e-3f + r + c; if (r < PR) rq[r][c] = 0.01f * (r + c); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 15 / PR; ++rep) {
#pragma unroll
      for (int a = 0; a < PR; ++a) group<MODE>(ph[a == 0 ? PR : a - 1], ph[a], ph[a + 1 == PR ? PR : a + 1], rq[a]);
#pragma unroll
      for (int a = PR - 1; a >= 0; --a) group<MODE>(ph[a], ph[a == 0 ? PR : a - 1], ph[a == 0 ? PR - 1 : (a == 1 ? PR : a - 2)], rq[a]);
    }
  }
  float sum = 0;
#pragma unroll
  for (int r = 0; r <= PR; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) sum += ph[r][c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int MODE, int NT, int PR>
void runpr(const char* name) {
  float* out;
  (void)hipMalloc(&out, (size_t)256 * NT * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((kpr<MODE, NT, PR>), dim3(256), dim3(NT), 0, 0, out, 50);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((kpr<MODE, NT, PR>), dim3(256), dim3(NT), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9 / (2.0 * iters);
  printf("%-34s rows=%2d waves/SIMD=%d  %.2f cycles per wave-instruction\n", name, PR, NT / 256, cyc / ((15 / PR) * PR * 16.0 * (NT / 256)));
  (void)hipFree(out);
}

template <int MODE, int NT>
void run(const char* name) {
  float* out;
  (void)hipMalloc(&out, (size_t)256 * NT * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, out, 50);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9 / (2.0 * iters);
  printf("%-44s waves/SIMD=%d  %.2f cycles per wave-instruction\n", name, NT / 256, cyc / (15.0 * 16 * (NT / 256)));
  (void)hipFree(out);
}

int main() {
  runpair<3, 512>("pair block, no lane crossing");
  runpair<0, 512>("pair block as in the kernels");
  runpair<1, 512>("pair block, s_nop 0 after each DPP");
  runpair<2, 512>("pair block, s_nop 0 after each DPP pair");
  runpair<4, 512>("pair block, s_nop 0 before and after each DPP");
  runpair<5, 512>("pair block, s_nop 1 after each DPP");
  runpair<6, 512>("pair block, s_nop 0 before each DPP");
  runpair<7, 512>("pair block, two s_nop 0 after each DPP");
  runpair<3, 1024>("pair block, no lane crossing");
  runpair<0, 1024>("pair block as in the kernels");
  runpair<1, 1024>("pair block, s_nop 0 after each DPP");
  runpair<2, 1024>("pair block, s_nop 0 after each DPP pair");
  return 0;
  runpr<0, 512, 15>("plain");
  runpr<1, 512, 15>("DPP, no nop");
  runpr<20, 512, 15>("s_nop 0 before");
  runpr<21, 512, 15>("s_nop 1 before");
  runpr<22, 512, 15>("s_nop 2 before");
  runpr<23, 512, 15>("s_nop 3 before");
  runpr<25, 512, 15>("s_nop 5 before");
  runpr<27, 512, 15>("s_nop 7 before");
  runpr<40, 512, 15>("s_nop 0 after");
  runpr<42, 512, 15>("s_nop 2 after");
  runpr<60, 512, 15>("s_nop 0 before+after");
  runpr<61, 512, 15>("s_nop 1 before+after");
  runpr<63, 512, 15>("s_nop 3 before+after");
  runpr<1, 1024, 15>("DPP, no nop");
  runpr<21, 1024, 15>("s_nop 1 before");
  runpr<23, 1024, 15>("s_nop 3 before");
  runpr<61, 1024, 15>("s_nop 1 before+after");
  return 0;
  runpr<9, 512, 15>("all 8-byte, no crossing");
  runpr<8, 512, 15>("all 8-byte, 2 DPP of 16");
  runpr<10, 512, 15>("s_nop before each DPP");
  runpr<9, 1024, 15>("all 8-byte, no crossing");
  runpr<8, 1024, 15>("all 8-byte, 2 DPP of 16");
  runpr<10, 1024, 15>("s_nop before each DPP");
  runpr<0, 512, 3>("no crossing");
  runpr<1, 512, 3>("2 DPP of 16");
  runpr<0, 512, 5>("no crossing");
  runpr<1, 512, 5>("2 DPP of 16");
  runpr<0, 512, 7>("no crossing");
  runpr<1, 512, 7>("2 DPP of 16");
  runpr<0, 512, 15>("no crossing");
  runpr<1, 512, 15>("2 DPP of 16");
  runpr<0, 1024, 3>("no crossing");
  runpr<1, 1024, 3>("2 DPP of 16");
  runpr<0, 1024, 7>("no crossing");
  runpr<1, 1024, 7>("2 DPP of 16");
  return 0;
  run<0, 512>("no lane crossing");
  run<1, 512>("2 DPP adds of 16 (kernel order)");
  run<2, 512>("2 DPP adds reading never-written registers");
  run<3, 512>("2 DPP adds first, back to back");
  run<4, 512>("16 DPP adds of 16");
  run<5, 512>("2 DPP quad_perm identity");
  run<6, 512>("2 DPP adds, v_sub instead of v_fma");
  run<7, 512>("no crossing, v_sub instead of v_fma");
  run<102, 512>("DPP in every 2nd group");
  run<104, 512>("DPP in every 4th group");
  run<108, 512>("DPP in every 8th group");
  run<115, 512>("DPP in 1 group of 15");
  run<50, 512>("odd waves DPP groups, even waves plain");
  run<0, 256>("no lane crossing");
  run<1, 256>("2 DPP adds of 16 (kernel order)");
  run<0, 1024>("no lane crossing");
  run<1, 1024>("2 DPP adds of 16 (kernel order)");
  run<4, 1024>("16 DPP adds of 16");
  run<6, 1024>("2 DPP adds, v_sub instead of v_fma");
  run<7, 1024>("no crossing, v_sub instead of v_fma");
  return 0;
}
