// Micro-benchmark: issue cost of the DPP forms used for neighbour-lane operands on gfx950.  Every variant runs the same
// 16-instruction group (14 plain v_add/v_fma + 2 "neighbour" adds) 15 times per iteration, 3 waves per SIMD, no barrier.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_dpp tools/ubench_dpp.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__device__ __forceinline__ void group(float (&s)[4], const float (&x)[4], const float (&n)[4], const float (&q)[4]) {
  if constexpr (MODE == 0) {          // plain adds only (no lane crossing)
    asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\tv_add_f32 %0, %7, %0\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %4, %3\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15"
                 : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3])
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "s"(0.25f));
  } else if constexpr (MODE == 1) {   // wave_shr / wave_shl (what the kernels use)
    asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\tv_add_f32_dpp %0, %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32_dpp %3, %4, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15"
                 : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3])
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "s"(0.25f));
  } else if constexpr (MODE == 2) {   // row_shr / row_shl (16-lane rows only)
    asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\tv_add_f32_dpp %0, %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32_dpp %3, %4, %3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15"
                 : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3])
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "s"(0.25f));
  } else {                            // quad_perm (within 4 lanes)
    asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\tv_add_f32_dpp %0, %7, %0 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32_dpp %3, %4, %3 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                 "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15"
                 : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3])
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "s"(0.25f));
  }
}

// all 2 x 15 lane-crossing values of a sweep fetched back to back (v_mov_b32_dpp), then 15 plain groups
__device__ __forceinline__ void group_pre(float (&s)[4], const float (&x)[4], const float (&n)[4], const float (&q)[4], float l, float r) {
  asm volatile("v_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2\n\tv_add_f32 %3, %6, %3\n\tv_add_f32 %0, %17, %0\n\t"
               "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %18, %3\n\t"
               "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
               "v_fma_f32 %0, %0, %16, -%12\n\tv_fma_f32 %1, %1, %16, -%13\n\tv_fma_f32 %2, %2, %16, -%14\n\tv_fma_f32 %3, %3, %16, -%15"
               : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3])
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "s"(0.25f),
                 "v"(l), "v"(r));
}
__device__ __forceinline__ float dpp_left(float v) {
  float r;
  asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ float dpp_right(float v) {
  float r;
  asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v));
  return r;
}

__device__ __forceinline__ float bperm(int addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}
template <int PAT>
__device__ __forceinline__ float swz(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), PAT));
}

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(float* out, int iters) {
  constexpr int PR = 15;
  float ph[PR + 1][4], rq[PR][4];
#pragma unroll
  for (int r = 0; r <= PR; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) { ph[r][c] = threadIdx.x * 1e-3f + r + c; if (r < PR) rq[r][c] = 0.01f * (r + c); }
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 5 || MODE == 6) {
      // neighbour-lane values through the LDS crossbar (ds_bpermute_b32 / ds_swizzle_b32), requested one group ahead
      const int al = (((threadIdx.x & 63) + 63) & 63) * 4, ar = (((threadIdx.x & 63) + 1) & 63) * 4;
      auto L = [&](float v) { return MODE == 5 ? bperm(al, v) : swz<0x8000 | (31 << 5) | 0>(v); };
      auto R = [&](float v) { return MODE == 5 ? bperm(ar, v) : swz<0x8000 | (1 << 5) | 0>(v); };
      float l0 = L(ph[0][3]), r0 = R(ph[0][0]);
#pragma unroll
      for (int a = 0; a < PR; ++a) {
        float l1 = 0, r1 = 0;
        if (a + 1 < PR) { l1 = L(ph[a + 1][3]); r1 = R(ph[a + 1][0]); }
        asm volatile("" ::: "memory");
        group_pre(ph[a == 0 ? PR : a - 1], ph[a], ph[a + 1 == PR ? PR : a + 1], rq[a], l0, r0);
        l0 = l1; r0 = r1;
      }
      l0 = L(ph[PR - 2][3]); r0 = R(ph[PR - 2][0]);
#pragma unroll
      for (int a = PR - 1; a >= 0; --a) {
        float l1 = 0, r1 = 0;
        if (a >= 1) { const int c = (a - 1 == 0) ? PR : a - 2; l1 = L(ph[c][3]); r1 = R(ph[c][0]); }
        asm volatile("" ::: "memory");
        group_pre(ph[a], ph[a == 0 ? PR : a - 1], ph[a == 0 ? PR - 1 : (a == 1 ? PR : a - 2)], rq[a], l0, r0);
        l0 = l1; r0 = r1;
      }
    } else if constexpr (MODE == 4) {
      float hl[PR], hr[PR];
#pragma unroll
      for (int a = 0; a < PR; ++a) { hl[a] = dpp_left(ph[a][3]); hr[a] = dpp_right(ph[a][0]); }
#pragma unroll
      for (int a = 0; a < PR; ++a) group_pre(ph[a == 0 ? PR : a - 1], ph[a], ph[a + 1 == PR ? PR : a + 1], rq[a], hl[a], hr[a]);
#pragma unroll
      for (int a = 0; a < PR; ++a) { hl[a] = dpp_left(ph[a == 0 ? PR : a - 1][3]); hr[a] = dpp_right(ph[a == 0 ? PR : a - 1][0]); }
#pragma unroll
      for (int a = PR - 1; a >= 0; --a) group_pre(ph[a], ph[a == 0 ? PR : a - 1], ph[a == 0 ? PR - 1 : (a == 1 ? PR : a - 2)], rq[a], hl[a], hr[a]);
    } else {
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

template <int MODE, int NT>
void run(const char* name) {
  float* out;
  hipMalloc(&out, (size_t)256 * NT * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, out, 50);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9 / (2.0 * iters);     // per sweep-equivalent (15 groups per wave)
  printf("%-24s waves/SIMD=%d  %.0f cycles at 2.4 GHz per 15 groups per wave-slot set = %.2f cycles per wave-instruction\n", name, NT / 256, cyc,
         cyc / (15.0 * 16 * (NT / 256)));
  hipFree(out);
}

int main() {
  for (int rep = 0; rep < 1; ++rep) {
    run<0, 512>("no lane crossing");
    run<1, 512>("wave_shr/wave_shl");
    run<5, 512>("ds_bpermute one group ahead");
    run<6, 512>("ds_swizzle one group ahead");
    run<5, 1024>("ds_bpermute one group ahead");
    run<6, 1024>("ds_swizzle one group ahead");
  }
  run<0, 512>("no lane crossing");
  run<1, 512>("wave_shr/wave_shl");
  run<2, 512>("row_shr/row_shl");
  run<3, 512>("quad_perm");
  run<0, 768>("no lane crossing");
  run<1, 768>("wave_shr/wave_shl");
  run<2, 768>("row_shr/row_shl");
  run<3, 768>("quad_perm");
  run<4, 768>("30 v_mov_dpp up front");
  run<4, 1024>("30 v_mov_dpp up front");
  run<0, 1024>("no lane crossing");
  run<1, 1024>("wave_shr/wave_shl");
  run<2, 1024>("row_shr/row_shl");
  return 0;
}
