// Micro-benchmark: cost of cross-lane primitives on gfx950, in SIMD cycles per wave-instruction (2.4 GHz nominal).
// Each kernel issues 8 independent operations per iteration per wave; WPS waves per SIMD (one workgroup per CU).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_xlane tools/ubench_xlane.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a) {
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  const int addr_l = (((threadIdx.x & 63) + 63) & 63) * 4, addr_r = (((threadIdx.x & 63) + 1) & 63) * 4;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[i]) : "v"(x[(i + 4) & 7]));
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[i]) : "v"(x[(i + 4) & 7]));
    } else if constexpr (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[i]) : "v"(x[(i + 4) & 7]));
    } else if constexpr (MODE == 4) {     // ds_bpermute_b32: LDS crossbar, no LDS memory
      float y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(i & 1 ? addr_l : addr_r, __builtin_bit_cast(int, x[i])));
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = y[i];
    } else if constexpr (MODE == 5) {     // 4 plain adds + 4 bpermutes whose results are consumed one iteration later
      float y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr_l, __builtin_bit_cast(int, x[i])));
#pragma unroll
      for (int i = 4; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = y[i];
    } else if constexpr (MODE == 6) {     // v_permlane32_swap (gfx950): swaps the upper half of src0 with the lower half of src1
#pragma unroll
      for (int i = 0; i < 8; i += 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[i + 1]));
#pragma unroll
      for (int i = 0; i < 8; i += 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i + 1]), "+v"(x[i]));
    } else if constexpr (MODE == 7) {     // v_permlane16_swap
#pragma unroll
      for (int i = 0; i < 8; i += 2) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[i + 1]));
#pragma unroll
      for (int i = 0; i < 8; i += 2) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[i + 1]), "+v"(x[i]));
    } else if constexpr (MODE == 8) {     // quad_perm dpp
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[i]) : "v"(x[(i + 4) & 7]));
    } else if constexpr (MODE == 9) {     // row_ror (rotate within 16 lanes)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(x[(i + 4) & 7]));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int wps) {
  float* out;
  hipMalloc(&out, (size_t)256 * 1024 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, 100, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s waves/SIMD=%d  %.2f cycles per wave-instruction\n", name, wps, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wps));
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_add_f32", w);
    run<1>("v_add_f32_dpp wave_shr", w);
    run<2>("v_add_f32_dpp row_shr", w);
    run<8>("v_add_f32_dpp quad_perm", w);
    run<9>("v_add_f32_dpp row_ror", w);
    run<3>("v_mov_b32_dpp wave_shr", w);
    run<4>("ds_bpermute_b32 x8", w);
    run<5>("4 ds_bpermute + 4 v_add", w);
    run<6>("v_permlane32_swap", w);
    run<7>("v_permlane16_swap", w);
  }
  return 0;
}
