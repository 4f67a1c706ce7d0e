// Does the cost of a DPP operation depend on how many distinct registers a wave touches?  N v_mov_b32(_dpp) from N distinct
// sources to N distinct destinations per iteration (no arithmetic), 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N, bool DPP>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  float x[N], y[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { x[i] = threadIdx.x + i; y[i] = 0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (DPP) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(y[i]) : "v"(x[i]));
      else asm volatile("v_mov_b32 %0, %1" : "=v"(y[i]) : "v"(x[i]));
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (DPP) asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(x[i]) : "v"(y[i]));
      else asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) s += x[i] + y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int N, bool DPP>
void run() {
  float* out;
  hipMalloc(&out, (size_t)256 * 1024 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<N, DPP>), dim3(256), dim3(1024), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<N, DPP>), dim3(256), dim3(1024), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("N=%2d %s  %.2f cycles per wave-instruction (4 waves/SIMD)\n", N, DPP ? "v_mov_b32_dpp" : "v_mov_b32    ", ms * 1e-3 * 2.4e9 / ((double)iters * 2 * N * 4));
  hipFree(out);
}
int main() {
  run<4, false>(); run<4, true>(); run<8, false>(); run<8, true>(); run<16, false>(); run<16, true>(); run<30, false>(); run<30, true>(); run<48, false>(); run<48, true>();
  return 0;
}
