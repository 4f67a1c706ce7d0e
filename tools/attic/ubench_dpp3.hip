// Extra SIMD time of ONE DPP operation embedded in a stream of plain VALU operations, as a function of the number K of plain
// operations between two DPP operations (4 waves per SIMD, independent registers, no data dependence on the DPP result
// unless DEP).  cycles(K, with dpp) - cycles(K+1 plain) = extra cost of making one of the K+1 operations a DPP one.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K, int KIND>   // KIND 0: all plain; 1: one v_add_f32_dpp per K plain adds, independent; 2: the next add consumes the DPP result
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a) {
  constexpr int N = 16;
  float x[N];
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
      for (int j = 0; j < K; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(rep * (K + 1) + j) % N]) : "v"(a));
      const int d = (rep * (K + 1) + K) % N;
      if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[d]) : "v"(a));
      else if (KIND == 1) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[d]) : "v"(x[(d + 8) % N]));
      else {
        asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[d]) : "v"(x[(d + 8) % N]));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K, int KIND>
double run() {
  float* out;
  hipMalloc(&out, (size_t)256 * 1024 * 4);
  const int iters = 3000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<K, KIND>), dim3(256), dim3(1024), 0, 0, out, 100, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<K, KIND>), dim3(256), dim3(1024), 0, 0, out, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms * 1e-3 * 2.4e9 / ((double)iters * 8 * 4);     // cycles per (K plain + 1) group per wave
}
template <int K>
void both() {
  const double p = run<K, 0>(), d = run<K, 1>();
  printf("K=%2d plain ops per DPP: group of K+1 plain = %.1f cycles, with one DPP = %.1f cycles -> extra %.1f cycles per DPP\n", K, p, d, d - p);
}
int main() {
  both<0>(); both<1>(); both<3>(); both<7>(); both<15>(); both<31>();
  return 0;
}
