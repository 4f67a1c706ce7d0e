// Cost of the two lane-crossing adds per patch row as a function of the row width: NP plain VALU operations + 2 DPP adds per
// group (NP = 14: 4 cells per lane and row, NP = 30: 8 cells), 8 dependency chains, 3 or 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NP, bool DPP, int NT>
__global__ __launch_bounds__(NT) void k(float* out, int iters, float a) {
  constexpr int N = 48;
  float x[N];
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
      for (int j = 0; j < NP / 2; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(g * 5 + j) % N]) : "v"(a));
      if (DPP) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[(g * 5 + 17) % N]) : "v"(x[(g * 5 + 29) % N]));
      else asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[(g * 5 + 17) % N]) : "v"(x[(g * 5 + 29) % N]));
#pragma unroll
      for (int j = 0; j < NP / 2; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[(g * 5 + j + 20) % N]) : "v"(a));
      if (DPP) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[(g * 5 + 41) % N]) : "v"(x[(g * 5 + 11) % N]));
      else asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[(g * 5 + 41) % N]) : "v"(x[(g * 5 + 11) % N]));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NP, bool DPP, int NT>
double run() {
  float* out;
  hipMalloc(&out, (size_t)256 * NT * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NP, DPP, NT>), dim3(256), dim3(NT), 0, 0, out, 100, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NP, DPP, NT>), dim3(256), dim3(NT), 0, 0, out, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms * 1e-3 * 2.4e9 / ((double)iters * 8 * (NT / 256));     // cycles per group per wave
}
template <int NP, int NT>
void both() {
  const double p = run<NP, false, NT>(), d = run<NP, true, NT>();
  printf("waves/SIMD=%d  %2d plain + 2 crossing: all plain %.1f cycles per group, with DPP %.1f -> %.1f extra cycles per DPP (%.2f -> %.2f cycles per instruction)\n",
         NT / 256, NP, p, d, (d - p) / 2, p / (NP + 2), d / (NP + 2));
}
int main() {
  both<14, 768>(); both<30, 768>(); both<62, 768>(); both<14, 1024>(); both<30, 1024>(); both<62, 1024>();
  return 0;
}
