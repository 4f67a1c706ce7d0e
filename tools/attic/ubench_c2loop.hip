// Micro-benchmark: the C2 sub-step loop (EPL = 4, 25 VALU + loop control per sub-step) in isolation, to separate the loop's
// issue rate from the kernel's prologue / epilogue and to test loop shapes (unroll, alignment, scalar loop control).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o tools/ubench_c2loop tools/ubench_c2loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float shl1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

template <int UNROLL, int ALIGN, bool NODPP>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int nsub) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  float x[4], c[4], fe[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { x[e] = in[e] + tid * 1e-6f; c[e] = in[4 + e]; fe[e] = in[8 + e]; }
  auto sub = [&]() {
    const float xl = NODPP ? x[3] : shr1(x[3]);
    const float xr = NODPP ? x[0] : shl1(x[0]);
    float t2[4], t3[4], t4[4], t5[4], t7[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) t2[e] = (e == 0) ? xl + (-2.0f * x[0]) : __builtin_fmaf(-2.0f, x[e], x[e - 1]);
#pragma unroll
    for (int e = 0; e < 4; ++e) t3[e] = t2[e] + ((e == 3) ? xr : x[e + 1]);
#pragma unroll
    for (int e = 0; e < 4; ++e) t7[e] = c[e] * x[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) t4[e] = fe[e] * t3[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) t5[e] = x[e] + t4[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = t5[e] + t7[e];
  };
  if (ALIGN) asm volatile(".p2align %0" ::"n"(ALIGN));
  int i = 0;
  for (; i + UNROLL - 1 < nsub; i += UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) sub();
  }
  out[tid] = x[0] + x[1] + x[2] + x[3];
}

template <int UNROLL, int ALIGN, bool NODPP>
void run(const char* name, float* out, float* in) {
  for (int w : {1, 2, 3, 4, 6, 8}) {
    const int blocks = 256 * w, nsub = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<UNROLL, ALIGN, NODPP>), dim3(blocks), dim3(256), 0, 0, out, in, 100);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<UNROLL, ALIGN, NODPP>), dim3(blocks), dim3(256), 0, 0, out, in, nsub);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%-26s waves/SIMD=%d  %7.1f ns per sub-step of the SIMD's waves  (%.1f ns per wave-sub-step)\n", name, w,
           best * 1e6 / nsub, best * 1e6 / nsub / w);
  }
}

int main() {
  float *out, *in;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipMalloc(&in, 64);
  const float h[12] = {0.1f, 0.2f, 0.3f, 0.4f, 1e-4f, 2e-4f, 3e-4f, 4e-4f, 0.25f, 0.25f, 0.25f, 0.25f};
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  run<1, 0, false>("unroll1", out, in);
  run<1, 6, false>("unroll1 align64", out, in);
  run<1, 7, false>("unroll1 align128", out, in);
  run<2, 0, false>("unroll2", out, in);
  run<2, 7, false>("unroll2 align128", out, in);
  run<4, 0, false>("unroll4", out, in);
  run<1, 0, true>("unroll1 nodpp", out, in);
  run<2, 0, true>("unroll2 nodpp", out, in);
  return 0;
}
