// Micro-benchmark: issue cost of f32 VALU forms on gfx950 (design input for the stepper kernels).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  f2 q0 = p0 + 1.f, q1 = p1 + 1.f, q2 = p2 + 1.f, q3 = p3 + 1.f;
  const f2 aa = {a, a}, bb = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent scalar add
      x0 += a; x1 += a; x2 += a; x3 += a; x4 += a; x5 += a; x6 += a; x7 += a;
    } else if (MODE == 1) {  // 8 packed add (16 flops / lane)
      p0 += aa; p1 += aa; p2 += aa; p3 += aa; q0 += aa; q1 += aa; q2 += aa; q3 += aa;
    } else if (MODE == 2) {  // 8 scalar mul
      x0 *= a; x1 *= a; x2 *= a; x3 *= a; x4 *= a; x5 *= a; x6 *= a; x7 *= a;
    } else if (MODE == 3) {  // 8 scalar fma
      x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
      x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    } else if (MODE == 4) {  // 8 packed fma
      p0 = __builtin_elementwise_fma(p0, aa, bb); p1 = __builtin_elementwise_fma(p1, aa, bb);
      p2 = __builtin_elementwise_fma(p2, aa, bb); p3 = __builtin_elementwise_fma(p3, aa, bb);
      q0 = __builtin_elementwise_fma(q0, aa, bb); q1 = __builtin_elementwise_fma(q1, aa, bb);
      q2 = __builtin_elementwise_fma(q2, aa, bb); q3 = __builtin_elementwise_fma(q3, aa, bb);
    } else if (MODE == 5) {  // 8 DPP moves + adds
      x0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x1), 0x138, 0xf, 0xf, false));
      x1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x2), 0x138, 0xf, 0xf, false));
      x2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x3), 0x138, 0xf, 0xf, false));
      x3 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x0), 0x138, 0xf, 0xf, false));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y +
                                               p3.x + p3.y + q0.x + q0.y + q1.x + q1.y + q2.x + q2.y + q3.x + q3.y;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int ninstr) {
  float* out;
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // cycles per wave-instruction per SIMD at 2.4 GHz nominal
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * ninstr * waves_per_simd);
  printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 6, 8}) {
    run<0>("v_add_f32 x8", w, 8);
    run<1>("v_pk_add_f32 x8", w, 8);
    run<2>("v_mul_f32 x8", w, 8);
    run<3>("v_fma_f32 x8", w, 8);
    run<4>("v_pk_fma_f32 x8", w, 8);
    run<5>("dpp mov+add x4 (8 instr)", w, 8);
  }
  return 0;
}
