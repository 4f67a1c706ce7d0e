// Micro-benchmark for VERDICT r3 item 3 ("8-column patches"): the Jacobi two-row block of the float32 NavierStokes2D kernels
// (pdegym_ns_common.h: jacobi_pair_up -- 32 arithmetic instructions for eight cells, FOUR of them v_add_f32_dpp with an s_nop 0 ahead,
// as shipped) against the same 32 instructions for ONE row of eight cells per lane, which needs only TWO lane crossings (West of
// the first cell, East of the last), and against the block without any crossing.  Registers only -- no LDS halo traffic, no
// barrier: the upper bound of what wider patches could buy in the sweeps.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_dpp_8wide tools/ubench_dpp_8wide.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define D1 " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define D2 " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
// V = 0: no crossing; 1: four DPP adds, s_nop 0 ahead of each (shipped 4-wide rows); 2: two DPP adds (8-wide row), s_nop 0 ahead
template <int V>
__device__ __forceinline__ void block(float (&da)[4], float (&db)[4], const float (&xb)[4], const float (&nb)[4],
                                      const float (&rqa)[4], const float (&rqb)[4]) {
  float t0, t1, t2, t3;
  asm volatile(
      "v_add_f32 %1, %4, %1\n\t"
      "v_add_f32 %9, %12, %5\n\t"
      "v_add_f32 %2, %5, %2\n\t"
      "v_add_f32 %10, %13, %6\n\t"
      "v_add_f32 %3, %6, %3\n\t"
      "v_add_f32 %11, %14, %7\n\t"
      ".if %29 == 0\n\tv_add_f32 %0, %7, %0\n\tv_add_f32 %8, %15, %4\n\t.endif\n\t"
      ".if %29 == 1\n\ts_nop 0\n\tv_add_f32_dpp %0, %7, %0" D1 "\n\ts_nop 0\n\tv_add_f32_dpp %8, %15, %4" D1 "\n\t.endif\n\t"
      ".if %29 == 2\n\ts_nop 0\n\tv_add_f32_dpp %0, %7, %0" D1 "\n\tv_add_f32 %8, %15, %4\n\t.endif\n\t"
      "v_add_f32 %0, %0, %5\n\t"
      "v_add_f32 %8, %8, %13\n\t"
      "v_add_f32 %1, %1, %6\n\t"
      "v_add_f32 %9, %9, %14\n\t"
      "v_add_f32 %2, %2, %7\n\t"
      "v_add_f32 %10, %10, %15\n\t"
      ".if %29 == 0\n\tv_add_f32 %3, %4, %3\n\tv_add_f32 %11, %12, %11\n\t.endif\n\t"
      ".if %29 == 1\n\ts_nop 0\n\tv_add_f32_dpp %3, %4, %3" D2 "\n\ts_nop 0\n\tv_add_f32_dpp %11, %12, %11" D2 "\n\t.endif\n\t"
      ".if %29 == 2\n\tv_add_f32 %3, %4, %3\n\ts_nop 0\n\tv_add_f32_dpp %11, %12, %11" D2 "\n\t.endif\n\t"
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
__global__ __launch_bounds__(NT) void kblock(float* out, int iters) {
  constexpr int PR = 8;        // 8 rows x 4 columns per lane, as ns_tile_step<8, 4> (32 cells: the same registers hold 4 rows x 8 columns)
  float ph[PR + 2][4], rq[PR][4];
#pragma unroll
  for (int r = 0; r <= PR + 1; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) { ph[r][c] = threadIdx.x * 1e-3f + r + c; if (r < PR) rq[r][c] = 0.01f * (r + c); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int a = 0; a < PR; a += 2) block<V>(ph[a == 0 ? PR : a - 1], ph[a], ph[a + 1], ph[a + 2 == PR ? PR + 1 : a + 2], rq[a], rq[a + 1]);
  }
  float sum = 0;
#pragma unroll
  for (int r = 0; r <= PR + 1; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) sum += ph[r][c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int V, int NT>
void run(const char* name) {
  float* out;
  (void)hipMalloc(&out, (size_t)256 * NT * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((kblock<V, NT>), dim3(256), dim3(NT), 0, 0, out, 100);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kblock<V, NT>), dim3(256), dim3(NT), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  // per SIMD: (NT / 256) waves x 16 blocks x 32 arithmetic instructions per iteration; time in ns per iteration
  const double ns_iter = best * 1e6 / iters;
  printf("%-52s waves/SIMD=%d  %7.1f ns per 16 blocks and wave set  = %.3f ns per arithmetic wave-instruction and SIMD\n", name, NT / 256,
         ns_iter, ns_iter / (16.0 * 32 * (NT / 256)));
  (void)hipFree(out);
}

int main() {
  run<0, 512>("no lane crossing");
  run<1, 512>("4-wide rows: 4 DPP adds per block (shipped)");
  run<2, 512>("8-wide row: 2 DPP adds per block");
  run<0, 1024>("no lane crossing");
  run<1, 1024>("4-wide rows: 4 DPP adds per block (shipped)");
  run<2, 1024>("8-wide row: 2 DPP adds per block");
  return 0;
}
