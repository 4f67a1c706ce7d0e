// Micro-benchmark (design input for the NS slab Jacobi pass): W waves per SIMD in ONE workgroup per CU run a barrier-
// synchronised loop; every iteration issues ROWS x 19 VALU instructions per wave (the shape of a Jacobi sweep over ROWS
// patch rows of 4 cells: 4 independent chains, stage-major) and then meets all other waves at __syncthreads().  The work
// per SIMD and iteration is the same for every W (ROWS x W = 40), so the time per iteration shows which wave count makes
// the best use of a SIMD when all waves must rendezvous after every sweep.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_barrier tools/ubench_barrier.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ROWS, int NT, bool BARRIER>
__global__ __launch_bounds__(NT) void k(float* out, int iters, float a, float b) {
  float x[ROWS][4], q[ROWS][4];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) { x[r][c] = threadIdx.x + r + c; q[r][c] = b * (r + c); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      float nv[4];
      const int up = r == 0 ? ROWS - 1 : r - 1, dn = r == ROWS - 1 ? 0 : r + 1;
#pragma unroll
      for (int c = 0; c < 4; ++c) nv[c] = x[r][(c + 3) & 3] + x[up][c];
#pragma unroll
      for (int c = 0; c < 4; ++c) nv[c] = nv[c] + x[r][(c + 1) & 3];
#pragma unroll
      for (int c = 0; c < 4; ++c) nv[c] = nv[c] + x[dn][c];
#pragma unroll
      for (int c = 0; c < 4; ++c) nv[c] = __builtin_fmaf(a, nv[c], -q[r][c]);
#pragma unroll
      for (int c = 0; c < 4; ++c) x[up][c] = nv[c];
    }
    if (BARRIER) __syncthreads();
  }
  float s = 0;
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) s += x[r][c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ROWS, int NT, bool BARRIER>
void run(const char* name) {
  float* out;
  const int blocks = 256;
  hipMalloc(&out, (size_t)blocks * NT * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ROWS, NT, BARRIER>), dim3(blocks), dim3(NT), 0, 0, out, 100, 0.25f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<ROWS, NT, BARRIER>), dim3(blocks), dim3(NT), 0, 0, out, iters, 0.25f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns_iter = ms * 1e6 / iters;
  printf("%-36s waves/SIMD=%d rows/wave=%2d  %.1f ns per iteration = %.0f cycles at 2.4 GHz (%.2f cycles per wave-instruction-slot)\n", name,
         NT / 256, ROWS, ns_iter, ns_iter * 2.4, ns_iter * 2.4 / (16.0 * ROWS * (NT / 256)));
  hipFree(out);
}

int main() {
  run<40, 256, true>("1 wave/SIMD, barrier");
  run<20, 512, true>("2 waves/SIMD, barrier");
  run<13, 768, true>("3 waves/SIMD (39 rows), barrier");
  run<10, 1024, true>("4 waves/SIMD, barrier");
  run<20, 512, false>("2 waves/SIMD, no barrier");
  run<13, 768, false>("3 waves/SIMD (39 rows), no barrier");
  run<10, 1024, false>("4 waves/SIMD, no barrier");
  return 0;
}
