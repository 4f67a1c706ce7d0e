/* c_abi_example.c -- the C ABI of include/pdegym.h driven from plain C: no Python, no PyTorch.
 *
 * Steps a batch of ReactionDiffusionPDE1D instances (the reference's parabolic.py:126-189 loop) on the GPU with
 * buffers from hipMalloc and prints the rewards.  Build and run on an MI355X:
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/c_abi_example.c -Iinclude -I/opt/rocm/include \
 *       -Lpdecontrolgym_amd/lib -lpdegym_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/pdecontrolgym_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -o /tmp/c_abi_example && /tmp/c_abi_example
 *
 * Exit code 0 and a line "ok ..." mean: every call returned 0, the time index advanced by the sub-step count, the
 * boundary node carries the commanded value, and a second identical batch produced identical bits (determinism).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pdegym.h"

#define CHECK_HIP(x)                                                             \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                                  \
    }                                                                            \
  } while (0)
#define CHECK_PDE(x)                                                             \
  do {                                                                           \
    int rc_ = (x);                                                               \
    if (rc_ != 0) {                                                              \
      fprintf(stderr, "pdegym error %d: %s\n", rc_, pdegym_last_error());        \
      return 3;                                                                  \
    }                                                                            \
  } while (0)

static void* dalloc(size_t bytes) {
  void* p = NULL;
  if (hipMalloc(&p, bytes) != hipSuccess) return NULL;
  hipMemset(p, 0, bytes);
  return p;
}

int main(void) {
  enum { B = 6, NX = 64, N = NX + 1, S = 20, STEPS = 5 };
  if (pdegym_abi_version() != PDEGYM_ABI_VERSION) {
    fprintf(stderr, "ABI mismatch: library %d, header %d\n", pdegym_abi_version(), PDEGYM_ABI_VERSION);
    return 1;
  }
  const double dx = 1.0 / NX, dt = 0.25 * dx * dx;
  pdegym_params1d P;
  memset(&P, 0, sizeof P);
  P.n = N;
  P.nt = STEPS * S + 1;
  P.substeps = S;
  P.control_type = PDEGYM_CONTROL_DIRICHLET;
  P.normalize = 0;
  P.sensing = PDEGYM_SENSE_FULL;
  P.limit_state = 1;
  P.reward_kind = PDEGYM_REWARD_TUNED1D;
  P.reward_nt = P.nt - 1;
  P.dt = (float)dt;
  P.dx = (float)dx;
  P.F = (float)(dt / (dx * dx));
  P.max_control = 20.0f;
  P.max_state = 1e10f;
  P.truncate_penalty = -1e3f;
  P.terminate_reward = 3e2f;
  P.rdx = 1.0 / (double)(float)dx;

  float h_init[B * N], h_beta[N], h_act[B];
  for (int j = 0; j < N; ++j) h_beta[j] = 8.0f * cosf(2.0f * acosf((float)j / NX));
  for (int b = 0; b < B; ++b) {
    h_act[b] = 0.25f * (float)(b - 2);
    for (int j = 0; j < N; ++j) h_init[b * N + j] = 1.0f + 0.5f * (float)b;
  }

  float h_rew[2][STEPS][B], h_row[2][B * N];
  for (int pass = 0; pass < 2; ++pass) {
    pdegym_bufs1d Q;
    memset(&Q, 0, sizeof Q);
    float* d_init = (float*)dalloc(sizeof h_init);
    Q.u = (float*)dalloc(sizeof h_init);
    Q.beta = (const float*)dalloc(sizeof h_beta);
    Q.beta_stride = 0; /* one beta row shared by the batch */
    Q.action = (const float*)dalloc(sizeof h_act);
    Q.time_index = (int32_t*)dalloc(B * sizeof(int32_t));
    Q.bsum = (double*)dalloc(B * sizeof(double));
    Q.ring = (float*)dalloc((size_t)B * PDEGYM_RING * sizeof(float));
    Q.obs = (float*)dalloc(sizeof h_init);
    Q.reward = (float*)dalloc(B * sizeof(float));
    Q.norm_now = (float*)dalloc(B * sizeof(float));
    Q.norm_back = (float*)dalloc(B * sizeof(float));
    Q.terminated = (uint8_t*)dalloc(B);
    Q.truncated = (uint8_t*)dalloc(B);
    if (!d_init || !Q.u || !Q.beta || !Q.action || !Q.time_index || !Q.bsum || !Q.ring || !Q.obs || !Q.reward ||
        !Q.norm_now || !Q.norm_back || !Q.terminated || !Q.truncated)
      return 2;
    CHECK_HIP(hipMemcpy(d_init, h_init, sizeof h_init, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy((void*)Q.beta, h_beta, sizeof h_beta, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy((void*)Q.action, h_act, sizeof h_act, hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_PDE(pdegym_reset1d_masked(&P, &Q, d_init, NULL, B, st));
    for (int k = 0; k < STEPS; ++k) {
      CHECK_PDE(pdegym_parabolic_step(&P, &Q, B, st));
      CHECK_HIP(hipMemcpyAsync(h_rew[pass][k], Q.reward, B * sizeof(float), hipMemcpyDeviceToHost, st));
    }
    int32_t h_t[B];
    uint8_t h_term[B];
    CHECK_HIP(hipMemcpyAsync(h_row[pass], Q.u, sizeof h_init, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipMemcpyAsync(h_t, Q.time_index, sizeof h_t, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipMemcpyAsync(h_term, Q.terminated, sizeof h_term, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));
    for (int b = 0; b < B; ++b) {
      if (h_t[b] != STEPS * S || !h_term[b]) {
        fprintf(stderr, "instance %d: time_index %d terminated %d\n", b, h_t[b], h_term[b]);
        return 4;
      }
      if (h_row[pass][b * N + N - 1] != h_act[b] || h_row[pass][b * N] != 0.0f) {
        fprintf(stderr, "instance %d: boundary nodes %g / %g\n", b, h_row[pass][b * N], h_row[pass][b * N + N - 1]);
        return 5;
      }
    }
    /* a bad argument must come back as an error code with a message, not a crash */
    pdegym_params1d bad = P;
    bad.n = 1 << 20;
    if (pdegym_parabolic_step(&bad, &Q, B, st) == 0 || strlen(pdegym_last_error()) == 0) return 6;
    CHECK_HIP(hipStreamDestroy(st));
  }
  if (memcmp(h_rew[0], h_rew[1], sizeof h_rew[0]) != 0 || memcmp(h_row[0], h_row[1], sizeof h_row[0]) != 0) {
    fprintf(stderr, "two identical batches differ\n");
    return 7;
  }
  /* The same episode as ONE launch: pdegym_parabolic_rollout reads the rows from observation slot t, the commands from
   * actions row t, and writes slot t + 1 and row t of the reward / flag arrays -- bit-identical to the step calls above. */
  {
    pdegym_bufs1d Q;
    memset(&Q, 0, sizeof Q);
    pdegym_rollout1d R;
    memset(&R, 0, sizeof R);
    float h_acts[STEPS][B], h_rrew[STEPS][B], h_last[B * N];
    for (int k = 0; k < STEPS; ++k) memcpy(h_acts[k], h_act, sizeof h_act);
    float* d_init = (float*)dalloc(sizeof h_init);
    R.T = STEPS;
    R.obs = (float*)dalloc((size_t)(STEPS + 1) * sizeof h_init);
    R.actions = (float*)dalloc(sizeof h_acts);
    R.rewards = (float*)dalloc(sizeof h_rrew);
    R.terminated = (uint8_t*)dalloc((size_t)STEPS * B);
    R.truncated = (uint8_t*)dalloc((size_t)STEPS * B);
    Q.u = NULL; /* full-state sensing: the rows live in the observation slots */
    Q.obs = R.obs;
    Q.beta = (const float*)dalloc(sizeof h_beta);
    Q.action = R.actions;
    Q.time_index = (int32_t*)dalloc(B * sizeof(int32_t));
    Q.bsum = (double*)dalloc(B * sizeof(double));
    Q.ring = (float*)dalloc((size_t)B * PDEGYM_RING * sizeof(float));
    Q.reward = R.rewards;
    Q.norm_now = (float*)dalloc(B * sizeof(float));
    Q.norm_back = (float*)dalloc(B * sizeof(float));
    Q.terminated = R.terminated;
    Q.truncated = R.truncated;
    if (!d_init || !R.obs || !R.actions || !R.rewards || !R.terminated || !R.truncated || !Q.beta || !Q.time_index || !Q.bsum ||
        !Q.ring || !Q.norm_now || !Q.norm_back)
      return 2;
    CHECK_HIP(hipMemcpy(d_init, h_init, sizeof h_init, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy((void*)Q.beta, h_beta, sizeof h_beta, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(R.actions, h_acts, sizeof h_acts, hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_PDE(pdegym_reset1d_masked(&P, &Q, d_init, NULL, B, st)); /* writes observation slot 0 */
    CHECK_PDE(pdegym_parabolic_rollout(&P, &Q, &R, B, st));
    CHECK_HIP(hipMemcpyAsync(h_rrew, R.rewards, sizeof h_rrew, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipMemcpyAsync(h_last, R.obs + (size_t)STEPS * B * N, sizeof h_last, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));
    if (memcmp(h_rrew, h_rew[0], sizeof h_rrew) != 0 || memcmp(h_last, h_row[0], sizeof h_last) != 0) {
      fprintf(stderr, "the one-launch rollout differs from the step calls\n");
      return 8;
    }
    CHECK_HIP(hipStreamDestroy(st));
  }
  printf("ok abi=%d rewards(step %d):", pdegym_abi_version(), STEPS);
  for (int b = 0; b < B; ++b) printf(" %.6g", h_rew[0][STEPS - 1][b]);
  printf("\n");
  return 0;
}
