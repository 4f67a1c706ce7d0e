/* abi_validation.c -- drives the argument-validation and dispatch paths of every C-ABI entry point (include/pdegym.h)
 * WITHOUT a GPU: null descriptors, out-of-range sizes, missing buffers, inconsistent options.  Every call must come back with a
 * negative code and a message in pdegym_last_error(); nothing may crash.  tests/test_sanitizer.py links it against a host-side
 * AddressSanitizer + UndefinedBehaviorSanitizer build of the library ("sanitizers on the host shim"), so a
 * stray read of a descriptor field, an overflow in a size computation or a misaligned access in these paths fails the CPU suite.
 * The last group hands over well-formed descriptors with fake device addresses: the host dispatch code then runs up to the
 * launch, which fails cleanly on a machine without a device (the pointers are never dereferenced on the host). */
#include <stdio.h>
#include <string.h>

#include "pdegym.h"

static int n_calls = 0, n_bad = 0;

static void expect_error(const char* what, int rc) {
  const char* msg = pdegym_last_error();
  ++n_calls;
  if (rc >= 0 || msg == NULL || msg[0] == '\0') {
    ++n_bad;
    printf("UNEXPECTED %s -> %d \"%s\"\n", what, rc, msg ? msg : "(null)");
  } else {
    printf("%-44s %5d  %s\n", what, rc, msg);
  }
}

#define FAKE(k) ((void*)(uintptr_t)(0x7f0000000000ull + 4096ull * (k)))

static pdegym_params1d good1d(void) {
  pdegym_params1d P;
  memset(&P, 0, sizeof P);
  P.n = 257; P.nt = 1001; P.substeps = 100; P.sensing = PDEGYM_SENSE_FULL; P.reward_kind = PDEGYM_REWARD_TUNED1D;
  P.reward_nt = 1000; P.dt = 1e-6f; P.dx = 1.0f / 256; P.F = 0.25f; P.max_control = 20; P.max_state = 1e10f; P.rdx = 256.0;
  P.dt64 = 1e-6; P.dx64 = 1.0 / 256; P.max_control64 = 20;
  return P;
}

static pdegym_bufs1d good_bufs1d(void) {
  pdegym_bufs1d b;
  memset(&b, 0, sizeof b);
  b.u = FAKE(1); b.beta = FAKE(2); b.beta_stride = 257; b.action = FAKE(3); b.time_index = FAKE(4); b.bsum = FAKE(5);
  b.ring = FAKE(6); b.obs = FAKE(7); b.reward = FAKE(8); b.norm_now = FAKE(9); b.norm_back = FAKE(10);
  b.terminated = FAKE(11); b.truncated = FAKE(12);
  return b;
}

int main(void) {
  if (pdegym_abi_version() != PDEGYM_ABI_VERSION) { printf("ABI mismatch\n"); return 1; }
  pdegym_params1d P = good1d(), Q;
  pdegym_bufs1d b = good_bufs1d(), c;
  pdegym_rollout1d ro;
  memset(&ro, 0, sizeof ro);

  /* ---- 1D step / reset / rollout ---- */
  expect_error("transport_step(NULL, NULL)", pdegym_transport_step(NULL, NULL, 4, NULL));
  expect_error("parabolic_step(P, NULL)", pdegym_parabolic_step(&P, NULL, 4, NULL));
  Q = P; Q.n = 2;
  expect_error("parabolic_step n=2", pdegym_parabolic_step(&Q, &b, 4, NULL));
  Q = P; Q.n = PDEGYM_MAX_N1D_WIDE + 1;
  expect_error("transport_step n=8193", pdegym_transport_step(&Q, &b, 4, NULL));
  Q = P; Q.nt = 1;
  expect_error("parabolic_step nt=1", pdegym_parabolic_step(&Q, &b, 4, NULL));
  c = b; c.obs = NULL;
  expect_error("parabolic_step obs=NULL", pdegym_parabolic_step(&P, &c, 4, NULL));
  c = b; c.reward = NULL;
  expect_error("parabolic_step reward=NULL", pdegym_parabolic_step(&P, &c, 4, NULL));
  c = b; c.state_in = c.obs;
  expect_error("parabolic_step state_in aliases obs", pdegym_parabolic_step(&P, &c, 4, NULL));
  c = b; c.state_in = FAKE(20); c.history = FAKE(21);
  expect_error("parabolic_step state_in + history", pdegym_parabolic_step(&P, &c, 4, NULL));
  Q = P; Q.sensing = PDEGYM_SENSE_LAST; c = b; c.state_in = FAKE(20);
  expect_error("parabolic_step state_in + scalar sensing", pdegym_parabolic_step(&Q, &c, 4, NULL));
  Q = P; Q.action_kind = 7;
  expect_error("transport_step action_kind=7", pdegym_transport_step(&Q, &b, 4, NULL));
  Q = P; Q.reward_kind = PDEGYM_REWARD_NORM_L2; Q.reward_horizon = 5;
  expect_error("transport_step reward_horizon=5", pdegym_transport_step(&Q, &b, 4, NULL));
  expect_error("reset1d_masked(NULL...)", pdegym_reset1d_masked(NULL, NULL, NULL, NULL, 4, NULL));
  Q = P; Q.n = 1;
  expect_error("reset1d_masked n=1", pdegym_reset1d_masked(&Q, &b, FAKE(30), NULL, 4, NULL));
  Q = P; Q.sensing = PDEGYM_SENSE_FIRST; c = b; c.u = NULL;
  expect_error("reset1d_masked u=NULL + scalar sensing", pdegym_reset1d_masked(&Q, &c, FAKE(30), NULL, 4, NULL));
  expect_error("transport_rollout(NULL...)", pdegym_transport_rollout(NULL, NULL, NULL, 4, NULL));
  ro.T = 5; ro.obs = FAKE(40); ro.actions = FAKE(41); ro.rewards = FAKE(42); ro.terminated = FAKE(43); ro.truncated = FAKE(44);
  Q = P; Q.n = PDEGYM_MAX_N1D + 1;
  expect_error("parabolic_rollout n=2049", pdegym_parabolic_rollout(&Q, &b, &ro, 4, NULL));
  Q = P; Q.beta_f64 = 1;
  expect_error("parabolic_rollout beta_f64", pdegym_parabolic_rollout(&Q, &b, &ro, 4, NULL));
  c = b; c.history = FAKE(21);
  expect_error("parabolic_rollout history", pdegym_parabolic_rollout(&P, &c, &ro, 4, NULL));
  { pdegym_rollout1d r2 = ro; r2.obs = NULL;
    expect_error("parabolic_rollout obs=NULL", pdegym_parabolic_rollout(&P, &b, &r2, 4, NULL)); }
  expect_error("selftest_quotient(NULL)", pdegym_selftest_quotient(NULL, 0.01f, 100.0, NULL, 8, NULL));
  expect_error("rownorm2_f32(NULL)", pdegym_rownorm2_f32(NULL, NULL, 100, 4, NULL));

  /* ---- policy network ---- */
  pdegym_mlp net;
  memset(&net, 0, sizeof net);
  expect_error("mlp_forward(NULL)", pdegym_mlp_forward(NULL, NULL, 0, NULL, 0, 4, NULL));
  expect_error("mlp_forward n_layers=0", pdegym_mlp_forward(&net, FAKE(50), 257, FAKE(51), 1, 4, NULL));
  net.n_layers = PDEGYM_MLP_MAX_LAYERS + 1;
  expect_error("mlp_forward n_layers=max+1", pdegym_mlp_forward(&net, FAKE(50), 257, FAKE(51), 1, 4, NULL));
  net.n_layers = 2;
  expect_error("mlp_forward w=NULL", pdegym_mlp_forward(&net, FAKE(50), 257, FAKE(51), 1, 4, NULL));
  net.layer[0].w = FAKE(52); net.layer[0].b = FAKE(53); net.layer[0].in_dim = 257; net.layer[0].out_dim = 64; net.layer[0].act = PDEGYM_MLP_TANH;
  net.layer[1].w = FAKE(54); net.layer[1].b = FAKE(55); net.layer[1].in_dim = 63; net.layer[1].out_dim = 1;
  expect_error("mlp_forward layer mismatch", pdegym_mlp_forward(&net, FAKE(50), 257, FAKE(51), 1, 4, NULL));
  net.layer[1].in_dim = 64; net.layer[1].act = 9;
  expect_error("mlp_forward bad activation", pdegym_mlp_forward(&net, FAKE(50), 257, FAKE(51), 1, 4, NULL));
  net.layer[1].act = PDEGYM_MLP_IDENTITY;
  expect_error("mlp_forward B<0", pdegym_mlp_forward(&net, FAKE(50), 257, FAKE(51), 1, -1, NULL));
  expect_error("mlp_forward short row stride", pdegym_mlp_forward(&net, FAKE(50), 100, FAKE(51), 1, 4, NULL));
  net.clamp = 1; net.lo = 1.0f; net.hi = -1.0f;
  expect_error("mlp_forward lo>hi", pdegym_mlp_forward(&net, FAKE(50), 257, FAKE(51), 1, 4, NULL));
  net.layer[0].out_dim = PDEGYM_MLP_MAX_WIDTH + 1;
  expect_error("mlp_forward too wide", pdegym_mlp_forward(&net, FAKE(50), 257, FAKE(51), 1, 4, NULL));

  /* ---- Navier-Stokes ---- */
  pdegym_params_ns2d N2;
  pdegym_bufs_ns2d nb;
  memset(&N2, 0, sizeof N2);
  memset(&nb, 0, sizeof nb);
  expect_error("ns2d_step_f32(NULL)", pdegym_ns2d_step_f32(NULL, NULL, 2, NULL));
  N2.nx = 2; N2.ny = 2;
  expect_error("ns2d_step_f64 2x2 grid", pdegym_ns2d_step_f64(&N2, &nb, 2, NULL));
  N2.nx = N2.ny = 21; N2.nt = 200; N2.iters = -1; N2.action_dim = 1; N2.dt = 1e-3; N2.dx = N2.dy = 0.05; N2.viscosity = 0.1; N2.density = 1;
  expect_error("ns2d_step_f64 iters<0", pdegym_ns2d_step_f64(&N2, &nb, 2, NULL));
  N2.iters = 50; N2.action_dim = 5;
  expect_error("ns2d_step_f64 action_dim=5", pdegym_ns2d_step_f64(&N2, &nb, 2, NULL));
  N2.action_dim = 1; N2.bc[2][1] = 3;
  expect_error("ns2d_step_f32 bad bc code", pdegym_ns2d_step_f32(&N2, &nb, 2, NULL));
  N2.bc[2][1] = 0;
  expect_error("ns2d_step_f32 bufs=NULL", pdegym_ns2d_step_f32(&N2, NULL, 2, NULL));
  expect_error("ns2d_step_f32 empty bufs", pdegym_ns2d_step_f32(&N2, &nb, 2, NULL));
  nb.p = FAKE(60); nb.scratch = FAKE(61); nb.action = FAKE(62); nb.U_ref = FAKE(63); nb.action_ref = FAKE(64); nb.obs = FAKE(65);
  nb.reward = FAKE(66); nb.time_index = FAKE(67); nb.terminated = FAKE(68); nb.nt_ref = 200;
  expect_error("ns2d_step_f64 no state", pdegym_ns2d_step_f64(&N2, &nb, 2, NULL));
  nb.u = FAKE(69);
  expect_error("ns2d_step_f64 u without v", pdegym_ns2d_step_f64(&N2, &nb, 2, NULL));
  nb.u = NULL; nb.state_in = nb.obs;
  expect_error("ns2d_step_f64 state_in aliases obs", pdegym_ns2d_step_f64(&N2, &nb, 2, NULL));
  nb.state_in = FAKE(70); nb.nt_ref = 0;
  expect_error("ns2d_step_f64 nt_ref=0", pdegym_ns2d_step_f64(&N2, &nb, 2, NULL));
  nb.nt_ref = 200; nb.p_out = nb.p;
  expect_error("ns2d_step_f64 p_out aliases p", pdegym_ns2d_step_f64(&N2, &nb, 2, NULL));
  nb.p_out = NULL; nb.reset_u0 = FAKE(71);
  expect_error("ns2d_step_f64 reset_u0 alone", pdegym_ns2d_step_f64(&N2, &nb, 2, NULL));
  nb.reset_u0 = NULL;
  expect_error("ns2d_solve_pressure_f64 NULL fields", pdegym_ns2d_solve_pressure_f64(&N2, NULL, NULL, NULL, NULL, NULL, 2, NULL));
  expect_error("ns2d_solve_pressure_f32 NULL params", pdegym_ns2d_solve_pressure_f32(NULL, FAKE(1), FAKE(2), FAKE(3), FAKE(4), FAKE(5), 2, NULL));
  expect_error("ns2d_reset_masked_f32 NULL", pdegym_ns2d_reset_masked_f32(&N2, NULL, NULL, NULL, NULL, NULL, 2, NULL));
  expect_error("ns2d_reset_masked_f64 NULL params", pdegym_ns2d_reset_masked_f64(NULL, &nb, FAKE(1), FAKE(2), FAKE(3), NULL, 2, NULL));
  {
    pdegym_rollout_ns2d nr;
    memset(&nr, 0, sizeof nr);
    nr.T = 4;
    expect_error("ns2d_rollout_f64 NULL descriptor", pdegym_ns2d_rollout_f64(&N2, &nb, NULL, 2, NULL));
    expect_error("ns2d_rollout_f32 empty rollout buffers", pdegym_ns2d_rollout_f32(&N2, &nb, &nr, 2, NULL));
    nr.obs = FAKE(80); nr.actions = FAKE(81); nr.rewards = FAKE(82); nr.terminated = FAKE(83);
    N2.nx = N2.ny = 128;
    expect_error("ns2d_rollout_f32 128x128 (no column kernel)", pdegym_ns2d_rollout_f32(&N2, &nb, &nr, 2, NULL));
    N2.nx = N2.ny = 21;
    expect_error("ns2d_rollout_f64 21x21 well-formed, no device", pdegym_ns2d_rollout_f64(&N2, &nb, &nr, 2, NULL));
  }
  expect_error("debug_set(-1)", pdegym_debug_set(-1, 0));

  /* ---- traffic ---- */
  pdegym_params_traffic TP;
  pdegym_bufs_traffic tb;
  pdegym_rollout_traffic tr;
  memset(&TP, 0, sizeof TP);
  memset(&tb, 0, sizeof tb);
  memset(&tr, 0, sizeof tr);
  expect_error("traffic_step(NULL)", pdegym_traffic_step(NULL, NULL, 4, NULL));
  TP.M = 3;
  expect_error("traffic_step M=3", pdegym_traffic_step(&TP, &tb, 4, NULL));
  TP.M = PDEGYM_TRAFFIC_MAX_M + 1;
  expect_error("traffic_step M=1025", pdegym_traffic_step(&TP, &tb, 4, NULL));
  TP.M = 51; TP.control_freq = 0;
  expect_error("traffic_step control_freq=0", pdegym_traffic_step(&TP, &tb, 4, NULL));
  TP.control_freq = 2; TP.sim = 9;
  expect_error("traffic_step sim=9", pdegym_traffic_step(&TP, &tb, 4, NULL));
  TP.sim = 0;
  expect_error("traffic_step empty bufs", pdegym_traffic_step(&TP, &tb, 4, NULL));
  expect_error("traffic_reset_masked empty bufs", pdegym_traffic_reset_masked(&TP, &tb, NULL, NULL, 4, NULL));
  expect_error("traffic_rollout ro=NULL", pdegym_traffic_rollout(&TP, &tb, NULL, 4, NULL));
  expect_error("traffic_rollout empty", pdegym_traffic_rollout(&TP, &tb, &tr, 4, NULL));

  /* ---- brain tumour ---- */
  pdegym_params_tumor UP;
  pdegym_bufs_tumor ub;
  memset(&UP, 0, sizeof UP);
  memset(&ub, 0, sizeof ub);
  expect_error("tumor_step(NULL)", pdegym_tumor_step(NULL, NULL, 4, NULL));
  expect_error("tumor_step empty", pdegym_tumor_step(&UP, &ub, 4, NULL));
  expect_error("tumor_advance(NULL)", pdegym_tumor_advance(NULL, NULL, 0, 10, 4, NULL));
  expect_error("tumor_advance empty", pdegym_tumor_advance(&UP, &ub, 0, 10, 4, NULL));
  expect_error("tumor_reset_masked(NULL)", pdegym_tumor_reset_masked(NULL, NULL, NULL, 0, NULL, 4, NULL));

  /* ---- well-formed descriptors, no device: the host dispatch runs up to the launch and reports the failure ---- */
  expect_error("parabolic_step well-formed, no device", pdegym_parabolic_step(&P, &b, 4096, NULL));
  Q = P; Q.n = 512; Q.beta_f64 = 1; Q.action_kind = PDEGYM_ACTION_WEAK;
  expect_error("transport_step mixed precision, no device", pdegym_transport_step(&Q, &b, 64, NULL));
  Q = P; Q.n = 4000;
  expect_error("transport_step wide rows, no device", pdegym_transport_step(&Q, &b, 64, NULL));
  c = b; c.u = NULL; c.state_in = FAKE(20);
  expect_error("parabolic_rollout well-formed, no device", pdegym_parabolic_rollout(&P, &c, &ro, 4096, NULL));
  nb.state_in = FAKE(70);
  expect_error("ns2d_step_f64 21x21 well-formed, no device", pdegym_ns2d_step_f64(&N2, &nb, 3072, NULL));
  N2.nx = N2.ny = 128;
  expect_error("ns2d_step_f32 128x128 well-formed, no device", pdegym_ns2d_step_f32(&N2, &nb, 512, NULL));
  N2.nx = N2.ny = 256;
  expect_error("ns2d_step_f32 256x256 well-formed, no device", pdegym_ns2d_step_f32(&N2, &nb, 512, NULL));
  expect_error("ns2d_step_f64 256x256 well-formed, no device", pdegym_ns2d_step_f64(&N2, &nb, 512, NULL));
  N2.nx = N2.ny = 100;
  expect_error("ns2d_step_f64 100x100 well-formed, no device", pdegym_ns2d_step_f64(&N2, &nb, 8, NULL));
  net.clamp = 0; net.layer[0].out_dim = 64;
  expect_error("mlp_forward well-formed, no device", pdegym_mlp_forward(&net, FAKE(50), 257, FAKE(51), 1, 4096, NULL));

  printf("%s %d calls, %d unexpected\n", n_bad ? "VALIDATION-FAILED" : "VALIDATION-OK", n_calls, n_bad);
  return n_bad ? 1 : 0;
}
