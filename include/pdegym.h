/*
 * pdegym.h -- C ABI of the MI355X (gfx950) batched PDE-environment stepper.
 *
 * Drop-in boundary for the hot path of lukebhan/PDEControlGym (reference checkout paths below are
 * relative to that repository):
 *
 *   pdegym_transport_step   replaces TransportPDE1D.step          environments1d/hyperbolic.py:126-169
 *                           + terminate()/truncate()              environments1d/hyperbolic.py:171-194
 *                           + TunedReward1D.reward                rewards/tuned_reward_1d.py:25-40
 *   pdegym_parabolic_step   replaces ReactionDiffusionPDE1D.step  environments1d/parabolic.py:126-164
 *                           + terminate()/truncate()              environments1d/parabolic.py:166-189
 *   pdegym_reset1d_masked   replaces the state part of reset()    hyperbolic.py:214-219, parabolic.py:208-213
 *   pdegym_ns2d_step_f32/_f64  replaces NavierStokes2D.step       environments2d/navier_stokes2D.py:118-157
 *                           (apply_boundary :68-91, solve_pressure :94-116, NSReward ns_reward.py:28)
 *   pdegym_ns2d_solve_pressure_f32/_f64  replaces NavierStokes2D.solve_pressure  navier_stokes2D.py:94-116
 *                           (public API: examples/NavierStokes/NS2Doptimization.py:97)
 *   pdegym_ns2d_reset_masked_f32/_f64    replaces the state part of NavierStokes2D.reset  navier_stokes2D.py:186-192
 *   pdegym_rownorm2_f32     replaces np.linalg.norm(row, 2)       hyperbolic.py:190, parabolic.py:185
 *
 * One call advances EVERY instance of a batch by one env-step (S PDE sub-steps for the 1D envs, one
 * Chorin projection step with K Jacobi sweeps for NS2D).  Instances are independent.
 *
 * Conventions
 *   - every pointer is a DEVICE-ACCESSIBLE pointer owned by the caller (PyTorch-ROCm tensors in this repo): HBM, or pinned
 *     host memory mapped into the device's address space (hipHostMalloc).  The plant state belongs in HBM; per-call INPUTS
 *     (action, control, kill, t_benchmark) and pure OUTPUTS (obs, reward, norm_now, norm_back, terminated, truncated, done) may
 *     live in pinned host memory -- the kernels read / write them in place, which is how the batch-of-one faces of the Python
 *     layer take a command and hand back a result with one launch + one stream synchronisation and no copies
 *     (PDEBatch1D.enable_host_io; the reference's own callers step ONE environment: examples/transportPDE/transport1Dppo.py:59-90);
 *     the library allocates nothing and keeps no state besides a thread-local error string;
 *   - calls only ENQUEUE work on `stream` (a hipStream_t passed as void*); they never synchronise;
 *   - return value 0 = success, negative = error (message via pdegym_last_error()); nothing throws;
 *   - arrays are C-contiguous; the grid axis is the fastest axis.
 */
#ifndef PDEGYM_H
#define PDEGYM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDEGYM_ABI_VERSION 15
#define PDEGYM_RING 128          /* slots of the per-instance row-norm ring (look-back is 100 rows) */
#define PDEGYM_LOOKBACK 100      /* tuned_reward_1d.py:25,40: int(1/0.01) rows */
#define PDEGYM_MAX_N1D 2048      /* nodes per 1D row kept in registers by the wave-per-instance kernels */
#define PDEGYM_MAX_N1D_WIDE 8192 /* longer rows (up to this) ping-pong through wave-private LDS */

/* control_type (hyperbolic.py:66-124): the reference's (mis)spelling "Dirchilet" is kept in the Python layer */
enum { PDEGYM_CONTROL_DIRICHLET = 0, PDEGYM_CONTROL_NEUMANN = 1 };
enum { PDEGYM_FLUX_LINEAR = 0, PDEGYM_FLUX_BURGERS = 1 };   /* transport kernels only */
/* What the caller of the reference's step() passed as `control` -- it decides, under NumPy's promotion rules, in which
 * precision control_update()/normalize() are evaluated before the value is stored into the float32 row
 * (hyperbolic.py:143-145, parabolic.py:148-150, base_env_1d.py:36-39): */
enum {
  PDEGYM_ACTION_F32 = 0,   /* float32 array / np.float32 (what SB3 passes): everything in float32; action = float[B]     */
  PDEGYM_ACTION_F64 = 1,   /* np.float64 scalar or array (and a Python float under NumPy 1.x value-based casting):
                              control*dx + neighbour and (a+1)*max-max in double, rounded once on store; action = double[B] */
  PDEGYM_ACTION_WEAK = 2   /* Python float / int under NumPy >= 2 (NEP 50 weak scalar): Dirichlet as F64; Neumann forms
                              control*dx in double, then joins the float32 neighbour in float32; action = double[B]        */
};
/* sensing_update variants (hyperbolic.py:72-116, parabolic.py:72-116) */
enum {
  PDEGYM_SENSE_FULL = 0,        /* obs = row                              obs_dim = n */
  PDEGYM_SENSE_LAST = 1,        /* obs = row[-1]                          (Neumann control, collocated) */
  PDEGYM_SENSE_LAST_DERIV = 2,  /* obs = (row[-1]-row[-2])/dx             (Dirichlet control, collocated) */
  PDEGYM_SENSE_FIRST_DERIV = 3, /* obs = (row[1]-row[0])/dx               (opposite, Neumann sensing) */
  PDEGYM_SENSE_FIRST = 4        /* obs = row[0]                           (opposite, Dirichlet sensing) */
};
/* reward evaluated inside the step kernel */
enum {
  PDEGYM_REWARD_NONE = 0,       /* reward[] untouched; norms/flags still produced (host-side custom rewards) */
  PDEGYM_REWARD_TUNED1D = 1,    /* rewards/tuned_reward_1d.py:25-40 */
  PDEGYM_REWARD_NORM_L1 = 2,    /* rewards/norm_reward.py "temporal" intent (parity unpinned: reference class raises) */
  PDEGYM_REWARD_NORM_L2 = 3,
  PDEGYM_REWARD_NORM_LINF = 4
};
/* NormReward horizon (rewards/norm_reward.py:52-73) */
enum {
  PDEGYM_HORIZON_TEMPORAL = 0,     /* -||u[t]||                                                                */
  PDEGYM_HORIZON_DIFFERENTIAL = 1, /* +||u[t] - u[t-1]|| over fine-time rows (t > 0), evaluated by pdegym_step1d_* */
  PDEGYM_HORIZON_T = 2             /* "t-horizon": -(||u[t]|| + ... + ||u[t-k+1]||) / k over fine-time rows, k =
                                      min(reward_t_horizon, t + 1), one float32 chain starting at row t; pdegym_*_step only.
                                      The ring then holds the reward's own norms (norm_back is meaningless)        */
};

/* Scalars of one 1D environment family (same for every instance of the batch). */
typedef struct pdegym_params1d {
  int32_t n;                /* nodes per row: nx (transport) or nx+1 (parabolic ghost point, parabolic.py:124) */
  int32_t nt;               /* rows per episode, int(round(T/dt)+1)                       base_env_1d.py:23 */
  int32_t substeps;         /* S = int(round(control_sample_rate/dt))                     hyperbolic.py:137 */
  int32_t control_type;     /* PDEGYM_CONTROL_*                                                              */
  int32_t normalize;        /* (a+1)*max-max if non-zero                                  base_env_1d.py:36-39 */
  int32_t sensing;          /* PDEGYM_SENSE_*                                                                */
  int32_t limit_state;      /* limit_pde_state_size                                       hyperbolic.py:188-191 */
  int32_t reward_kind;      /* PDEGYM_REWARD_*                                                               */
  int32_t reward_nt;        /* TunedReward1D(nt, ...) ctor argument                       tuned_reward_1d.py:17-23 */
  float dt;                 /* float32 cast of the Python doubles: NumPy casts them where they meet a float32 array */
  float dx;
  float F;                  /* (float)(dt/dx**2)                                          parabolic.py:138 */
  float max_control;        /* max_control_value */
  float max_state;          /* max_state_value */
  float truncate_penalty;   /* TunedReward1D / NormReward arguments */
  float terminate_reward;
  double rdx;               /* 1.0/(double)(float)dx: the transport quotient (u[j+1]-u[j])/dx is formed as
                               (float)((double)d * rdx), which equals the IEEE float32 division bit for bit */
  int32_t flux;             /* PDEGYM_FLUX_LINEAR = the reference's transport term; PDEGYM_FLUX_BURGERS = extension,
                               NOT in the reference (parity unpinned): n[j] = p[j] + dt*(p[j]*((p[j+1]-p[j])/dx) + (p[0]*beta)[j]) */
  int32_t beta_f64;         /* non-zero: bufs.beta is double[] -- the reference's arithmetic when reset_recirculation_func returns
                               float64 (e.g. np.ones(nx), docs/source/guide/quickstart.rst:27-28): u[0]*beta, dt*beta*u and the
                               sums they enter are evaluated in double and rounded once when the row is stored
                               (hyperbolic.py:146-155, parabolic.py:143-144)                                              */
  int32_t action_kind;      /* PDEGYM_ACTION_*                                                                            */
  int32_t reward_horizon;   /* PDEGYM_HORIZON_* (NormReward kinds only; pdegym_step1d, not the rollout entry points)      */
  double dt64, dx64;        /* the Python doubles themselves (used where they meet a float64 operand)                     */
  double max_control64;
  int32_t reward_t_horizon; /* PDEGYM_HORIZON_T: t_horizon_length of NormReward (norm_reward.py:19), 1 .. PDEGYM_RING             */
  int32_t reserved1_;
} pdegym_params1d;

/* Per-instance device buffers of a 1D batch (B instances). */
typedef struct pdegym_bufs1d {
  float* u;                 /* [B, n]   live row (in/out)                                                     */
  const void* beta;         /* [B, n] or [n]   plant parameter beta(x) / lambda(x): float, or double when beta_f64    */
  int64_t beta_stride;      /* elements between instances; 0 = one shared row                                 */
  const void* action;       /* [B]      control input of this env-step: float, or double when action_kind != F32      */
  int32_t* time_index;      /* [B]      in/out                                                                */
  double* bsum;             /* [B]      running sum |u[tau,-1]| over written rows (in/out)  tuned_reward_1d.py:37 */
  float* ring;              /* [B, PDEGYM_RING]  row norms that a later look-back will read (in/out)          */
  float* obs;               /* [B, obs_dim] out                                                               */
  float* reward;            /* [B] out                                                                        */
  float* norm_now;          /* [B] out  ||u_t||_2                                                             */
  float* norm_back;         /* [B] out  ||u_{t-100}||_2 (0 for an unwritten row)                              */
  uint8_t* terminated;      /* [B] out                                                                        */
  uint8_t* truncated;       /* [B] out                                                                        */
  float* history;           /* optional [B, nt, n] full trajectory (NULL = keep only the live row)            */
  const float* reset_init;  /* optional [B, n] pool of next initial conditions: when non-NULL an instance whose
                               step ends terminated|truncated is restarted INSIDE the same launch (state := pool row,
                               time_index := 0) and obs[b] is the first observation of the new episode            */
  float* final_obs;         /* optional [B, obs_dim]: last observation of the finished episode (written only for
                               instances that were auto-reset in this call; SB3's "terminal_observation")          */
  /* The reference redraws BOTH the initial condition and beta at every reset (hyperbolic.py:207-209).  With a pool of
   * reset_pool_rows >= B rows, the k-th restart of instance b takes pool row (b + k*B) mod reset_pool_rows, so consecutive
   * episodes of one instance start from different rows without any host work; reset_pool_rows == 0 means B (row b always). */
  const void* reset_beta;   /* optional [reset_pool_rows, n] (same element type as beta): on an auto-reset beta[b] := the pool row
                               (needs beta_stride != 0, i.e. per-instance beta, and a writable beta buffer)        */
  int32_t* reset_count;     /* optional [B] in/out: restarts of each instance so far (k above); NULL = always row b  */
  int32_t reset_pool_rows;  /* rows of reset_init / reset_beta (0 = B)                                            */
  int32_t reserved_;
  /* With full-state sensing the observation IS the row (hyperbolic.py:72-75, parabolic.py:74-77).  state_in, when non-NULL
   * (needs sensing == PDEGYM_SENSE_FULL and history == NULL): the rows are READ from state_in [B, n] -- the obs buffer of the
   * previous call -- and the new rows are written to obs only; u is not touched and may be NULL.  One row store per instance
   * and env-step instead of two; obs must not alias state_in (double-buffer the observations).  pdegym_reset1d_masked with
   * u == NULL likewise writes the reset rows to obs only. */
  const float* state_in;
} pdegym_bufs1d;

int pdegym_abi_version(void);
const char* pdegym_last_error(void);

int pdegym_transport_step(const pdegym_params1d* prm, const pdegym_bufs1d* buf, int32_t B, void* stream);
int pdegym_parabolic_step(const pdegym_params1d* prm, const pdegym_bufs1d* buf, int32_t B, void* stream);

/* T env-steps in ONE launch (the on-device rollout of SURVEY.md section 8f rank 1 without a kernel boundary per step):
 * replaces the loop of env.step calls that SB3's model.learn() drives (examples/transportPDE/transport1Dppo.py:88-90,
 * examples/reactionDiffusionPDE/reactionDiffusion1Dppo.py), i.e. T x (hyperbolic.py:126-194 / parabolic.py:126-189).
 * Step t reads the rows from obs slot t, the commands from actions row t, and writes obs slot t + 1, rewards / terminated /
 * truncated row t; everything else (beta, time_index, bsum, ring, norm_now, norm_back, the auto-reset pools, final_obs,
 * reset_count) comes from the pdegym_bufs1d of the call and behaves as in T consecutive pdegym_*_step calls with state_in --
 * the results are bit-identical to those calls.  bufs->state_in / obs / action / reward / terminated / truncated / history are
 * ignored.  Every control / sensing combination of the reference's table (hyperbolic.py:66-124, parabolic.py:66-122):
 *   - full-state sensing: the observation slots [T + 1, B, n] hold the state (slot t in, slot t + 1 out); bufs->u is ignored.
 *     Dirichlet actuation keeps the row in registers across the T env-steps; Neumann actuation passes it through the slots;
 *   - scalar sensing (PDEGYM_SENSE_LAST / _LAST_DERIV / _FIRST_DERIV / _FIRST): the state lives in bufs->u [B, n] (required,
 *     advanced in place) and the observation slots are [T + 1, B, 1]: slot t + 1 receives the value sensed after step t
 *     (slot 0 is only read by a policy).
 * Needs float32 beta and actions, the temporal reward horizon, n <= 2048. */
typedef struct pdegym_rollout1d {
  int32_t T;                /* env-steps per call                                                              */
  int32_t reserved_;
  float* obs;               /* [T + 1, B, n]  slot 0 = the input rows; slots 1 .. T are written                 */
  float* actions;           /* [T, B]  read; with a policy: WRITTEN (the command the policy issued, after noise and clamp) */
  float* rewards;           /* [T, B]  (may be NULL with PDEGYM_REWARD_NONE)                                   */
  uint8_t* terminated;      /* [T, B]                                                                          */
  uint8_t* truncated;       /* [T, B]                                                                          */
  /* Optional policy (HOST pointer, read during the call; see pdegym_mlp below): evaluated INSIDE the launch on observation
   * slot t -- the caller of env.step of an SB3 rollout (transport1Dppo.py:88-90) without a launch of its own.  First in_dim ==
   * the observation row (n, or 1 with scalar sensing; n <= 513), one output.  Layers of up to 64 units: the whole network must
   * fit into 160 KB of LDS next to 16 observation rows; same arithmetic as pdegym_mlp_forward except for the order of the
   * additions inside a group of 16 inputs.  With a layer of 65 .. 256 units (SB3's 256-256 actors, reactionDiffusion1Dsac.py:95)
   * the 16 waves of a workgroup evaluate the network together with pdegym_mlp_forward's own MFMA reduction, weights streamed from
   * L2: the commands equal pdegym_mlp_forward's BIT FOR BIT.  noise, when given, is [T, B] (row t, instance b at
   * noise[(t * B + b) * noise_stride]), added before the clamp. */
  const struct pdegym_mlp_s* policy;
  /* The sensing-noise hook (hyperbolic.py:160-164: the agent sees sensing_noise_func(observation)) for the policy inside the
   * launch, as noise the caller drew ahead: the policy of step t reads obs[t] + obs_noise[t]; obs itself stays clean (with
   * full-state sensing it is the plant state).  Both optional, both only with a policy:                                    */
  const float* obs_noise;   /* [T, B, obs_dim] added to observation slot t on its way into the policy                  */
  float* obs_seen;          /* [T, B, obs_dim] receives what the policy read (obs[t] + obs_noise[t]): the rollout's
                               training input                                                                          */
} pdegym_rollout1d;

int pdegym_transport_rollout(const pdegym_params1d* prm, const pdegym_bufs1d* buf, const pdegym_rollout1d* ro, int32_t B,
                             void* stream);
int pdegym_parabolic_rollout(const pdegym_params1d* prm, const pdegym_bufs1d* buf, const pdegym_rollout1d* ro, int32_t B,
                             void* stream);

/* Where mask[b] != 0 (or mask == NULL): u[b] = init[b], beta untouched, time_index = 0, bsum = |init[b,-1]|,
 * ring[b, 0] = ||init[b]|| , obs[b] = sensing(init[b]).  (hyperbolic.py:214-227) */
int pdegym_reset1d_masked(const pdegym_params1d* prm, const pdegym_bufs1d* buf, const float* init,
                          const uint8_t* mask, int32_t B, void* stream);

/* Self-test of the transport quotient: counts i where (float)((double)a[i] * rdx) != a[i] / dx (IEEE division),
 * bitwise, NaNs compared as equal.  *mismatches (device uint32) must be zeroed by the caller. */
int pdegym_selftest_quotient(const float* a, float dx, double rdx, uint32_t* mismatches, int32_t n, void* stream);

/* out[b] = ||rows[b, 0:n]||_2 */
int pdegym_rownorm2_f32(const float* rows, float* out, int32_t n, int32_t B, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Navier-Stokes 2D (collocated grid, Chorin projection, Jacobi pressure Poisson)
 * ------------------------------------------------------------------------------------------------ */
enum { PDEGYM_BC_NEUMANN = 0, PDEGYM_BC_DIRICHLET = 1, PDEGYM_BC_CONTROLLABLE = 2 };
/* edge order of apply_boundary (navier_stokes2D.py:76): lower (row 0), upper (row ny-1), left (col 0), right (col nx-1) */
enum { PDEGYM_EDGE_LOWER = 0, PDEGYM_EDGE_UPPER = 1, PDEGYM_EDGE_LEFT = 2, PDEGYM_EDGE_RIGHT = 3 };

typedef struct pdegym_params_ns2d {
  int32_t nx, ny;           /* int(round(X/dx+1)), int(round(Y/dy+1))                      base_env_2d.py:28-29 */
  int32_t nt;               /* int(round(T/dt))                                           base_env_2d.py:27 */
  int32_t iters;            /* maximum_pressure_iteration (Jacobi sweeps per step)        navier_stokes2D.py:104 */
  int32_t action_dim;       /* 1 = one scalar per instance; nx (== ny) = one value per edge node */
  int32_t bc[4][2];         /* [edge][component u,v] -> PDEGYM_BC_*                        navier_stokes2D.py:61-91 */
  double dt, dx, dy;        /* the float32 entry points cast them */
  double viscosity, density;
  double gamma;             /* NSReward(gamma)                                             ns_reward.py:15-28 */
} pdegym_params_ns2d;

/* T = float (f32 entry points) or double (f64 entry points). Fields are [B, ny, nx], row = y, col = x. */
typedef struct pdegym_bufs_ns2d {
  void* u;                  /* [B, ny, nx] in/out; may be NULL (together with v) when state_in is given        */
  void* v;                  /* [B, ny, nx] in/out                                                             */
  void* p;                  /* [B, ny, nx] in/out (warm start of the next step, navier_stokes2D.py:115)        */
  void* scratch;            /* [B, 4, ny, nx] work space (u*, v*, rhs, p')                                    */
  const void* action;       /* [B, action_dim]                                                                */
  int32_t* time_index;      /* [B] in/out                                                                     */
  const void* U_ref;        /* [nt_ref, ny, nx, 2] shared reference trajectory (indexed AFTER the increment)  */
  const void* action_ref;   /* [>= nt] shared reference actions                                               */
  int32_t nt_ref;           /* rows of U_ref / action_ref (index is clamped to nt_ref-1)                       */
  void* obs;                /* [B, ny, nx, 2] out (u,v interleaved; base_env_2d.py:50)                         */
  void* reward;             /* [B] out                                                                        */
  uint8_t* terminated;      /* [B] out; truncated is always False in the reference (navier_stokes2D.py:155)    */
  void* p_out;              /* optional [B, ny, nx]: where the solved pressure of this step goes (must not alias p).  NULL = in place
                               (p is overwritten).  With it the caller ping-pongs two pressure tensors: the 256x256 pipeline
                               then needs no copy of its last pass back into p                                             */
  const void* state_in;     /* optional [B, ny, nx, 2]: the observation written by the PREVIOUS call.  When non-NULL the
                               velocity state is read from it and written only to obs (the reference's obs is the state,
                               navier_stokes2D.py:147-154), which saves one write of u and v per step; must not alias obs */
  /* Fused VecEnv auto-reset: when reset_u0 is non-NULL, an instance whose step ends terminated
   * (time_index >= nt-1, navier_stokes2D.py:159-168) restarts inside the same call, without a host round trip: its observation
   * is first copied to final_obs (if given), then (u, v, p) := pool row, time_index := 0, obs := (u0, v0).  The k-th restart of
   * instance b takes pool row (b + k*B) mod reset_pool_rows (k from reset_count, see pdegym_bufs1d). */
  const void* reset_u0;     /* optional [reset_pool_rows, ny, nx]                                                            */
  const void* reset_v0;     /* [reset_pool_rows, ny, nx]  (required with reset_u0)                                            */
  const void* reset_p0;     /* [reset_pool_rows, ny, nx]  (required with reset_u0)                                            */
  void* final_obs;          /* optional [B, ny, nx, 2]: last observation of the finished episode ("terminal_observation")    */
  int32_t* reset_count;     /* optional [B] in/out                                                                           */
  int32_t reset_pool_rows;  /* rows of the pools (0 = B)                                                                     */
  int32_t reserved_;
} pdegym_bufs_ns2d;

int pdegym_ns2d_step_f32(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, int32_t B, void* stream);
int pdegym_ns2d_step_f64(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, int32_t B, void* stream);

/* T env-steps in ONE launch for small grids (the reference's shipped 21 x 21 example; grids of 8, 11, 16, 21, 26, 31 or 32 rows and at
 * most 64 columns): the loop of env.step calls of examples/NavierStokes/NS2Dppo.py:52-66 / the forward sweeps of the adjoint example
 * (NS2Doptimization.py:74-76) with the commands given ahead.  Step t reads the state from obs slot t and the command from actions row
 * t, writes obs slot t + 1 and rewards / terminated row t; p, time_index, U_ref / action_ref, the auto-reset pools, final_obs and
 * reset_count come from the pdegym_bufs_ns2d of the call and behave as in T consecutive pdegym_ns2d_step calls with state_in (fused
 * auto-reset included; the pressure stays in bufs->p).  Observation slots, pressure, flags, time index and restart counters are
 * bit-identical to those calls.  The REWARDS are bit-identical to step calls that take the same (column-per-lane) kernel -- the
 * default for float32 at any batch and for float64 batches of at least 400 instances per 1024 SIMDs of the device; a smaller float64
 * batch steps on the workgroup-per-instance kernel, which adds the reward's squared distances in another order: same fields, rewards
 * equal to ~1e-15 relative (so a float64 instance's reward BITS depend on whether its batch is below or above that size).
 * bufs->u / v / state_in / obs / action / reward / terminated / p_out are ignored. */
typedef struct pdegym_rollout_ns2d {
  int32_t T;                /* env-steps per call                                                              */
  int32_t reserved_;
  void* obs;                /* [T + 1, B, ny, nx, 2]  slot 0 = the input state; slots 1 .. T are written       */
  const void* actions;      /* [T, B, action_dim]                                                             */
  void* rewards;            /* [T, B]                                                                         */
  uint8_t* terminated;      /* [T, B]                                                                         */
} pdegym_rollout_ns2d;
int pdegym_ns2d_rollout_f32(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, const pdegym_rollout_ns2d* ro, int32_t B,
                            void* stream);
int pdegym_ns2d_rollout_f64(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, const pdegym_rollout_ns2d* ro, int32_t B,
                            void* stream);

/* p_out = K Jacobi sweeps from p_in with rhs = rho/dt*(d/dx u + d/dy v)  (navier_stokes2D.py:94-116).
 * u, v, p_in, p_out: [B, ny, nx]; scratch: [B, 2, ny, nx]. p_out may alias p_in. */
int pdegym_ns2d_solve_pressure_f32(const pdegym_params_ns2d* prm, const void* u, const void* v, const void* p_in,
                                   void* p_out, void* scratch, int32_t B, void* stream);
int pdegym_ns2d_solve_pressure_f64(const pdegym_params_ns2d* prm, const void* u, const void* v, const void* p_in,
                                   void* p_out, void* scratch, int32_t B, void* stream);

/* Where mask[b] != 0 (or mask == NULL): u,v,p[b] = u0,v0,p0[b]; time_index = 0; obs[b] = (u0,v0). */
int pdegym_ns2d_reset_masked_f32(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, const void* u0,
                                 const void* v0, const void* p0, const uint8_t* mask, int32_t B, void* stream);
int pdegym_ns2d_reset_masked_f64(const pdegym_params_ns2d* prm, const pdegym_bufs_ns2d* buf, const void* u0,
                                 const void* v0, const void* p0, const uint8_t* mask, int32_t B, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Traffic ARZ 1D: Aw-Rascle-Zhang freeway model in (r, y), float64,
 * two-step Lax-Wendroff with relaxation, flux boundary control.
 *   pdegym_traffic_step          replaces TrafficPDE1D.step + terminate/truncate   environments1d/traffic_arz_env.py:103-233
 *                                + TrafficARZReward.reward                          rewards/traffic_arz_reward.py:12-22
 *   pdegym_traffic_reset_masked  replaces the state part of TrafficPDE1D.reset     traffic_arz_env.py:245-260
 * ------------------------------------------------------------------------------------------------ */
enum { PDEGYM_TRAFFIC_INLET = 0, PDEGYM_TRAFFIC_OUTLET = 1, PDEGYM_TRAFFIC_BOTH = 2, PDEGYM_TRAFFIC_OUTLET_TRAIN = 3 };
#define PDEGYM_TRAFFIC_MAX_M 1024  /* nodes per freeway (the reference notebook uses 51; up to 64 stay in registers) */

typedef struct pdegym_params_traffic {
  int32_t M;               /* len(np.arange(0, X+dx, dx))                                   traffic_arz_env.py:77-79 */
  int32_t control_freq;    /* PDE sub-steps per step() call                                  :41-42 */
  int32_t sim;             /* PDEGYM_TRAFFIC_*  (simulation_type)                            :44-57 */
  int32_t limit;           /* limit_pde_state_size                                           :124 */
  double dt, dx, T;        /* T in seconds; time advances by dt per step() (not per sub-step) :146 */
  double vm, rm, tau;      /* v_max, ro_max, relaxation time */
} pdegym_params_traffic;

typedef struct pdegym_bufs_traffic {
  double* r;               /* [B, M] density (in/out)                                                       */
  double* y;               /* [B, M] relative flow y = r (v - Veq(r)) (in/out)                              */
  const double* action;    /* [B, action_stride] inlet / outlet flux command; column 1 is used by 'both' only */
  double* time;            /* [B] simulated seconds ("time_index" of the reference) in/out                  */
  double* rs;              /* [B] steady-state density of each instance (vs, qs follow the equilibrium law); rewritten
                              for an instance that the fused auto-reset restarts (reset_rs below)                     */
  const double* qs_clip;   /* [B] qs the action bounds [0.8 qs, 1.2 qs] were built from (:97-100)           */
  double* obs;             /* [B, 2M] out: (r, v), or ((r-rs)/rs, (v-vs)/vs) for outlet-train (:227-230)    */
  double* reward;          /* [B] out                                                                       */
  uint8_t* done;           /* [B] out: terminate() or reward > -0.00023 (:230)                              */
  uint8_t* truncated;      /* [B] out                                                                       */
  int32_t action_stride;   /* elements between the commands of consecutive instances: 2 (or 0 = 2), or 1 for the
                              single-command simulation types (the caller's [B] / [B,1] tensor is used as it is)      */
  int32_t reserved_;
  /* Fused VecEnv auto-reset (as for the 1D and NS engines): when reset_rs is non-NULL, an
   * instance whose step ends done | truncated restarts INSIDE the same launch the way TrafficPDE1D.reset does
   * (traffic_arz_env.py:245-260): its observation goes to final_obs (if given), then rs[b] := reset_rs[(b + k*B) mod
   * reset_pool_rows] (k from reset_count: the reference redraws the steady state in 'outlet-train'), r = rs * profile,
   * y = qs - vm r + vm/rm r^2, time = 0, and obs[b] is the first observation (r, v) of the new episode.  done / truncated /
   * reward still report the finished step.  qs_clip (the construction-time action bounds) is left alone. */
  const double* reset_rs;      /* optional [reset_pool_rows] steady-state densities of the coming episodes              */
  const double* reset_profile; /* [M] sin(3 x/L pi)*0.1 + 1 as for pdegym_traffic_reset_masked (required with reset_rs)  */
  double* final_obs;           /* optional [B, 2M]: last observation of the finished episode ("terminal_observation")   */
  int32_t* reset_count;        /* optional [B] in/out: restarts so far; NULL = always pool row b mod reset_pool_rows     */
  int32_t reset_pool_rows;     /* rows of reset_rs (0 = B)                                                              */
  int32_t reserved2_;
} pdegym_bufs_traffic;

int pdegym_traffic_step(const pdegym_params_traffic* prm, const pdegym_bufs_traffic* buf, int32_t B, void* stream);

/* T env-steps in ONE launch (M <= 64; T x traffic_arz_env.py:131-233, the loop of the reference's "RL control" notebook):
 * step t takes the command(s) from actions row t and writes observation slot t + 1,
 * rewards / done / truncated row t; r, y, time of the pdegym_bufs_traffic are read once and written back at the end
 * (bufs->action / obs / reward / done / truncated are ignored).  Bit-identical to T pdegym_traffic_step calls; like them it
 * keeps stepping a finished episode.  With a policy (HOST pointer; layers of <= 256 units -- more than 64: evaluated cooperatively with
 * pdegym_mlp_forward's MFMA reduction, see pdegym_rollout1d.policy --, 2M inputs, action_stride outputs)
 * the command of step t is computed inside the launch from observation slot t (slot 0 = the caller's current observation)
 * rounded to float32, plus noise [T, B, action_stride] (row stride noise_stride), clamped, widened and stored to actions. */
typedef struct pdegym_rollout_traffic {
  int32_t T;
  int32_t reserved_;
  double* obs;              /* [T + 1, B, 2M]  slot 0 is read only by a policy; slots 1 .. T are written         */
  double* actions;          /* [T, B, action_stride]  read, or written when a policy is given                   */
  double* rewards;          /* [T, B]                                                                          */
  uint8_t* done;            /* [T, B]                                                                          */
  uint8_t* truncated;       /* [T, B]                                                                          */
  const struct pdegym_mlp_s* policy;
} pdegym_rollout_traffic;

int pdegym_traffic_rollout(const pdegym_params_traffic* prm, const pdegym_bufs_traffic* buf, const pdegym_rollout_traffic* ro,
                           int32_t B, void* stream);
/* Where mask[b] != 0 (or mask == NULL): rs[b] is taken as given, r = rs*profile, y = qs - vm r + vm/rm r^2, time = 0,
 * obs = (r, v).  profile[M] = sin(3 x/L pi)*0.1 + 1 is computed by the caller in NumPy (libm sin, bit parity). */
int pdegym_traffic_reset_masked(const pdegym_params_traffic* prm, const pdegym_bufs_traffic* buf, const double* profile,
                                const uint8_t* mask, int32_t B, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Brain tumour 1D (float64) -- SURVEY.md section 8f rank 3
 *
 *   pdegym_tumor_step          replaces BrainTumor1D.step + _update_fd + _compute_radiation_field + getTumorRadius
 *                              + terminate/truncate                               environments1d/brain_tumor_env.py:106-352
 *                              + BrainTumorReward.reward                          rewards/brain_tumor_reward.py:30-73
 *   pdegym_tumor_reset_masked  replaces the state part of BrainTumor1D.reset      brain_tumor_env.py:354-384
 * ------------------------------------------------------------------------------------------------------------------ */
enum { PDEGYM_TUMOR_GROWTH = 0, PDEGYM_TUMOR_THERAPY = 1, PDEGYM_TUMOR_POST = 2 };   /* "Growth" / "Therapy" / "Post-Therapy" */
enum { PDEGYM_TUMOR_DAY_GROWTH = 0, PDEGYM_TUMOR_DAY_THERAPY = 1, PDEGYM_TUMOR_DAY_POST = 2, PDEGYM_TUMOR_DAY_SIM = 3,
       PDEGYM_TUMOR_DAY_DEATH = 4, PDEGYM_TUMOR_DAYS = 5 };
enum { PDEGYM_TUMOR_OUT_T1 = 0, PDEGYM_TUMOR_OUT_T2 = 1, PDEGYM_TUMOR_OUT_TREAT = 2, PDEGYM_TUMOR_OUT_DOSE = 3,
       PDEGYM_TUMOR_OUTS = 4 };

typedef struct pdegym_params_tumor {
  int32_t nx;              /* int(round(X/dx)+1)                                             brain_tumor_env.py:51 */
  int32_t nt;              /* int(round(T/dt)+1)                                             base_env_1d.py:23 */
  double dt, dx, dx2;      /* dx2 = dx**2 as the caller's Python evaluates it                :226 */
  double D, rho, alpha, alpha_beta_ratio, k;
  double thr_t1, thr_t2;   /* detection_threshold * k  (caller's double product)             :114 */
  double detect_radius;    /* t1_detection_radius: Growth -> Therapy                          :151 */
  double death_radius;     /* t1_death_radius: truncation                                    :327 */
  double total_dosage, dose_end;   /* total_dosage, dosage_termination_threshold             :159, :171 */
  double margin;           /* 25 (mm) added to the T2 radius for the treated region          :257 */
} pdegym_params_tumor;

typedef struct pdegym_bufs_tumor {
  double* u;               /* [B, nx] live density row, updated in place (it IS the observation)               */
  const double* xscale;    /* [nx] np.linspace(0, X, nx) from the caller                                  :56 */
  const double* control;   /* [B] proportion of total_dosage requested this day (used in Therapy only)         */
  const double* kill;      /* [B] or NULL: 1 - exp(-alpha*BED) precomputed by the caller (bit parity with
                              NumPy's exp); NULL = the kernel evaluates exp itself (<= 1 ulp apart)   :260-262 */
  int32_t* time_index;     /* [B] in/out                                                                       */
  int32_t* stage;          /* [B] in/out PDEGYM_TUMOR_*                                                        */
  double* remaining;       /* [B] in/out remaining_dosage                                                      */
  int32_t* days;           /* [B, 5] in/out growthDays, therapyDays, postTherapyDays, simulationDays, cDeathDay (-1 = None) */
  const double* t_benchmark; /* [B] NaN = not set (reward 0)                                                   */
  double* reward;          /* [B] out                                                                          */
  uint8_t* terminated;     /* [B] out                                                                          */
  uint8_t* truncated;      /* [B] out                                                                          */
  double* out;             /* [B, 4] out: T1 radius and T2 radius of the new row (NaN = invisible), treatment radius and
                              applied dose of this step (0 outside Therapy)                                    */
  const uint8_t* active;   /* [B] or NULL: instances with active[b] == 0 are left untouched (no output is written)    */
  double* history;         /* [B, nt, nx] or NULL: the row of every simulated day t is stored at [b, t] (env.u)   :143 */
  double* t1_log;          /* [B, nt] or NULL: T1 radius / dx of every simulated day (t1_radius_idx_vs_time) :271-273  */
} pdegym_bufs_tumor;

/* Runs of days inside one launch (the loops of TherapyWrapper, brain_tumor_env.py:385-505) */
enum {
  PDEGYM_TUMOR_RUN_ONE_DAY = 0, /* == pdegym_tumor_step                                                                 */
  PDEGYM_TUMOR_RUN_GROWTH = 1,  /* instances in Growth: step(0) until the stage changes or the episode ends  :409-428   */
  PDEGYM_TUMOR_RUN_POST = 2,    /* instances in Post-Therapy: step(0) until terminated or truncated          :437-446   */
  PDEGYM_TUMOR_RUN_TO_END = 3   /* every live instance: step(0) until terminated or truncated (benchmark())  :488-503   */
};

/* One day per call.  Instances with time_index >= nt-1 are left untouched (reward 0, flags 0).  nx <= 4096. */
int pdegym_tumor_step(const pdegym_params_tumor* prm, const pdegym_bufs_tumor* buf, int32_t B, void* stream);
/* Up to max_days days per participating instance inside ONE launch (row and stage machine stay on chip between days);
 * control is 0 on every day of modes 1-3; reward / flags / out are those of the LAST simulated day; instances that do not
 * take part (wrong stage, inactive, or past nt-1) are left untouched.  kill is only honoured by RUN_ONE_DAY. */
int pdegym_tumor_advance(const pdegym_params_tumor* prm, const pdegym_bufs_tumor* buf, int32_t mode, int32_t max_days,
                         int32_t B, void* stream);
/* Where mask[b] != 0 (or mask == NULL): u = init (init_stride = 0 broadcasts one row), time_index = 0, stage = Growth,
 * remaining = total_dosage, days = (0,0,0,0,-1). */
int pdegym_tumor_reset_masked(const pdegym_params_tumor* prm, const pdegym_bufs_tumor* buf, const double* init,
                              int64_t init_stride, const uint8_t* mask, int32_t B, void* stream);

/* ---- policy network on the device (the caller of env.step inside an on-device rollout) ------------------------------
 * The reference evaluates an SB3 "MlpPolicy" (two hidden layers of 64 tanh units by default) between two env.step calls:
 * model = PPO("MlpPolicy", env) / model.predict(obs), examples/transportPDE/transport1Dppo.py:88-90 and
 * transport1DtestAlgorithm.py (the RL controllers).  pdegym_mlp_forward evaluates such a network for B observation rows
 * in ONE launch (float32; weights in a blocked transpose of torch.nn.Linear.weight W[out_dim, in_dim]:
 * w[(k / 4) * out_dim * 4 + n * 4 + k % 4] = W[n][k], zero-padded to a multiple of four k, 16-byte aligned -- one 16-byte
 * load per lane brings four consecutive inputs of its neuron and a wave's load is contiguous), optionally clamping the
 * output to the action box, so that policy + step + auto-reset of a rollout are two launches per env-step.
 * Summation order is k ascending with the bias added last (not a BLAS order): float32-rounding agreement with torch. */
#define PDEGYM_MLP_MAX_LAYERS 4
#define PDEGYM_MLP_MAX_WIDTH 256   /* widest layer output                          */
#define PDEGYM_MLP_MAX_INPUT 8192  /* widest observation row (first layer in_dim)  */
enum { PDEGYM_MLP_IDENTITY = 0, PDEGYM_MLP_TANH = 1, PDEGYM_MLP_RELU = 2 };

typedef struct {
  const float* w;      /* [ceil(in_dim / 4), out_dim, 4] blocked transpose, see above  */
  const float* b;      /* [out_dim] or NULL                                         */
  int32_t in_dim, out_dim;
  int32_t act;         /* PDEGYM_MLP_* applied to this layer's output               */
  int32_t reserved_;
} pdegym_mlp_layer;

typedef struct pdegym_mlp_s {
  int32_t n_layers;    /* 1 .. PDEGYM_MLP_MAX_LAYERS                                 */
  int32_t clamp;       /* nonzero: the last layer's output is clamped to [lo, hi]   */
  float lo, hi;
  int32_t x_f64;       /* nonzero: x holds float64 rows (TrafficPDE1D, BrainTumor1D, float64 NavierStokes2D observations),
                          rounded to float32 as they are read -- what SB3 does before it evaluates its policy           */
  int32_t y_f64;       /* nonzero: y receives float64 (the float32 result widened: those environments' action dtype)   */
  const float* noise;  /* [B, out_dim] float32 or NULL: added to the last layer's output before the clamp -- the exploration
                          noise of a stochastic policy (SB3's Gaussian MlpPolicy samples mean + std * eps), drawn by the caller */
  int64_t noise_stride; /* floats between consecutive rows of noise                                                     */
  pdegym_mlp_layer layer[PDEGYM_MLP_MAX_LAYERS];
} pdegym_mlp;

/* y[b, :] = net(x[b, :]) for b < B; x_stride / y_stride = ELEMENTS between consecutive rows (>= the row lengths). */
int pdegym_mlp_forward(const pdegym_mlp* net, const void* x, int64_t x_stride, void* y, int64_t y_stride, int32_t B,
                       void* stream);

/* ---- test-only: kernel dispatch overrides -------------------------------------------------------------------------
 * Nothing on the product path calls this (the reference has no counterpart); the parity tests use it to run the SAME step
 * through two kernel families and compare the results bit for bit, developer tools for A/B timings.  Process-wide; returns
 * the previous value, or a negative error code for an unknown key. */
enum {
  PDEGYM_DEBUG_NS_GENERIC = 0,        /* != 0: every NavierStokes2D step takes ns_generic_step (workgroup per instance)     */
  PDEGYM_DEBUG_NS_NO_COL = 1,         /* != 0: small grids skip the column-per-lane kernel                                   */
  PDEGYM_DEBUG_NS_COL_MIN_BATCH = 2,  /* >= 0: smallest batch the column-per-lane kernel takes (-1: the built-in rule)       */
  PDEGYM_DEBUG_NS_NO_LDS_JACOBI = 3,  /* != 0: ns_generic_step keeps the pressure in global memory during the sweeps         */
  PDEGYM_DEBUG_COUNT = 4
};
int32_t pdegym_debug_set(int32_t key, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* PDEGYM_H */
