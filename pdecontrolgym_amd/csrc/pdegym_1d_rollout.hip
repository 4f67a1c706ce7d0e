// pdegym_1d_rollout.hip -- T env-steps of the 1D transport / reaction-diffusion environments in ONE launch (pdegym_*_rollout),
// optionally with the MLP policy that commands them evaluated inside the launch (pdegym_policy.h).  The env-step itself is
// step1d_body of pdegym_1d_body.h: every value equals what T step calls produce, bit for bit.
#include "pdegym_1d_body.h"
#include "pdegym_policy.h"

namespace {

// T env-steps of one instance by one wave in ONE launch (pdegym_*_rollout): iteration t is the step kernel's body with the
// row written to observation slot t + 1, action / reward / flags taken from / written to row t of the [T, B] rollout arrays
// -- every value equals what T separate step calls produce, bit for bit.  What it removes is the kernel boundary between
// env-steps: no dispatch gap, no load phase (the state stays in registers, Carry), and the waves of a SIMD drift apart
// instead of finishing in two generations (docs/HISTORY.md section 3.2).
// FULL: the row fills the wave exactly (n - J0 == 64 EPL): the slot masks fold away (pdegym_1d_body.h: run_substeps).
template <int EPL, bool PARABOLIC, bool BURGERS, bool FULL>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void rollout1d_kernel(pdegym_params1d P, pdegym_bufs1d Bf, pdegym_rollout1d Ro,
                                                                         int B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int inst = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (inst >= B) return;  // wave-uniform
  const int n = FULL ? kWave * EPL + (PARABOLIC ? 1 : 0) : P.n;
  const size_t slot = (size_t)B * n;
  // The state -- row, beta, time index, sums AND the norm ring -- stays in registers over the T env-steps (Carry).  Round 5: the
  // steady state of the loop has NO load at all: the commands of 64 env-steps arrive in one load per lane (lane l holds step
  // t0 + l; handed out by v_readlane), the ring is two registers per lane, and the per-instance
  // state words that every step would overwrite are stored by the last step only.  gfx9 counts loads and stores in one counter
  // (vmcnt), so the single ring / command load per env-step of the round-4 loop made every step wait for the stores of the step
  // before (SQ_WAIT_ANY 57 % of the wave cycles at S = 1); now only the batch hand-over, once per 64 env-steps, waits.
  Carry<EPL> C;
  carry_load<EPL, PARABOLIC, FULL>(C, P, Bf, Ro.obs, inst, lane);
  // the per-step views of the rollout arrays advance by one slot / row per env-step (no 64-bit multiplications in the loop)
  pdegym_bufs1d S = Bf;
  S.u = nullptr;
  S.history = nullptr;
  S.state_in = Ro.obs;
  S.obs = Ro.obs + slot;
  S.action = Ro.actions;
  S.reward = Ro.rewards;
  S.terminated = Ro.terminated;
  S.truncated = Ro.truncated;
  auto command_batch = [&](int t0) { return (t0 + lane < Ro.T) ? Ro.actions[(size_t)(t0 + lane) * B + inst] : 0.f; };
  float a_cur = command_batch(0);
  drain_vmem();
  for (int t = 0; t < Ro.T; ++t) {
    const int j = t & (kWave - 1);
    if (j == 0 && t) {      // wave-uniform; once per 64 env-steps, the loop's only wait for memory
      a_cur = command_batch(t);
      drain_vmem();
    }
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_cur), j));
    step1d_body<EPL, PARABOLIC, false, false, BURGERS, false, true, true, FULL>(P, S, B, inst, lane, &a, &C, t == Ro.T - 1);
    S.state_in = S.obs;
    S.obs += slot;
    S.action = static_cast<const float*>(S.action) + B;
    S.reward += B;
    S.terminated += B;
    S.truncated += B;
  }
  carry_store_ring<EPL>(C, Bf, inst, lane);
}


// The general form (round 4): Neumann actuation and / or scalar sensing -- what the reference's control / sensing table offers
// beyond the Dirichlet / full-state corner (hyperbolic.py:66-124, parabolic.py:66-122).  Iteration t is the step kernel's body
// exactly as step1d_kernel instantiates it (select form for Neumann, the fast form otherwise), with the state going through
// memory between iterations: full-state sensing keeps it in the observation slots (slot t in, slot t + 1 out), scalar sensing in
// bufs.u (in place) while the observation slots receive the sensed value.  A wave re-reads what its own lanes stored (rows, time
// index, sums, ring), so iterations are separated by a workgroup-scope release / acquire pair, nothing more.
template <int EPL, bool PARABOLIC, bool NEUMANN, bool BURGERS>
__device__ __forceinline__ void rollout1d_general_step(const pdegym_params1d& P, const pdegym_bufs1d& Bf, const pdegym_rollout1d& Ro, int B,
                                                        int inst, int lane, int t, const float* command) {
  const bool full = P.sensing == PDEGYM_SENSE_FULL;
  const size_t slot = (size_t)B * (full ? P.n : 1);
  pdegym_bufs1d S = Bf;
  S.history = nullptr;
  if (full) {
    S.u = nullptr;
    S.state_in = Ro.obs + (size_t)t * slot;
  } else {
    S.state_in = nullptr;
  }
  S.obs = Ro.obs + (size_t)(t + 1) * slot;
  S.action = Ro.actions + (size_t)t * B;
  S.reward = Ro.rewards + (size_t)t * B;
  S.terminated = Ro.terminated + (size_t)t * B;
  S.truncated = Ro.truncated + (size_t)t * B;
  step1d_body<EPL, PARABOLIC, NEUMANN, false, BURGERS>(P, S, B, inst, lane, command);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int EPL, bool PARABOLIC, bool NEUMANN, bool BURGERS>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void rollout1d_general_kernel(pdegym_params1d P, pdegym_bufs1d Bf, pdegym_rollout1d Ro,
                                                                                 int B) {
  const int lane = threadIdx.x & (kWave - 1);
  const int inst = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (inst >= B) return;  // wave-uniform
  for (int t = 0; t < Ro.T; ++t) rollout1d_general_step<EPL, PARABOLIC, NEUMANN, BURGERS>(P, Bf, Ro, B, inst, lane, t, nullptr);
}


// ================================================================================================
// ---- the policy inside the rollout kernel (pdegym_policy.h) -----------------------------------------------------------------
// The sensing-noise hook of the reference (hyperbolic.py:160-164: the agent sees sensing_noise_func(observation)) as pre-drawn
// additive noise: the wave's LDS copy of observation t (od values) becomes obs + obs_noise[t], which is what the policy reads
// and what obs_seen[t] receives; the observation slots themselves -- the plant state with full-state sensing -- stay clean.
__device__ __forceinline__ void sense_noise(const float* obs_noise, float* obs_seen, float* xw, int od, int B, int inst, int lane, int t) {
  if (!obs_noise && !obs_seen) return;     // wave-uniform
  const size_t base = ((size_t)t * B + inst) * od;
  for (int j = lane; j < od; j += kWave) {
    float v = xw[j];
    if (obs_noise) v += obs_noise[base + j];
    xw[j] = v;
    if (obs_seen) obs_seen[base + j] = v;
  }
  pdegym_policy::wave_lds_sync();
}

// WIDE: a network with a layer of more than 64 units, evaluated by the 16 waves together (pdegym_policy.h: eval_wide) -- every wave of
// the workgroup, with or without an instance, runs all T iterations because of its barriers.
template <int EPL, bool PARABOLIC, bool BURGERS, bool WIDE, bool FULL>
__global__ __launch_bounds__(kWave* pdegym_policy::kWaves) void rollout1d_policy_kernel(pdegym_params1d P, pdegym_bufs1d Bf,
                                                                                        pdegym_rollout1d Ro, pdegym_mlp N, int B) {
  namespace pol = pdegym_policy;
  extern __shared__ __attribute__((aligned(16))) float pol_smem[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int inst = blockIdx.x * pol::kWaves + wave;
  constexpr int J0 = PARABOLIC ? 1 : 0;
  // (the policy's loops take the row length from the kernel argument even when FULL makes it a compile-time constant for the
  // env-step: with a constant trip count the compiler unrolls the whole reduction and spills 74 registers)
  const int n = P.n, xpad = pol::xpad(n);
  pol::Staged St;
  pol::Wide Wd;
  if constexpr (WIDE) Wd = pol::wide_setup(N, pol_smem, n, wave, lane);
  else St = pol::stage(N, pol_smem);      // the launch's only barrier
  const bool active = inst < B;           // wave-uniform
  if (!WIDE && !active) return;
  float* const xw = WIDE ? Wd.X + wave * Wd.ldx : pol_smem + St.end + wave * (xpad + 2 * pol::kMaxWidth);
  float* const hw = xw + xpad;
  const size_t slot = (size_t)B * n;
  const int ns = n - J0, s0 = lane * EPL;
  Carry<EPL> C;       // the state stays in registers over the T env-steps (see rollout1d_kernel)
  if (active) carry_load<EPL, PARABOLIC, FULL>(C, P, Bf, Ro.obs, inst, lane);
  if (!WIDE)
    for (int j = n + lane; j < xpad; j += kWave) xw[j] = 0.f;     // zero padding to a multiple of four: written once
  for (int t = 0; t < Ro.T; ++t) {
    if (active) {
      // observation of this instance -> LDS, straight from the carried row (slot t of Ro.obs holds the same values)
      if (PARABOLIC && lane == 0) xw[0] = C.bl;
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        if (s0 + e < ns) xw[J0 + s0 + e] = C.x[e];
      pol::wave_lds_sync();
      sense_noise(Ro.obs_noise, Ro.obs_seen, xw, n, B, inst, lane, t);
    }
    float a;
    if constexpr (WIDE) a = pol::eval_wide(N, Wd, n, wave, lane);
    else a = pol::lane_value(pol::eval(N, St, pol_smem, xw, hw, n, lane), 0);      // neuron 0 of the last layer
    if (!active) continue;
    if (N.noise) a += N.noise[((size_t)t * B + inst) * N.noise_stride];
    if (N.clamp) a = fminf(fmaxf(a, N.lo), N.hi);
    if (lane == 0) Ro.actions[(size_t)t * B + inst] = a;

    pdegym_bufs1d S = Bf;
    S.u = nullptr;
    S.history = nullptr;
    S.state_in = Ro.obs + (size_t)t * slot;
    S.obs = Ro.obs + (size_t)(t + 1) * slot;
    S.action = Ro.actions + (size_t)t * B;
    S.reward = Ro.rewards + (size_t)t * B;
    S.terminated = Ro.terminated + (size_t)t * B;
    S.truncated = Ro.truncated + (size_t)t * B;
    step1d_body<EPL, PARABOLIC, false, false, BURGERS, false, true, true, FULL>(P, S, B, inst, lane, &a, &C, t == Ro.T - 1);
  }
  if (active) carry_store_ring<EPL>(C, Bf, inst, lane);
}

// The policy in front of the general step (Neumann actuation / scalar sensing): its input is observation slot t as stored -- od = n
// values, or the one sensed value -- read back from memory after the previous iteration's fence.
template <int EPL, bool PARABOLIC, bool NEUMANN, bool BURGERS, bool WIDE>
__global__ __launch_bounds__(kWave* pdegym_policy::kWaves) void rollout1d_policy_general_kernel(pdegym_params1d P, pdegym_bufs1d Bf,
                                                                                                pdegym_rollout1d Ro, pdegym_mlp N, int B) {
  namespace pol = pdegym_policy;
  extern __shared__ __attribute__((aligned(16))) float pol_smem[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int inst = blockIdx.x * pol::kWaves + wave;
  const int od = P.sensing == PDEGYM_SENSE_FULL ? P.n : 1, xpad = pol::xpad(od);
  pol::Staged St;
  pol::Wide Wd;
  if constexpr (WIDE) Wd = pol::wide_setup(N, pol_smem, od, wave, lane);
  else St = pol::stage(N, pol_smem);      // the launch's only barrier
  const bool active = inst < B;           // wave-uniform
  if (!WIDE && !active) return;
  float* const xw = WIDE ? Wd.X + wave * Wd.ldx : pol_smem + St.end + wave * (xpad + 2 * pol::kMaxWidth);
  float* const hw = xw + xpad;
  if (!WIDE)
    for (int j = od + lane; j < xpad; j += kWave) xw[j] = 0.f;
  for (int t = 0; t < Ro.T; ++t) {
    if (active) {
      const float* orow = Ro.obs + ((size_t)t * B + inst) * od;
      for (int j = lane; j < od; j += kWave) xw[j] = orow[j];
      pol::wave_lds_sync();
      sense_noise(Ro.obs_noise, Ro.obs_seen, xw, od, B, inst, lane, t);
    }
    float a;
    if constexpr (WIDE) a = pol::eval_wide(N, Wd, od, wave, lane);
    else a = pol::lane_value(pol::eval(N, St, pol_smem, xw, hw, od, lane), 0);
    if (!active) continue;
    if (N.noise) a += N.noise[((size_t)t * B + inst) * N.noise_stride];
    if (N.clamp) a = fminf(fmaxf(a, N.lo), N.hi);
    if (lane == 0) Ro.actions[(size_t)t * B + inst] = a;
    rollout1d_general_step<EPL, PARABOLIC, NEUMANN, BURGERS>(P, Bf, Ro, B, inst, lane, t, &a);
  }
}

template <bool PARABOLIC, bool BURGERS = false>
int launch_rollout(const pdegym_params1d* prm, const pdegym_bufs1d* buf, const pdegym_rollout1d* ro, int B, void* stream) {
  if (!prm || !buf || !ro) return pdegym::fail(-1, "null params/bufs/rollout");
  if (B <= 0 || ro->T <= 0) return 0;
  const pdegym_params1d& P = *prm;
  if (P.n < 3 || P.n > PDEGYM_MAX_N1D) return pdegym::fail(-2, "rollout: n must be in [3, 2048] (register-resident rows)");
  if (P.nt < 2) return pdegym::fail(-2, "nt must be >= 2");
  if (P.sensing < PDEGYM_SENSE_FULL || P.sensing > PDEGYM_SENSE_FIRST) return pdegym::fail(-2, "bad sensing code");
  if (P.sensing != PDEGYM_SENSE_FULL && !buf->u) return pdegym::fail(-3, "rollout with scalar sensing: the state lives in bufs.u (obs slots hold the sensed values)");
  if (P.control_type != PDEGYM_CONTROL_DIRICHLET && P.control_type != PDEGYM_CONTROL_NEUMANN) return pdegym::fail(-2, "bad control_type");
  if (P.beta_f64 || P.action_kind != PDEGYM_ACTION_F32) return pdegym::fail(-2, "rollout: float32 beta and actions only");
  if (buf->history) return pdegym::fail(-2, "rollout cannot record a history buffer");
  if (P.reward_horizon != PDEGYM_HORIZON_TEMPORAL) return pdegym::fail(-2, "rollout: only the temporal reward horizon is evaluated in the rollout kernels");
  if (!buf->beta || !buf->time_index || !buf->bsum || !buf->ring || !buf->norm_now || !buf->norm_back)
    return pdegym::fail(-3, "null device buffer");
  if (!ro->obs || !ro->actions || !ro->terminated || !ro->truncated) return pdegym::fail(-3, "null rollout buffer");
  if (P.reward_kind != PDEGYM_REWARD_NONE && !ro->rewards) return pdegym::fail(-3, "null reward buffer");
  hipStream_t st = (hipStream_t)stream;
  const int nslots = P.n - (PARABOLIC ? 1 : 0);
  const int epl = (nslots + kWave - 1) / kWave;
  // the carried, register-resident form is the Dirichlet / full-state corner; everything else takes the general kernels
  const bool neumann = P.control_type == PDEGYM_CONTROL_NEUMANN;
  const bool general = neumann || P.sensing != PDEGYM_SENSE_FULL;
  if ((ro->obs_noise || ro->obs_seen) && !ro->policy) return pdegym::fail(-2, "rollout: obs_noise / obs_seen belong to the policy's input (policy is NULL)");
  if (ro->policy) {
    const pdegym_mlp& N = *ro->policy;
    const int od = P.sensing == PDEGYM_SENSE_FULL ? P.n : 1;
    if (const char* why = pdegym_policy::check(N, od, 1, true)) return pdegym::fail(-2, why);
    if (N.x_f64 || N.y_f64) return pdegym::fail(-2, "policy inside the 1D rollout kernel: float32 observations and commands");
    if (epl > 8) return pdegym::fail(-2, "policy inside the rollout kernel: rows of up to 513 nodes");
    const dim3 pgrid((B + pdegym_policy::kWaves - 1) / pdegym_policy::kWaves), pblock(kWave * pdegym_policy::kWaves);
    bool ok = true;
    // (round 5, measured again after the loop lost its waits: the cooperative MFMA evaluation forced onto a 64-64 network is slower
    // than one fma chain per neuron -- 0.99 against 0.73 ms per 100 env-steps at S = 1, 0.55 against 0.45 ms per 25 at S = 100: four
    // workgroup barriers per env-step and twelve of sixteen waves without an output tile)
    const bool wide = pdegym_policy::is_wide(N);
    const int lds_bytes = pdegym_policy::lds_floats(N, od) * (int)sizeof(float);
    auto launch_pol = [&](auto kernel, signed char (&attr)[pdegym::kMaxDevices]) {
      ok = pdegym::raise_dynamic_lds_limit(reinterpret_cast<const void*>(kernel), pdegym_policy::kMaxLdsBytes, attr);
      if (ok) hipLaunchKernelGGL(kernel, pgrid, pblock, lds_bytes, st, P, *buf, *ro, N, B);
    };
    auto gop = [&](auto tag) {
      constexpr int E = decltype(tag)::value;
      static signed char attr[8][pdegym::kMaxDevices] = {};
      if (!general) {
        constexpr bool kHasFull = E == 1 || E == 2 || E == 4 || E == 8;      // rows of 64 / 128 / 256 / 512 slots
        if (kHasFull && nslots == kWave * E) {
          if (wide) launch_pol(&rollout1d_policy_kernel<E, PARABOLIC, BURGERS, true, kHasFull>, attr[6]);
          else launch_pol(&rollout1d_policy_kernel<E, PARABOLIC, BURGERS, false, kHasFull>, attr[7]);
        } else {
          if (wide) launch_pol(&rollout1d_policy_kernel<E, PARABOLIC, BURGERS, true, false>, attr[0]);
          else launch_pol(&rollout1d_policy_kernel<E, PARABOLIC, BURGERS, false, false>, attr[1]);
        }
      } else if (neumann) {
        if (wide) launch_pol(&rollout1d_policy_general_kernel<E, PARABOLIC, true, BURGERS, true>, attr[2]);
        else launch_pol(&rollout1d_policy_general_kernel<E, PARABOLIC, true, BURGERS, false>, attr[3]);
      } else {
        if (wide) launch_pol(&rollout1d_policy_general_kernel<E, PARABOLIC, false, BURGERS, true>, attr[4]);
        else launch_pol(&rollout1d_policy_general_kernel<E, PARABOLIC, false, BURGERS, false>, attr[5]);
      }
    };
    if (epl <= 1) gop(std::integral_constant<int, 1>{});
    else if (epl <= 2) gop(std::integral_constant<int, 2>{});
    else if (epl <= 3) gop(std::integral_constant<int, 3>{});
    else if (epl <= 4) gop(std::integral_constant<int, 4>{});
    else if (epl <= 5) gop(std::integral_constant<int, 5>{});
    else if (epl <= 6) gop(std::integral_constant<int, 6>{});
    else gop(std::integral_constant<int, 8>{});
    if (!ok) return pdegym::fail(-4, "cannot raise the dynamic LDS limit of rollout1d_policy_kernel");
    return pdegym::check_launch("rollout1d_policy");
  }
  const dim3 grid((B + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
  auto go = [&](auto tag) {
    constexpr int E = decltype(tag)::value;
    constexpr bool kHasFull = E == 1 || E == 2 || E == 4 || E == 8;      // rows of 64 / 128 / 256 / 512 slots
    if (!general && kHasFull && nslots == kWave * E) hipLaunchKernelGGL((rollout1d_kernel<E, PARABOLIC, BURGERS, kHasFull>), grid, block, 0, st, P, *buf, *ro, B);
    else if (!general) hipLaunchKernelGGL((rollout1d_kernel<E, PARABOLIC, BURGERS, false>), grid, block, 0, st, P, *buf, *ro, B);
    else if (neumann) hipLaunchKernelGGL((rollout1d_general_kernel<E, PARABOLIC, true, BURGERS>), grid, block, 0, st, P, *buf, *ro, B);
    else hipLaunchKernelGGL((rollout1d_general_kernel<E, PARABOLIC, false, BURGERS>), grid, block, 0, st, P, *buf, *ro, B);
  };
  // the same slots-per-lane choice as launch_step: the norm reductions (hence rewards) depend on the layout
  if (epl <= 1) go(std::integral_constant<int, 1>{});
  else if (epl <= 2) go(std::integral_constant<int, 2>{});
  else if (epl <= 3) go(std::integral_constant<int, 3>{});
  else if (epl <= 4) go(std::integral_constant<int, 4>{});
  else if (epl <= 5) go(std::integral_constant<int, 5>{});
  else if (epl <= 6) go(std::integral_constant<int, 6>{});
  else if (epl <= 8) go(std::integral_constant<int, 8>{});
  else if (epl <= 12) go(std::integral_constant<int, 12>{});
  else if (epl <= 16) go(std::integral_constant<int, 16>{});
  else if (epl <= 24) go(std::integral_constant<int, 24>{});
  else go(std::integral_constant<int, 32>{});
  return pdegym::check_launch("rollout1d");
}

}  // namespace

extern "C" {

int pdegym_transport_rollout(const pdegym_params1d* prm, const pdegym_bufs1d* buf, const pdegym_rollout1d* ro, int32_t B,
                             void* stream) {
  if (prm && prm->flux == PDEGYM_FLUX_BURGERS) return launch_rollout<false, true>(prm, buf, ro, B, stream);
  return launch_rollout<false, false>(prm, buf, ro, B, stream);
}

int pdegym_parabolic_rollout(const pdegym_params1d* prm, const pdegym_bufs1d* buf, const pdegym_rollout1d* ro, int32_t B,
                             void* stream) {
  return launch_rollout<true>(prm, buf, ro, B, stream);
}

}  // extern "C"
