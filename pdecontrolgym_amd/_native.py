"""ctypes binding of libpdegym_hip.so (C ABI declared in include/pdegym.h).

There is NO fallback: if the shared library is missing, or a call is made with tensors that are not on
a HIP device, this module raises.  PyTorch is used only as the owner of device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libpdegym_hip.so")

ABI_VERSION = 15
RING = 128
LOOKBACK = 100
MAX_N1D = 8192
MAX_N1D_REG = 2048      # rows kept in registers (PDEGYM_MAX_N1D); the rollout kernel handles these

CONTROL = {"Dirchilet": 0, "Neumann": 1}
FLUX_LINEAR, FLUX_BURGERS = 0, 1
ACTION_F32, ACTION_F64, ACTION_WEAK = 0, 1, 2
SENSE_FULL, SENSE_LAST, SENSE_LAST_DERIV, SENSE_FIRST_DERIV, SENSE_FIRST = range(5)
REWARD_NONE, REWARD_TUNED1D, REWARD_NORM_L1, REWARD_NORM_L2, REWARD_NORM_LINF = range(5)
HORIZON_TEMPORAL, HORIZON_DIFFERENTIAL, HORIZON_T = 0, 1, 2      # NormReward horizon evaluated by the step kernels
BC = {"Neumann": 0, "Dirchilet": 1, "Controllable": 2}
EDGES = ("lower", "upper", "left", "right")

EXPORTS = [
    "pdegym_abi_version", "pdegym_last_error", "pdegym_transport_step", "pdegym_parabolic_step",
    "pdegym_reset1d_masked", "pdegym_rownorm2_f32", "pdegym_selftest_quotient", "pdegym_ns2d_step_f32", "pdegym_ns2d_step_f64", "pdegym_ns2d_rollout_f32", "pdegym_ns2d_rollout_f64",
    "pdegym_ns2d_solve_pressure_f32", "pdegym_ns2d_solve_pressure_f64", "pdegym_ns2d_reset_masked_f32",
    "pdegym_ns2d_reset_masked_f64", "pdegym_traffic_step", "pdegym_traffic_reset_masked",
    "pdegym_tumor_step", "pdegym_tumor_advance", "pdegym_tumor_reset_masked", "pdegym_mlp_forward",
    "pdegym_transport_rollout", "pdegym_parabolic_rollout", "pdegym_traffic_rollout", "pdegym_debug_set",
]
# keys of pdegym_debug_set (test-only dispatch overrides, include/pdegym.h)
DEBUG_NS_GENERIC, DEBUG_NS_NO_COL, DEBUG_NS_COL_MIN_BATCH, DEBUG_NS_NO_LDS_JACOBI = range(4)
MLP_MAX_LAYERS, MLP_MAX_WIDTH, MLP_MAX_INPUT = 4, 256, 8192
MLP_IDENTITY, MLP_TANH, MLP_RELU = 0, 1, 2


class Params1D(C.Structure):
    _fields_ = [("n", C.c_int32), ("nt", C.c_int32), ("substeps", C.c_int32), ("control_type", C.c_int32),
                ("normalize", C.c_int32), ("sensing", C.c_int32), ("limit_state", C.c_int32),
                ("reward_kind", C.c_int32), ("reward_nt", C.c_int32), ("dt", C.c_float), ("dx", C.c_float),
                ("F", C.c_float), ("max_control", C.c_float), ("max_state", C.c_float),
                ("truncate_penalty", C.c_float), ("terminate_reward", C.c_float), ("rdx", C.c_double),
                ("flux", C.c_int32), ("beta_f64", C.c_int32), ("action_kind", C.c_int32), ("reward_horizon", C.c_int32),
                ("dt64", C.c_double), ("dx64", C.c_double), ("max_control64", C.c_double),
                ("reward_t_horizon", C.c_int32), ("reserved1_", C.c_int32)]


class Bufs1D(C.Structure):
    _fields_ = [("u", C.c_void_p), ("beta", C.c_void_p), ("beta_stride", C.c_int64), ("action", C.c_void_p),
                ("time_index", C.c_void_p), ("bsum", C.c_void_p), ("ring", C.c_void_p), ("obs", C.c_void_p),
                ("reward", C.c_void_p), ("norm_now", C.c_void_p), ("norm_back", C.c_void_p),
                ("terminated", C.c_void_p), ("truncated", C.c_void_p), ("history", C.c_void_p),
                ("reset_init", C.c_void_p), ("final_obs", C.c_void_p), ("reset_beta", C.c_void_p),
                ("reset_count", C.c_void_p), ("reset_pool_rows", C.c_int32), ("reserved_", C.c_int32),
                ("state_in", C.c_void_p)]


class Rollout1D(C.Structure):
    _fields_ = [("T", C.c_int32), ("reserved_", C.c_int32), ("obs", C.c_void_p), ("actions", C.c_void_p),
                ("rewards", C.c_void_p), ("terminated", C.c_void_p), ("truncated", C.c_void_p), ("policy", C.c_void_p),
                ("obs_noise", C.c_void_p), ("obs_seen", C.c_void_p)]


class RolloutNS2D(C.Structure):
    _fields_ = [("T", C.c_int32), ("reserved_", C.c_int32), ("obs", C.c_void_p), ("actions", C.c_void_p), ("rewards", C.c_void_p),
                ("terminated", C.c_void_p)]


class ParamsNS2D(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nt", C.c_int32), ("iters", C.c_int32),
                ("action_dim", C.c_int32), ("bc", (C.c_int32 * 2) * 4), ("dt", C.c_double), ("dx", C.c_double),
                ("dy", C.c_double), ("viscosity", C.c_double), ("density", C.c_double), ("gamma", C.c_double)]


class BufsNS2D(C.Structure):
    _fields_ = [("u", C.c_void_p), ("v", C.c_void_p), ("p", C.c_void_p), ("scratch", C.c_void_p),
                ("action", C.c_void_p), ("time_index", C.c_void_p), ("U_ref", C.c_void_p),
                ("action_ref", C.c_void_p), ("nt_ref", C.c_int32), ("obs", C.c_void_p), ("reward", C.c_void_p),
                ("terminated", C.c_void_p), ("p_out", C.c_void_p), ("state_in", C.c_void_p), ("reset_u0", C.c_void_p),
                ("reset_v0", C.c_void_p), ("reset_p0", C.c_void_p), ("final_obs", C.c_void_p), ("reset_count", C.c_void_p),
                ("reset_pool_rows", C.c_int32), ("reserved_", C.c_int32)]


TRAFFIC_SIM = {"inlet": 0, "outlet": 1, "both": 2, "outlet-train": 3}


class ParamsTraffic(C.Structure):
    _fields_ = [("M", C.c_int32), ("control_freq", C.c_int32), ("sim", C.c_int32), ("limit", C.c_int32),
                ("dt", C.c_double), ("dx", C.c_double), ("T", C.c_double), ("vm", C.c_double), ("rm", C.c_double),
                ("tau", C.c_double)]


class BufsTraffic(C.Structure):
    _fields_ = [("r", C.c_void_p), ("y", C.c_void_p), ("action", C.c_void_p), ("time", C.c_void_p), ("rs", C.c_void_p),
                ("qs_clip", C.c_void_p), ("obs", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p),
                ("truncated", C.c_void_p), ("action_stride", C.c_int32), ("reserved_", C.c_int32),
                ("reset_rs", C.c_void_p), ("reset_profile", C.c_void_p), ("final_obs", C.c_void_p), ("reset_count", C.c_void_p),
                ("reset_pool_rows", C.c_int32), ("reserved2_", C.c_int32)]


class RolloutTraffic(C.Structure):
    _fields_ = [("T", C.c_int32), ("reserved_", C.c_int32), ("obs", C.c_void_p), ("actions", C.c_void_p),
                ("rewards", C.c_void_p), ("done", C.c_void_p), ("truncated", C.c_void_p), ("policy", C.c_void_p)]


TUMOR_GROWTH, TUMOR_THERAPY, TUMOR_POST = range(3)
TUMOR_STAGE_NAMES = ("Growth", "Therapy", "Post-Therapy")
TUMOR_RUN_ONE_DAY, TUMOR_RUN_GROWTH, TUMOR_RUN_POST, TUMOR_RUN_TO_END = range(4)


class ParamsTumor(C.Structure):
    _fields_ = [("nx", C.c_int32), ("nt", C.c_int32)] + [(k, C.c_double) for k in (
        "dt", "dx", "dx2", "D", "rho", "alpha", "alpha_beta_ratio", "k", "thr_t1", "thr_t2", "detect_radius",
        "death_radius", "total_dosage", "dose_end", "margin")]


class BufsTumor(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("u", "xscale", "control", "kill", "time_index", "stage", "remaining", "days",
                                          "t_benchmark", "reward", "terminated", "truncated", "out", "active", "history",
                                          "t1_log")]


class MlpLayer(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("in_dim", C.c_int32), ("out_dim", C.c_int32), ("act", C.c_int32),
                ("reserved_", C.c_int32)]


class Mlp(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("clamp", C.c_int32), ("lo", C.c_float), ("hi", C.c_float),
                ("x_f64", C.c_int32), ("y_f64", C.c_int32), ("noise", C.c_void_p), ("noise_stride", C.c_int64),
                ("layer", MlpLayer * MLP_MAX_LAYERS)]


class NativeError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen libpdegym_hip.so (once) and declare the prototypes. Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: build it with `python -m pdecontrolgym_amd.build` "
            "(there is no CPU fallback for the PDE steppers)")
    import torch  # noqa: F401  (first: the library must bind to the HIP runtime torch has loaded, not to a second copy of libamdhip64)
    lib = C.CDLL(LIB_PATH)
    lib.pdegym_abi_version.restype = C.c_int
    lib.pdegym_last_error.restype = C.c_char_p
    for name in ("pdegym_transport_step", "pdegym_parabolic_step"):
        f = getattr(lib, name)
        f.argtypes = [C.POINTER(Params1D), C.POINTER(Bufs1D), C.c_int32, C.c_void_p]
        f.restype = C.c_int
    for f in (lib.pdegym_transport_rollout, lib.pdegym_parabolic_rollout):
        f.argtypes = [C.POINTER(Params1D), C.POINTER(Bufs1D), C.POINTER(Rollout1D), C.c_int32, C.c_void_p]
        f.restype = C.c_int
    lib.pdegym_reset1d_masked.argtypes = [C.POINTER(Params1D), C.POINTER(Bufs1D), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.pdegym_reset1d_masked.restype = C.c_int
    lib.pdegym_rownorm2_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.pdegym_rownorm2_f32.restype = C.c_int
    lib.pdegym_selftest_quotient.argtypes = [C.c_void_p, C.c_float, C.c_double, C.c_void_p, C.c_int32, C.c_void_p]
    lib.pdegym_selftest_quotient.restype = C.c_int
    for sfx in ("f32", "f64"):
        f = getattr(lib, "pdegym_ns2d_step_" + sfx)
        f.argtypes = [C.POINTER(ParamsNS2D), C.POINTER(BufsNS2D), C.c_int32, C.c_void_p]
        f.restype = C.c_int
        f = getattr(lib, "pdegym_ns2d_rollout_" + sfx)
        f.argtypes = [C.POINTER(ParamsNS2D), C.POINTER(BufsNS2D), C.POINTER(RolloutNS2D), C.c_int32, C.c_void_p]
        f.restype = C.c_int
        f = getattr(lib, "pdegym_ns2d_solve_pressure_" + sfx)
        f.argtypes = [C.POINTER(ParamsNS2D), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        f.restype = C.c_int
        f = getattr(lib, "pdegym_ns2d_reset_masked_" + sfx)
        f.argtypes = [C.POINTER(ParamsNS2D), C.POINTER(BufsNS2D), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        f.restype = C.c_int
    lib.pdegym_traffic_rollout.argtypes = [C.POINTER(ParamsTraffic), C.POINTER(BufsTraffic), C.POINTER(RolloutTraffic), C.c_int32, C.c_void_p]
    lib.pdegym_traffic_rollout.restype = C.c_int
    lib.pdegym_traffic_step.argtypes = [C.POINTER(ParamsTraffic), C.POINTER(BufsTraffic), C.c_int32, C.c_void_p]
    lib.pdegym_traffic_step.restype = C.c_int
    lib.pdegym_traffic_reset_masked.argtypes = [C.POINTER(ParamsTraffic), C.POINTER(BufsTraffic), C.c_void_p, C.c_void_p,
                                                C.c_int32, C.c_void_p]
    lib.pdegym_traffic_reset_masked.restype = C.c_int
    lib.pdegym_tumor_step.argtypes = [C.POINTER(ParamsTumor), C.POINTER(BufsTumor), C.c_int32, C.c_void_p]
    lib.pdegym_tumor_step.restype = C.c_int
    lib.pdegym_tumor_advance.argtypes = [C.POINTER(ParamsTumor), C.POINTER(BufsTumor), C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.pdegym_tumor_advance.restype = C.c_int
    lib.pdegym_tumor_reset_masked.argtypes = [C.POINTER(ParamsTumor), C.POINTER(BufsTumor), C.c_void_p, C.c_int64, C.c_void_p,
                                              C.c_int32, C.c_void_p]
    lib.pdegym_tumor_reset_masked.restype = C.c_int
    lib.pdegym_mlp_forward.argtypes = [C.POINTER(Mlp), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    lib.pdegym_mlp_forward.restype = C.c_int
    lib.pdegym_debug_set.argtypes = [C.c_int32, C.c_int32]
    lib.pdegym_debug_set.restype = C.c_int32
    if lib.pdegym_abi_version() != ABI_VERSION:
        raise NativeError(f"ABI mismatch: library {lib.pdegym_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


class ns_dispatch:
    """Test-only context manager around pdegym_debug_set: run NavierStokes2D steps through another kernel family
    (generic=True: workgroup-per-instance kernel; no_col=True: no column-per-lane kernel; col_min_batch=n; no_lds_jacobi=True)."""

    def __init__(self, generic=None, no_col=None, col_min_batch=None, no_lds_jacobi=None):
        self.want = {DEBUG_NS_GENERIC: generic, DEBUG_NS_NO_COL: no_col, DEBUG_NS_COL_MIN_BATCH: col_min_batch,
                     DEBUG_NS_NO_LDS_JACOBI: no_lds_jacobi}
        self.old = {}

    def __enter__(self):
        lib = load()
        for k, v in self.want.items():
            if v is not None:
                self.old[k] = lib.pdegym_debug_set(k, int(v))
        return self

    def __exit__(self, *exc):
        lib = load()
        for k, v in self.old.items():
            lib.pdegym_debug_set(k, v)
        return False


def check(rc: int, what: str):
    if rc != 0:
        raise NativeError(f"{what} failed ({rc}): {load().pdegym_last_error().decode()}")


def dptr(t, dtype=None):
    """Device pointer of a torch tensor, with the checks the C side cannot do."""
    if t is None:
        return None
    if not t.is_cuda and not t.is_pinned():
        # pinned host memory (hipHostMalloc: mapped into the device's address space, fine-grained coherent) is device-accessible:
        # the batch-of-one faces pass their command and take their results through it (PDEBatch1D.enable_host_io)
        raise NativeError("pdegym kernels need tensors on a HIP device or in pinned host memory (got a pageable CPU tensor); "
                          "there is no CPU path")
    if not t.is_contiguous():
        raise NativeError("pdegym kernels need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise NativeError(f"expected dtype {dtype}, got {t.dtype}")
    return t.data_ptr()


def current_stream_ptr(device=None):
    import torch
    return torch.cuda.current_stream(device).cuda_stream
