"""Constructor kwargs of every golden case (mirrors tests/golden/make_golden.py; data lives in the .npz)."""
import numpy as np

T_BASE = dict(T=1, dt=1e-4, X=1, dx=1e-2, normalize=False, sensing_loc="full", control_type="Dirchilet",
              sensing_type=None, limit_pde_state_size=True, max_state_value=1e10, max_control_value=20,
              control_sample_rate=0.1)
_dx512 = 1.0 / 512
_dt512 = 0.5 * _dx512

TRANSPORT_CASES = {"H1": dict(T_BASE)}
for _n, (_ct, _sl, _st) in {
    "neu_full": ("Neumann", "full", None), "neu_col": ("Neumann", "collocated", None),
    "neu_opp_neu": ("Neumann", "opposite", "Neumann"), "neu_opp_dir": ("Neumann", "opposite", "Dirchilet"),
    "dir_col": ("Dirchilet", "collocated", None), "dir_opp_neu": ("Dirchilet", "opposite", "Neumann"),
    "dir_opp_dir": ("Dirchilet", "opposite", "Dirchilet"),
}.items():
    TRANSPORT_CASES[f"H2_{_n}"] = dict(T_BASE, T=0.3, control_type=_ct, sensing_loc=_sl, sensing_type=_st,
                                       normalize=True, control_sample_rate=0.05)
TRANSPORT_CASES["H3"] = dict(T_BASE, T=700 * _dt512, dt=_dt512, dx=_dx512, control_sample_rate=100 * _dt512)
TRANSPORT_CASES["R_s30"] = dict(T_BASE, T=0.0400, dt=1e-4, control_sample_rate=30e-4)
TRANSPORT_CASES["R_s30_small"] = dict(T_BASE, T=0.0400, dt=1e-4, control_sample_rate=30e-4)
TRANSPORT_CASES["R_trunc"] = dict(T_BASE, T=1, max_state_value=46.5)
TRANSPORT_CASES["R_tiny"] = dict(T_BASE, T=0.0090, dt=1e-4, control_sample_rate=15e-4)

P_BASE = dict(T=1, dt=1e-5, X=1, dx=5e-3, normalize=False, sensing_loc="full", control_type="Dirchilet",
              sensing_type=None, limit_pde_state_size=True, max_state_value=1e10, max_control_value=20,
              control_sample_rate=1e-3)
_dx256 = 1.0 / 256
_dt256 = 0.25 * _dx256 * _dx256
PARABOLIC_CASES = {"P1": dict(P_BASE)}
for _n, _S, _ct, _norm in [("P2_s100", 100, "Dirchilet", False), ("P2_s1", 1, "Dirchilet", False),
                           ("P2_s100_neu", 100, "Neumann", False), ("P2_s1_neu", 1, "Neumann", False),
                           ("P2_s1_neu_norm", 1, "Neumann", True), ("P2_s100_dir_norm", 100, "Dirchilet", True)]:
    PARABOLIC_CASES[_n] = dict(P_BASE, T=1000 * _dt256, dt=_dt256, dx=_dx256, control_sample_rate=_S * _dt256,
                               control_type=_ct, normalize=_norm)
for _n, (_ct, _sl, _st) in {"col_neu": ("Neumann", "collocated", None), "col_dir": ("Dirchilet", "collocated", None),
                            "opp_neu": ("Dirchilet", "opposite", "Neumann")}.items():
    PARABOLIC_CASES[f"P3_{_n}"] = dict(P_BASE, T=600 * _dt256, dt=_dt256, dx=_dx256, control_sample_rate=50 * _dt256,
                                       control_type=_ct, sensing_loc=_sl, sensing_type=_st)

# ---- float64 beta / float64 or Python-scalar control inputs (the reference's mixed-precision arithmetic) -----------------
# name -> (kind, kwargs, action_as, beta_kind, steps); action_as in {"f32arr", "npf64", "pyfloat", "pyint"},
# beta_kind in {"ones64", "cos64", "int", "cos32"} (tests/golden/make_golden.py gen_mixed holds the data recipe)
_Q = dict(T=5, dt=1e-4, X=1, dx=1e-2, normalize=None, sensing_loc="full", control_type="Dirchilet", sensing_type=None,
          limit_pde_state_size=True, max_state_value=1e10, max_control_value=20, control_sample_rate=0.1)
MIXED_CASES = {
    # docs/source/guide/quickstart.rst:26-70 verbatim: beta = np.ones(nx) (float64), env.step(0), normalize=None
    "Q_quick": ("transport", dict(_Q), "pyint", "ones64", 51),
    "Q_t_b64_f32": ("transport", dict(T_BASE, T=0.5), "f32arr", "cos64", 6),
    "Q_t_b64_py_norm": ("transport", dict(T_BASE, T=0.5, normalize=True), "pyfloat", "cos64", 6),
    "Q_t_b64_neu_py_norm": ("transport", dict(T_BASE, T=0.3, control_type="Neumann", normalize=True, control_sample_rate=0.05), "pyfloat", "cos64", 6),
    "Q_t_b32_neu_np64_norm": ("transport", dict(T_BASE, T=0.3, control_type="Neumann", normalize=True, control_sample_rate=0.05), "npf64", "cos32", 6),
    "Q_t_b32_py_norm": ("transport", dict(T_BASE, T=0.5, normalize=True), "pyfloat", "cos32", 5),
    "Q_t_int_beta": ("transport", dict(T_BASE, T=0.3), "f32arr", "int", 4),
    "Q_t512_b64": ("transport", dict(T_BASE, T=700 * _dt512, dt=_dt512, dx=_dx512, control_sample_rate=100 * _dt512), "npf64", "cos64", 8),
    "Q_p_b64_f32": ("parabolic", dict(P_BASE, T=1000 * _dt256, dt=_dt256, dx=_dx256, control_sample_rate=100 * _dt256), "f32arr", "cos64", 8),
    "Q_p_b64_py_norm": ("parabolic", dict(P_BASE, T=1000 * _dt256, dt=_dt256, dx=_dx256, control_sample_rate=100 * _dt256, normalize=True), "pyfloat", "cos64", 8),
    "Q_p_b64_neu_py": ("parabolic", dict(P_BASE, T=1000 * _dt256, dt=_dt256, dx=_dx256, control_sample_rate=100 * _dt256, control_type="Neumann"), "pyfloat", "cos64", 8),
    "Q_p_b32_neu_np64": ("parabolic", dict(P_BASE, T=1000 * _dt256, dt=_dt256, dx=_dx256, control_sample_rate=50 * _dt256, control_type="Neumann"), "npf64", "cos32", 8),
    "Q_p_b64_neu_np64_norm": ("parabolic", dict(P_BASE, T=1000 * _dt256, dt=_dt256, dx=_dx256, control_sample_rate=1 * _dt256, control_type="Neumann", normalize=True), "npf64", "cos64", 6),
    "Q_p_b32_neu_py_norm": ("parabolic", dict(P_BASE, T=1000 * _dt256, dt=_dt256, dx=_dx256, control_sample_rate=1 * _dt256, control_type="Neumann", normalize=True), "pyfloat", "cos32", 6),
    "Q_p200_b64_col": ("parabolic", dict(P_BASE, T=0.02, sensing_loc="collocated"), "npf64", "cos64", 8),
}
ACTION_KIND = {"f32arr": "f32", "npf64": "f64", "pyfloat": "weak", "pyint": "weak"}

NS_BC = {"upper": ["Controllable", "Dirchilet"], "lower": ["Dirchilet", "Dirchilet"],
         "left": ["Dirchilet", "Dirchilet"], "right": ["Dirchilet", "Dirchilet"]}


def ns_bc_from_array(arr):
    arr = [str(x) for x in arr]
    return {k: [arr[2 * i], arr[2 * i + 1]] for i, k in enumerate(("upper", "lower", "left", "right"))}
