"""Editable / in-tree install of the drop-in package (the reference ships `setup.py:1-8`: name "pdecontrolgym", no build step).

    pip install -e .          # puts this checkout on sys.path; `build_py` / `develop` / `editable_wheel` compile the gfx950 library in place
    python setup.py build_py  # the same build without installing anything
    python setup.py develop   # the legacy editable path (does not run build_py by itself: hooked below)

The HIP library stays IN THE TREE (pdecontrolgym_amd/lib/libpdegym_hip.so, next to its fingerprint stamp): `pdecontrolgym_amd._native`
loads it from there and refuses to run without it -- there is no CPU fallback.  `gymnasium` / `stable_baselines3` are optional at
import time (pde_control_gym/_compat.py) and therefore not install requirements here; torch (ROCm build) is the only hard dependency
besides NumPy.
"""
from setuptools import find_packages, setup
from setuptools.command.build_py import build_py


class BuildWithHip(build_py):
    def run(self):
        from pdecontrolgym_amd import build as hip_build      # hipcc --offload-arch=gfx950 (cross-compiles without a GPU)
        hip_build.build()
        super().run()


def _with_hip(base):
    class Cmd(base):
        def run(self):
            from pdecontrolgym_amd import build as hip_build
            hip_build.build()
            super().run()
    Cmd.__name__ = base.__name__
    return Cmd


_CMDS = {"build_py": BuildWithHip}
try:        # the editable paths do not go through build_py on every setuptools: hook them as well
    from setuptools.command.develop import develop
    _CMDS["develop"] = _with_hip(develop)
except ImportError:
    pass
try:
    from setuptools.command.editable_wheel import editable_wheel
    _CMDS["editable_wheel"] = _with_hip(editable_wheel)
except ImportError:
    pass

setup(
    name="pdecontrolgym",
    version="0.6.0",
    description="MI355X-native batched stepper behind the PDEControlGym Gymnasium API",
    packages=find_packages(include=["pde_control_gym*", "pdecontrolgym_amd*"]),
    package_data={"pdecontrolgym_amd": ["lib/*.so", "lib/*.stamp", "csrc/*"]},
    data_files=[("include", ["include/pdegym.h"])],
    install_requires=["numpy"],
    extras_require={"rl": ["gymnasium", "stable_baselines3"]},
    cmdclass=_CMDS,
)
