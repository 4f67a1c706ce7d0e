"""pdecontrolgym_amd -- MI355X-native batched PDE-environment stepper.

Hand-written HIP kernels (csrc/) behind a C ABI (include/pdegym.h), and the Python host layer that
mirrors lukebhan/PDEControlGym's environment interface on top of them.  The drop-in package that user
code imports is ``pde_control_gym`` (same names as the reference); this package is the engine.
"""
__version__ = "0.1.0"
