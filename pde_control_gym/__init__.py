"""Drop-in replacement for lukebhan/PDEControlGym's ``pde_control_gym`` package (MI355X-native engine).

Registers the reference's environment ids (pde_control_gym/__init__.py:3-18 there; the reference's file has
a syntax error at :11-14 that merges two ``register`` calls -- fixed here) when gymnasium is importable.
"""
