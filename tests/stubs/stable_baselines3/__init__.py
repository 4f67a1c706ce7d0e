"""Stand-in for stable_baselines3 (see tests/stubs/README.md)."""
__version__ = "0.0+contract-stub"
