import numpy as np


def batch_space(space, n=1):
    from gymnasium.spaces import Box
    if isinstance(space, Box):
        return Box(np.stack([space.low] * n), np.stack([space.high] * n), dtype=space.dtype)
    raise TypeError(f"batch_space: unsupported space {type(space).__name__}")
