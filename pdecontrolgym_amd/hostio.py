"""Host <-> device hand-over for the batch-of-one faces: several small tensors per step, ONE stream synchronisation.

``t.cpu()`` / ``float(t[0])`` / ``torch.as_tensor(numpy_array, device=...)`` each go through pageable memory and block the host
until the copy has finished -- a single environment that reads seven scalars that way pays seven round trips per ``step()``.
``HostFetch`` copies any number of device tensors into pinned staging buffers asynchronously and synchronises once; ``PackLayout``
lays several small tensors out in ONE allocation so that "any number" becomes one copy; ``PinnedInputs`` holds per-call inputs in pinned
host memory that the kernels read in place (no upload).  (The transport / reaction-diffusion / Navier-Stokes single environments go one
step further: their kernels also WRITE their results into pinned host memory, ``PDEBatch1D.enable_host_io``.)
"""
from __future__ import annotations


class HostFetch:
    def __init__(self, device):
        import torch
        self.device = torch.device(device)
        self._pins = {}

    def __call__(self, tensors):
        """Device tensors -> NumPy views of pinned buffers (valid until the next call with a tensor of the same position, shape
        and dtype: copy what you keep)."""
        import torch
        if self.device.type != "cuda":
            return [t.detach().numpy() for t in tensors]
        out = []
        for i, t in enumerate(tensors):
            key = (i, tuple(t.shape), t.dtype)
            slot = self._pins.get(key)
            if slot is None:
                pin = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                slot = self._pins[key] = (pin, pin.numpy())
            slot[0].copy_(t, non_blocking=True)
            out.append(slot[1])
        torch.cuda.current_stream(self.device).synchronize()
        return out


class PackLayout:
    """Several small per-instance tensors as views of ONE allocation, so that a host-facing caller fetches all of them with a
    single device-to-host copy (``HostFetch``) and names them again on the host side with ``numpy_views``.  Segments start on
    64-byte boundaries."""

    _NP = {"float64": "f8", "float32": "f4", "int32": "i4", "uint8": "u1"}

    def __init__(self, spec):
        """spec: [(name, shape, torch dtype)]"""
        self.items, off = [], 0
        for name, shape, dtype in spec:
            n = 1
            for d in shape:
                n *= int(d)
            nbytes = n * {"float64": 8, "float32": 4, "int32": 4, "uint8": 1}[str(dtype).replace("torch.", "")]
            self.items.append((name, tuple(int(d) for d in shape), dtype, off, nbytes))
            off += (nbytes + 63) // 64 * 64
        self.nbytes = off

    def allocate(self, device):
        """(pack, {name: tensor view}) -- zero-filled."""
        import torch
        pack = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        return pack, {name: pack[off:off + nb].view(dtype).view(shape) for name, shape, dtype, off, nb in self.items}

    def numpy_views(self, raw):
        """The same names on a uint8 NumPy array holding a copy of the pack."""
        return {name: raw[off:off + nb].view(self._NP[str(dtype).replace("torch.", "")]).reshape(shape)
                for name, shape, dtype, off, nb in self.items}


def as_kernel_input(x, dtype, device, shape):
    """What an engine hands the kernels for a per-call input ``x``: a tensor of the right dtype and shape that is already
    device-accessible -- in HBM, or in PINNED host memory (read by the kernel over the bus: the batch-of-one faces) -- is used in
    place; anything else is copied to the device."""
    import torch
    if torch.is_tensor(x) and x.dtype == dtype and (x.device == device or (x.device.type == "cpu" and device.type == "cuda" and x.is_pinned())):
        return x.reshape(shape).contiguous()
    return torch.as_tensor(x, dtype=dtype, device=device).reshape(shape).contiguous()


class PinnedInputs:
    """Named pinned host tensors + their NumPy views: per-call inputs of a batch-of-one face, written on the host and read by the
    kernel in place (no upload)."""

    def __init__(self, device):
        import torch
        self.device = torch.device(device)
        self._slots = {}

    def __call__(self, name, array, dtype):
        import numpy as np
        import torch
        npdt = {torch.float64: np.float64, torch.float32: np.float32}[dtype]
        a = np.asarray(array, dtype=npdt)
        slot = self._slots.get((name, a.shape))
        if slot is None:
            t = torch.empty(a.shape, dtype=dtype, pin_memory=self.device.type == "cuda")
            slot = self._slots[(name, a.shape)] = (t, t.numpy())
        slot[1][...] = a
        return slot[0]
