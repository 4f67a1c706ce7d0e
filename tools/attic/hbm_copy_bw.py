#!/usr/bin/env python3
"""Measured device-to-device copy bandwidth of the box (SURVEY.md section 8d asks for it next to the 8 TB/s spec figure)."""
import torch

for mb in (64, 512, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"copy {mb} MiB: {ms*1e3:.1f} us, read+write {2 * n * 4 / ms / 1e6:.0f} GB/s")
props = torch.cuda.get_device_properties(0)
print(props.name, "CUs", props.multi_processor_count, "mem GiB", props.total_memory // 2**30)
