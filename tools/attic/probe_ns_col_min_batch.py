import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench_ns_example as bx
bx._dbg("DEBUG_NS_COL_MIN_BATCH", 0)
for B in (1, 3, 16, 64, 128, 256, 512, 768):
    a, b = bx.run(B, False), bx.run(B, True)
    print(f"21x21 K=2000 float64 B={B}: column kernel (forced) {a*1e3:.3f} ms/step | workgroup kernel {b*1e3:.3f} ms/step", flush=True)
