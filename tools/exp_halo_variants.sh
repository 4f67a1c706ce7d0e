#!/bin/bash
# Round-6 bounded experiment: what does the per-sweep inter-wave halo exchange of ns_tile_step_f64 cost?
# Builds two TIMING-ONLY variants of the library from a patched copy of csrc/ (results are wrong by construction; nothing here ships):
#   nosync : halo_tb_f64 without its workgroup barrier (LDS write + read kept)      -> cost of the barrier
#   nohalo : halo_tb_f64 without LDS traffic and without the barrier               -> cost of the whole exchange = upper bound of what
#            ANY halo-blocking scheme (depth-2 temporal blocking halves the exchanges) can recover
# usage: bash tools/exp_halo_variants.sh   (then: bash tools/ab_libs.sh "ns2d_c4_f64_b4096 ns2d_c4_f64" - pdecontrolgym_amd/lib/ab/libnosync.so pdecontrolgym_amd/lib/ab/libnohalo.so)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
for v in nosync nohalo; do
  W=/tmp/pdegym_exp_$v
  rm -rf $W && mkdir -p $W && cp -r $R/pdecontrolgym_amd/csrc $W/csrc
  python3 - $W/csrc/pdegym_ns2d.hip $v <<'PY'
import sys
p, v = sys.argv[1], sys.argv[2]
s = open(p).read()
i = s.index("__device__ __forceinline__ void halo_tb_f64(")
j = s.index("template <int PR>", i)
body = s[i:j]
if v == "nosync":
    new = body.replace("__syncthreads();", "__builtin_amdgcn_wave_barrier();")
else:
    k = body.index("{", body.index("int ty)"))
    new = body[:k] + "{\n  ht[0] = top[0]; ht[1] = top[1]; hb[0] = bot[0]; hb[1] = bot[1]; ++xc; (void)lds; (void)tid; (void)ty;\n}\n\n"
assert new != body
open(p, "w").write(s[:i] + new + s[j:])
PY
  mkdir -p $R/pdecontrolgym_amd/lib/ab/$v
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -falign-loops=32 -fPIC -I$R/include -I$W/csrc"
  hipcc $FLAGS -c $W/csrc/pdegym_ns2d.hip -o $R/pdecontrolgym_amd/lib/ab/$v/pdegym_ns2d.o
  objs=""
  for f in pdegym_abi pdegym_1d pdegym_1d_rollout pdegym_ns256 pdegym_ns256_f64 pdegym_traffic pdegym_tumor pdegym_mlp; do objs="$objs $R/pdecontrolgym_amd/lib/$f.o"; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/pdecontrolgym_amd/lib/ab/lib$v.so $R/pdecontrolgym_amd/lib/ab/$v/pdegym_ns2d.o $objs
  rm -rf $R/pdecontrolgym_amd/lib/ab/$v $W
  echo built pdecontrolgym_amd/lib/ab/lib$v.so
done
