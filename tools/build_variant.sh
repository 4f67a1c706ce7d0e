#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags]: builds pdecontrolgym_amd/lib/ab/lib<name>.so (A/B runs: tools/ab_libs.sh)
set -e
cd "$(dirname "$0")/../pdecontrolgym_amd"
name=$1; shift
mkdir -p lib/ab/$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -falign-loops=32 -fPIC -I../include -Icsrc"
pids=()
for f in pdegym_abi pdegym_1d pdegym_1d_rollout pdegym_ns2d pdegym_ns256 pdegym_ns256_f64 pdegym_traffic pdegym_tumor pdegym_mlp; do
  hipcc $FLAGS "$@" -c csrc/$f.hip -o lib/ab/$name/$f.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o lib/ab/lib$name.so lib/ab/$name/*.o
rm -rf lib/ab/$name
echo lib/ab/lib$name.so
