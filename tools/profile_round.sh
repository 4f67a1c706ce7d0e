#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01c
# Writes gpurun_out/<tag>/... ; tools/summarize_profiles.py then distils the files that are committed under profiles/.
# Counters are collected in their own passes with --kernel-trace only (never combined with sys/hip traces).
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for wl in parabolic_c2 transport_c3 burgers_c3 ns2d_c4 ns2d_c5 traffic_arz brain_tumor; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -o p -- python3 $R/bench.py --steps 200 --warmup 20 --no-also --no-cpu-baseline --workload $wl > $OUT/stats_$wl.json 2> $OUT/stats_$wl.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${wl}_$c -o p -- python3 $R/bench.py --steps 30 --warmup 5 --no-also --no-cpu-baseline --workload $wl > /dev/null 2>&1
  done
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq_parabolic_c2 -o p -- python3 $R/bench.py --steps 30 --warmup 5 --no-also --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq_ns2d_c4 -o p -- python3 $R/bench.py --steps 30 --warmup 5 --no-also --no-cpu-baseline --workload ns2d_c4 > /dev/null 2>&1
cd $R
python3 tools/summarize_profiles.py $TAG
