#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02a [workload ...]
# Writes gpurun_out/<tag>/... ; tools/summarize_profiles.py then distils the files that are committed under profiles/.
# Counters are collected in their own passes with --kernel-trace only (never combined with sys/hip traces).
TAG=${1:-rXX}
shift
WLS=${@:-parabolic_c2 parabolic_c2_s1 parabolic_c2_s1_open_loop_rollout parabolic_c2_s1_rollout parabolic_c2_policy_loop parabolic_c2_rollout parabolic_c2_policy_loop_256 parabolic_c2_rollout_256 parabolic_c2_open_loop_rollout transport_c3 burgers_c3 ns2d_c4 ns2d_c4_b4096 ns2d_c5 ns2d_c4_f64 ns2d_c4_f64_b4096 ns2d_c5_f64 ns2d_example traffic_arz traffic_arz_rollout brain_tumor}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
# float64 arithmetic occupies a SIMD for 4 cycles per wave-instruction (16 lanes per clock) against 2 for float32 / integer: counted apart
SQ3="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES"
for wl in $WLS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -o p -- python3 $R/bench.py --steps 200 --warmup 20 --repeats 2 --no-also --no-cpu-baseline --workload $wl > $OUT/stats_$wl.json 2> $OUT/stats_$wl.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${wl}_$c -o p -- python3 $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-also --no-cpu-baseline --workload $wl > /dev/null 2>&1
  done
  rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/pmc_sq1_$wl -o p -- python3 $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-also --no-cpu-baseline --workload $wl > /dev/null 2>&1
  rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/pmc_sq2_$wl -o p -- python3 $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-also --no-cpu-baseline --workload $wl > /dev/null 2>&1
  rocprofv3 --pmc $SQ3 --kernel-trace --output-format csv -d $OUT/pmc_sq3_$wl -o p -- python3 $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-also --no-cpu-baseline --workload $wl > /dev/null 2>&1
done
# gpurun merges at most 64 MiB back: the per-dispatch traces are not read by tools/summarize_profiles.py (it takes p_kernel_stats.csv and
# p_counter_collection.csv), so they stay on the box
find $OUT -name p_kernel_trace.csv -delete
find $OUT -name p_agent_info.csv -delete
cd $R
python3 tools/summarize_profiles.py $TAG
# what is judged travels home small: the summaries the line above wrote under profiles/ (which gpurun does not merge back) are copied
# into gpurun_out/<tag>_profiles/; the raw per-dispatch counter tables are dropped when they would push gpurun_out/ past the 64 MiB
# gpurun merges (round 6: the whole directory was refused once)
mkdir -p $R/gpurun_out/${TAG}_profiles
cp $R/profiles/${TAG}_* $R/profiles/counters_latest.json $R/gpurun_out/${TAG}_profiles/
if [ "$(du -sm $R/gpurun_out | cut -f1)" -gt 40 ]; then
  find $OUT -name p_counter_collection.csv -delete
  find $OUT -name '*.csv' -size +1M -delete
fi
du -sm $R/gpurun_out
