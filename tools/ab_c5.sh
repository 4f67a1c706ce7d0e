#!/bin/bash
# A/B inside ONE gpurun call (boxes differ in clocks): the 256x256 second-generation pipeline vs the round-1 pipeline
# (PDEGYM_NS256_OLD=1), parity tests first, then bench lines and the per-kernel times under rocprofv3.
python -m pytest tests/test_gpu_ns2d.py -m gpu -x -q -k "256 or c5 or tiled or full_size" 2>&1 | tail -2
for old in 0 1; do
  echo -n "PDEGYM_NS256_OLD=$old: "
  PDEGYM_NS256_OLD=$old python bench.py --steps 50 --warmup 5 --repeats 3 --no-also --no-cpu-baseline --workload ns2d_c5 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['step_ms']*1e3,1),'us/step', round(d['value']/1e3,1),'k env-steps/s')"
done
cd /tmp; export TMPDIR=/tmp
for old in 0 1; do
PDEGYM_NS256_OLD=$old rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5old$old -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --repeats 2 --no-also --no-cpu-baseline --workload ns2d_c5 > /dev/null 2>&1
python3 - $GRAFT_REPO_ROOT/gpurun_out/c5old$old/p_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'ns' in r['Name'] and 'native' not in r['Name']: print('  ', r['Name'].replace('(anonymous namespace)::','')[:50], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us', r['Percentage'])
PY
done
