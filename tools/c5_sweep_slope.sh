#!/bin/bash
# per-sweep cost of the 256x256 slab pass: time the pass for several sweep counts K (slope = one sweep, intercept = load + store)
cd /tmp; export TMPDIR=/tmp
for K in 2 14 26 38 50; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5k_$K -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --repeats 1 --no-also --no-cpu-baseline --workload ns2d_c5 --substeps $K > /dev/null 2>&1
python3 - $GRAFT_REPO_ROOT/gpurun_out/c5k_$K/p_kernel_stats.csv $K <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'slab' in r['Name']: print('K',sys.argv[2], r['Name'].replace('(anonymous namespace)::','')[5:32], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
done
