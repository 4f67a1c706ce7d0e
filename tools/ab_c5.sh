python -m pytest tests/test_gpu_ns2d.py -m gpu -x -q -k "256 or c5 or tiled" 2>&1 | tail -3
for v in 0 1 2; do
  PDEGYM_NS256_VARIANT=$v python -m pytest tests/test_gpu_ns2d.py -m gpu -x -q -k "256 or c5 or tiled" 2>&1 | tail -1
  echo -n "variant $v: "; PDEGYM_NS256_VARIANT=$v python bench.py --steps 50 --warmup 5 --repeats 3 --no-also --no-cpu-baseline --workload ns2d_c5 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['step_ms']*1e3,1),'us/step', round(d['value']/1e3,1),'k env-steps/s')"
done
echo -n "old pipeline: "; PDEGYM_NS256_OLD=1 python bench.py --steps 50 --warmup 5 --repeats 3 --no-also --no-cpu-baseline --workload ns2d_c5 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['step_ms']*1e3,1),'us/step', round(d['value']/1e3,1),'k env-steps/s')"
cd /tmp; export TMPDIR=/tmp
for v in 0 1 2; do
PDEGYM_NS256_VARIANT=$v rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5v$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --repeats 2 --no-also --no-cpu-baseline --workload ns2d_c5 > /dev/null 2>&1
python3 - $GRAFT_REPO_ROOT/gpurun_out/c5v$v/p_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'ns' in r['Name'] and 'native' not in r['Name']: print('  ', r['Name'].replace('(anonymous namespace)::','')[:50], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us', r['Percentage'])
PY
done
