#!/bin/bash
# us per step of the NS workloads given as arguments (default: all of them)
for w in ${@:-ns2d_c5 ns2d_c5_f64 ns2d_c4 ns2d_c4_f64 ns2d_c4_b4096 ns2d_c4_f64_b4096}; do
python3 bench.py --steps 20 --warmup 5 --repeats 3 --no-also --no-cpu-baseline --workload $w 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w','us/step',round(d['ms_per_step']*1e3,1))"
done
