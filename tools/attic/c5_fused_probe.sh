#!/bin/bash
# fused 256x256 step: launch time against the sweep count K (slope = one sweep, intercept = front + back) and the batch
for B in 256 512; do for K in 1 2 26 50 98; do
python3 bench.py --steps 20 --warmup 3 --repeats 3 --no-also --no-cpu-baseline --workload ns2d_c5 --substeps $K --batch $B 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B',$B,'K',$K,'us/step',round(d['ms_per_step']*1e3,1))"
done; done
