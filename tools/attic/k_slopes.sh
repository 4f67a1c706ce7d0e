#!/bin/bash
# Split of an NS env-step into its sweep part and the rest: step time for several Jacobi sweep counts K (hipGraph replay,
# bench.py events) -> T(K) = intercept (predictor, boundary rules, corrector, observation, reward, loads/stores) + slope * K.
for wl in ns2d_c4 ns2d_c4_f64 ns2d_c5; do
  for K in 2 26 50 74 98; do
    python bench.py --steps 60 --warmup 10 --repeats 5 --no-also --no-cpu-baseline --workload $wl --substeps $K | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl K=%3d  %7.1f us per step of %d instances' % (d['config']['jacobi_sweeps_per_step'], d['roofline']['kernel_avg_ns']/1e3, d['config']['batch_per_gpu']))"
  done
done
