#!/bin/bash
# Developer A/B inside ONE gpurun call (boxes differ in clocks): current library vs alternative builds under tools/tim/
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4g' % d['value'], '%.5f ms' % d['ms_per_step'])"; }
ALT=${ALT:-tools/tim/libpdegym_old.so}
for rep in 1 2; do
for w in ${WORKLOADS:-parabolic_c2 transport_c3}; do
    python bench.py --no-also --no-cpu-baseline --workload $w 2>/dev/null | line "new $w"
    for a in $ALT; do PDEGYM_LIB=$a python tools/ab_lib.py --no-also --no-cpu-baseline --workload $w 2>/dev/null | line "$a $w"; done
done
done
