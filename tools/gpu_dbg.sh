#!/bin/bash
cd "$(dirname "$0")/.."
for r in 1 2 3; do for v in 0 1; do
if [ $v = 1 ]; then export ULCX_NOISE_LOWPRI=1; else unset ULCX_NOISE_LOWPRI; fi
ULCX_BENCH_TIMING=1 timeout 300 python bench.py --no-cpu --steps 20 --warmup 3 --mode encode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('lowpri=$v ms_per_step %.3f |' % d['ms_per_step'], ' '.join('%s %.2f' % (n.split('.')[1][2:], v) for n, v in k.items() if v > 0.05))"; done; done
