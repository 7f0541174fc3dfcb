#!/bin/bash
cd "$(dirname "$0")/.."
ULCX_NBARK_ASIDE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
for r in 1 2 3; do for v in 0 1; do
if [ $v = 1 ]; then export ULCX_NBARK_ASIDE=1; else unset ULCX_NBARK_ASIDE; fi
ULCX_BENCH_TIMING=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --mode encode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('nbarkAside=$v', 'step %.3f |' % (d['ms_per_step']), ' '.join('%s %.2f' % (n.split('.')[1][2:], v) for n, v in k.items() if v > 0.05))"; done; done
