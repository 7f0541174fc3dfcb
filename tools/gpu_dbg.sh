#!/bin/bash
# scratch: parity + A/B of early k_nbark launches (environment switch, one box)
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2 3; do for v in 0 1; do
if [ $v = 1 ]; then export ULCX_NBARK_LATE=1; else unset ULCX_NBARK_LATE; fi
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; w=d['whole_pipeline']
print('late=$v', 'step %.3f enc %.3f dec %.3f |' % (d['ms_per_step'], w['encode_ms'], w['decode_ms']), ' '.join('%s %.2f' % (n.split('.')[1][2:], v) for n, v in k.items() if v > 0.05))"; done; done
