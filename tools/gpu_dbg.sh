#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python bench.py --mode decode --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('decode ms_per_step %.3f' % d['ms_per_step'], {k.split('.')[1]: round(v,3) for k,v in d['kernels_ms'].items()})"; done
timeout 300 python bench.py --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('both ms_per_step %.3f value %.0f' % (d['ms_per_step'], d['value']), d['whole_pipeline'])"
