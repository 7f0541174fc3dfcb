#!/bin/bash
# scratch: decoder without the FIFO shuffle: parity, fuzz, A/B
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -2
bash tools/ab_dec.sh
timeout 600 python bench.py --config wswitch_4096 --steps 3 --warmup 1 --no-cpu --mode decode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('wswitch decode ms', d['ms_per_step'])"
