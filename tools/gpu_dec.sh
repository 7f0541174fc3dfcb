#!/bin/bash
# decoder iteration: decode-side parity tests, then a short bench (no cpu baseline)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "decode or roundtrip or packed or unusual or pcm16 or config" 2>&1 | tail -40 > gpurun_out/pytest_dec.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
tail -5 gpurun_out/pytest_dec.txt; tail -2 gpurun_out/bench_quick.err
python3 -c "
import json; d=json.load(open('gpurun_out/bench_quick.json')); print(d['ms_per_step'], d['whole_pipeline']); print({k:v for k,v in d['kernels_ms'].items() if k.startswith('dec')})"
