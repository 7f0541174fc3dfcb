#!/bin/bash
# quick GPU iteration: parity tests + bench (no cpu baseline)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -25 > gpurun_out/pytest_gpu.txt
python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
