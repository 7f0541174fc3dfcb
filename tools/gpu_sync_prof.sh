#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
export ULCX_ASYNC_FB=0
python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/bench_prof.json 2> gpurun_out/prof.err
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats.csv
