#!/bin/bash
# GPU session: smoke, tests, bench, rocprofv3 kernel trace of the same bench command
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.txt
python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/pytest_gpu.txt 2>&1
python bench.py --steps 5 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/bench_prof.json 2> gpurun_out/prof.err
find gpurun_out/prof -name "*stats*" | head
