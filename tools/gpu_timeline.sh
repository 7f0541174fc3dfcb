#!/bin/bash
# kernel timeline of the last bench step (async exact path as shipped)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
python tools/timeline.py $(find gpurun_out/tl -name "*kernel_trace.csv" | head -1) > gpurun_out/timeline.txt
