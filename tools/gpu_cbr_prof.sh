#!/bin/bash
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; rm -rf gpurun_out/cbrprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cbrprof -- python3 tools/cbr_time.py > gpurun_out/cbr_time.txt 2>&1
cp $(find gpurun_out/cbrprof -name "*kernel_stats.csv" | head -1) gpurun_out/cbr_kernel_stats.csv
