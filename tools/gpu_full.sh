#!/bin/bash
# round artefacts: smoke, GPU tests, bench with cpu baseline, rocprof kernel stats (same command), PMC traffic passes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.txt
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.txt
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
rm -rf gpurun_out/prof gpurun_out/pmc_rd gpurun_out/pmc_wr && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --no-cpu > gpurun_out/bench_prof.json 2> gpurun_out/prof.err
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats.csv
# HBM traffic: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_rd -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/pmc_rd.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_wr -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/pmc_wr.err
cp $(find gpurun_out/pmc_rd -name "*counter_collection.csv" | head -1) gpurun_out/pmc_fetch.csv
cp $(find gpurun_out/pmc_wr -name "*counter_collection.csv" | head -1) gpurun_out/pmc_write.csv
python tools/pmc_summary.py gpurun_out/pmc_fetch.csv gpurun_out/pmc_write.csv gpurun_out/pmc_summary.json > gpurun_out/pmc_summary.txt
