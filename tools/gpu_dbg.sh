#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r02d2; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --mode encode > $O/bench_encode.json 2>> $O/bench.err
python bench.py --no-cpu --pmc-summary profiles/r02_d_pmc_summary.json > $O/bench_with_traffic.json 2>> $O/bench.err
python bench.py --config wswitch_4096 --steps 3 --warmup 1 > $O/bench_wswitch_4096.json 2>> $O/bench.err
head -c 300 $O/bench.json; tail -2 $O/bench.err
