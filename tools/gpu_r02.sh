#!/bin/bash
# round-2 artefacts: smoke, bench lines (headline, per mode, per config), rocprof kernel stats of the headline command,
# PMC traffic passes of the same command, kernel timeline.  Everything lands in gpurun_out/r02/.
cd "$(dirname "$0")/.."
O=gpurun_out/r02i; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --mode encode > $O/bench_encode.json 2>> $O/bench.err
python bench.py --mode decode > $O/bench_decode.json 2>> $O/bench.err
python bench.py --config cbr64_48k --steps 3 --warmup 1 > $O/bench_cbr64_48k.json 2>> $O/bench.err
python bench.py --config wswitch_4096 --steps 3 --warmup 1 > $O/bench_wswitch_4096.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu > $O/bench_under_rocprof.json 2> $O/prof.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
# HBM traffic: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_rd -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $O/pmc_rd.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_wr -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $O/pmc_wr.err
cp $(find $O/pmc_rd -name "*counter_collection.csv" | head -1) $O/pmc_fetch.csv
cp $(find $O/pmc_wr -name "*counter_collection.csv" | head -1) $O/pmc_write.csv
python tools/pmc_summary.py $O/pmc_fetch.csv $O/pmc_write.csv $O/pmc_summary.json > $O/pmc_summary.txt
python bench.py --no-cpu --pmc-summary $O/pmc_summary.json > $O/bench_with_traffic.json 2>> $O/bench.err
rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
python tools/timeline.py $(find $O/tl -name "*kernel_trace.csv" | head -1) > $O/timeline.txt
rm -rf $O/prof $O/pmc_rd $O/pmc_wr $O/tl
ls -la $O; head -c 400 $O/bench.json; echo; cat $O/smoke.txt | tail -2; head -12 $O/pmc_summary.txt
python tools/dropin_rate.py > $O/dropin_rate.txt 2>&1
