#!/bin/bash
# Round artefacts -> gpurun_out/$TAG (copy what is to be judged into profiles/): smoke, bench lines (headline, per mode,
# K = 32, the other configurations), rocprofv3 kernel stats of the headline command, PMC traffic passes of the same
# command (FETCH_SIZE / WRITE_SIZE in separate runs), SQ counter passes (single-stream mode: clean per-kernel numbers),
# one step's kernel timeline, the drop-in's single-stream rate.   usage: GIT_REV=$(git rev-parse --short HEAD) gpurun ... tools/gpu_r04.sh [tag]
# (the GPU box has no .git: the revision the numbers belong to is handed in)
cd "$(dirname "$0")/.."
TAG=${1:-r04}; O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --mode encode --no-cpu > $O/bench_encode.json 2>> $O/bench.err
timeout 600 python bench.py --mode decode --no-cpu > $O/bench_decode.json 2>> $O/bench.err
timeout 600 python bench.py --blocks 16 --no-cpu > $O/bench_blocks16.json 2>> $O/bench.err
timeout 600 python bench.py --config cbr64_48k --steps 3 --warmup 1 --no-cpu > $O/bench_cbr64_48k.json 2>> $O/bench.err
timeout 600 python bench.py --config wswitch_4096 --steps 3 --warmup 1 --no-cpu > $O/bench_wswitch_4096.json 2>> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu > $O/bench_under_rocprof.json 2> $O/prof.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_rd -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $O/pmc_rd.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_wr -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $O/pmc_wr.err
cp $(find $O/pmc_rd -name "*counter_collection.csv" | head -1) $O/pmc_fetch.csv
cp $(find $O/pmc_wr -name "*counter_collection.csv" | head -1) $O/pmc_write.csv
python tools/pmc_summary.py $O/pmc_fetch.csv $O/pmc_write.csv $O/pmc_summary.json config=vbr50 blocks=32 streams=4096 mode=both git=${GIT_REV:-unknown} > $O/pmc_summary.txt
timeout 600 python bench.py --no-cpu --pmc-summary $O/pmc_summary.json > $O/bench_with_traffic.json 2>> $O/bench.err
# SQ counters, everything on one stream (ULCX_ASYNC_FB=0 ULCX_WC_PIPE=1): two passes of 8 counters
export ULCX_ASYNC_FB=0 ULCX_WC_PIPE=1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $O/sq1.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $O/sq2.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu > /dev/null 2> $O/prof1.err
cp $(find $O/sq1 -name "*counter_collection.csv" | head -1) $O/sq_pass1.csv
cp $(find $O/sq2 -name "*counter_collection.csv" | head -1) $O/sq_pass2.csv
cp $(find $O/prof1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_single_stream.csv
unset ULCX_ASYNC_FB ULCX_WC_PIPE
python tools/bounds_table.py $O/sq_pass1.csv $O/sq_pass2.csv $O/kernel_stats_single_stream.csv $O/pmc_summary.json > $O/bounds.md 2> $O/bounds.err
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
python tools/timeline.py $(find $O/tl -name "*kernel_trace.csv" | head -1) > $O/timeline.txt
rm -rf $O/prof $O/prof1 $O/pmc_rd $O/pmc_wr $O/tl $O/sq1 $O/sq2
timeout 600 python tools/dropin_rate.py > $O/dropin_rate.txt 2>&1
ls -la $O; head -c 600 $O/bench.json; echo; tail -2 $O/smoke.txt; head -12 $O/pmc_summary.txt
# keep only this library's kernels in the counter files (the PyTorch kernels of the input generator are most of the rows)
python3 - $O/sq_pass1.csv $O/sq_pass2.csv <<'PY'
import csv, sys
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if r["Kernel_Name"].replace("void ", "").startswith("k_")]
    w = csv.DictWriter(open(f, "w", newline=""), fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
PY
