#!/bin/bash
# Round artefacts -> gpurun_out/$TAG (copy what is to be judged into profiles/): smoke, bench lines (headline, per mode, K = 16,
# the other configurations), rocprofv3 kernel stats of the headline command, PMC traffic passes of the same command (FETCH_SIZE /
# WRITE_SIZE in separate runs), SQ counter passes (single-stream mode: clean per-kernel numbers), one step's kernel timeline, the
# drop-in's single-stream rate, the issue-rate microbenchmarks.
#   usage: GIT_REV=$(git rev-parse --short HEAD) gpurun ... tools/gpu_r06.sh [tag]
# (the GPU box has no .git: the commit the numbers belong to is handed in; the SOURCE revision comes from the library itself)
# Every step records its exit code in $O/steps.txt; a missing required artefact makes the script exit non-zero.
cd "$(dirname "$0")/.."
TAG=${1:-r06}; O=gpurun_out/$TAG; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
FAIL=0
step() { local name=$1; shift; "$@"; local rc=$?; echo "$name rc=$rc" >> "$O/steps.txt"; [ $rc -ne 0 ] && FAIL=1; return $rc; }
need() { if [ ! -s "$1" ]; then echo "MISSING $1" >> "$O/steps.txt"; FAIL=1; return 1; fi; return 0; }
first() { find "$1" -name "$2" 2>/dev/null | head -1; }
copy_first() { local f; f=$(first "$1" "$2"); if [ -n "$f" ]; then cp "$f" "$3"; else echo "MISSING $2 under $1" >> "$O/steps.txt"; FAIL=1; fi; }
SRC_REV=$(python -c "import sys; sys.path.insert(0, 'ulc-codec_amd'); import ulc_amd; print(ulc_amd.build_rev())")
echo "src_rev=$SRC_REV git=${GIT_REV:-unknown}" > "$O/steps.txt"

# the issue-rate microbenchmarks are sources only in the tree: build them here (a fresh checkout has no binaries)
for u in issue_rate valu_cost dep_chain; do
  step build_$u bash -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o tools/ubench/$u > $O/build_$u.log 2>&1"
done
step smoke bash -c "timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.txt 2>&1"
step bench bash -c "timeout 900 python bench.py > $O/bench.json 2> $O/bench.err"; need $O/bench.json
step bench_encode bash -c "timeout 600 python bench.py --mode encode --no-cpu > $O/bench_encode.json 2>> $O/bench.err"
step bench_decode bash -c "timeout 600 python bench.py --mode decode --no-cpu > $O/bench_decode.json 2>> $O/bench.err"
step bench_blocks16 bash -c "timeout 600 python bench.py --blocks 16 --no-cpu > $O/bench_blocks16.json 2>> $O/bench.err"
step bench_cbr bash -c "timeout 600 python bench.py --config cbr64_48k --steps 3 --warmup 1 --no-cpu > $O/bench_cbr64_48k.json 2>> $O/bench.err"
step bench_wswitch bash -c "timeout 600 python bench.py --config wswitch_4096 --steps 3 --warmup 1 --no-cpu > $O/bench_wswitch_4096.json 2>> $O/bench.err"
step prof bash -c "timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu --no-secondary > $O/bench_under_rocprof.json 2> $O/prof.err"
copy_first $O/prof "*kernel_stats.csv" $O/kernel_stats.csv
step pmc_rd bash -c "timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_rd -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $O/pmc_rd.err"
step pmc_wr bash -c "timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_wr -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $O/pmc_wr.err"
copy_first $O/pmc_rd "*counter_collection.csv" $O/pmc_fetch.csv
copy_first $O/pmc_wr "*counter_collection.csv" $O/pmc_write.csv
if need $O/pmc_fetch.csv && need $O/pmc_write.csv; then
  step pmc_summary bash -c "python tools/pmc_summary.py $O/pmc_fetch.csv $O/pmc_write.csv $O/pmc_summary.json config=vbr50 blocks=32 streams=4096 mode=both git=${GIT_REV:-unknown} src_rev=$SRC_REV > $O/pmc_summary.txt"
  step bench_traffic bash -c "timeout 600 python bench.py --no-cpu --no-secondary --pmc-summary $O/pmc_summary.json > $O/bench_with_traffic.json 2>> $O/bench.err"
fi
# SQ counters, everything on one stream (ULCX_ASYNC_FB=0 ULCX_WC_PIPE=1): three passes
export ULCX_ASYNC_FB=0 ULCX_WC_PIPE=1
step sq1 bash -c "timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $O/sq1.err"
step sq2 bash -c "timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $O/sq2.err"
step sq3 bash -c "timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $O/sq3 -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2> $O/sq3.err"
step prof1 bash -c "timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-secondary > /dev/null 2> $O/prof1.err"
copy_first $O/sq1 "*counter_collection.csv" $O/sq_pass1.csv
copy_first $O/sq2 "*counter_collection.csv" $O/sq_pass2.csv
copy_first $O/sq3 "*counter_collection.csv" $O/sq_pass3.csv
copy_first $O/prof1 "*kernel_stats.csv" $O/kernel_stats_single_stream.csv
unset ULCX_ASYNC_FB ULCX_WC_PIPE
# keep only this library's kernels in the counter files (the PyTorch kernels of the input generator are most of the rows)
python3 - $O/sq_pass1.csv $O/sq_pass2.csv $O/sq_pass3.csv <<'PY'
import csv, os, sys
for f in sys.argv[1:]:
    if not os.path.exists(f): continue
    rows = list(csv.DictReader(open(f)))
    if not rows: continue
    keep = [r for r in rows if r["Kernel_Name"].replace("void ", "").startswith("k_")]
    w = csv.DictWriter(open(f, "w", newline=""), fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
PY
if need $O/sq_pass1.csv && need $O/sq_pass2.csv && need $O/kernel_stats_single_stream.csv; then
  S3=""; [ -s $O/sq_pass3.csv ] && S3=$O/sq_pass3.csv
  step bounds bash -c "python tools/bounds_table.py $O/sq_pass1.csv $O/sq_pass2.csv $O/kernel_stats_single_stream.csv $O/pmc_summary.json profiles/r05_isa_mix.json $S3 > $O/bounds.md 2> $O/bounds.err"
fi
step timeline_run bash -c "timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1"
TL=$(first $O/tl "*kernel_trace.csv"); if [ -n "$TL" ]; then python tools/timeline.py "$TL" > $O/timeline.txt; else echo "MISSING kernel_trace.csv" >> $O/steps.txt; FAIL=1; fi
rm -rf $O/prof $O/prof1 $O/pmc_rd $O/pmc_wr $O/tl $O/sq1 $O/sq2 $O/sq3
step dropin_rate bash -c "timeout 600 python tools/dropin_rate.py > $O/dropin_rate.txt 2>&1"
step issue_rate bash -c "timeout 300 tools/ubench/issue_rate > $O/issue_rate.md 2>&1"
step valu_cost bash -c "timeout 400 tools/ubench/valu_cost > $O/valu_cost.md 2>&1"
step dep_chain bash -c "timeout 120 tools/ubench/dep_chain > $O/dep_chain.txt 2>&1"
ls -la $O; cat $O/steps.txt; head -c 700 $O/bench.json; echo; tail -2 $O/smoke.txt; head -12 $O/pmc_summary.txt
exit $FAIL
