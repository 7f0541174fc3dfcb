#!/bin/bash
# scratch: per-launch kernel durations of one CBR encode call (65536 blocks)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp; rm -rf gpurun_out/cbr_tr
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cbr_tr -- python3 bench.py --config cbr64_48k --streams 4096 --mode encode --steps 1 --warmup 1 --no-cpu > gpurun_out/cbr_tr.txt 2>&1
f=$(find gpurun_out/cbr_tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r['Kernel_Name'].replace('void ','').startswith('k_')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last call: find last k_cplx
idx = max(i for i, r in enumerate(rows) if 'k_cplx' in r['Kernel_Name'])
t0 = int(rows[idx]['Start_Timestamp'])
out = []
for r in rows[idx:]:
    n = r['Kernel_Name'].replace('void ','').split('(')[0]
    out.append((n, (int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
for n, s, d in out:
    if d > 0.02: print("%-28s start %8.3f dur %7.3f" % (n[:28], s, d))
PY
