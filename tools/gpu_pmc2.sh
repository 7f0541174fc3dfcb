#!/bin/bash
# PMC pass with memory-instruction counters (own run, kernel-trace only), serial launch mode for clean per-kernel numbers
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmc && mkdir -p gpurun_out/pmc
export ULCX_ASYNC_FB=0 ULCX_WC_PIPE=1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu --streams ${1:-4096} --blocks 16 > gpurun_out/pmc_bench.json 2> gpurun_out/pmc.err
cp $(find gpurun_out/pmc -name "*counter_collection.csv" | head -1) gpurun_out/pmc_counters.csv
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/pmc_counters.csv')))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name'].replace('void ', '')
    if not n.startswith('k_'): continue
    k = n.split('(')[0]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    agg[k]['dur_us'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
out = open('gpurun_out/pmc2.txt', 'w')
for k,v in sorted(agg.items(), key=lambda kv: -sum(kv[1]['dur_us'])/len(kv[1]['dur_us'])):
    m = {c: sum(x)/len(x) for c,x in v.items()}
    w = max(m.get('SQ_WAVES',1),1)
    print(f"{k:22s} dur={m['dur_us']:8.1f}us waves={int(w):8d} valu/w={m.get('SQ_INSTS_VALU',0)/w:8.0f} salu/w={m.get('SQ_INSTS_SALU',0)/w:8.0f} vmemW/w={m.get('SQ_INSTS_VMEM_WR',0)/w:7.0f} vmemR/w={m.get('SQ_INSTS_VMEM_RD',0)/w:7.0f} wavecyc/w={m.get('SQ_WAVE_CYCLES',0)/w:9.0f} busy={m.get('SQ_BUSY_CYCLES',0):10.0f} waitinst/w={m.get('SQ_WAIT_INST_ANY',0)/w:9.0f}", file=out)
PY
