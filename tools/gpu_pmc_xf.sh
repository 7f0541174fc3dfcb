#!/bin/bash
# SQ counters of the encoder's transform kernel on the bench batch (two PMC passes, kernel-trace only, everything on one stream)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp ULCX_ASYNC_FB=0
rm -rf gpurun_out/pmcx1 gpurun_out/pmcx2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcx1 -- python3 bench.py --mode encode --steps 2 --warmup 1 --no-cpu > gpurun_out/pmcx1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmcx2 -- python3 bench.py --mode encode --steps 2 --warmup 1 --no-cpu > gpurun_out/pmcx2.txt 2>&1
python3 - <<'PY'
import csv, collections, glob
for d in ("pmcx1", "pmcx2"):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        n = r['Kernel_Name'].replace('void ', '')
        k = n.split('(')[0]
        if not (k.startswith('k_xf') or k.startswith('k_select_wave') or k.startswith('k_encode_wave<true') or k.startswith('k_gapsums')): continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[k]['dur_us'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k, v in agg.items():
        m = {c: sum(x[-4:])/len(x[-4:]) for c, x in v.items()}
        w = max(m.get('SQ_WAVES', 1), 1)
        print(d, k, " ".join(f"{c}={m[c]/w:.0f}/w" if c.startswith('SQ_') and c not in ('SQ_WAVES','SQ_BUSY_CYCLES') else f"{c}={m[c]:.0f}" for c in sorted(m)))
PY
