#!/bin/bash
# SQ counters of the decoder kernels on the bench batch (two PMC passes, kernel-trace only)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmcd1 gpurun_out/pmcd2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcd1 -- python3 tools/dec_bench.py > gpurun_out/pmcd1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmcd2 -- python3 tools/dec_bench.py > gpurun_out/pmcd2.txt 2>&1
python3 - <<'PY'
import csv, collections, glob
for d in ("pmcd1", "pmcd2"):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        n = r['Kernel_Name'].replace('void ', '')
        if not n.startswith('k_d'): continue
        k = n.split('(')[0]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[k]['dur_us'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
        agg[k]['vgpr'].append(int(r['VGPR_Count'])); agg[k]['lds'].append(int(r['LDS_Block_Size'])); agg[k]['scr'].append(int(r.get('Scratch_Size', 0) or 0))
    for k, v in agg.items():
        m = {c: sum(x[-3:])/len(x[-3:]) for c, x in v.items()}
        w = max(m.get('SQ_WAVES', 1), 1)
        print(d, k, " ".join(f"{c}={m[c]/w:.0f}/w" if c.startswith('SQ_') and c not in ('SQ_WAVES','SQ_BUSY_CYCLES') else f"{c}={m[c]:.0f}" for c in sorted(m)))
PY
