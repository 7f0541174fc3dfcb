#!/bin/bash
# k_encode_wave with/without the on-the-fly noise sums of phase F (ULCX_DBG_SKIP bit 0x40: measurement only, wrong output)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
for m in 0 64; do
  rm -rf gpurun_out/wp; ULCX_DBG_SKIP=$m ULCX_ASYNC_FB=0 ULCX_WC_PIPE=1 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/wp -- python3 bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
  python3 - "$m" <<'PY'
import csv, sys, glob, collections
f = glob.glob('gpurun_out/wp/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if 'k_encode_wave<true>' in r['Kernel_Name'] and int(r['Grid_Size']) > 1000000:
        agg[r['Counter_Name']] += float(r['Counter_Value'])
w = agg['SQ_WAVES'] or 1
kt = glob.glob('gpurun_out/wp/**/*kernel_trace.csv', recursive=True)
d = [ (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in csv.DictReader(open(kt[0])) if 'k_encode_wave<true>' in r['Kernel_Name'] and int(r['Grid_Size_X']) > 1000000 ]
print("skip=%s ms=%.3f valu/w=%.0f salu/w=%.0f lds/w=%.0f" % (sys.argv[1], sum(d)/max(1,len(d)), agg['SQ_INSTS_VALU']/w, agg['SQ_INSTS_SALU']/w, agg['SQ_INSTS_LDS']/w))
PY
done
