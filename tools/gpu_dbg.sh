#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "cbr or abr or config4 or tie" 2>&1 | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --config cbr64_48k --streams 4096 --mode encode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cbr encode ms', d['ms_per_step'], d['kernels_ms'])"
rm -rf gpurun_out/tl; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 1 --warmup 1 --no-cpu --config cbr64_48k --streams 4096 --mode encode > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/tl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last encode call: from the last k_wc_ef group
idx = max(i for i, r in enumerate(rows) if 'k_state_update' in r['Kernel_Name'])
# walk back to the first kernel of that call: the k_wc_ef before
start = max(i for i, r in enumerate(rows[:idx]) if 'k_wc_ef' in r['Kernel_Name'] and (i == 0 or 'k_wc' not in rows[i-1]['Kernel_Name'] and 'k_xf' not in rows[i-1]['Kernel_Name']))
t0 = int(rows[start]['Start_Timestamp'])
agg = collections.OrderedDict()
for r in rows[start:]:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:22]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
print('call span ms', (max(int(r['End_Timestamp']) for r in rows[start:]) - t0) / 1e6)
for n, (c, t) in agg.items(): print('%-24s x%-3d %.3f ms' % (n, c, t))
PY
