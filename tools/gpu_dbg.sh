#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp; rm -rf gpurun_out/bk_tr
ULCX_ASYNC_FB=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bk_tr -- python3 bench.py --mode encode --steps 3 --warmup 1 --no-cpu > gpurun_out/bk_tr.txt 2>&1
f=$(find gpurun_out/bk_tr -name "*kernel_stats.csv" | head -1)
grep -i "bark\|nline" $f | awk -F, '{printf "%-50s calls %s avg %.3f ms\n", $1, $2, $4/1e6}'
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
for r in 1 2; do ULCX_BENCH_TIMING=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --mode encode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('step %.3f |' % (d['ms_per_step']), ' '.join('%s %.2f' % (n.split('.')[1][2:], v) for n, v in k.items() if v > 0.05))"; done
