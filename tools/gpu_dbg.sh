#!/bin/bash
# scratch: window-control step / transform chunk sweeps (environment switches)
cd "$(dirname "$0")/.."
run() { timeout 300 python bench.py --no-cpu --mode encode --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$1', 'ms_per_step %.3f' % d['ms_per_step'], 'xf %.2f exposed %.2f' % (k['enc.05_k_xf'] if 'enc.05_k_xf' in k else -1, [v for n,v in k.items() if 'exposed' in n][0]))"; }
for r in 1 2; do
unset ULCX_WC_STEPS ULCX_WC_PIPE; run "default(steps4,pipe4)"
ULCX_WC_STEPS=0 run "steps=chunks pipe4"
ULCX_WC_STEPS=0 ULCX_WC_PIPE=5 run "steps=chunks pipe5"
ULCX_WC_STEPS=0 ULCX_WC_PIPE=6 run "steps=chunks pipe6"
ULCX_WC_STEPS=8 ULCX_WC_PIPE=5 run "steps8 pipe5"
ULCX_WC_STEPS=5 ULCX_WC_PIPE=6 run "steps5 pipe6"
done
