#!/bin/bash
cd "$(dirname "$0")/.."
for m in 0 1 2 4 3 6 5 7; do
  echo "skip=$m" ; ULCX_DBG_SKIP=$m python bench.py --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels_ms']['dec.k_dimdct'])"
done
