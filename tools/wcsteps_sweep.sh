# bench step time vs number of uniform window-control steps (ULCX_WC_STEPS; 0 = the transform's own chunks)
cd "$(dirname "$0")/.."
for s in 0 4 8 16; do for r in 1 2; do ULCX_WC_STEPS=$s python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wcsteps=$s', round(d['ms_per_step'],3), round(d['whole_pipeline']['encode_ms'],3))"; done; done
