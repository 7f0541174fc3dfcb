# bench step time vs number of transform chunks (ULCX_WC_PIPE) and window-control steps (ULCX_WC_STEPS)
cd "$(dirname "$0")/.."
for p in 3 4 5 6 8; do for s in 8 16; do
  ULCX_WC_PIPE=$p ULCX_WC_STEPS=$s python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipe=$p steps=$s', round(d['ms_per_step'],3), round(d['whole_pipeline']['encode_ms'],3))"
done; done
