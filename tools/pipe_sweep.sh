for p in 1 2 3 4 5 6 8; do
  for r in 1 2; do
  ULCX_WC_PIPE=$p python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipe=$p', round(d['ms_per_step'],3), round(d['whole_pipeline']['encode_ms'],3))"
  done
done
