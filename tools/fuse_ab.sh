# the fused envelope+forward kernel (ULCX_WC_FUSE=1) against the two kernels: parity tests, then bench pairs
cd "$(dirname "$0")/.."
ULCX_WC_FUSE=1 timeout 300 python -m pytest tests -m gpu -x -q -k "vbr or many or switches or pcm16 or unusual" 2>&1 | tail -3
for r in 1 2 3; do for v in 0 1; do ULCX_WC_FUSE=$v timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('fuse=$v', round(d['ms_per_step'],3), round(d['whole_pipeline']['encode_ms'],3), 'xf', round(k['enc.k_xf'],3), 'exposed', round(k['enc.wc_pipeline_exposed'],3), d['whole_pipeline']['decode_ok'])"; done; done
