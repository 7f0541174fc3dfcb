# parity tests, then three default bench runs (step / encode / decode ms): the quick check after a kernel change
cd "$(dirname "$0")/.."
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for r in 1 2 3; do python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['whole_pipeline']['encode_ms'],3), round(d['whole_pipeline']['decode_ms'],3))"; done
