cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q -k "rate_search or cbr or abr or CBR or straddle or exact or pcm16 or tools" 2>&1 | tail -3
show='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], "step %.2f" % d["ms_per_step"], {k.split(".")[1]:round(v,2) for k,v in d["kernels_ms"].items() if v>0.5})'
for r in 1 2; do for f in ab/e_slots.so ab/f_rate.so; do ULC_AMD_LIB=$PWD/$f timeout 300 python bench.py --config cbr64_48k --mode encode --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "$show" $(basename $f .so); done; done
timeout 300 python tools/fuzz_parity.py 150 31 2>&1 | tail -2
