cd /root/repo
show='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], "step %.2f" % d["ms_per_step"], {k.split(".")[1]:round(v,2) for k,v in d["kernels_ms"].items() if v>0.5})'
for r in 1 2; do for f in ab/d_cbrwin.so ab/e_slots.so; do ULC_AMD_LIB=$PWD/$f timeout 300 python bench.py --config cbr64_48k --mode encode --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "$show" $(basename $f .so); done; done
