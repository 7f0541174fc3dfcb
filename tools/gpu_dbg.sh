#!/bin/bash
# scratch: CBR parity + timing of two builds
cd "$(dirname "$0")/.."
cp ulc-codec_amd/libulc_amd.so /tmp/lib_keep.so
for f in ab/*.so; do cp $f ulc-codec_amd/libulc_amd.so; echo "== $(basename $f) tests: $(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -1)"; done
for r in 1 2; do for f in ab/*.so; do cp $f ulc-codec_amd/libulc_amd.so; timeout 300 python bench.py --config cbr64_48k --streams 4096 --mode encode --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$(basename $f .so)', 'cbr encode step %.3f |' % (d['ms_per_step']), ' '.join('%s %.2f' % (n.split('.')[1][2:], v) for n, v in k.items() if v > 0.05))"; done; done
cp /tmp/lib_keep.so ulc-codec_amd/libulc_amd.so
