#!/bin/bash
# scratch: stand-alone kernel times (single stream) of the ab/ builds, then the usual A/B
cd "$(dirname "$0")/.."
cp ulc-codec_amd/libulc_amd.so /tmp/lib_keep.so
for f in ab/*.so; do cp $f ulc-codec_amd/libulc_amd.so; ULCX_ASYNC_FB=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --mode encode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$(basename $f .so) serial', 'step %.3f |' % (d['ms_per_step']), ' '.join('%s %.2f' % (n.split('.')[1][2:], v) for n, v in k.items() if v > 0.03))"; done
cp /tmp/lib_keep.so ulc-codec_amd/libulc_amd.so
bash tools/ab_all.sh
