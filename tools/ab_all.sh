#!/bin/bash
# A/B of library builds on one box: every ab/*.so in turn over the in-tree library: the parity tests once per build, then bench.py twice
cd "$(dirname "$0")/.."
cp ulc-codec_amd/libulc_amd.so /tmp/lib_keep.so
for f in ab/*.so; do cp $f ulc-codec_amd/libulc_amd.so; echo "== $(basename $f) tests: $(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1)"; done
for r in 1 2; do for f in ab/*.so; do cp $f ulc-codec_amd/libulc_amd.so; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; w=d['whole_pipeline']
print('$(basename $f .so)', 'step %.3f enc %.3f dec %.3f |' % (d['ms_per_step'], w['encode_ms'], w['decode_ms']), ' '.join('%s %.2f' % (n.split('.')[1][2:], v) for n, v in k.items() if v > 0.05))"; done; done
cp /tmp/lib_keep.so ulc-codec_amd/libulc_amd.so
