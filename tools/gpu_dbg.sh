#!/bin/bash
cd "$(dirname "$0")/.."
ULCX_ASYNC_FB=0 timeout 600 python bench.py --config wswitch_4096 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; w=d['whole_pipeline']
print('wswitch serial step %.3f enc %.3f dec %.3f |' % (d['ms_per_step'], w['encode_ms'], w['decode_ms']), ' '.join('%s %.2f' % (n.split('.')[1][2:], v) for n, v in k.items() if v > 0.05))"
ULCX_ASYNC_FB=0 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; w=d['whole_pipeline']
print('vbr50 serial step %.3f enc %.3f dec %.3f |' % (d['ms_per_step'], w['encode_ms'], w['decode_ms']), ' '.join('%s %.2f' % (n.split('.')[1][2:], v) for n, v in k.items() if v > 0.05))"
