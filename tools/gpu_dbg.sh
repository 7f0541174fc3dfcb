#!/bin/bash
# scratch: sequential vs overlapped steps
cd "$(dirname "$0")/.."
for r in 1 2; do for f in "" "--overlap"; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); w=d['whole_pipeline']
print('overlap=%s' % d['config']['overlap'], 'value %.0f step %.3f enc %.3f dec %.3f ok %s' % (d['value'], d['ms_per_step'], w['encode_ms'], w['decode_ms'], w['decode_ok']))"; done; done
