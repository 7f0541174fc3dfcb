#!/bin/bash
cd "$(dirname "$0")/.."
for r in 1 2 3; do for v in 0 1; do
if [ $v = 1 ]; then export ULCX_BENCH_TIMING=1; else unset ULCX_BENCH_TIMING; fi
timeout 300 python bench.py --no-cpu --steps 20 --warmup 3 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('events_in_timed_region=$v both ms_per_step %.3f value %.0f' % (d['ms_per_step'], d['value']), 'enc %.3f dec %.3f' % (d['whole_pipeline']['encode_ms'], d['whole_pipeline']['decode_ms']))"; done; done
