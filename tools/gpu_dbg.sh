#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q -k "unusual or decode_block_reads" 2>&1 | grep -v "^  \|^$" | tail -25
