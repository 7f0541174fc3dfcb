#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "^  \|^$" | tail -30
