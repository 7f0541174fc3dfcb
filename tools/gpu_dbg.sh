#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "corrupt or packed" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "roundtrip or packed" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hand_assembled or packed" 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15
