#!/bin/bash
# first-contact GPU run: build is prebuilt in-tree; run the parity tests verbosely
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx9|Compute Unit" | head -6 > gpurun_out/rocminfo.txt 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.txt
