#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/dsyn_stamps.py 256 2>&1 | grep -v amdgpu.ids
