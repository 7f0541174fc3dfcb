#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -6
