#!/bin/bash
cd "$(dirname "$0")/.."
for m in ${XF_MODES:-0 1 2 4 3 6 7}; do
  echo "skip=$m" ; ULCX_DBG_SKIP=$m ULCX_ASYNC_FB=0 python tools/fb_count.py 2>/dev/null | tail -1 | grep -o "'k_xf': [0-9.]*"
done
