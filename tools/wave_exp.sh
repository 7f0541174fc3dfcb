#!/bin/bash
cd "$(dirname "$0")/.."
for m in 0 256 512 768 1024 1280; do
  echo "skip=$m" ; ULCX_DBG_SKIP=$m ULCX_ASYNC_FB=0 ULCX_WC_PIPE=1 python tools/fb_count.py 2>/dev/null | tail -1 | grep -o "'k_encode_wave': [0-9.]*"
done
