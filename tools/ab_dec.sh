#!/bin/bash
# A/B of decoder builds on one box: every ab/*.so in turn over the in-tree library (decode-only timing + output hash)
cd "$(dirname "$0")/.."
cp ulc-codec_amd/libulc_amd.so /tmp/lib_keep.so
for r in 1 2; do for f in ab/*.so; do cp $f ulc-codec_amd/libulc_amd.so; timeout 120 python tools/dec_bench.py $(basename $f .so) 2>&1 | grep -v amdgpu.ids | tail -1; done; done
cp /tmp/lib_keep.so ulc-codec_amd/libulc_amd.so
