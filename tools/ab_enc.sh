#!/bin/bash
# A/B of encoder builds on ONE box: every ab/*.so in turn through ULC_AMD_LIB, the encode-only timing of the bench batch
# (4096 streams x 32 blocks) twice per build, alternating; the md5 of slots + sizes must agree across builds.
cd "$(dirname "$0")/.."
for r in 1 2; do for f in ab/*.so; do
  ULC_AMD_LIB=$PWD/$f timeout 300 python tools/enc_bench.py "$(basename $f .so)" ${AB_K:-32} 2>/dev/null | tail -1
done; done
