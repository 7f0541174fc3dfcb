#!/bin/bash
# registers / scratch / LDS of every kernel of the objects of a build directory (default: ulc-codec_amd/build); optional filter
# usage: tools/kregs.sh [build dir] [pattern]
DIR=${1:-ulc-codec_amd/build}; PAT=${2:-.}
T=$(mktemp -d); trap 'rm -rf $T' EXIT
for o in $DIR/ulcx_enc_wc.o $DIR/ulcx_enc_xf.o $DIR/ulcx_enc_psy.o $DIR/ulcx_enc_wr.o $DIR/ulcx_dec.o; do
  [ -f $o ] || continue
  objcopy -O binary --only-section=.hip_fatbin $o $T/fat.bin
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | awk '
    /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {sg=$2} /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {l=$2}
    /\.wavefront_size:/ {printf "%-60s vgpr %3d sgpr %3d scratch %5d lds %6d\n", substr(name,1,60), v, sg, p, l}' | grep -E "$PAT" | sort
done
