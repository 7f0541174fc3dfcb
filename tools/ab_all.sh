#!/bin/bash
# A/B of library builds on ONE box (boxes of the pool differ by +-1.5 %): every ab/*.so in turn, selected through
# ULC_AMD_LIB (ulc_amd.py loads that file instead of the in-tree library; nothing is overwritten): the parity tests once
# per build, then bench.py twice per build, alternating.  AB_TESTS=0 skips the tests, AB_ARGS adds bench arguments.
cd "$(dirname "$0")/.."
show='
import json,sys
d=json.loads(sys.stdin.read()); k=d["kernels_ms"]; w=d["whole_pipeline"]
print(sys.argv[1], "step %.3f enc %.3f dec %.3f |" % (d["ms_per_step"], w["encode_ms"] or 0, w["decode_ms"] or 0), " ".join("%s %.2f" % (n.split(".")[1][2:], v) for n, v in k.items() if v > 0.05))'
if [ "${AB_TESTS:-1}" != 0 ]; then
  for f in ab/*.so; do echo "== $(basename $f) tests: $(ULC_AMD_LIB=$PWD/$f timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1)"; done
fi
for r in 1 2; do for f in ab/*.so; do
  ULC_AMD_LIB=$PWD/$f timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu $AB_ARGS 2>/dev/null | python -c "$show" "$(basename $f .so)"
done; done
