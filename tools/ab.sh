# A/B two builds of the library on the same box: ab/lib_A.so vs ab/lib_B.so (copied over the in-tree library in turn)
cd "$(dirname "$0")/.."
cp ulc-codec_amd/libulc_amd.so /tmp/lib_keep.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for r in 1 2 3; do for v in A B; do cp ab/lib_$v.so ulc-codec_amd/libulc_amd.so; python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$v', round(d['ms_per_step'],3), round(d['whole_pipeline']['encode_ms'],3), round(d['whole_pipeline']['decode_ms'],3), 'xf', round(k['enc.k_xf'],3), 'exposed', round(k['enc.wc_pipeline_exposed'],3), 'select', round(k['enc.k_select'],3))"; done; done
cp /tmp/lib_keep.so ulc-codec_amd/libulc_amd.so
