cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rm -rf gpurun_out/pf; ULCX_WC_FUSE=1 ULCX_WC_PIPE=1 ULCX_ASYNC_FB=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf -- python3 bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/pf/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.reader(open(f)):
    if 'k_wc' in r[0]: print(r[0][:40], r[1], float(r[3]) / 1e3, 'us')
PY
