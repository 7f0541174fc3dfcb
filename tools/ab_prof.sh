# usage: bash tools/ab_prof.sh "<bench args>" "<kernel regex>"  -- per-kernel average durations under rocprofv3 for every ab/*.so
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
ARGS=${1:---mode decode}; PAT=${2:-k_d}
for f in ab/*.so; do
  n=$(basename $f .so); rm -rf /tmp/prof_$n
  ULC_AMD_LIB=$PWD/$f timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o x --output-format csv -- python bench.py $ARGS --steps ${AB_STEPS:-10} --warmup ${AB_WARM:-2} --no-cpu > /tmp/prof_$n.log 2>&1
  echo "== $n: $(grep -o '"ms_per_step": [0-9.]*' /tmp/prof_$n.log | head -1)"
  python - "$n" "$PAT" <<'PY'
import csv,glob,sys,re
n,pat=sys.argv[1],sys.argv[2]
fs=glob.glob(f"/tmp/prof_{n}/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(fs[0])):
    if re.search(pat, r["Name"]): print("   %-60s calls %5s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
