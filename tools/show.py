import csv, json, sys
try:
    print(open('gpurun_out/pytest_gpu.txt').read().strip().splitlines()[-1])
    d = json.load(open('gpurun_out/bench_quick.json'))
    print("value %.0f Msamples/s  step %.2f ms  enc %.2f  dec %.2f" % (d['value'], d['ms_per_step'], d['whole_pipeline']['encode_ms'], d['whole_pipeline']['decode_ms']))
    rows = list(csv.reader(open('gpurun_out/kernel_stats.csv')))
    out = []
    for r in rows[1:40]:
        if 'k_' in r[0][:20] or 'k_select' in r[0]:
            out.append("%s=%.2f(x%s)" % (r[0].split('(')[0].replace('void ', ''), float(r[3]) / 1e6, r[1]))
    print("  ".join(out))
except Exception as e:
    print("show failed:", e)
