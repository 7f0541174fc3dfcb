#!/usr/bin/env python3
"""Builds profiles/<prefix>_pmc_summary.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

Per kernel and per bench step (a kernel launched several times per step, e.g. the chunked k_xf, is summed):
  hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE      (both counters are in KiB)
FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-byte
requests at 64 bytes for wide coalesced reads; other access widths are uncalibrated - the same guide).
The number of steps in a run = launches of k_state_update<float> (encoder) / k_dsyn (decoder)."""
import csv, json, sys, collections

def load(path, counter):
    tot = collections.Counter(); cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        tot[k] += float(r["Counter_Value"]) * 1024.0
        cnt[k] += 1
    return tot, cnt

def main():
    fetch_csv, write_csv, out = sys.argv[1], sys.argv[2], sys.argv[3]
    meta = dict(kv.split("=", 1) for kv in sys.argv[4:])          # config=vbr50 blocks=32 streams=4096 mode=both git=abc123: what the passes ran
    for k in ("blocks", "streams"):
        if k in meta: meta[k] = int(meta[k])
    f, fc = load(fetch_csv, "FETCH_SIZE")
    w, wc = load(write_csv, "WRITE_SIZE")
    steps_f = max(fc.get("k_state_update<float>", 0), 1); steps_w = max(wc.get("k_state_update<float>", 0), 1)
    res = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"): continue
        fb = f.get(k, 0.0) / steps_f; wb = w.get(k, 0.0) / steps_w
        res[k] = {"FETCH_SIZE_bytes_raw_per_step": fb, "WRITE_SIZE_bytes_per_step": wb,
                  "hbm_bytes_per_launch": 2.0 * fb + wb,
                  "launches_per_step": fc.get(k, 0) / steps_f,
                  "note": "per bench step (all launches of the kernel in one step summed); FETCH_SIZE doubled per MI355X_MICROARCH.md"}
    # stage-name aliases used by bench.py
    for alias, real in (("k_select", "k_select_wave<64, 11, 0>"), ("k_encode_wave", "k_encode_wave<true>"), ("k_xf", "k_xf<true, float>"),
                        ("k_wc_energy", "k_wc_energy<float>"), ("k_state_update", "k_state_update<float>"), ("k_dsyn", "k_dsyn<float, 16, true, false, 2048>"), ("k_dsyn", "k_dsyn<float, 16, true, true, 2048>"), ("k_wc_forward", "k_wc_ef<9, float>")):
        if real in res: res[alias] = res[real]                        # (k_dsyn: one workgroup per stream / a cut launch - whichever the batch took)
    if meta: res["_meta"] = meta
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(((k, v) for k, v in res.items() if k != "_meta"), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print("%-24s %8.1f MB/step  (fetch x2 %8.1f + write %8.1f)  launches/step %.1f" % (k, v["hbm_bytes_per_launch"] / 1e6, 2 * v["FETCH_SIZE_bytes_raw_per_step"] / 1e6, v["WRITE_SIZE_bytes_per_step"] / 1e6, v["launches_per_step"]))

if __name__ == "__main__":
    main()
