#!/usr/bin/env python3
"""Per-kernel own-bound table (profiles/rNN_bounds.md) from two SQ counter passes, the kernel stats of a run with
everything on one stream, and the HBM traffic summary.

usage: bounds_table.py sq_pass1.csv sq_pass2.csv kernel_stats.csv pmc_summary.json > bounds.md

For every kernel of the step, per bench step (launches of one step summed):
  ms            device time alone on the GPU (kernel stats of the single-stream run; as shipped several overlap)
  hbm_ms        measured HBM bytes / 6.3 TB/s (the copy rate MI355X_MICROARCH.md measures; the 8 TB/s spec is the roofline's peak)
  valu_ms       cycles the vector units were occupied (SQ_ACTIVE_INST_VALU, quad-cycles x 4) / (1024 SIMDs x 2.4 GHz): the
                floor if every SIMD issued vector work back to back.  On these kernels that is 4.0 cycles per vector
                instruction (round 4: k_select_wave runs at 94 % of it, and no kernel here has ever run faster than it), not
                the 2 a wave64 instruction needs on a SIMD-32 on paper
  chain_ms      wave lifetime: SQ_WAVE_CYCLES per wave (quad-cycles x 4) / 2.4 GHz x waves per SIMD slot in sequence
                = what the kernel takes if it is bound by how long ONE wave lives (dependent chains, waits); for kernels with
                at most one wave per SIMD this is the kernel's time
  bound         the largest of the three; slack = ms / that
Counters are averaged over the launches of the last step in the file(s)."""
import csv
import collections
import json
import sys

CLK = 2.4e9
SIMDS = 1024
HBM_ACHIEVABLE = 6.3e12


def short(name):
    return name.replace("void ", "").split("(")[0]


def load_counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    sq1, sq2, stats, pmcj = sys.argv[1:5]
    a1, a2 = load_counters(sq1), load_counters(sq2)
    nsteps = max(len(a1.get("k_state_update<float>", {}).get("SQ_WAVES", [])), 1)
    dur = {}
    for r in csv.DictReader(open(stats)):
        k = short(r["Name"])
        if k.startswith("k_"):
            dur[k] = (float(r["TotalDurationNs"]), int(r["Calls"]))
    steps_stats = max(dur.get("k_state_update<float>", (0, 1))[1], 1)
    try:
        traffic = json.load(open(pmcj))
    except Exception:
        traffic = {}
    rows = []
    for k, (tot, calls) in dur.items():
        ms = tot / steps_stats / 1e6
        c = {}
        for src in (a1, a2):
            for name, vals in src.get(k, {}).items():
                c[name] = sum(vals) / nsteps              # per step
        waves = c.get("SQ_WAVES", 0.0)
        hbm = traffic.get(k, {}).get("hbm_bytes_per_launch")
        hbm_ms = hbm / HBM_ACHIEVABLE * 1e3 if hbm else None
        valu = c.get("SQ_INSTS_VALU", 0.0)
        valu_ms = (c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 or valu * 4.0) / (SIMDS * CLK) * 1e3
        wave_cyc = c.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / waves if waves else 0.0       # cycles one wave lives
        rounds = max(1.0, waves / calls * steps_stats / (SIMDS * 8.0)) if calls else 1.0   # wave slots: 8 per SIMD
        per_launch_chain = wave_cyc / CLK * 1e3 * rounds
        chain_ms = per_launch_chain * (calls / steps_stats)
        busy = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c.get("SQ_WAVE_CYCLES", 1.0) if c.get("SQ_WAVE_CYCLES") else 0.0
        wait = c.get("SQ_WAIT_ANY", 0.0) / c.get("SQ_WAVE_CYCLES", 1.0) if c.get("SQ_WAVE_CYCLES") else 0.0
        ldsc = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c.get("SQ_LDS_IDX_ACTIVE", 1.0) if c.get("SQ_LDS_IDX_ACTIVE") else 0.0
        cands = {"hbm": hbm_ms or 0.0, "valu issue": valu_ms, "wave lifetime": chain_ms if waves / max(calls / steps_stats, 1) <= SIMDS * 1.5 else 0.0}
        bname, bval = max(cands.items(), key=lambda kv: kv[1])
        rows.append((ms, k, calls / steps_stats, waves, valu / waves if waves else 0, c.get("SQ_INSTS_SALU", 0) / waves if waves else 0,
                     c.get("SQ_INSTS_LDS", 0) / waves if waves else 0, (c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)) / waves if waves else 0,
                     wave_cyc, busy, wait, ldsc, hbm, hbm_ms, valu_ms, chain_ms, bname, bval))
    rows.sort(reverse=True)
    print("| kernel | launches/step | ms/step alone | waves/step | VALU/wave | SALU/wave | LDS/wave | VMEM/wave | cycles a wave lives | VALU-active share of wave cycles | waiting share | LDS conflict / LDS active | HBM MB/step | hbm_ms | valu_ms | wave-lifetime ms | binding resource | bound ms | measured / bound |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for (ms, k, lps, waves, vw, sw, lw, mw, wc, busy, wait, ldsc, hbm, hbm_ms, valu_ms, chain_ms, bname, bval) in rows[:16]:
        print("| `%s` | %.0f | %.3f | %.0f | %.0f | %.0f | %.0f | %.0f | %.0f | %.2f | %.2f | %.2f | %s | %s | %.3f | %.3f | %s | %.3f | %s |" % (
            k, lps, ms, waves, vw, sw, lw, mw, wc, busy, wait, ldsc,
            ("%.0f" % (hbm / 1e6)) if hbm else "-", ("%.3f" % hbm_ms) if hbm_ms else "-", valu_ms, chain_ms, bname, bval,
            ("%.1fx" % (ms / bval)) if bval > 0 else "-"))


if __name__ == "__main__":
    main()
