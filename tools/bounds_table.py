#!/usr/bin/env python3
"""Per-kernel own-bound table (profiles/rNN_bounds.md) from the SQ counter passes, the kernel stats of a run with everything on
one stream, the HBM traffic summary and the static instruction mix.

usage: bounds_table.py sq_pass1.csv sq_pass2.csv kernel_stats.csv pmc_summary.json [isa_mix.json] [sq_pass3.csv] > bounds.md

For every kernel of the step, per bench step (launches of one step summed):
  ms            device time alone on the GPU (kernel stats of the single-stream run; as shipped several overlap)
  hbm_ms        the kernel's MEASURED HBM bytes / 6.29 TB/s (the float4-copy rate MI355X_MICROARCH.md measures; the 8 TB/s spec is
                the roofline's peak): what its actual traffic costs at copy bandwidth
  valu_ms       vector instructions x the cycles a SIMD needs per instruction of THIS kernel's mix / (1024 SIMDs x 2.4 GHz).
                The price per instruction is measured (tools/ubench/valu_cost.hip, profiles/r05_valu_cost.md, >= 2 waves per
                SIMD): 2.3 full-rate (add / mul / fma / logic / mov), 4.2 half-rate (compares, selects, shifts, min / max, DPP,
                lane ops, conversions, binary64), 5.6 packed binary32, 8.2 transcendental; the mix is the kernel's static one
                (tools/isa_mix.py over `hipcc -S`).  (Round 4 charged 4.0 for everything, from SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU.)
  salu_ms       scalar instructions / (256 CUs x 0.90 per cycle x 2.4 GHz): a CU retires at most 0.90-0.95 scalar instructions per
                cycle whatever the number of waves (tools/ubench/issue_rate.hip)
  lanes         active-lane fraction of the vector instructions: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) (pass 3)
  chain_ms      wave lifetime: SQ_WAVE_CYCLES per wave (quad-cycles x 4) / 2.4 GHz x waves per SIMD slot in sequence
                = what the kernel takes if it is bound by how long ONE wave lives (dependent chains, waits); only for kernels
                with at most ~one wave per SIMD
  bound         the largest of the four; measured / bound = how far the kernel runs above it
Counters are averaged over the launches of the last step in the file(s)."""
import csv
import collections
import json
import sys

CLK = 2.4e9
SIMDS = 1024
CUS = 256
HBM_ACHIEVABLE = 6.29e12
SALU_PER_CYCLE_CU = 0.90
DEFAULT_VEC_CYCLES = 3.6


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
    mixj = sys.argv[5] if len(sys.argv) > 5 else None
    sq3 = sys.argv[6] if len(sys.argv) > 6 else None
    passes = [load_counters(sq1), load_counters(sq2)] + ([load_counters(sq3)] if sq3 else [])
    nsteps = max(len(passes[0].get("k_state_update<float>", {}).get("SQ_WAVES", [])), 1)
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
    try:
        mix = json.load(open(mixj)) if mixj else {}
    except Exception:
        mix = {}
    rows = []
    for k, (tot, calls) in dur.items():
        ms = tot / steps_stats / 1e6
        c = {}
        for src in passes:
            n = max(len(src.get("k_state_update<float>", {}).get("SQ_WAVES", [])), 1)
            for name, vals in src.get(k, {}).items():
                c.setdefault(name, sum(vals) / n)         # per step
        waves = c.get("SQ_WAVES", 0.0)
        hbm = traffic.get(k, {}).get("hbm_bytes_per_launch")
        hbm_ms = hbm / HBM_ACHIEVABLE * 1e3 if hbm else None
        valu = c.get("SQ_INSTS_VALU", 0.0)
        cyc = mix.get(k, {}).get("cycles_per_vector", DEFAULT_VEC_CYCLES)
        valu_ms = valu * cyc / (SIMDS * CLK) * 1e3
        salu = c.get("SQ_INSTS_SALU", 0.0)
        salu_ms = salu / (CUS * SALU_PER_CYCLE_CU * CLK) * 1e3
        act = c.get("SQ_ACTIVE_INST_VALU", 0.0)
        lanes = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * act) if act and c.get("SQ_THREAD_CYCLES_VALU") else None
        wave_cyc = c.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / waves if waves else 0.0       # cycles one wave lives
        rounds = max(1.0, waves / calls * steps_stats / (SIMDS * 8.0)) if calls else 1.0   # wave slots: 8 per SIMD
        chain_ms = wave_cyc / CLK * 1e3 * rounds * (calls / steps_stats)
        wait = c.get("SQ_WAIT_ANY", 0.0) / c.get("SQ_WAVE_CYCLES", 1.0) if c.get("SQ_WAVE_CYCLES") else 0.0
        ldsc = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c.get("SQ_LDS_IDX_ACTIVE", 1.0) if c.get("SQ_LDS_IDX_ACTIVE") else 0.0
        cands = {"hbm": hbm_ms or 0.0, "valu issue": valu_ms, "salu issue": salu_ms,
                 "wave lifetime": chain_ms if waves / max(calls / steps_stats, 1) <= SIMDS * 1.5 else 0.0}
        bname, bval = max(cands.items(), key=lambda kv: kv[1])
        rows.append((ms, k, calls / steps_stats, waves, valu / waves if waves else 0, salu / waves if waves else 0,
                     c.get("SQ_INSTS_LDS", 0) / waves if waves else 0, (c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)) / waves if waves else 0,
                     wave_cyc, cyc, lanes, wait, ldsc, hbm, hbm_ms, valu_ms, salu_ms, chain_ms, bname, bval))
    rows.sort(reverse=True)
    print("| kernel | launches/step | ms/step alone | waves/step | VALU/wave | SALU/wave | LDS/wave | VMEM/wave | cycles a wave lives | cycles per vector instruction (mix) | active lanes | waiting share | LDS conflict / LDS active | HBM MB/step | hbm_ms | valu_ms | salu_ms | wave-lifetime ms | binding resource | bound ms | measured / bound |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for (ms, k, lps, waves, vw, sw, lw, mw, wc, cyc, lanes, wait, ldsc, hbm, hbm_ms, valu_ms, salu_ms, chain_ms, bname, bval) in rows[:18]:
        print("| `%s` | %.0f | %.3f | %.0f | %.0f | %.0f | %.0f | %.0f | %.0f | %.2f | %s | %.2f | %.2f | %s | %s | %.3f | %.3f | %.3f | %s | %.3f | %s |" % (
            k, lps, ms, waves, vw, sw, lw, mw, wc, cyc, ("%.2f" % lanes) if lanes is not None else "-", wait, ldsc,
            ("%.0f" % (hbm / 1e6)) if hbm else "-", ("%.3f" % hbm_ms) if hbm_ms else "-", valu_ms, salu_ms, chain_ms, bname, bval,
            ("%.1fx" % (ms / bval)) if bval > 0 else "-"))


if __name__ == "__main__":
    main()
