#!/usr/bin/env python3
"""Static instruction mix of one kernel of a `hipcc -S --cuda-device-only` dump, priced with the issue costs measured by
tools/ubench/valu_cost.hip (profiles/r05_valu_cost.md): per basic block (label) the count of full-rate / half-rate /
packed / transcendental vector instructions, scalar, LDS, memory instructions and waits.
usage: isa_mix.py dump.s <kernel-name-substring> [--blocks]"""
import re, sys, collections
FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_fma_f32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_mov_b32", "v_not_b32", "v_mac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
def cls(op):
    b = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith("_dpp") or op.endswith("_sdwa"): return "vhalf"
    if b.startswith("v_pk_"): return "vpk"
    if b in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"): return "vtrans"
    if b.startswith("v_") and b.endswith("f64"): return "vhalf"
    if b in FULL: return "vfull"
    if b.startswith("v_"): return "vhalf"
    if b.startswith("s_waitcnt"): return "wait"
    if b.startswith("s_barrier"): return "barrier"
    if b.startswith("s_load") or b.startswith("s_buffer_load"): return "smem"
    if b.startswith("s_"): return "salu"
    if b.startswith("ds_"): return "lds"
    if b.startswith("global_") or b.startswith("buffer_") or b.startswith("flat_") or b.startswith("scratch_"): return "vmem"
    return "other"
COST = {"vfull": 2.3, "vhalf": 4.2, "vpk": 5.6, "vtrans": 8.2}
def all_kernels(paths):
    """{demangled kernel name: {class: count}} for every kernel of the dumps (llvm-cxxfilt names, 'void ' and arguments stripped)."""
    import subprocess
    out = {}
    for path in paths:
        lines = open(path).read().split("\n")
        name, cnt = None, None
        for l in lines:
            t = l.strip()
            if re.match(r"^_Z[\w]+:", l) and "@" in l:
                name = l.split(":")[0]; cnt = collections.Counter(); continue
            if name is None: continue
            if t.startswith(".Lfunc_end"):
                out[name] = dict(cnt); name = None; continue
            if not t or t.startswith(";") or t.startswith(".") or re.match(r"^\.?LBB", t): continue
            cnt[cls(t.split()[0])] += 1
    names = list(out)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    res = {}
    for m, d in zip(names, dem):
        short = d.replace("void ", "").split("(")[0]
        c = out[m]
        v = sum(c.get(k, 0) for k in COST)
        if v == 0: continue
        res[short] = dict(c, vector=v, cycles_per_vector=sum(c.get(k, 0) * COST[k] for k in COST) / v)
    return res
def main():
    if sys.argv[1] == "--json":
        import json
        print(json.dumps({"_meta": {"cost_cycles": COST, "note": "static instruction mix per kernel (hipcc -S), priced with profiles/r05_valu_cost.md at >= 2 waves per SIMD"},
                          **all_kernels(sys.argv[2:])}, indent=1))
        return
    path, name = sys.argv[1], sys.argv[2]
    per_block = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().split(":")[0].endswith(name.split("$")[-1]) or (l.startswith("_Z") and name in l and ":" in l))
    tot = collections.Counter(); blocks = []; cur = ("entry", collections.Counter())
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"): break
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur); cur = (t.split(":")[0], collections.Counter()); continue
        if not t or t.startswith(";") or t.startswith("."): continue
        op = t.split()[0]
        c = cls(op); tot[c] += 1; cur[1][c] += 1; cur[1]["op:" + re.sub(r"_(e32|e64)$", "", op)] += 1
    blocks.append(cur)
    v = sum(tot[k] for k in COST)
    print("kernel %s: %d instructions; vector %d (full %d, half %d, packed %d, trans %d) = %.0f issue cycles at >= 2 waves/SIMD; salu %d lds %d vmem %d smem %d waits %d barriers %d" % (
        name, sum(tot.values()), v, tot["vfull"], tot["vhalf"], tot["vpk"], tot["vtrans"], sum(tot[k] * COST[k] for k in COST), tot["salu"], tot["lds"], tot["vmem"], tot["smem"], tot["wait"], tot["barrier"]))
    if per_block:
        for lab, c in blocks:
            n = sum(v for k, v in c.items() if not k.startswith("op:"))
            if n < 24: continue
            top = sorted(((v, k[3:]) for k, v in c.items() if k.startswith("op:")), reverse=True)[:6]
            print("  %-12s %5d  vfull %4d vhalf %4d vpk %4d salu %4d lds %4d vmem %3d wait %3d | %s" % (lab, n, c["vfull"], c["vhalf"], c["vpk"], c["salu"], c["lds"], c["vmem"], c["wait"], " ".join("%s:%d" % (k, v) for v, k in top)))
if __name__ == "__main__":
    main()
