#!/usr/bin/env python3
"""Issue-cycle model of a gfx950 ISA listing (hipcc --save-temps .s), per line range.

Costs per wave64 instruction are the ones measured by scripts/ubench/ubench*.hip on an MI355X at
4 waves/SIMD (MEASUREMENTS.md 4.4): simple f32 / logic ops with VGPR operands ~2.9 cycles, with an SGPR or literal
operand 4.4-5.5, conversions / integer multiplies / packed ops ~4.6, transcendentals ~8.45.
Usage: isa_cost.py file.s start:end[:label[:weight]] ...
"""
import re, sys
FAST = {"v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_xor_b32", "v_and_b32", "v_or_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_fmamk_f32", "v_fmaak_f32", "v_bitop3_b32", "v_add_f32_e64"}
TRANS = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32"}
def cost(op, args):
    base = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", op)
    sg = bool(re.search(r"(^|[ ,\-|])s\d+|s\[\d+:\d+\]|0x[0-9a-f]{5,}", args))
    if base in TRANS: return 8.45, "trans"
    if base == "v_mad_u64_u32": return 4.85, "mad64"
    if base in ("v_fmamk_f32", "v_fmaak_f32"): return 2.85, "fast"
    if base in FAST:
        if not sg: return 2.9, "fast"
        if base in ("v_fma_f32", "v_fmac_f32"): return 5.5, "fast+sgpr"
        return 4.45, "fast+sgpr"
    if base.startswith("v_mfma"): return 0.0, "mfma"
    if base.startswith("v_"): return 4.6, "medium"
    if base.startswith("ds_"): return 0.0, "lds"
    if base.startswith(("global_", "flat_", "scratch_", "buffer_")): return 0.0, "vmem"
    return 0.0, "scalar"
def main():
    lines = open(sys.argv[1]).read().split("\n")
    for spec in sys.argv[2:]:
        p = spec.split(":")
        a, b = int(p[0]), int(p[1]); label = p[2] if len(p) > 2 else spec; w = float(p[3]) if len(p) > 3 else 1.0
        tot = {}; cnt = {}
        for ln in lines[a - 1:b]:
            ln = ln.split(";")[0].strip()
            if not ln or ln.startswith(".") or ln.endswith(":"): continue
            m = re.match(r"(\S+)\s*(.*)", ln)
            c, k = cost(m.group(1), m.group(2))
            tot[k] = tot.get(k, 0) + c; cnt[k] = cnt.get(k, 0) + 1
        cyc = sum(tot.values())
        print(f"{label:28s} x{w:<5g} VALU cycles {cyc:8.0f} -> {cyc*w:9.0f}   " +
              " ".join(f"{k}:{cnt[k]}({tot[k]:.0f})" for k in sorted(cnt)))
if __name__ == "__main__":
    main()
