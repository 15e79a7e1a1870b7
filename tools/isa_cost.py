#!/usr/bin/env python3
"""dev aid: static issue-cost estimate of a kernel's ISA by instruction FORM (tools/ubench/valu_forms.hip, gfx950:
plain v_add/sub/mul/fma/fmac/fmamk/fmaak_f32 on VGPR or literal operands 2.2 cycles per wave-instruction per SIMD, the
same with an SGPR operand / DPP / SDWA / every other VALU operation 4.4, transcendentals 8.8, 4x4x1 MFMA 8,
16x16x4 MFMA 32).  usage: isa_cost.py file.s [first_line last_line]   (lines of the .s, default the whole file)"""
import re, sys, collections
FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32"}
TRANS = {"v_sqrt_f32", "v_log_f32", "v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32"}
def classify(line):
    t = line.split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        return None
    op = t.split()[0]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.startswith("v_mfma"):
        return ("mfma", 32.0 if "16x16x4" in op else 8.0, op)
    if op.startswith("v_"):
        if base in TRANS:
            return ("trans", 8.8, base)
        rest = t[len(op):]
        slow = op.endswith("_dpp") or op.endswith("_sdwa") or " row_" in t or "quad_perm" in t or "sdwa" in t
        # operands: an SGPR / vcc / exec source (destinations of compares excluded)
        ops = [o.strip() for o in rest.split(",")]
        srcs = ops[1:]
        sg = any(re.match(r"^-?\|?(s\d+|s\[\d+:\d+\]|vcc|exec|m0)", o) for o in srcs)
        if base in FAST and not slow and not sg:
            return ("fast", 2.2, base)
        if base in FAST:
            return ("fast-op slow-form", 4.4, base + (" sgpr" if sg else " dpp/sdwa"))
        return ("other valu", 4.4, base)
    if op.startswith("ds_"):
        return ("lds", 0.0, op)
    if op.startswith("s_"):
        return ("salu", 0.0, op)
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return ("vmem", 0.0, op)
    return ("?", 0.0, op)
def main():
    lines = open(sys.argv[1]).read().split("\n")
    a = int(sys.argv[2]) - 1 if len(sys.argv) > 2 else 0
    b = int(sys.argv[3]) if len(sys.argv) > 3 else len(lines)
    n = collections.Counter(); cyc = collections.Counter(); byop = collections.Counter(); cop = collections.Counter()
    for l in lines[a:b]:
        c = classify(l)
        if c is None: continue
        n[c[0]] += 1; cyc[c[0]] += c[1]
        if c[1] > 0: byop[(c[0], c[2])] += 1; cop[(c[0], c[2])] += c[1]
    tot = sum(cyc.values())
    print("lines %d-%d: %d instructions, estimated VALU issue %.0f cycles" % (a + 1, b, sum(n.values()), tot))
    for k, v in n.most_common():
        print("  %-20s %5d  %7.0f cycles" % (k, v, cyc[k]))
    print("  by operation (cycles):")
    for k, v in cop.most_common(28):
        print("    %-18s %-28s %4d  %6.0f" % (k[0], k[1], byop[k], v))
main()
