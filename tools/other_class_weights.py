"""Prices the VALU instructions that no PMC instruction class counts ("other" = SQ_INSTS_VALU minus the ADD/MUL/FMA/TRANS_F64, *_F32, INT32,
INT64 and CVT counters): selects, moves, compares, lane reads, the helpers of the division sequence. Their share of each kernel's STATIC
instruction mix (hipcc --save-temps assembly of msdf_capi.hip) is weighted with the cycles per opcode measured by tools/valu_calib.hip
(profiles/r03_valu_calibration.json: opcodes_outside_the_pmc_classes) -> profiles/r03_other_class_weights.json, read by tools/pmc_report.py.
Static, not dynamic: the weight only has to say where between 2.2 (v_cndmask vcc) and 4.4 (v_cmp_f64, readlane) the mix of a kernel sits --
it comes out at 3.6-3.7 cycles for every kernel of this library.

    python tools/other_class_weights.py [tag]        (writes profiles/<tag>_other_class_weights.json, default tag r03)
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

W_OTHER = {"v_cndmask_b32_e32": 2.25, "v_cndmask_b32_e64": 4.19, "v_mov_b32_e32": 2.38, "v_mov_b64_e32": 4.30, "v_cmp": 4.40, "v_readlane_b32": 4.42,
           "v_writelane_b32": 4.42, "v_readfirstlane_b32": 4.42, "v_div_scale_f64": 4.53, "v_div_fixup_f64": 4.27, "v_div_fmas_f64": 4.51, "v_ldexp_f64": 4.22,
           "v_max_f64": 4.29, "v_min_f64": 4.29, "v_frexp": 4.33, "dpp": 4.20, "v_bfrev_b32_e32": 2.38, "v_xor_b32_e32": 2.5, "v_and_b32_e32": 2.5, "v_or_b32_e32": 2.5,
           "v_lshl": 2.5}
KERNELS = (("k_distanceILi3ELb1ELb0E", "k_distance<3,true,false>"), ("k_distanceILi3ELb0ELb0E", "k_distance<3,false,false>"),
           ("k_distanceILi3ELb1ELb1E", "k_distance<3,true,true>"), ("k_ec_fastILi3E", "k_ec_fast<3>"), ("k_ec_queryILi3ELb1E", "k_ec_query<3,true>"))
INT32 = re.compile(r"v_(add|sub|subrev|mul_lo|mul_hi|mad|lshl|lshr|ashr|and|or|xor|bfe|add3|lshl_add|add_lshl|lshl_or|and_or|or3|min|max|bfi|not|bcnt|mbcnt|subb|addc|subbrev|mul)"
                   r"_?(u|i|b|co_u|lo_u|hi_u)?(32|24|16|_u32|_i32)")


def pmc_class(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64")):
        return "FMA_F64"
    if op.startswith("v_add_f64"):
        return "ADD_F64"
    if op.startswith("v_mul_f64"):
        return "MUL_F64"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return "TRANS_F64"
    if op.startswith("v_cvt"):
        return "CVT"
    if re.match(r"v_(add|sub|subrev)_f32", op):
        return "ADD_F32"
    if op.startswith("v_mul_f32"):
        return "MUL_F32"
    if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mad_f32")):
        return "FMA_F32"
    if op.startswith(("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32")):
        return "TRANS_F32"
    if INT32.match(op) and "f64" not in op and "f32" not in op and "64" not in op.split("_")[-1]:
        return "INT32"
    if re.search(r"(u64|i64|b64)", op) and not op.startswith("v_mov_b64") and not op.startswith("v_cmp"):
        return "INT64"
    return "other"


def weight_other(op):
    if "dpp" in op:
        return W_OTHER["dpp"]
    if op.startswith("v_cmp"):
        return W_OTHER["v_cmp"]
    for k, v in W_OTHER.items():
        if op.startswith(k):
            return v
    return 3.0


def main():
    from msdfgen_amd import build as B
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([B.hipcc()]+B.HIPCC_FLAGS+["--save-temps", os.path.join(B.CSRC, "msdf_capi.hip"), "-o", os.path.join(tmp, "x.so")], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open([os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]).read()
    out = {}
    for key, label in KERNELS:
        m = re.search(r"^(_ZN\w*%s\w*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(key), text, re.S | re.M)
        ops = collections.Counter()
        for ln in m.group(2).split("\n"):
            t = ln.strip()
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            op = t.split()[0]
            if op.startswith("v_"):
                ops[op+(" dpp" if "dpp" in t or "row_" in t else "")] += 1
        oth = [(op, n) for op, n in ops.items() if pmc_class(op) == "other"]
        tot = sum(n for _, n in oth)
        w = sum(weight_other(op)*n for op, n in oth)/tot
        out[label] = {"static_valu": sum(ops.values()), "static_other": tot, "other_weight_cycles": round(w, 2), "top_other": sorted(oth, key=lambda x: -x[1])[:8]}
        print(label, out[label]["static_valu"], tot, round(w, 2))
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_other_class_weights.json" % tag), "w"), indent=1)


if __name__ == "__main__":
    main()
