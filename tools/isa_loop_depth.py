"""Where a kernel's SGPR-spill lane moves (v_readlane / v_writelane) sit: static gfx950 ISA of ONE k_distance instantiation, instructions and lane moves by LOOP
DEPTH (from the compiler's own loop annotations in the assembly), and the instruction mix of the innermost edge loop (the loop holding the hand-placed
s_load_dwordx16 batches). No GPU needed: hipcc cross-compiles a one-kernel translation unit in a few seconds.

    python tools/isa_loop_depth.py [SEL OVERLAP GRES TPW]        default 3 true false 4 = the LDS class of the msdf bench step
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msdfgen_amd import build as B  # noqa: E402


def analyse(sel="3", overlap="true", gres="false", tpw="4"):
    """-> {"resources": {...}, "instructions": {depth: n}, "lane_moves": {depth: n}, "edge_loop": {class: n} or None}"""
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "probe.hip")
        open(src, "w").write('#include "msdf_kernels.hpp"\nusing namespace msdfhip;\n'
                             "template __global__ void msdfhip::k_distance<%s, %s, %s, %s>(int, const int32_t *, const int32_t *, const EdgeRec *, const int8_t *, const MsdfHipGlyph *, "
                             "int, int, int, int, int, float *, int, unsigned, double *, size_t, const int *, int, unsigned *, unsigned);\n" % (sel, overlap, gres, tpw))
        flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]+os.environ.get("MSDF_ISA_FLAGS", "").split()   # e.g. MSDF_ISA_FLAGS=-DMSDF_DISTANCE_WAVES_PER_SIMD=3
        r = subprocess.run([B.hipcc()] + flags + ["-I", B.CSRC, "-I", os.path.join(ROOT, "include"), "-c", "--cuda-device-only", "-save-temps", "-Rpass-analysis=kernel-resource-usage",
                            src, "-o", os.path.join(d, "probe.o")], capture_output=True, text=True, cwd=d)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        asm = open(os.path.join(d, "probe-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
        res = {}
        for line in r.stderr.splitlines():
            m = re.search(r"remark: [^ ]+ +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
            if m and "k_distance" in r.stderr[:r.stderr.find(line)].rsplit("Function Name:", 1)[-1]:
                res[m.group(1).strip()] = m.group(2)
    start = next(i for i, l in enumerate(asm) if re.match(r"_ZN7msdfhip10k_distanceI", l))
    end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
    lines = asm[start:end+1]
    depth, d = [], 0
    for i, l in enumerate(lines):
        m = re.match(r"\.LBB\d+_\d+:\s*;(.*)", l)
        if m:
            txt, j = m.group(1), i+1
            while j < len(lines) and lines[j].lstrip().startswith(";"):
                txt += lines[j]
                j += 1
            m2 = re.search(r"(?:This (?:Inner )?Loop Header: |in Loop: Header=BB\d+_\d+ )Depth=(\d+)", txt)
            d = int(m2.group(1)) if m2 else (d if re.search(r"Loop", txt) else 0)
        depth.append(d)
    instr = [i for i, l in enumerate(lines) if re.match(r"\s+[a-z]", l)]
    tot, moves, scratch = collections.Counter(), collections.Counter(), collections.Counter()
    for i in instr:
        tot[depth[i]] += 1
        if "v_readlane" in lines[i] or "v_writelane" in lines[i]:
            moves[depth[i]] += 1
        if re.match(r"\s+scratch_", lines[i]):
            scratch[depth[i]] += 1
    # the edge loop: the contiguous run of instructions, around the first hand-placed s_load_dwordx16 batch, that sit at that batch's loop depth or deeper
    edge = None
    first = next((i for i in instr if "s_load_dwordx16" in lines[i]), None)
    if first is not None and depth[first] > 0:
        d0 = depth[first]
        lo = hi = instr.index(first)
        while lo > 0 and depth[instr[lo-1]] >= d0:
            lo -= 1
        while hi+1 < len(instr) and depth[instr[hi+1]] >= d0:
            hi += 1
        body = [lines[i] for i in instr[lo:hi+1]]
        mix = [("instructions", r"."), ("f64 arithmetic", r"_f64"), ("v_cndmask", r"v_cndmask"), ("v_mov", r"v_mov_b"), ("v_cmp", r"v_cmp"), ("lane moves", r"v_readlane|v_writelane"),
               ("scratch", r"^\s+scratch_"), ("salu", r"^\s+s_(?!waitcnt|load|cbranch|branch|nop)"), ("branches", r"s_c?branch"), ("s_load", r"s_load"), ("lds", r"\bds_")]
        edge = {n: sum(1 for l in body if re.search(p, l)) for n, p in mix}
    return {"resources": res, "instructions": dict(tot), "lane_moves": dict(moves), "scratch_ops": dict(scratch), "edge_loop": edge}


def main():
    sel, overlap, gres, tpw = (sys.argv[1:5] + ["3", "true", "false", "4"][len(sys.argv[1:5]):])
    a = analyse(sel, overlap, gres, tpw)
    res = a["resources"]
    print("k_distance<%s, %s, %s, %s>: %s" % (sel, overlap, gres, tpw, ", ".join("%s %s" % (k, res[k]) for k in ("VGPRs", "TotalSGPRs", "SGPRs Spill", "VGPRs Spill", "Occupancy") if k in res)))
    print("loop depth   instructions   v_readlane + v_writelane   scratch loads + stores")
    for k in sorted(a["instructions"]):
        print("%10d %14d %10d %22d" % (k, a["instructions"][k], a["lane_moves"].get(k, 0), a["scratch_ops"].get(k, 0)))
    if a["edge_loop"]:
        print("edge loop: " + ", ".join("%s %d" % kv for kv in a["edge_loop"].items()))


if __name__ == "__main__":
    main()
