"""Exact dynamic instruction attribution of the production kernels: a basic-block execution counter inserted into the compiler's own gfx950 assembly.

VERDICT r5 (#1): "49.5 % of k_distance's VALU instructions are v_cndmask / v_mov / v_cmp / lane moves and nobody has measured where". There is no thread-trace
decoder on this pool, and source-level counters would change the code they count. This tool counts at the ISA level instead:

  build    hipcc -S (device only, -gline-tables-only: identical instruction stream, checked) -> for every basic block of the chosen kernels a five-instruction
           bump of ONE LANE of a few extra VGPRs (v_readlane / s_add / v_writelane, SCC saved and restored; exec-neutral), the lanes flushed into
           msdfhip_bbcount[] with atomics at s_endpgm -> assembled, linked, bundled and wrapped into variants/bbcount.so by the normal host compile.
           The kernels' own instructions are the production build's, in the same order, with the same registers (the extra VGPRs sit above them: one
           wavefront per SIMD fewer, which changes no instruction count). Also writes variants/bbcount_map.json: per block its instructions, each with the
           inline stack of source functions it came from (llvm-symbolizer over the instrumented code object's own line tables).
  run      (GPU) loads the variant, renders the bench workload once, reads the counters -> gpurun_out/<tag>_bbcount_raw.json
  report   counts x per-block instruction mix -> instructions per launch by source REGION and instruction CLASS; compared with the PMC totals of the
           production kernels when a profiles/*_pmc_bench.json is given -> profiles/<tag>_valu_attribution.json + a table on stdout

    [BBCOUNT_NAME=<variant> BBCOUNT_FLAGS="-D..."] python tools/isa_bbcount.py build [kernel-regex ...]
    python tools/isa_bbcount.py run <tag>
    python tools/isa_bbcount.py report <tag> [profiles/r05_pmc_bench.json]
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msdfgen_amd import build as B  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"
NAME = os.environ.get("BBCOUNT_NAME", "bbcount")          # several instrumented builds side by side (A/B of kernel variants): BBCOUNT_NAME=..., BBCOUNT_FLAGS="-D..."
EXTRA_FLAGS = os.environ.get("BBCOUNT_FLAGS", "").split()
WORK = os.path.join(ROOT, "variants", "bbcount_work", NAME)
OUT_SO = os.path.join(ROOT, "variants", NAME+".so")
OUT_MAP = os.path.join(ROOT, "variants", NAME+"_map.json")
NCOUNTERS = 16384                                    # msdfhip_bbcount[]: 64 counters per extra VGPR, the kernels' slices one after the other
DEFAULT_KERNELS = [r"k_distanceILi3ELb1ELb0ELi4E", r"k_distanceILi3ELb1ELb1ELi1E", r"k_distanceILi3ELb0ELb0ELi4E", r"k_ec_fastILi3E", r"k_ec_queryILi3ELb1E"]


def run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if r.returncode != 0:
        raise RuntimeError("%s\n%s" % (" ".join(cmd)[:400], r.stderr[-3000:]))
    return r


def bump(vgpr, lane, t0, t1):
    # (the s_nop's: the assembler inserts no hazard padding; VALU-written SGPR -> SALU and SALU-written SGPR -> v_writelane data are not listed hazards on
    # gfx9, the padding costs nothing that is measured here)
    return ("\ts_cselect_b32 %s, 1, 0\n\tv_readlane_b32 %s, v%d, %d\n\ts_nop 1\n\ts_add_u32 %s, %s, 1\n\ts_nop 1\n\tv_writelane_b32 v%d, %s, %d\n\ts_cmp_lg_u32 %s, 0\n"
            % (t1, t0, vgpr, lane, t0, t0, vgpr, t0, lane, t1))


def instrument(lines, patterns):
    """-> (new lines, kernels: [{symbol, base, blocks, vgprs}])"""
    out, kernels, base = [], [], 0
    i, n = 0, len(lines)
    while i < n:
        l = lines[i]
        m = re.match(r"(_Z\w+):", l)
        if not (m and any(re.search(p, m.group(1)) for p in patterns) and ".amdhsa_kernel " + m.group(1) in KERNEL_DESCS):
            out.append(l)
            i += 1
            continue
        sym = m.group(1)
        j = i+1
        while not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i+1:j]
        text = "".join(body)
        nfv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", KERNEL_DESCS[".amdhsa_kernel "+sym]).group(1))
        nfs = int(re.search(r"\.amdhsa_next_free_sgpr (\d+)", KERNEL_DESCS[".amdhsa_kernel "+sym]).group(1))
        if nfs > 100:
            raise RuntimeError("%s uses %d SGPRs: no two free ones for the counter bump" % (sym, nfs))
        if re.search(r"\bs10[01]\b|s\[100:101\]", text):
            raise RuntimeError("%s already uses s100 / s101" % sym)
        t0, t1 = "s100", "s101"
        vbase = (nfv+3)//4*4
        starts = [k for k, b in enumerate(body) if re.match(r"\.LBB\d+_\d+:|; %bb\.\d+:", b)]
        nblocks = len(starts)
        nv = (nblocks+63)//64
        if vbase+nv > 256:
            raise RuntimeError("%s: %d VGPRs + %d counter registers" % (sym, nfv, nv))
        new = [l]
        blk = -1
        for k, b in enumerate(body):
            if re.match(r"\s+s_endpgm", b):
                fl = "\ts_mov_b64 exec, -1\n\tv_mbcnt_lo_u32_b32 v0, -1, 0\n\tv_mbcnt_hi_u32_b32 v0, -1, v0\n\tv_lshlrev_b32_e32 v0, 2, v0\n\ts_getpc_b64 s[0:1]\n"
                fl += "\ts_add_u32 s0, s0, msdfhip_bbcount@rel32@lo+4\n\ts_addc_u32 s1, s1, msdfhip_bbcount@rel32@hi+12\n"
                fl += "\ts_add_u32 s0, s0, %d\n\ts_addc_u32 s1, s1, 0\n" % (base*4)
                for v in range(nv):
                    fl += "\ts_nop 4\n\tglobal_atomic_add v0, v%d, s[0:1]\n\ts_add_u32 s0, s0, 256\n\ts_addc_u32 s1, s1, 0\n" % (vbase+v)
                new.append(fl)
                new.append(b)
                continue
            new.append(b)
            if k in starts:
                blk += 1
                if blk == 0:
                    new.append("".join("\tv_mov_b32_e32 v%d, 0\n" % (vbase+v) for v in range(nv)))
                new.append(bump(vbase+blk//64, blk % 64, t0, t1))
        out.extend(new)
        kernels.append({"symbol": sym, "base": base, "blocks": nblocks, "vgpr_base": vbase, "vgprs": nv, "orig_vgprs": nfv})
        base += nv*64
        i = j
    if base > NCOUNTERS:
        raise RuntimeError("%d counters needed" % base)
    # kernel descriptors: more VGPRs, two more SGPRs
    text = "".join(out)
    for kd in kernels:
        key = ".amdhsa_kernel "+kd["symbol"]
        old = KERNEL_DESCS[key]
        tot = kd["vgpr_base"]+kd["vgprs"]
        tot4 = (tot+3)//4*4
        newd = re.sub(r"\.amdhsa_next_free_vgpr \d+", ".amdhsa_next_free_vgpr %d" % tot4, old)
        newd = re.sub(r"\.amdhsa_accum_offset \d+", ".amdhsa_accum_offset %d" % tot4, newd)
        newd = re.sub(r"\.amdhsa_next_free_sgpr \d+", ".amdhsa_next_free_sgpr 102", newd)
        assert old in text
        text = text.replace(old, newd)
        # the code-object metadata (YAML note) repeats the counts: the loader sizes the wavefront from the descriptor, the note is informational -- patched too
        text = re.sub(r"(\.name:\s+%s\n(?:.*\n)*?\s+\.sgpr_count:\s+)\d+" % re.escape(kd["symbol"]), lambda mm: mm.group(1)+"108", text, count=1)
        text = re.sub(r"(\.name:\s+%s\n(?:.*\n)*?\s+\.vgpr_count:\s+)\d+" % re.escape(kd["symbol"]), lambda mm: mm.group(1)+str(tot4), text, count=1)
    return text, kernels


KERNEL_DESCS = {}


def classify(mn):
    """PMC class of a VALU mnemonic (SQ_INSTS_VALU_*), 'other' for what no class counter takes; non-VALU: salu / smem / lds / vmem / branch / wait."""
    if mn.startswith("v_"):
        base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", mn)
        if re.match(r"v_(cndmask|mov_b|readlane|writelane|readfirstlane|accvgpr|swap|perm|bfi|mbcnt|nop|permlane)", base):
            return "other"
        if re.match(r"v_cmpx?_", base):
            return "other"
        if re.match(r"v_cvt_", base):
            return "CVT"
        if base.endswith("_f64"):
            if re.match(r"v_(add|sub)_f64", base):
                return "ADD_F64"
            if re.match(r"v_mul_f64", base):
                return "MUL_F64"
            if re.match(r"v_(fma|fmac|div_fmas)_f64", base):      # (checked against SQ_INSTS_VALU_FMA_F64 of the same kernel: exact with these three)
                return "FMA_F64"
            if re.match(r"v_(rcp|rsq|sqrt)_f64", base):
                return "TRANS_F64"
            return "other"                                   # div_scale / div_fmas / div_fixup / min / max / ldexp / frexp / fract / rndne / trig_preop ...
        if base.endswith("_f32"):
            if re.match(r"v_(add|sub|subrev)_f32", base):
                return "ADD_F32"
            if re.match(r"v_mul(_legacy)?_f32", base):
                return "MUL_F32"
            if re.match(r"v_(fma|mad|mac|fmac|fmaak|fmamk)_f32", base):
                return "FMA_F32"
            if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag|_legacy)?_f32", base):
                return "TRANS_F32"
            return "other"
        if re.search(r"_[iu]64$|_b64$", base) or re.match(r"v_mad_[iu]64", base):
            return "INT64"
        if re.search(r"_(u32|i32|b32|u16|i16|b16|u24|i24|co_u32|co_ci_u32)$", base) or re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co", base):
            return "INT32"
        return "other"
    if mn.startswith("s_load") or mn.startswith("s_buffer") or mn.startswith("s_dcache") or mn.startswith("s_memtime") or mn.startswith("s_memrealtime") or mn.startswith("s_store") or mn.startswith("s_atc"):
        return "smem"
    if re.match(r"s_c?branch|s_endpgm|s_setpc|s_swappc|s_call", mn):
        return "branch"
    if re.match(r"s_waitcnt|s_nop|s_barrier|s_sleep|s_sethalt|s_setprio", mn):
        return "wait"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("ds_"):
        return "lds"
    if re.match(r"global_|flat_|buffer_|scratch_", mn):
        return "vmem_scratch" if mn.startswith("scratch_") else "vmem"
    return "misc"


def build(patterns):
    os.makedirs(WORK, exist_ok=True)
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]+EXTRA_FLAGS
    src = os.path.join(B.CSRC, "msdf_capi.hip")
    asm = os.path.join(WORK, "dev.s")
    run([B.hipcc()] + flags + ["-gline-tables-only", "-DMSDF_BBCOUNT=%d" % NCOUNTERS, "-S", "--cuda-device-only", src, "-o", asm])
    lines = open(asm).read().splitlines(keepends=True)
    text = "".join(lines)
    for m in re.finditer(r"(\.amdhsa_kernel \S+)\n(?:.*\n)*?\s*\.end_amdhsa_kernel", text):
        KERNEL_DESCS[m.group(1)] = m.group(0)
    new, kernels = instrument(lines, patterns)
    inst = os.path.join(WORK, "dev_inst.s")
    open(inst, "w").write(new)
    obj, out, fb = os.path.join(WORK, "dev.o"), os.path.join(WORK, "dev.out"), os.path.join(WORK, "dev.hipfb")
    run([LLVM+"/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", inst, "-o", obj])
    run([LLVM+"/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", out, obj])
    run([LLVM+"/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input="+out,
         "-output="+fb])
    run([B.hipcc()] + B.HIPCC_FLAGS + EXTRA_FLAGS + ["-DMSDF_BBCOUNT=%d" % NCOUNTERS, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, src, "-o", OUT_SO])
    # the map: disassemble the instrumented kernels, split at the bumps, symbolize every instruction
    kmap = []
    for kd in kernels:
        dis = run([LLVM+"/llvm-objdump", "-d", "--disassemble-symbols="+kd["symbol"], out]).stdout.splitlines()
        ins = []
        for l in dis:
            m = re.match(r"\s+(\S+)(.*?)//\s*([0-9A-Fa-f]+):", l)
            if m:
                ins.append((int(m.group(3), 16), m.group(1), m.group(2).strip()))
        sym = run([LLVM+"/llvm-symbolizer", "--obj="+out, "--inlines", "--output-style=JSON"], input="\n".join(hex(a) for a, _, _ in ins)+"\n").stdout
        stacks = {}
        for line in sym.splitlines():
            if not line.strip():
                continue
            d = json.loads(line)
            stacks[int(d["Address"], 16)] = [(re.sub(r"\(.*", "", s.get("FunctionName", "?").replace("msdfhip::", "")), os.path.basename(s.get("FileName", "?")), s.get("Line", 0))
                                              for s in d.get("Symbol", [])]
        blocks, cur, k = [], None, 0
        t0 = "s100"
        while k < len(ins):
            a, mn, ops = ins[k]
            # a bump: s_cselect_b32 s101 | v_readlane_b32 s100, vN, lane | s_nop | s_add_u32 | s_nop | v_writelane_b32 | s_cmp_lg_u32
            if mn == "s_cselect_b32" and ops.startswith("s101") and k+6 < len(ins) and ins[k+1][1] == "v_readlane_b32" and ins[k+1][2].startswith(t0):
                mm = re.match(r"s100, v(\d+), (\d+)", ins[k+1][2])
                bid = (int(mm.group(1))-kd["vgpr_base"])*64+int(mm.group(2))
                cur = {"id": bid, "ins": []}
                blocks.append(cur)
                k += 7
                continue
            if cur is not None:
                cur["ins"].append([mn, classify(mn), stacks.get(a, [])])
            k += 1
        # drop the init moves of block 0's predecessor (they precede the first bump: cur is None there) and the flush before s_endpgm
        for b in blocks:
            cut = next((q for q, it in enumerate(b["ins"]) if it[0] == "s_mov_b64" and q+1 < len(b["ins"]) and b["ins"][q+1][0].startswith("v_mbcnt_lo")), None)
            if cut is not None:
                b["ins"] = b["ins"][:cut]+[["s_endpgm", "branch", []]]
        assert len(blocks) == kd["blocks"], (kd["symbol"], len(blocks), kd["blocks"])
        kd = dict(kd)
        kd["block_list"] = blocks
        kd["static_instructions"] = sum(len(b["ins"]) for b in blocks)
        kmap.append(kd)
        print("%s: %d blocks, %d instructions, counters %d..%d, VGPRs %d -> %d" % (kd["symbol"][:60], kd["blocks"], kd["static_instructions"], kd["base"], kd["base"]+kd["vgprs"]*64-1,
                                                                                   kd["orig_vgprs"], kd["vgpr_base"]+kd["vgprs"]))
    json.dump({"source_hash": B.source_hash(), "flags": EXTRA_FLAGS, "counters": NCOUNTERS, "kernels": kmap}, open(OUT_MAP, "w"))
    print("->", OUT_SO, OUT_MAP)


def gpu_run(tag):
    os.environ["MSDFGEN_HIP_LIB"] = OUT_SO
    import ctypes as C
    import numpy as np
    import msdfgen_amd as M
    from msdfgen_amd import lib as L
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_npz
    from msdfgen_amd.shape import ShapeBatch
    M.init(0)
    lib = L.load()
    import hashlib
    buf = (C.c_uint32*NCOUNTERS)()
    if os.environ.get("BBCOUNT_WORKLOAD") == "cjk":                  # BASELINE config 4's CJK-like set (tools/bench_configs.py), pinned by tests/golden/cjk512.npz
        from msdfgen_amd import synth
        from msdfgen_amd.shape import autoframe
        zc = load_npz("cjk512.npz")
        base = [synth.cjk_like_shape(20000+i) for i in range(512)]
        batch = ShapeBatch.from_shapes([base[i % 512] for i in range(8192)])
        xf = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in base])[np.arange(8192) % 512]
        gb = M.GlyphBatch(batch)
        gb.generate(M.MODE_MSDF, 48, 48, xf)
        assert lib.msdfhip_debug_bbcount(buf, NCOUNTERS, 1) == NCOUNTERS
        out = gb.generate(M.MODE_MSDF, 48, 48, xf).cpu().numpy()
        n = lib.msdfhip_debug_bbcount(buf, NCOUNTERS, 1)
        sha = zc["sha48"] if "sha48" in zc else zc["sha"]
        ok = all((np.frombuffer(hashlib.sha256(np.ascontiguousarray(out[g]).tobytes()).digest(), np.uint8) == sha[g % 512]).all() for g in range(0, 8192, 97))
        name = "8192 CJK-like synthetic glyphs (512 distinct), msdf 48x48, library defaults: ONE step (digest + distance + error correction)"
    else:
        z = load_npz("dejavu8192.npz")
        batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                           z["colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
        gb = M.GlyphBatch(batch)
        gb.generate(M.MODE_MSDF, 64, 64, z["xf64"])                      # warm-up: class lists, workspaces
        assert lib.msdfhip_debug_bbcount(buf, NCOUNTERS, 1) == NCOUNTERS
        out = gb.generate(M.MODE_MSDF, 64, 64, z["xf64"]).cpu().numpy()
        n = lib.msdfhip_debug_bbcount(buf, NCOUNTERS, 1)
        ok = all((np.frombuffer(hashlib.sha256(np.ascontiguousarray(out[g]).tobytes()).digest(), np.uint8) == z["sha64"][g]).all() for g in range(0, 8192, 97))
        name = "8192 distinct DejaVu glyphs, msdf 64x64, library defaults: ONE step (digest + distance + error correction)"
    res = {"workload": name, "tiles_match_reference_sha": bool(ok),
           "source_hash": json.load(open(OUT_MAP))["source_hash"], "counts": list(buf[:n])}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "%s_bbcount_raw.json" % tag), "w"))
    print("counters read: %d, nonzero %d, tiles match the reference: %s" % (n, sum(1 for v in buf[:n] if v), ok))


# Source regions of k_distance (innermost match wins from the TOP of the list: an instruction inlined from vlen() into sdLinear() into selAddEdge() is 'linear').
REGIONS = [
    ("eval: sdLinear", r"^sdLinear"), ("eval: sdQuadratic (incl. cos / cbrt kernels)", r"^(sdQuadratic|cosThirdsOne|cosZeroToPi|cosKernel|sinKernel|reduceFrom|powThird|solveQuadratic)"),
    ("eval: sdCubic", r"^sdCubic"), ("eval: distanceToPerpendicular (new nearest edge)", r"^distanceToPerpendicular"),
    ("selector update: selAddEdge (replace tests, perpendicular ends, pbAddPerpIf)", r"^(selAddEdge|sdReplacesWave|pbAddPerpIf|getPerpendicularDistance)"),
    ("relevance: box test + wave vote", r"^selEdgeRelevantBox"), ("relevance: wedges", r"^selEdgeRelevantWedges"),
    ("survivor walk: list entry, record batch loads, loop", r"^selAddContour"),
    ("per contour: selInit / selDistance / selMerge / res[] / member counts", r"^(selInit|selDistance|pbCompute|selMerge|pbMergeWave|pbInit|resolve|median)"),
    ("combiner: contour / pass loops + epilogue over res[]", r"^(shapeDistanceOverlap|combinerEpilogue)"), ("simple combiner", r"^shapeDistanceSimple"),
    ("phase 1: cull (bounds, test, rank, compaction)", r"^(cull|rowRank|rowMinNonNegative|waveMinNonNegative|packEntry|floatAbove)"),
]


def region_of(stack, kernel):
    names = [s[0] for s in stack]
    for title, pat in REGIONS:
        for nm in names:
            if re.search(pat, nm):
                return title
    # what is left sits in distanceBody itself: split by line (phase 1 / tile prologue + stores) using the outermost msdf_kernels.hpp frame
    for nm, f, line in stack:
        if f == "msdf_kernels.hpp" and nm.startswith("distanceBody"):
            if line < 534:
                return "wavefront prologue (work item, offsets, windings, transform)"
            if line < 660:
                return "phase 1: cull (bounds, test, rank, compaction)"
            if line < 688:
                return "phase 1 -> 2 hand-off"
            return "tile prologue (texel position) + mapDistance + stores"
    return "other: " + (names[0] if names else "?")


def report(tag, pmc_path=None):
    kmap = json.load(open(OUT_MAP))
    raw = json.load(open(os.path.join(ROOT, "gpurun_out", "%s_bbcount_raw.json" % tag)))
    counts = raw["counts"]
    pmc = json.load(open(pmc_path))["kernels"] if pmc_path else {}
    result = {"method": __doc__.split("\n\n")[1].strip(), "workload": raw["workload"], "source_hash": raw["source_hash"], "tiles_match_reference_sha": raw["tiles_match_reference_sha"], "kernels": {}}
    for kd in kmap["kernels"]:
        sym = run(["c++filt", kd["symbol"]]).stdout.strip()
        short = re.sub(r"\(.*", "", sym.replace("void msdfhip::", ""))
        by_region = collections.defaultdict(lambda: collections.Counter())
        by_mn = collections.Counter()
        waves = counts[kd["base"]]                                       # block 0 runs once per wavefront
        for b in kd["block_list"]:
            c = counts[kd["base"]+b["id"]]
            if not c:
                continue
            for mn, cls, stack in b["ins"]:
                reg = region_of(stack, short) if "k_distance" in short else (stack[0][0] if stack else "?")
                by_region[reg][cls] += c
                if mn.startswith("v_"):
                    by_mn[re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)] += c
        valu_classes = ["ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT", "other"]
        tot = collections.Counter()
        rows = {}
        for reg, c in by_region.items():
            valu = sum(c[x] for x in valu_classes)
            f64 = c["ADD_F64"]+c["MUL_F64"]+c["FMA_F64"]+c["TRANS_F64"]
            rows[reg] = {"valu": valu, "valu_f64_arith": f64, "valu_other": c["other"], "valu_int_cvt_f32": valu-f64-c["other"], "salu": c["salu"], "smem": c["smem"], "lds": c["lds"],
                         "vmem": c["vmem"], "scratch": c["vmem_scratch"], "branch": c["branch"], "wait": c["wait"]}
            tot.update(c)
        valu_total = sum(tot[x] for x in valu_classes)
        k = {"wavefronts": waves, "valu_insts": valu_total, "valu_other": tot["other"], "valu_other_frac": round(tot["other"]/valu_total, 4) if valu_total else None,
             "class_counts": {x: tot[x] for x in valu_classes if tot[x]}, "salu_insts": tot["salu"], "smem_insts": tot["smem"], "lds_insts": tot["lds"],
             "regions": dict(sorted(rows.items(), key=lambda kv: -kv[1]["valu"])), "top_valu_opcodes": dict(by_mn.most_common(40))}
        # the PMC counts of the production kernel (same source, no counters)
        pk = next((v for n, v in pmc.items() if n.replace(" ", "") == short.replace(" ", "")), None)
        if pk:
            k["pmc"] = {"valu_insts": pk["valu_insts"], "valu_other": pk["valu_other"], "class_counts": pk["valu_class_counts"], "salu_insts": pk["salu_insts"], "smem_insts": pk["smem_insts"], "lds_insts": pk["lds_insts"]}
            k["accounted_valu_frac"] = round(valu_total/pk["valu_insts"], 4)
            k["accounted_other_frac"] = round(tot["other"]/pk["valu_other"], 4) if pk["valu_other"] else None
        result["kernels"][short] = k
        print("\n== %s: %d wavefronts, VALU %.1f M (other %.1f M = %.3f), SALU %.1f M%s" % (short, waves, valu_total/1e6, tot["other"]/1e6, tot["other"]/max(valu_total, 1), tot["salu"]/1e6,
              ("; PMC of the production kernel: VALU %.1f M, other %.1f M -> accounted %.3f / %.3f" % (pk["valu_insts"]/1e6, pk["valu_other"]/1e6, k["accounted_valu_frac"], k["accounted_other_frac"])) if pk else ""))
        print("%-78s %9s %6s %9s %9s %8s %8s" % ("region", "VALU M", "share", "f64 M", "other M", "oth/VALU", "SALU M"))
        for reg, r in k["regions"].items():
            if r["valu"]+r["salu"] < 0.002*(valu_total+tot["salu"]):
                continue
            print("%-78s %9.2f %6.3f %9.2f %9.2f %8.3f %8.2f" % (reg[:78], r["valu"]/1e6, r["valu"]/valu_total, r["valu_f64_arith"]/1e6, r["valu_other"]/1e6, r["valu_other"]/max(r["valu"], 1), r["salu"]/1e6))
    path = os.path.join(ROOT, "profiles", "%s_valu_attribution.json" % tag)
    json.dump(result, open(path, "w"), indent=1)
    print("->", path)


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    if sys.argv[1] == "build":
        build(sys.argv[2:] or DEFAULT_KERNELS)
    elif sys.argv[1] == "run":
        gpu_run(sys.argv[2])
    elif sys.argv[1] == "report":
        report(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
