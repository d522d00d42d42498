"""Per-kernel counter report of one workload from the rocprofv3 passes of tools/profile_calib.sh / tools/profile_round.sh, with the VALU busy
fraction priced by MEASURED cycles per instruction class (tools/valu_calib.hip -> profiles/r03_valu_calibration.json) instead of a flat
"4 cycles per SQ_INSTS_VALU" (VERDICT r2, weak #2):

    busy = sum_class(count_class x cycles_class) / (1024 SIMDs x kernel duration x shader clock)

  * class counts: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64, _{ADD,MUL,FMA,TRANS}_F32, _INT32, _INT64, _CVT (passes p1 / p2); what is left of
    SQ_INSTS_VALU ("other": v_cndmask / v_mov / v_cmp / readlane / div helpers ...) is priced by the kernel's STATIC mix of those opcodes
    (profiles/r03_other_class_weights.json, from hipcc --save-temps assembly) -- 3.6-3.7 cycles for every kernel of this library;
  * shader clock during the kernel: GRBM_GUI_ACTIVE / 8 XCDs / duration (pass p4) -- the fp64 kernels run at 2.1-2.35 GHz, not at the 2.4 GHz peak;
  * durations are those of the counter passes (rocprofv3 serialises kernels under --pmc): each kernel ALONE on the device.

    python tools/pmc_report.py <tag> <workload: bench|cfg4|cfg5> [kernel-substring ...]     (reads gpurun_out/<tag>_<workload>_p{1..4}/**/*.db)
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 1024
CLASSES = ["ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT"]


def load_weights():
    p = os.path.join(ROOT, "profiles", "r03_valu_calibration.json")
    w = json.load(open(p))["cycles_per_wave64_instruction"]
    other = json.load(open(os.path.join(ROOT, "profiles", "r03_other_class_weights.json")))
    return w, other


def short(name):
    n = name.replace("void msdfhip::", "")
    return n.split("(")[0]


def rows(path):
    cur = sqlite3.connect(path).cursor()
    out = {}
    for name, counter, n, total in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        out.setdefault(short(name), {})[counter] = (n, total)
    dur = {}
    for name, n, total in cur.execute("select name, count(*), sum(end-start) from kernels group by name"):
        dur[short(name)] = (n, total)
    return out, dur


def main():
    tag, workload = sys.argv[1], sys.argv[2]
    only = sys.argv[3:]
    weights, other_w = load_weights()
    counters, durations = {}, {}
    for p in (1, 2, 3, 4):
        hits = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s_p%d" % (tag, workload, p), "**", "*.db"), recursive=True)
        if not hits:
            raise SystemExit("no database for %s_%s_p%d" % (tag, workload, p))
        c, d = rows(hits[0])
        for k, v in c.items():
            counters.setdefault(k, {}).update(v)
        for k, v in d.items():
            durations.setdefault(k, []).append(v)
    report = {}
    for k in sorted(counters, key=lambda k: -sum(t for _, t in durations.get(k, [(0, 0)]))):
        if only and not any(o in k for o in only):
            continue
        c = counters[k]
        if "SQ_INSTS_VALU" not in c:
            continue
        per = lambda name: c[name][1]/c[name][0] if name in c and c[name][0] else 0.    # per dispatch
        ms = sum(t/n for n, t in durations[k])/len(durations[k])/1e6
        if ms < 0.02:
            continue
        clock = per("GRBM_GUI_ACTIVE")/8/(ms*1e-3) if per("GRBM_GUI_ACTIVE") else 2.3e9
        valu = per("SQ_INSTS_VALU")
        cls = {x: per("SQ_INSTS_VALU_"+x) for x in CLASSES}
        other = max(valu-sum(cls.values()), 0.)
        wo = other_w.get(k.replace(", ", ","), {}).get("other_weight_cycles", 3.7)
        cycles = sum(cls[x]*weights[x] for x in CLASSES)+other*wo
        avail = SIMDS*ms*1e-3*clock
        wave_cycles = per("SQ_WAVE_CYCLES")
        line = {"ms_alone": round(ms, 4), "dispatches": c["SQ_INSTS_VALU"][0], "shader_clock_ghz": round(clock/1e9, 3), "waves": round(per("SQ_WAVES")),
                "valu_insts": round(valu), "valu_class_counts": {x: round(v) for x, v in cls.items() if v}, "valu_other": round(other), "other_weight_cycles": wo,
                "valu_busy_frac_calibrated": round(cycles/avail, 4), "valu_busy_frac_flat4_r2_method": round(4*valu/(SIMDS*ms*1e-3*2.4e9), 4),
                "salu_insts": round(per("SQ_INSTS_SALU")), "salu_pipe_frac": round(per("SQ_INSTS_SALU")*weights["SALU"]/avail, 4),
                "smem_insts": round(per("SQ_INSTS_SMEM")), "lds_insts": round(per("SQ_INSTS_LDS")), "vmem_rd": round(per("SQ_INSTS_VMEM_RD")), "vmem_wr": round(per("SQ_INSTS_VMEM_WR")),
                "wait_any_over_wave_cycles": round(per("SQ_WAIT_ANY")/wave_cycles, 4) if wave_cycles else None,
                "wait_inst_any_over_wave_cycles": round(per("SQ_WAIT_INST_ANY")/wave_cycles, 4) if wave_cycles else None,
                "active_inst_any_over_wave_cycles": round(per("SQ_ACTIVE_INST_ANY")/wave_cycles, 4) if wave_cycles else None,
                "scalar_cache_miss_rate": round(per("SQC_DCACHE_MISSES")/per("SQC_DCACHE_REQ"), 4) if per("SQC_DCACHE_REQ") else None,
                "lds_bank_conflict_over_idx_active": round(per("SQ_LDS_BANK_CONFLICT")/per("SQ_LDS_IDX_ACTIVE"), 4) if per("SQ_LDS_IDX_ACTIVE") else None,
                "cycles_per_wave": round(4*wave_cycles/per("SQ_WAVES")) if per("SQ_WAVES") else None,
                "insts_per_wave": round((valu+per("SQ_INSTS_SALU")+per("SQ_INSTS_SMEM")+per("SQ_INSTS_LDS")+per("SQ_INSTS_VMEM_RD")+per("SQ_INSTS_VMEM_WR"))/per("SQ_WAVES")) if per("SQ_WAVES") else None}
        report[k] = line
    out = {"tag": tag, "workload": workload, "method": __doc__.split("\n\n")[1].strip(), "kernels": report}
    path = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (tag, workload))
    json.dump(out, open(path, "w"), indent=1)
    print("%-34s %8s %6s %9s %9s %7s %7s %7s %7s %7s" % ("kernel", "ms alone", "GHz", "VALU M", "SALU M", "busy", "flat4", "wait", "sqc miss", "lds cfl"))
    for k, l in report.items():
        print("%-34s %8.3f %6.2f %9.1f %9.1f %7.3f %7.3f %7.3f %7s %7s" % (k[:34], l["ms_alone"], l["shader_clock_ghz"], l["valu_insts"]/1e6, l["salu_insts"]/1e6,
              l["valu_busy_frac_calibrated"], l["valu_busy_frac_flat4_r2_method"], l["wait_any_over_wave_cycles"] or 0,
              l["scalar_cache_miss_rate"], l["lds_bank_conflict_over_idx_active"]))
    print("->", path)


if __name__ == "__main__":
    main()
