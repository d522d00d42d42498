"""Turns the rocprofv3 passes of `bench.py` (tools/profile_round.sh) into profiles/pmc_traffic.json -- what bench.py quotes for the dominant
pass (the distance field = every k_distance<...> launch of a step; up to three instantiations running concurrently):

  * HBM bytes per step from the separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes. Units and corrections as CALIBRATED on this device
    (tools/valu_calib.hip --fetch, profiles/r03_valu_calibration.json): both counters count KiB; WRITE_SIZE is exact for all three store
    patterns of this library; FETCH_SIZE tallies vector loads at HALF and wave-uniform scalar loads at 1.13x of the bytes moved. k_distance
    reads through both (gathers in phase 1, s_load in phase 2), so the read side is reported as a RANGE [x1, x2] and `hbm_bytes_per_launch`
    takes the x2 end (the guide's rule for wide loads; an upper bound here);
  * the pass's duration = union of the launches' intervals in the --kernel-trace --stats pass;
  * the VALU busy fraction of the distance kernels from tools/pmc_report.py (profiles/<tag>_pmc_bench.json: per-class instruction counts x
    measured cycles per class, shader clock from GRBM_GUI_ACTIVE), weighted by each kernel's own time.

    python tools/pmc_traffic.py <tag> <commit>
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def db(tag, kind):
    hits = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, kind), "**", "*.db"), recursive=True)
    if not hits:
        raise SystemExit("no database for %s_%s" % (tag, kind))
    return sqlite3.connect(hits[0]).cursor()


def counter_per_step(cur, counter, like):
    rows = cur.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    steps = max(n for name, _, n in rows if "k_ec_fast" in name)              # one k_ec_fast launch per step
    return sum(v for name, v, _ in rows if like in name)/steps, steps


TILES = 8192*64*64*3*4
ALGORITHMIC = 8192*(64*64*3*4+48)+72*189191


def main():
    tag, commit = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "?"
    like = "k_distance"
    fetch, steps = counter_per_step(db(tag, "fetch"), "FETCH_SIZE", like)
    write, _ = counter_per_step(db(tag, "write"), "WRITE_SIZE", like)
    st = db(tag, "stats")
    rows = st.execute("select name, sum(end-start), count(*) from kernels group by name").fetchall()
    nsteps = max(n for name, _, n in rows if "k_ec_fast" in name)
    spans = sorted(st.execute("select start, end from kernels where name like ?", ("%"+like+"%",)).fetchall())
    busy, (lo, hi) = 0, spans[0]
    for a, b in spans[1:]:
        if a > hi:
            busy, lo, hi = busy+hi-lo, a, b
        else:
            hi = max(hi, b)
    dist_ns = (busy+hi-lo)/nsteps
    per_kernel = {name.split("(")[0].replace("void msdfhip::", ""): round(t/nsteps/1e6, 4) for name, t, _ in rows if t/nsteps > 2000}
    rep = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_bench.json" % tag)))["kernels"]
    dk = {k: v for k, v in rep.items() if like in k}
    # VALU busy of the PASS: every class kernel's busy SIMD time (busy fraction x its duration alone, from the serialized counter passes) over the duration
    # of the pass with the launches concurrent. (Round 3 weighted the per-kernel fractions by their durations alone; since round 4 the global-scratch
    # class runs as a small persistent grid that is slow BY DESIGN when profiled alone, which that average would count as an idle device.)
    busy_w = sum(v["valu_busy_frac_calibrated"]*v["ms_alone"] for v in dk.values())/(dist_ns/1e6)
    from msdfgen_amd.build import source_hash
    # fp64 operations the distance kernels actually EXECUTED per step (PMC class counts x 64 lanes, FMA = 2; masked-off lanes included, so an
    # upper bound of the useful ones) over the pass's duration with the three launches concurrent: the achieved fp64 rate, not an estimate
    flops = sum((v["valu_class_counts"].get("ADD_F64", 0)+v["valu_class_counts"].get("MUL_F64", 0)+2*v["valu_class_counts"].get("FMA_F64", 0))*64. for v in dk.values())
    other = sum(v["valu_other"] for v in dk.values())/max(1., sum(v["valu_insts"] for v in dk.values()))
    out = {"workload": "dejavu8192", "source_hash": source_hash(), "fp64_gflops_pmc": round(flops/(dist_ns*1e-9)/1e9, 1),
           "fp64_note": "(SQ_INSTS_VALU_ADD_F64 + MUL_F64 + 2 x FMA_F64) x 64 lanes of the three k_distance launches / duration of the pass (launches concurrent)",
           "valu_other_over_valu_insts": round(other, 4), "glyphs_per_gpu": 8192, "tile": [64, 64], "commit": commit, "steps_profiled": steps,
           "kernels": "every k_distance<...> launch of a step (1-contour / LDS-scratch / global-scratch classes)",
           "distance_pass_ms": round(dist_ns/1e6, 4), "kernel_ms_per_step_concurrent": per_kernel,
           "FETCH_SIZE_raw_KiB": fetch, "WRITE_SIZE_raw_KiB": write,
           "hbm_read_bytes_range": [fetch*1024, 2*fetch*1024], "hbm_write_bytes": write*1024,
           "hbm_bytes_per_launch": 2*fetch*1024+write*1024,
           # W*H*N*4 + 72*E + 48 per glyph = 8192 x (64*64*3*4 + 48) + 72 x 189 191 edges of the bench workload
           "algorithmic_bytes_per_launch": ALGORITHMIC, "tile_bytes_per_launch": TILES,
           "traffic_over_algorithmic": round((2*fetch*1024+write*1024)/ALGORITHMIC, 3), "write_over_algorithmic": round(write*1024/ALGORITHMIC, 3),
           "non_tile_write_bytes": write*1024-TILES,
           "non_tile_write_note": "WRITE_SIZE minus the tiles: scratch (spill) stores of the two 128-VGPR instantiations + the global-scratch class's workspace slices; "
                                  "tools/scratch_by_source.py splits the spill part by source function from basic-block counts",
           "correction": "counters in KiB (calibrated: profiles/r03_valu_calibration.json); WRITE_SIZE exact; FETCH_SIZE x2 for vector loads, x0.88 for scalar loads -- "
                         "the kernel mixes both, the x2 end is used",
           "valu_busy_frac_calibrated": round(busy_w, 4),
           "valu_busy_per_kernel": {k: {"ms_alone": v["ms_alone"], "busy": v["valu_busy_frac_calibrated"], "wait_any": v["wait_any_over_wave_cycles"],
                                        "scalar_cache_miss_rate": v["scalar_cache_miss_rate"], "shader_clock_ghz": v["shader_clock_ghz"]} for k, v in dk.items()},
           "valu_busy_note": "sum over instruction classes of (PMC count x cycles per wave64 instruction measured with tools/valu_calib.hip) / (1024 SIMDs x kernel time x "
                             "shader clock from GRBM_GUI_ACTIVE); per kernel: each alone on the device (rocprofv3 serialises kernels under --pmc); for the pass: the class kernels' busy "
                             "SIMD time over the pass's duration with the launches concurrent; commit %s." % commit}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out)[:1500])


if __name__ == "__main__":
    main()
