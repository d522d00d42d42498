"""Turns the rocprofv3 passes of `bench.py` (tools/profile_round.sh) into profiles/pmc_traffic.json, the counters bench.py quotes for the
dominant pass (the distance field = every k_distance<...> launch of a step: up to three instantiations, the global-scratch one chunked):

  * HBM bytes per step from the separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, corrected as /opt/skills/guides/MI355X_MICROARCH.md
    (HBM section) prescribes: both counters are in units of 1024 B, and on gfx950 FETCH_SIZE reports half of the bytes of a wide
    coalesced read (so the read side is doubled; WRITE_SIZE is uncalibrated, taken as is);
  * the pass's duration = union of the launches' intervals in the --kernel-trace --stats pass (the classes run concurrently);
  * VALU issue utilisation of the pass from the SQ pass: SQ_INSTS_VALU wave-instructions x 4 cycles (a wave64 instruction occupies the
    16-lane SIMD for at least 4 cycles; fp64 ops take longer, so this is a LOWER bound of the busy fraction) over 1024 SIMDs x the pass's
    duration (--kernel-trace --stats pass) at the 2.4 GHz peak engine clock; SQ_WAIT_ANY / SQ_WAVE_CYCLES = share of a wavefront's
    life spent parked in s_waitcnt.

    python tools/pmc_traffic.py <tag> <commit>        (reads gpurun_out/<tag>_{stats,fetch,write,sq}/**/*.db)
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOCK_HZ = 2.4e9
SIMDS = 256*4


def db(tag, kind):
    hits = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, kind), "**", "*.db"), recursive=True)
    if not hits:
        raise SystemExit("no database for %s_%s" % (tag, kind))
    return sqlite3.connect(hits[0]).cursor()


def counter_per_step(cur, counter, like):
    rows = cur.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    steps = max(n for name, _, n in rows if "k_ec_fast" in name)              # one k_ec_fast launch per step
    return sum(v for name, v, _ in rows if like in name)/steps, steps


def main():
    tag, commit = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "?"
    like = "k_distance"
    fetch, steps = counter_per_step(db(tag, "fetch"), "FETCH_SIZE", like)
    write, _ = counter_per_step(db(tag, "write"), "WRITE_SIZE", like)
    sq = db(tag, "sq")
    c = {name: counter_per_step(sq, name, like)[0] for name in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_WAVES",
                                                                 "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")}
    st = db(tag, "stats")
    rows = st.execute("select name, sum(end-start), count(*) from kernels group by name").fetchall()
    nsteps = max(n for name, _, n in rows if "k_ec_fast" in name)
    # the glyph classes' launches run concurrently (side streams): the pass lasts as long as the UNION of their intervals
    spans = sorted(st.execute("select start, end from kernels where name like ?", ("%"+like+"%",)).fetchall())
    busy, (lo, hi) = 0, spans[0]
    for a, b in spans[1:]:
        if a > hi:
            busy, lo, hi = busy+hi-lo, a, b
        else:
            hi = max(hi, b)
    dist_ns = (busy+hi-lo)/nsteps
    dist_sum_ns = sum(t for name, t, _ in rows if like in name)/nsteps
    per_kernel = {name.split("(")[0].replace("void msdfhip::", ""): round(t/nsteps/1e6, 4) for name, t, _ in rows if t/nsteps > 2000}
    cycles = dist_ns*1e-9*CLOCK_HZ*SIMDS
    out = {"workload": "dejavu8192", "glyphs_per_gpu": 8192, "tile": [64, 64], "commit": commit, "steps_profiled": steps,
           "kernels": "every k_distance<...> launch of a step (1-contour / LDS-scratch / global-scratch classes)",
           "distance_pass_ms": round(dist_ns/1e6, 4), "distance_kernels_sum_ms": round(dist_sum_ns/1e6, 4), "kernel_ms_per_step": per_kernel,
           "FETCH_SIZE_raw": fetch, "WRITE_SIZE_raw": write, "hbm_read_bytes": 2*fetch*1024, "hbm_write_bytes": write*1024,
           "hbm_bytes_per_launch": 2*fetch*1024+write*1024,
           "correction": "x1024 B per counter unit; FETCH_SIZE x2 (gfx950 half-count of wide coalesced reads, MI355X_MICROARCH.md)",
           "valu_issue_frac": round(4*c["SQ_INSTS_VALU"]/cycles, 4),
           "valu_issue_note": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x distance-pass duration x 2.4 GHz), separate rocprofv3 --pmc pass at commit %s; "
                              "fp64 instructions occupy the SIMD longer than 4 cycles, so this is a lower bound of the busy fraction" % commit,
           "valu_active_frac": round(4*c["SQ_ACTIVE_INST_VALU"]/cycles, 4),
           "wait_any_over_wave_cycles": round(c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"], 4),
           "wait_inst_any_over_wave_cycles": round(c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"], 4),
           "lds_bank_conflict_over_idx_active": round(c["SQ_LDS_BANK_CONFLICT"]/max(c["SQ_LDS_IDX_ACTIVE"], 1), 5),
           "sq_counters_per_step": {k: round(v) for k, v in c.items()}}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
