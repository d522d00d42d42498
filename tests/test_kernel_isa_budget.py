"""Static budget of the dominant kernel's inner loop (no GPU: hipcc cross-compiles one instantiation in a few seconds).

k_distance holds a survivor's record in ~80 SGPRs inside its per-edge loop and sits at its register caps (106 SGPRs, 128 VGPRs at 4 wavefronts per SIMD). Its 90-odd SGPR
spills are harmless while they stay OUTSIDE that loop (profiles/r05_ab_notes.md: none inside); round 5 saw a ten-line change in a workgroup's EXIT path move 592
v_readlane / v_writelane INTO the loop and cost the global-scratch class 40 % of its speed (CJK-like set 13.6 -> 18.2 ms) with every parity test still green. This
test pins the property per instantiation of the bench step: no spill lane moves and no scratch traffic at the depth of the edge loop or below, occupancy as designed."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")


@pytest.mark.parametrize("inst,occupancy,max_vgpr_spill", [(("3", "true", "false", "4"), 4, 40),    # LDS class: 2+ contours, combiner scratch in LDS, four tiles per wavefront
                                                             (("3", "true", "true", "1"), 4, 40),     # global-scratch class (persistent grid): many contours
                                                             (("3", "false", "false", "4"), 5, 0)])    # one-contour class: simple combiner
def test_edge_loop_of_k_distance_has_no_spills(inst, occupancy, max_vgpr_spill):
    """The overlapping-combiner instantiations run at FOUR wavefronts per SIMD (128 VGPRs) and spill 10-24 dwords to scratch -- which pays only because every
    spill sits outside the edge loop (5 / 6 wavefronts per SIMD put a handful inside: 7.5 / 8.9 instead of 5.4 ms per step), and only since the combiner's pass
    results are no longer loop-carried across the walk (94-164 dwords before: 2.4 GB of scratch stores per pass)."""
    from isa_loop_depth import analyse
    a = analyse(*inst)
    res = a["resources"]
    assert int(res["Occupancy"]) == occupancy, res
    assert int(res["VGPRs Spill"]) <= max_vgpr_spill, res
    deep_moves = {d: n for d, n in a["lane_moves"].items() if d >= 4 and n}
    deep_scratch = {d: n for d, n in a["scratch_ops"].items() if d >= 4 and n}
    assert sum(deep_moves.values()) <= 4, "SGPR-spill lane moves inside the edge loop: %s (by loop depth; resources %s)" % (deep_moves, res)
    assert not deep_scratch, "scratch loads / stores inside the edge loop: %s (by loop depth; resources %s)" % (deep_scratch, res)
    e = a["edge_loop"]
    assert e is not None and e["lane moves"] <= 4 and e["scratch"] == 0 and e["f64 arithmetic"] > 300, e
