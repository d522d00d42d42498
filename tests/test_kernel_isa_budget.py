"""Static budget of the dominant kernel's inner loop (no GPU: hipcc cross-compiles one instantiation in a few seconds).

k_distance holds a survivor's record in ~80 SGPRs inside its per-edge loop and sits at its register caps (106 SGPRs, 154-168 VGPRs, 3 wavefronts per SIMD). Its 90-odd SGPR
spills are harmless while they stay OUTSIDE that loop (profiles/r05_ab_notes.md: none inside); round 5 saw a ten-line change in a workgroup's EXIT path move 592
v_readlane / v_writelane INTO the loop and cost the global-scratch class 40 % of its speed (CJK-like set 13.6 -> 18.2 ms) with every parity test still green. This
test pins the property per instantiation of the bench step: no spill lane moves at the depth of the edge loop or below, occupancy and VGPR spills unchanged."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")


@pytest.mark.parametrize("inst,occupancy,max_vgpr_spill", [(("3", "true", "false", "4"), 3, 0),      # LDS class: 2+ contours, combiner scratch in LDS, four tiles per wavefront
                                                             (("3", "true", "true", "1"), 3, 2),       # global-scratch class (persistent grid): many contours
                                                             (("3", "false", "false", "4"), 5, 0)])    # one-contour class: simple combiner
def test_edge_loop_of_k_distance_has_no_spill_lane_moves(inst, occupancy, max_vgpr_spill):
    from isa_loop_depth import analyse
    a = analyse(*inst)
    res = a["resources"]
    assert int(res["Occupancy"]) == occupancy, res
    assert int(res["VGPRs Spill"]) <= max_vgpr_spill and int(res.get("ScratchSize", "0")) <= 16, res
    deep = {d: n for d, n in a["lane_moves"].items() if d >= 4 and n}
    assert sum(deep.values()) <= 2, "SGPR-spill lane moves inside the edge loop: %s (by loop depth; resources %s)" % (deep, res)
    assert a["edge_loop"] is not None and a["edge_loop"]["lane moves"] <= 2 and a["edge_loop"]["f64 arithmetic"] > 300, a["edge_loop"]
