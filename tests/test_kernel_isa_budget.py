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


@pytest.mark.parametrize("inst,occupancy,max_vgpr_spill", [(("3", "true", "false", "4"), 4, 56),    # LDS class: 2+ contours, combiner scratch in LDS, four tiles per wavefront
                                                             (("3", "true", "true", "1"), 4, 96),     # global-scratch class (persistent grid): many contours
                                                             (("3", "false", "false", "4"), 5, 0)])    # one-contour class: simple combiner
def test_edge_loop_of_k_distance_has_no_spills(inst, occupancy, max_vgpr_spill):
    """The overlapping-combiner instantiations run at FOUR wavefronts per SIMD (128 VGPRs) and spill a few dozen dwords to scratch -- which pays only because
    every spill sits outside the HOT edge loop (5 / 6 wavefronts per SIMD put a handful inside: 7.5 / 8.9 instead of 5.4 ms per step). Round 6: the combiner has
    two instances of the contour loop (msdf_device.hpp: shapeDistanceOverlapSplit) -- the hot one, whose edge loop holds the hand-placed s_load_dwordx16 batches,
    and the rare second walks, which may spill as they like. The static spill count went UP with the cold instance (24 -> 42 / 66 dwords) while the scratch bytes
    the pass really writes went DOWN (585 -> 334 MB per 8 192 glyphs, tools/isa_bbcount.py: dynamic counts): what is pinned here is the hot loop."""
    from isa_loop_depth import analyse
    a = analyse(*inst)
    res = a["resources"]
    assert int(res["Occupancy"]) == occupancy, res
    assert int(res["VGPRs Spill"]) <= max_vgpr_spill, res
    e = a["edge_loop"]                                   # the loop around the first s_load_dwordx16 batch = the hot walk
    assert e is not None and e["lane moves"] <= 4 and e["scratch"] == 0 and e["f64 arithmetic"] > 300, e
