"""CLI-level drop-in: the reference's UNMODIFIED main.cpp built twice by integration/build_cli_dropin.sh -- once against
msdfgen's own core (tests/cli/msdfgen_cpu) and once with INTEGRATION.md section 2 applied (tests/cli/msdfgen_hip: core/msdfgen.cpp and
core/msdf-error-correction.cpp guarded by integration/patch_msdfgen_for_hip.py, rasterization.cpp / render-sdf.cpp dropped, the HIP shim
linked). Both run the same command lines; the -format fl32 outputs (raw floats, save-fl32.cpp:12-32) must be byte-identical.
main.cpp:1233-1298 is the caller under test: generate* -> [distanceSignCorrection -> msdfErrorCorrection]."""
import os
import subprocess

import numpy as np
import pytest

from conftest import load_npz

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CPU, HIP = os.path.join(HERE, "cli", "msdfgen_cpu"), os.path.join(HERE, "cli", "msdfgen_hip")

TEARDROP = "{ 1471,0; 1149,0; 1021,333; 435,333; 314,0; 0,0; 571,1466; 884,1466; # }{ 926,580; 724,1124; 526,580; # }"          # README.md:214-218 style
BLOBS = ("{ 0,0; (2,3; 5,3); 6,0; (4,-2; 1,-2); # } { 2,0.5; 4,0.5; 3,2; # } { 7,1; (8,3); 10,1; (8,-1); # } "
         "{ 1,4; 3,4; 3,6; 1,6; # } { 1.5,4.5; 1.5,5.5; 2.5,5.5; 2.5,4.5; # }")

FLOWS = [
    # (name, mode, extra arguments)
    ("default flow (scanline pass on: no Skia in a core-only build)", "msdf", []),
    ("noscanline, library-default error correction with distance checks", "msdf", ["-noscanline"]),
    ("noscanline + overlap support", "msdf", ["-noscanline", "-overlap"]),
    ("mtsdf, asymmetric range", "mtsdf", ["-noscanline", "-arange", "-0.35", "0.6"]),
    ("mtsdf scanline flow, fill rule odd", "mtsdf", ["-scanline", "-fillrule", "odd"]),
    ("yflip", "msdf", ["-noscanline", "-yflip"]),
    ("yflip + scanline", "msdf", ["-yflip"]),
    ("sdf", "sdf", []),
    ("psdf noscanline", "psdf", ["-noscanline"]),
    ("error correction edge-fast", "msdf", ["-noscanline", "-errorcorrection", "edge-fast"]),
    ("error correction distance-full", "msdf", ["-noscanline", "-errorcorrection", "distance-full"]),
    ("error correction disabled", "mtsdf", ["-noscanline", "-errorcorrection", "disabled"]),
    ("legacy generators stay on the CPU in both builds", "msdf", ["-legacy"]),
]


def run(binary, mode, desc, out, extra, size=(56, 48)):
    cmd = [binary, mode, "-shapedesc", desc, "-dimensions", str(size[0]), str(size[1]), "-autoframe", "-pxrange", "4", "-format", "fl32", "-o", out]+extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (cmd, r.stdout, r.stderr)
    return open(out, "rb").read()


@pytest.mark.parametrize("shape", ["teardrop", "blobs", "a"])
def test_reference_cli_with_and_without_the_hip_shim(tmp_path, shape):
    if not (os.path.exists(CPU) and os.path.exists(HIP)):
        pytest.skip("tests/cli/ not built (integration/build_cli_dropin.sh needs the msdfgen sources)")
    text = {"teardrop": TEARDROP, "blobs": BLOBS, "a": str(load_npz("shape_a.npz")["desc"])}[shape]
    desc = tmp_path/"shape.txt"
    desc.write_text(text)
    for name, mode, extra in FLOWS:
        want = run(CPU, mode, str(desc), str(tmp_path/"cpu.fl32"), extra)
        got = run(HIP, mode, str(desc), str(tmp_path/"hip.fl32"), extra)
        assert want[:16] == got[:16], name                                                          # FL32 header: magic, height, width, channels
        a, b = np.frombuffer(want[16:], np.float32), np.frombuffer(got[16:], np.float32)
        assert len(a) == len(b) and len(a) > 0
        n = int((a.view(np.uint32) != b.view(np.uint32)).sum())
        assert n == 0, "%s, %s: %d of %d floats differ between the CPU and the HIP build of the CLI (max |delta| %.3g)" % (
            shape, name, n, len(a), float(np.abs(a.astype(np.float64)-b).max()))
