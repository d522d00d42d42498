"""The C++ drop-in (msdfgen_amd/shim/msdfgen_shim.cpp): a client written against msdfgen's own headers (tests/shim/shim_check.cpp,
after README.md:114-142) calls msdfgen::generateSDF/PSDF/MSDF/MTSDF with the reference's signatures and gets the HIP path.
The client binary is built in the authoring container (needs the msdfgen headers + the reference's non-hot-path objects) and
travels to the GPU box; the test is skipped where it does not exist."""
import os
import subprocess

import numpy as np
import pytest

from conftest import load_npz, bits
from msdfgen_amd.shape import FlatShape

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "shim", "shim_check")


@pytest.mark.parametrize("mode,ydown", [(1, 0), (2, 0), (3, 0), (4, 0), (3, 1)])
def test_reference_client_through_cpp_shim(tmp_path, oracle, mode, ydown):
    if not os.path.exists(BIN):
        pytest.skip("tests/shim/shim_check not built (needs the msdfgen headers)")
    z = load_npz("shape_a.npz")
    s = FlatShape(z["contour_offsets"], z["points"], z["types"], z["colors"])
    desc = tmp_path/"a.txt"
    desc.write_text(str(z["desc"]))
    out = tmp_path/"a.bin"
    w, h = 40, 32
    scale, tx, ty, rng = 2.75, .625, .71875, 1.5   # exactly representable / round-trip safe
    r = subprocess.run([BIN, str(desc), str(out), str(mode), str(w), str(h), repr(scale), repr(tx), repr(ty), repr(rng), str(ydown)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    n = {1: 1, 2: 1, 3: 3, 4: 4}[mode]
    got = np.fromfile(out, np.float32).reshape(h, w, n)
    want = oracle.generate(s, mode, w, h, [scale, scale, tx, ty, -.5*rng, .5*rng], y_down=bool(ydown))
    assert np.abs(got.astype(np.float64)-want).max() <= 1e-5
    print("bitwise differing:", int((bits(got) != bits(want)).sum()))


@pytest.mark.parametrize("mode,rule", [(1, 0), (3, 0), (4, 1), (2, 0), (3, 2)])
def test_reference_client_scanline_flow_through_cpp_shim(tmp_path, oracle, mode, rule):
    """msdfgen::distanceSignCorrection / rasterize (core/rasterization.h:13-27) through the shim: main.cpp's -scanline flow."""
    if not os.path.exists(BIN):
        pytest.skip("tests/shim/shim_check not built (needs the msdfgen headers)")
    z = load_npz("shape_a.npz")
    s = FlatShape(z["contour_offsets"], z["points"], z["types"], z["colors"])
    desc = tmp_path/"a.txt"
    desc.write_text(str(z["desc"]))
    out = tmp_path/"a.bin"
    w, h = 40, 32
    scale, tx, ty, rng = 2.75, .625, .71875, 1.5
    r = subprocess.run([BIN, str(desc), str(out), str(mode), str(w), str(h), repr(scale), repr(tx), repr(ty), repr(rng), "0", str(rule+1)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    n = {1: 1, 2: 1, 3: 3, 4: 4}[mode]
    got = np.fromfile(out, np.float32).reshape(h, w, n)
    xf = [scale, scale, tx, ty, -.5*rng, .5*rng]
    if mode == 2:
        want = oracle.rasterize(s, w, h, xf, rule)
    else:
        want = oracle.sign_correction(s, oracle.generate(s, mode, w, h, xf, overlap=False, ec_mode=0), xf, .5, rule)
        if mode >= 3:
            want = oracle.error_correction(s, want, xf, overlap=False, ec_mode=2, ec_dist=0)
    assert np.abs(got.astype(np.float64)-want).max() <= 1e-5
    assert (bits(got) == bits(want)).all()


@pytest.mark.parametrize("mode", [1, 3, 4])
def test_reference_client_render_sdf_through_cpp_shim(tmp_path, oracle, mode):
    """msdfgen::renderSDF / simulate8bit (core/render-sdf.h:12-22) through the shim: a 2x preview of the generated field."""
    if not os.path.exists(BIN):
        pytest.skip("tests/shim/shim_check not built (needs the msdfgen headers)")
    z = load_npz("shape_a.npz")
    s = FlatShape(z["contour_offsets"], z["points"], z["types"], z["colors"])
    desc = tmp_path/"a.txt"
    desc.write_text(str(z["desc"]))
    out = tmp_path/"a.bin"
    w, h = 40, 32
    scale, tx, ty, rng = 2.75, .625, .71875, 1.5
    r = subprocess.run([BIN, str(desc), str(out), str(mode), str(w), str(h), repr(scale), repr(tx), repr(ty), repr(rng), "0", "9"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, np.float32).reshape(2*h, 2*w, 1)
    field = oracle.generate(s, mode, w, h, [scale, scale, tx, ty, -.5*rng, .5*rng])
    want = oracle.simulate_8bit(oracle.render_sdf(field, 2*w, 2*h, 1, -.5*rng, .5*rng, .5))
    assert (bits(got) == bits(want)).all()


@pytest.mark.parametrize("mode,which", [(3, 7), (4, 7), (3, 8), (4, 8)])
def test_reference_client_shapeless_error_correction_through_cpp_shim(tmp_path, oracle, ref, mode, which):
    """msdfgen::msdfFastDistanceErrorCorrection / msdfFastEdgeErrorCorrection (core/msdf-error-correction.h:21-34) through the shim, all three
    overload families (SDFTransformation; Projection + Range; default and explicit minDeviationRatio) -- against the compiled reference."""
    if not os.path.exists(BIN):
        pytest.skip("tests/shim/shim_check not built (needs the msdfgen headers)")
    z = load_npz("shape_a.npz")
    s = FlatShape(z["contour_offsets"], z["points"], z["types"], z["colors"])
    desc = tmp_path/"a.txt"
    desc.write_text(str(z["desc"]))
    out = tmp_path/"a.bin"
    w, h = 40, 32
    scale, tx, ty, rng = 2.75, .625, .71875, 1.5
    r = subprocess.run([BIN, str(desc), str(out), str(mode), str(w), str(h), repr(scale), repr(tx), repr(ty), repr(rng), "0", str(which)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, np.float32).reshape(h, w, mode)
    xf = [scale, scale, tx, ty, -.5*rng, .5*rng]
    pre = oracle.generate(s, mode, w, h, xf, ec_mode=0)
    want = ref.fast_error_correction(pre, xf, 1.11111111111111111 if which == 7 else 1.5, which == 8)
    assert (bits(got) == bits(want)).all()


@pytest.mark.parametrize("mode,size,count", [(3, 64, 0), (4, 48, 300), (1, 32, 300), (2, 40, 200)])
def test_batch_entry_of_the_shim_equals_per_call_results(tmp_path, mode, size, count):
    """msdfgen_hip::generate*Batch() (include/msdfgen_hip_batch.hpp): a LIST of real msdfgen::Shape objects through ONE pipelined call --
    host threads flatten chunk k+1 into pinned staging while chunk k is uploaded, digested and rendered and chunk k-1 travels back. The client renders every
    glyph through its own generate*() call as well and compares: packed float tiles, float rectangles of an atlas (gutters, two cell sizes, both
    orientations, negative row strides) and an 8-bit atlas (pixelFloatToByte of the per-call floats; gutter bytes untouched) -- zero differing values.
    The 94 + 1 406 distinct glyphs: Basic Latin (Y-down flags as in the fixture) followed by DejaVu glyphs up to 543 edges / 43 contours."""
    if not os.path.exists(BIN):
        pytest.skip("tests/shim/shim_check not built (needs the msdfgen headers)")
    import json
    from msdfgen_amd.shape import ShapeBatch, autoframe
    z = load_npz("dejavu8192.npz")
    full = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                      z["colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
    idx = list(range(0, 8192, 5)) if count == 0 else list(range(3, 8192, 8192//count))[:count]
    sub = full.select(idx)
    sub.inverse_y = (np.arange(sub.n_glyphs) % 7 == 3).astype(np.uint8)          # some shapes Y-down: the flip flag is per glyph
    pinned = (mode, size) == (3, 64)                                               # the fixture holds the REFERENCE's sha256 per msdf 64x64 tile (xf64 = the same framing)
    xfs = z["xf64"][idx] if pinned else np.stack([autoframe(sub.shape(g).bounds(), size, size, 4) for g in range(sub.n_glyphs)])
    path = tmp_path/"shapes.bin"
    sub.dump(str(path), xfs)
    packed = tmp_path/"packed.f32"
    r = subprocess.run([BIN, "batch", str(path), str(mode), str(size), str(size)] + (["0", str(packed)] if pinned else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    res = json.loads(r.stdout.strip().splitlines()[-1])
    print(res)
    assert res["glyphs"] == sub.n_glyphs
    assert res["packed_values_differing"] == 0 and res["atlas_values_differing"] == 0 and res["byte_values_differing"] == 0 and res["gutter_bytes_touched"] == 0
    if pinned:
        # ... and not only against the HIP path's own per-call results: the batch entry's packed tiles against the compiled reference's
        # hashes of the same glyphs (tools/make_golden_full.py). A tile whose bitmap orientation differs from its shape's comes back with its rows reversed.
        import hashlib
        tiles = np.fromfile(packed, np.float32).reshape(sub.n_glyphs, size, size, 3)
        for g in range(sub.n_glyphs):
            flipped = bool(sub.inverse_y[g]) != (g % 3 == 1)                       # shim_check renders every third bitmap Y_DOWNWARD
            tile = np.ascontiguousarray(tiles[g, ::-1] if flipped else tiles[g])
            assert (np.frombuffer(hashlib.sha256(tile.tobytes()).digest(), np.uint8) == z["sha64"][idx[g]]).all(), (g, sub.names[g], flipped)
