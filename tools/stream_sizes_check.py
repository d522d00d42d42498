"""The chunk schedules of the host-output pipeline at list lengths and tile sizes other than the bench's (msdf_capi.hip: runPipelineOnce, `lengths`): the streamed
generator (float tiles and the 8-bit atlas) against the device batch, for 3 000 / 5 000 / 9 000 / 20 000 glyphs at 32x32 and 64x64 -- bytes must be identical."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import msdfgen_amd as M
    from msdfgen_amd.shape import ShapeBatch, autoframe
    M.init(0)
    z = np.load(os.path.join(ROOT, "tests", "golden", "dejavu8192.npz"))
    full = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                      z["colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
    bad = 0
    for size in (32, 64):
        frames = np.stack([autoframe(b, size, size, 4) for b in z["bounds"]]) if size != 64 else z["xf64"]
        for n in (3000, 5000, 9000, 20000):
            idx = [(7*i) % 8192 for i in range(n)]
            sub, xfs = full.select(idx), frames[idx]
            want = M.GlyphBatch(sub).generate(M.MODE_MSDF, size, size, xfs).cpu().numpy()
            got = M.generate_stream(sub, M.MODE_MSDF, size, size, xfs)
            ok_f = bool((got.view(np.uint32) == want.view(np.uint32)).all())
            a8 = np.zeros((n, size, size, 3), np.uint8)
            M.generate_stream(sub, M.MODE_MSDF, size, size, xfs, atlas=a8, out_offsets=np.arange(n, dtype=np.int64)*size*size*3, row_stride=size*3)
            conv = (255-(np.float32(255.5)-np.float32(255)*np.clip(want, np.float32(0), np.float32(1))).astype(np.int32)).astype(np.uint8)
            ok_b = bool((a8 == conv).all())
            bad += (not ok_f)+(not ok_b)
            print("size %d glyphs %d: float tiles %s, 8-bit atlas %s" % (size, n, "identical" if ok_f else "DIFFER", "identical" if ok_b else "DIFFER"), flush=True)
    print("stream_sizes_check: %d failure(s)" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
