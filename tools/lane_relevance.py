"""How much finer than a wavefront could relevance get? (tools only; tests/hostemu's emu_lane_relevance_stats: lockstep walk of phase 2 of k_distance.)
For every edge a wavefront evaluates (some lane finds it relevant), how many of its 64 lanes did -- i.e. the lower bound of ANY finer-grained
scheme (half tiles, quarter tiles, lanes = (texel, edge) pairs) against the wave-level count of today.

    python tools/lane_relevance.py
"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
from emu import Emu
from msdfgen_amd.shape import ShapeBatch, distance_mapping
e=Emu()
z=np.load(ROOT+'/tests/golden/dejavu8192.npz')
batch=ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32), z["colors"].astype(np.int32), np.zeros(8192,bool), [str(n) for n in z["names"]])
tot=np.zeros(8,np.int64); tiles=0
for g in range(0,8192,41):
    s=batch.shape(g); xf=z["xf64"][g]
    ms,mt=distance_mapping(xf[4],xf[5]); x6=np.array([xf[0],xf[1],xf[2],xf[3],ms,mt])
    keep,args=e._shape(s); out=np.zeros(8,np.int64)
    e.lib.emu_lane_relevance_stats(64,64,*args,x6.ctypes.data_as(C.POINTER(C.c_double)),out.ctypes.data_as(C.POINTER(C.c_long)))
    tot+=out; tiles+=64
print("wave evaluations/tile %.2f, relevant lanes per evaluation %.1f of 64, skipped by vote/tile %.2f, lane-evaluations per lane %.2f" % (tot[0]/tiles, tot[1]/tot[0], tot[2]/tiles, tot[1]/tiles/64))
