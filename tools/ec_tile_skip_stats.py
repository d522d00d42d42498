"""How many 8x8 tiles of the pre-correction field could skip stages of k_ec_fast wave-uniformly (costed before writing the kernel code):
  * protectEdges emits nothing if every texel of the tile's 10x10 halo has |median - .5| >= radius/2   (MSDFErrorCorrection.cpp:201, :217, :233)
  * findErrors emits nothing if, for each channel pair, the difference has one strict sign with a margin over the whole halo, or is zero everywhere
Compiled reference (oracle/_ref), every 16th glyph of the distinct DejaVu set at 64x64.   python tools/ec_tile_skip_stats.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Ref
from msdfgen_amd.shape import ShapeBatch
z=np.load(os.path.join(ROOT, 'tests', 'golden', 'dejavu8192.npz'))
batch=ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32), z["colors"].astype(np.int32), np.zeros(len(z["names"]),bool), [str(n) for n in z["names"]])
xfs=z["xf64"]
ref=Ref()
idx=list(range(0,8192,16))
shapes=[batch.shape(i) for i in idx]
out,secs=ref.generate_batch_timed(shapes,3,64,64,xfs[idx],ec_mode=0,threads=16)
print(out.shape, secs)
def med(a): return np.median(a,axis=-1)
tot=0; skipP=0; skipF=0; skipBoth=0; skipF_relaxed=0
for gi,g in enumerate(idx):
    xf=xfs[g]; dm=1/(xf[5]-xf[4])
    rH=1.001*abs(dm/xf[0]); rV=1.001*abs(dm/xf[1]); rD=1.001*np.hypot(dm/xf[0],dm/xf[1])
    rmax=np.float32(max(rH,rV,rD))
    f=out[gi]
    dev=np.abs(med(f)-np.float32(.5))
    d=[f[...,1]-f[...,0], f[...,2]-f[...,1], f[...,0]-f[...,2]]
    for ty in range(8):
        for tx in range(8):
            y0,y1=max(ty*8-1,0),min(ty*8+9,64); x0,x1=max(tx*8-1,0),min(tx*8+9,64)
            tot+=1
            sp = (dev[y0:y1,x0:x1] >= rmax*np.float32(.5)).all()
            sf=True; sfr=True
            for j in range(3):
                t=d[j][y0:y1,x0:x1]
                ok = (t==0).all() or ((t>2e-3)&(t<=100)).all() or ((t<-2e-3)&(t>=-100)).all()
                okr = (t==0).all() or (t>0).all() or (t<0).all()
                sf &= ok; sfr &= okr
            skipP+=sp; skipF+=sf; skipBoth+= (sp and sf); skipF_relaxed+=sfr
print("tiles",tot,"skipProtect %.3f skipFind %.3f (sign-only %.3f) both %.3f"%(skipP/tot,skipF/tot,skipF_relaxed/tot,skipBoth/tot))
