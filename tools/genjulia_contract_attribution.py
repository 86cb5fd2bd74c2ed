#!/usr/bin/env python3
"""Is the arithmetic contract's quat_pow (one shared log2, reciprocal products: DESIGN.md section 4) further from the
literal WGSL than any other legal evaluation is?  Three renderings of the generalised-Julia workload's march:
  literal   oracle/kifs_oracle_np.py: quaternions.wgsl:57-63 and gen_julia.wgsl:16 as written, NumPy / libm, no fma
  shared    the contract's FORM (d and |q|^2 once, L = log2(|q|^2) once, products with reciprocals) evaluated with the
            same NumPy / libm primitives -- so literal vs shared isolates the re-association alone
  C oracle  the contract as shipped (that form + pinned polynomial log2 / exp2 / acos / sin / cos + fma chains)
and the number of pixels whose march-step counter differs between each pair.  CPU only.
    python tools/genjulia_contract_attribution.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, dataclasses
import oracle as O
from oracle import kifs_oracle_np as NP
import kifs_raymarching_amd as K
from kifs_raymarching_amd.configs import WORKLOADS
F=np.float32
def quat_pow_shared(q,x):
    with np.errstate(all="ignore"):
        d=NP._dot(q[1:],q[1:]); qs=q[0]*q[0]+d
        L=np.log2(qs).astype(F)
        inv=F(1.0)/np.sqrt(qs); phi=np.arccos(q[0]*inv).astype(F)
        ninv=F(1.0)/np.sqrt(d); n=[c*ninv for c in q[1:]]
        pw=np.exp2(x*(F(0.5)*L)).astype(F); a=x*phi
        cs,sn=np.cos(a).astype(F),np.sin(a).astype(F)
        return [pw*cs]+[pw*(c*sn) for c in n]
def genjulia_sdf_shared(s,p):
    norm=NP._length(p); outside=norm>F(2.0)+s.epsilon; res=norm-F(2.0)
    idx=np.nonzero(~outside)[0]
    if idx.size:
        q=[p[0][idx],p[1][idx],p[2][idx],np.full(idx.size,0.1,dtype=F)]
        qs=NP._dot(q,q); dqs=np.ones(idx.size,dtype=F); live=np.ones(idx.size,dtype=bool)
        pp=s.power*s.power; pm1=s.power-F(1.0)
        with np.errstate(all="ignore"):
            for _ in range(s.sdf_iters):
                if not live.any(): break
                L=np.log2(qs).astype(F)
                dqs=np.where(live,dqs*(pp*np.exp2(pm1*L).astype(F)),dqs)
                nq=NP.quat_add(quat_pow_shared(q,s.power),s.c)
                q=[np.where(live,a,b) for a,b in zip(nq,q)]
                qs=np.where(live,NP._dot(q,q),qs)
                live=live&~(qs>s.max_distance)
            val=F(0.25)*np.log(qs)*np.sqrt(qs/dqs)
        res=res.copy(); res[idx]=val.astype(F)
    return res
base=WORKLOADS["n1_genjulia_1080p"]
ub=K.uniform_bytes
for name,wl in (("power 8",base),("power 3",dataclasses.replace(base,gui=dataclasses.replace(base.gui,power=3.0)))):
    s=O.from_bytes(O.Screen,ub(wl.screen.into_buffer_data())); c=O.from_bytes(O.Camera,ub(wl.camera.into_buffer_data())); o=O.from_bytes(O.Options,ub(wl.gui.into_buffer_data())); it=O.iters(*wl.iters)
    _,i_lit,hit_lit=NP.render(s,c,o,it)
    orig=NP.genjulia_sdf
    NP.genjulia_sdf=genjulia_sdf_shared
    _,i_sh,hit_sh=NP.render(s,c,o,it)
    NP.genjulia_sdf=orig
    _,steps_c,_=O.render_stats(s,c,o,it)
    print(name,"literal vs shared-form (both NumPy/libm): march steps differ on",int((i_lit!=i_sh).sum()),"pixels; hit flips",int((hit_lit!=hit_sh).sum()),
          "| C oracle vs literal:",int((steps_c.astype(int)!=i_lit).sum()),"| C oracle vs shared-form NumPy:",int((steps_c.astype(int)!=i_sh).sum()))
