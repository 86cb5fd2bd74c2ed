#!/usr/bin/env python3
"""CPU model (as tools/lane_study.py): round length of render_wave_kernel chosen by the tile's queue population
(short rounds while a tile has more than one wave of rays, long ones after), priced with the rounds' own overhead.
    python tools/lane_round_policy_study.py [workload]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import oracle as O
from oracle import kifs_oracle_np as NP
import kifs_raymarching_amd as K
from kifs_raymarching_amd.configs import WORKLOADS
F=np.float32
TRIP,TAIL,OUT,OVER=34.0,180.0,45.0,250.0
key=sys.argv[1] if len(sys.argv)>1 else "cfg2_julia_1080p"
w=WORKLOADS[key]; ub=K.uniform_bytes
s=NP.Scene(O.from_bytes(O.Screen,ub(w.screen.into_buffer_data())),O.from_bytes(O.Camera,ub(w.camera.into_buffer_data())),O.from_bytes(O.Options,ub(w.gui.into_buffer_data())),O.iters(*w.iters))
W,H=s.width,s.height
ys,xs=np.mgrid[0:H,0:W]
px,py=xs.ravel().astype(F)+F(0.5),ys.ravel().astype(F)+F(0.5)
uvx,uvy=F(2.0)*px/s.h-s.aspect,F(2.0)*py/s.h-F(1.0)
d=[uvx*s.m[1][k]-uvy*s.m[2][k]-s.m[0][k] for k in range(3)]
dirv=NP._normalize(d); o=s.origin
R2=F(1.1)*(F(2.0)+s.epsilon)**2
oo=sum(c*c for c in o); b=-(o[0]*dirv[0]+o[1]*dirv[1]+o[2]*dirv[2])
never=np.where(b<=0,oo>R2,(oo-b*b)>R2); live=~never
n=W*H; t=np.zeros(n,dtype=F); pos=[np.full(n,o[k],dtype=F) for k in range(3)]
tile=(ys.ravel()//8)*((W+31)//32)+xs.ravel()//32
ntiles=tile.max()+1
# policies: name -> function(n_live_in_tile) -> round length
POL={"const16":lambda m:np.full(m.shape,16),"const8":lambda m:np.full(m.shape,8),
     "8 if >64 else 16":lambda m:np.where(m>64,8,16),"8 if >64 else 32":lambda m:np.where(m>64,8,32),
     "4 if >128, 8 if >64, else 16":lambda m:np.where(m>128,4,np.where(m>64,8,16)),
     "8 if >64 else 24":lambda m:np.where(m>64,8,24),"12 if >64 else 32":lambda m:np.where(m>64,12,32),
     "16 if >64 else 32":lambda m:np.where(m>64,16,32),"6 if >64 else 16":lambda m:np.where(m>64,6,16)}
nxt={k:np.zeros(ntiles,dtype=np.int64) for k in POL}
wave={k:np.full(n,-1,dtype=np.int64) for k in POL}
cost={k:0.0 for k in POL}; over={k:0.0 for k in POL}
useful=0.0; step=0
while live.any() and step<s.max_iterations:
    idx=np.nonzero(live)[0]
    p=[c[idx] for c in pos]
    norm=NP._length(p); outside=norm>F(2.0)+s.epsilon
    trips=np.zeros(idx.size,dtype=np.int32)
    ins=np.nonzero(~outside)[0]
    q=[p[0][ins],p[1][ins],p[2][ins],np.full(ins.size,0.1,dtype=F)]
    qs=NP._dot(q,q); dqs=np.ones(ins.size,dtype=F); alive=np.ones(ins.size,dtype=bool)
    with np.errstate(all="ignore"):
        for _ in range(s.sdf_iters):
            if not alive.any(): break
            trips[ins[alive]]+=1
            dqs=np.where(alive,dqs*(F(4.0)*qs),dqs)
            nq=NP.quat_add(NP.quat_sq(q),s.c)
            q=[np.where(alive,a,c) for a,c in zip(nq,q)]
            qs=np.where(alive,NP._dot(q,q),qs)
            alive=alive&~(qs>s.max_distance)
        dist=norm-F(2.0)
        dist[ins]=(F(0.25)*np.log(qs)*np.sqrt(qs/dqs)).astype(F)
    useful+=float(np.where(outside,OUT,TRIP*trips+TAIL).sum())
    tl=tile[idx]
    cnt=np.bincount(tl,minlength=ntiles)
    for k,f in POL.items():
        due=nxt[k]<=step            # tiles whose round starts now
        sel=due[tl]
        if sel.any():
            ii=idx[sel]; tt=tl[sel]
            order=np.lexsort([ii,tt]); t2=tt[order]
            first=np.r_[0,np.nonzero(np.diff(t2))[0]+1]; start=np.zeros(t2.size,dtype=np.int64); start[first]=first; start=np.maximum.accumulate(start)
            wid=t2*1000+(np.arange(t2.size)-start)//64
            wave[k][ii[order]]=wid
            over[k]+=64.0*OVER*np.unique(wid).size
            nxt[k][due]=step+f(cnt[due])
        _,inv=np.unique(wave[k][idx],return_inverse=True)
        mt=np.zeros(inv.max()+1); np.maximum.at(mt,inv,trips)
        anyin=np.zeros(inv.max()+1,dtype=bool); np.logical_or.at(anyin,inv,~outside)
        anyout=np.zeros(inv.max()+1,dtype=bool); np.logical_or.at(anyout,inv,outside)
        cost[k]+=float((64*(np.where(anyin,TRIP*mt+TAIL,0)+np.where(anyout,OUT,0))).sum())
    with np.errstate(invalid="ignore"): hit=dist<s.epsilon
    go=idx[~hit]; t[go]=t[go]+dist[~hit]
    for k in range(3): pos[k][go]=o[k]+t[go]*dirv[k][go]
    live[idx[hit]]=False
    pg=[pos[k][go] for k in range(3)]
    leaving=(NP._dot(pg,pg)>R2)&(NP._dot(pg,[dirv[k][go] for k in range(3)])>0)
    with np.errstate(invalid="ignore"): live[go]=(t[go]<s.max_distance)&~leaving
    step+=1
base=cost["const16"]+over["const16"]
for k in POL: print(f"{k:32s} march {cost[k]/useful:.3f} + rounds {over[k]/useful:.3f} = {(cost[k]+over[k])/useful:.3f} x useful;  {base/(cost[k]+over[k]):.3f} x vs const16")
