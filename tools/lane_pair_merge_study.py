#!/usr/bin/env python3
"""CPU model (as tools/lane_study.py): workgroups of 2 / 4 / 8 tiles in which a tile whose queue has fallen to a
threshold pours its rays into the group's pool (VERDICT r02's proposal for the headline kernel), against one tile per
wave and against one pool per group from the start.    python tools/lane_pair_merge_study.py [workload]"""
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
tx=(W+31)//32
tile=(ys.ravel()//8)*tx+xs.ravel()//32
ntiles=tile.max()+1
R=16
# schemes: name -> (group size G of horizontally adjacent tiles, merge threshold thr)
SCH={"one tile per wave":(1,0),"pairs, merge when a queue <= 32":(2,32),"pairs, merge when a queue <= 48":(2,48),"pairs, always one pool":(2,10**9),
     "quads, merge when <= 32":(4,32),"quads, always one pool":(4,10**9), "8 tiles, merge when <= 32":(8,32)}
gid={k:(tile//tx)*tx + (tile%tx)//g*g for k,(g,_) in SCH.items()}   # group id per pixel (first tile of the group)
wave={k:np.full(n,-1,dtype=np.int64) for k in SCH}
cost={k:0.0 for k in SCH}; over={k:0.0 for k in SCH}
merged={k:np.zeros(ntiles,dtype=bool) for k in SCH}  # per tile: already poured into its group's pool
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
    if step%R==0:
        cnt=np.bincount(tl,minlength=ntiles)
        for k,(g,thr) in SCH.items():
            if g>1:
                merged[k]|=(cnt<=thr)&(cnt>0)   # a tile whose queue is small pours it into the group's pool (for good)
            pool=np.where(merged[k][tl],gid[k][idx]+10**7,tl)   # pooled rays share their group's waves; others stay per tile
            order=np.lexsort([idx,pool]); t2=pool[order]
            first=np.r_[0,np.nonzero(np.diff(t2))[0]+1]; start=np.zeros(t2.size,dtype=np.int64); start[first]=first; start=np.maximum.accumulate(start)
            wid=t2*1000+(np.arange(t2.size)-start)//64
            wave[k][idx[order]]=wid
            over[k]+=64.0*OVER*np.unique(wid).size
    for k in SCH:
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
base=cost["one tile per wave"]+over["one tile per wave"]
for k in SCH: print(f"{k:36s} march {cost[k]/useful:.3f} + rounds {over[k]/useful:.3f} = {(cost[k]+over[k])/useful:.3f} x useful;  {base/(cost[k]+over[k]):.3f} x fewer cycles")
