#!/usr/bin/env python3
"""CPU replay of the generalised-Julia workload's march: orbit trips per estimate (is there a tail of long orbits
that regrouping could collect?) and what capping a wave's trips would save.    python tools/genjulia_orbit_study.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import oracle as O
from oracle import kifs_oracle_np as NP
import kifs_raymarching_amd as K
from kifs_raymarching_amd.configs import WORKLOADS
F=np.float32
TRIP,TAIL,OUT=420.0,250.0,45.0
w=WORKLOADS["n1_genjulia_1080p"]; ub=K.uniform_bytes
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
step=0; useful=0; lock=0; lock_capped={4:0,8:0,12:0,16:0}; hist=np.zeros(101)
wave_of=np.full(n,-1,dtype=np.int64)
# what if a tile's rays were cut into waves by their LAST step's trip count (known at filing time) instead of pixel order?
ROUND=int(sys.argv[1]) if len(sys.argv)>1 else 16
wave_sorted=np.full(n,-1,dtype=np.int64); last_trips=np.zeros(n,dtype=np.int32); lock_sorted=0
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
            factor=s.power*s.power*np.power(qs,s.power-F(1.0)).astype(F)
            dqs=np.where(alive,dqs*factor,dqs)
            nq=NP.quat_add(NP.quat_pow(q,s.power),s.c)
            q=[np.where(alive,a,c) for a,c in zip(nq,q)]
            qs=np.where(alive,NP._dot(q,q),qs)
            alive=alive&~(qs>s.max_distance)
        dist=norm-F(2.0)
        dist[ins]=(F(0.25)*np.log(qs)*np.sqrt(qs/dqs)).astype(F)
    hist+=np.bincount(trips[~outside],minlength=101)[:101]
    useful+=float(np.where(outside,OUT,TRIP*trips+TAIL).sum())
    if step%ROUND==0:
        order=np.lexsort([idx,tile[idx]]); tl=tile[idx][order]
        first=np.r_[0,np.nonzero(np.diff(tl))[0]+1]; start=np.zeros(tl.size,dtype=np.int64); start[first]=first; start=np.maximum.accumulate(start)
        wid=(np.cumsum(np.r_[True,tl[1:]!=tl[:-1]])-1)*1000+(np.arange(tl.size)-start)//64
        wave_of[idx[order]]=wid
        order=np.lexsort([idx,-last_trips[idx],tile[idx]]); tl=tile[idx][order]
        first=np.r_[0,np.nonzero(np.diff(tl))[0]+1]; start=np.zeros(tl.size,dtype=np.int64); start[first]=first; start=np.maximum.accumulate(start)
        wave_sorted[idx[order]]=(np.cumsum(np.r_[True,tl[1:]!=tl[:-1]])-1)*1000+(np.arange(tl.size)-start)//64
    _,inv=np.unique(wave_of[idx],return_inverse=True)
    mt=np.zeros(inv.max()+1); np.maximum.at(mt,inv,trips)
    anyin=np.zeros(inv.max()+1,dtype=bool); np.logical_or.at(anyin,inv,~outside)
    anyout=np.zeros(inv.max()+1,dtype=bool); np.logical_or.at(anyout,inv,outside)
    lock+=float((64*(np.where(anyin,TRIP*mt+TAIL,0)+np.where(anyout,OUT,0))).sum())
    _,inv2=np.unique(wave_sorted[idx],return_inverse=True)
    mt2=np.zeros(inv2.max()+1); np.maximum.at(mt2,inv2,trips)
    ain=np.zeros(inv2.max()+1,dtype=bool); np.logical_or.at(ain,inv2,~outside)
    aout=np.zeros(inv2.max()+1,dtype=bool); np.logical_or.at(aout,inv2,outside)
    lock_sorted+=float((64*(np.where(ain,TRIP*mt2+TAIL,0)+np.where(aout,OUT,0))).sum())
    last_trips[idx]=np.where(outside,-1,trips)
    for cap in lock_capped:
        # trips beyond `cap` are done elsewhere at full lane utilisation: wave pays min(mt,cap); excess lane-trips paid at 64/64
        excess=np.maximum(trips-cap,0).sum()
        lock_capped[cap]+=float((64*(np.where(anyin,TRIP*np.minimum(mt,cap)+TAIL,0)+np.where(anyout,OUT,0))).sum())+TRIP*excess
    with np.errstate(invalid="ignore"): hit=dist<s.epsilon
    go=idx[~hit]; t[go]=t[go]+dist[~hit]
    for k in range(3): pos[k][go]=o[k]+t[go]*dirv[k][go]
    live[idx[hit]]=False
    pg=[pos[k][go] for k in range(3)]
    leaving=(NP._dot(pg,pg)>R2)&(NP._dot(pg,[dirv[k][go] for k in range(3)])>0)
    with np.errstate(invalid="ignore"): live[go]=(t[go]<s.max_distance)&~leaving
    step+=1
print("steps",step,"rounds of",ROUND,"lockstep/useful",lock/useful)
print("waves cut by the last step's trip count: cost/useful",lock_sorted/useful,"gain",lock/lock_sorted)
for cap,v in lock_capped.items(): print("cap",cap,"cost/useful",v/useful,"gain",lock/v)
tot=hist.sum(); cum=np.cumsum(hist*np.arange(101))
print("inside evaluations",int(tot),"mean trips",float((hist*np.arange(101)).sum()/tot))
for k in (2,3,4,6,8,12,16,32,64,99,100): print("trips<=",k,"evals",hist[:k+1].sum()/tot,"trip-work share",cum[k]/cum[-1])
