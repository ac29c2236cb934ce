#!/usr/bin/env python3
"""Dev tool (CPU only, round 6): census of the queries of a drive pair by the distance of their second ring (who asks for cells beyond their own
cell +- 1 in a wide-gate association round), and what a per-coarse-cell ring mask could prove for the queries with a single ring inside the gate.
scipy KD-trees per target ring on the pair tools/round_profile.py replays.  -> profiles/r06_far_query_census.txt"""
import sys, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import velo_amd
from velo_amd import synth
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation as Rot
k=3
plan=synth.drive(k+2, seed=0)
(txyz,toff),(sxyz,soff)=plan["frames"][k],plan["frames"][k+1]
x=np.asarray(plan["x_true"][k])
R=Rot.from_rotvec(x[:3]).as_matrix()
q=(sxyz[:,:3].astype(np.float64)@R.T+x[3:]).astype(np.float32)
t=txyz[:,:3]
nr=len(toff)-1
ring_of=np.repeat(np.arange(nr),np.diff(toff))
G=0.5**0.5
d=np.full((len(q),nr),np.inf,np.float32)
for s in range(nr):
    tr=cKDTree(t[toff[s]:toff[s+1]])
    dd,_=tr.query(q,k=1,distance_upper_bound=G*1.01)
    d[:,s]=dd
ds=np.sort(d,axis=1)
d1,d2=ds[:,0],ds[:,1]
h=0.1785
print("no ring in gate: %.3f"%np.mean(d1>G), " one ring only (type a): %.3f"%np.mean((d1<=G)&(d2>G)), " second ring in (h,G] (type b): %.3f"%np.mean((d2<=G)&(d2>h)), " second ring within h: %.3f"%np.mean(d2<=h))
# coarse ring-mask test for type a: box(q, G) in coarse cells of size c: all target points in those cells belong to ring r1?
a=(d1<=G)&(d2>G)
r1=np.argmin(d,axis=1)
for c in (4*h, 2*h, 3*h, 8*h):
    o=t.min(0)
    cc=np.floor((t-o)/c).astype(np.int64)
    n=cc.max(0)+1
    key=(cc[:,2]*n[1]+cc[:,1])*n[0]+cc[:,0]
    # per cell: mask as python dict of sets -> use sorting: cell -> (min ring, max ring) pair; "only ring r" iff min==max==r
    order=np.argsort(key); ks=key[order]; rs=ring_of[order]
    uk,start=np.unique(ks,return_index=True)
    mn=np.minimum.reduceat(rs,start); mx=np.maximum.reduceat(rs,start)
    cell_min=dict(zip(uk.tolist(),mn.tolist())); cell_max=dict(zip(uk.tolist(),mx.tolist()))
    idx=np.nonzero(a)[0]
    passed=0; ncell=0
    for i in idx[::7]:
        lo=np.floor((q[i]-G-o)/c).astype(int); hi=np.floor((q[i]+G-o)/c).astype(int)
        ok=True
        for z in range(lo[2],hi[2]+1):
            for y in range(lo[1],hi[1]+1):
                for xx in range(lo[0],hi[0]+1):
                    ncell+=1
                    kk=(z*n[1]+y)*n[0]+xx
                    if kk in cell_min and not (cell_min[kk]==r1[i] and cell_max[kk]==r1[i]): ok=False
        passed+=ok
    print("coarse cell %.3f m: type-a queries proven closed: %.3f (cells per query %.1f)"%(c,passed/len(idx[::7]),ncell/len(idx[::7])))
# per group of 64 in patch order? approximate with ring order groups: fraction of groups containing a type a / type b member
for name,m in (("a",a),("b",(d2<=G)&(d2>h))):
    g=m[:len(m)//64*64].reshape(-1,64).any(1)
    print("groups (ring order) with a type",name,"member: %.3f"%g.mean())
