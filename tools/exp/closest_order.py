#!/usr/bin/env python3
"""What does the reference's fixed visiting order (left child first, whatever the ray's direction) cost the closest-hit
rays?  Node visits and triangle tests of random diffuse rays leaving random surface points of the first BLAS of a scene,
under the reference's order and under near-child-first with the same running bound.  CPU only, first instance only.
usage: closest_order.py [scene] [rays]"""
import sys, os, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import webgpu_raytracer_amd as W, parity_util as pu
b = pu.bridge_for(W, sys.argv[1] if len(sys.argv)>1 else "sponza_like")
blas = np.asarray(b.blas).reshape(-1,8); u = blas.view(np.uint32)
topo = np.asarray(b.mesh_topology).reshape(-1,20)
V = np.asarray(b.vertices).reshape(-1,4)[:,:3]
tris = V[topo[:,0:3].astype(np.int64)]
root=0; base=0
rng = np.random.default_rng(5)
def tri_point(t):
    r1,r2 = rng.random(2); s=np.sqrt(r1); return t[0]*(1-s)+t[1]*(s*(1-r2))+t[2]*(s*r2)
def hit_tri(o,d,t):
    e1=t[1]-t[0]; e2=t[2]-t[0]; h=np.cross(d,e2); a=e1@h
    if abs(a)<1e-6: return None
    f=1/a; s=o-t[0]; uu=f*(s@h)
    if uu<0 or uu>1: return None
    q=np.cross(s,e1); vv=f*(d@q)
    if vv<0 or uu+vv>1: return None
    return f*(e2@q)
def box(n,o,inv,tmin,tmax):
    t1=(blas[n,0:3]-o)*inv; t2=(blas[n,4:7]-o)*inv
    near=max(tmin,np.minimum(t1,t2).max()); far=min(tmax,np.maximum(t1,t2).min())
    return near<=far, near
def traverse(o,d,order):
    inv=1/d; cnt=0; tcnt=0; stack=[root]; closest=1e30; best=-1; maxstack=0
    while stack:
        maxstack=max(maxstack,len(stack))
        n=stack.pop(); cnt+=1
        ok,near=box(n,o,inv,1e-3,closest)
        if not ok: continue
        data=int(u[n,7]); c=data&7
        if c>0:
            first=data>>3
            for k in range(c):
                tcnt+=1
                t=hit_tri(o,d,tris[first+k])
                if t is not None and 1e-3<t<closest: closest=t; best=first+k
        else:
            l=n+1; r=int(u[l,3])+base
            if order=="ref": stack.append(r); stack.append(l)
            else:
                okl,nl=box(l,o,inv,1e-3,closest); okr,nr=box(r,o,inv,1e-3,closest)
                # near-first using entry distances (children tested again when popped: count once here instead)
                if nl<=nr: stack.append(r); stack.append(l)
                else: stack.append(l); stack.append(r)
    return cnt,tcnt,best,maxstack
N=int(sys.argv[2]) if len(sys.argv)>2 else 300
acc={"ref":[0,0,0],"near":[0,0,0]}; same=0
for i in range(N):
    ti=rng.integers(0,len(tris)); t=tris[ti]; o=tri_point(t)
    n=np.cross(t[1]-t[0],t[2]-t[0]); n/=np.linalg.norm(n)+1e-30
    d=rng.normal(size=3); d/=np.linalg.norm(d)
    if d@n<0: d=-d
    d=np.where(d==0,1e-9,d); o2=o+n*1e-4
    r={}
    for k in acc:
        c,tc,bst,ms=traverse(o2,d,k); acc[k][0]+=c; acc[k][1]+=tc; acc[k][2]=max(acc[k][2],ms); r[k]=bst
    same+= r["ref"]==r["near"]
print({k:(v[0]/N,v[1]/N,v[2]) for k,v in acc.items()}, "same result", same, "/", N)
