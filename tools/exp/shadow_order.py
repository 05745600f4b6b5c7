#!/usr/bin/env python3
"""Would another visiting order help the shadow rays?  The any-hit answer does not depend on the order (boxes are culled
against the fixed [t_min, t_max] of the ray, a triangle is accepted or not on its own), so a non-counting build would be
free to choose one.  Counts node visits of random surface-to-light segments in one scene under the reference's order
(left child first), the mirrored order and near-child-first.  CPU only.  usage: shadow_order.py [scene] [rays]"""
import sys, numpy as np
import os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import webgpu_raytracer_amd as W, parity_util as pu
b = pu.bridge_for(W, sys.argv[1] if len(sys.argv)>1 else "sponza_like")
blas = np.asarray(b.blas).reshape(-1,8); u = blas.view(np.uint32)
inst = np.asarray(b.instances).reshape(-1,36)
topo = np.asarray(b.mesh_topology).reshape(-1,20)
V = np.asarray(b.vertices).reshape(-1,4)[:,:3]
print("instances", len(inst), "blas nodes", len(blas), "tris", len(topo))
# use instance 0 only if identity; find the biggest BLAS
iu = inst.view(np.uint32)
roots = iu[:,32]
print("roots", roots[:8])
# pick instance whose BLAS is largest: assume instance 0
M = inst[0,:16].reshape(4,4).T
print(M)
root = int(roots[0])
tris = V[topo[:,0:3].astype(np.int64)]  # (n,3,3)
lights = np.asarray(b.lights).reshape(-1,2)
rng = np.random.default_rng(3)
def tri_point(t):
    r1,r2 = rng.random(2); s=np.sqrt(r1); return t[0]*(1-s)+t[1]*(s*(1-r2))+t[2]*(s*r2)
def hit_tri(o,d,t,tmin,tmax):
    e1=t[1]-t[0]; e2=t[2]-t[0]; h=np.cross(d,e2); a=e1@h
    if abs(a)<1e-6: return False
    f=1/a; s=o-t[0]; uu=f*(s@h)
    if uu<0 or uu>1: return False
    q=np.cross(s,e1); vv=f*(d@q)
    if vv<0 or uu+vv>1: return False
    tt=f*(e2@q); return tmin<tt<tmax
def box(n,o,inv,tmin,tmax):
    t1=(blas[n,0:3]-o)*inv; t2=(blas[n,4:7]-o)*inv
    return max(tmin,np.minimum(t1,t2).max()) <= min(tmax,np.maximum(t1,t2).min())
def traverse(o,d,tmax,order):
    inv=1/d; cnt=0; stack=[root]
    while stack:
        n=stack.pop(); cnt+=1
        if not box(n,o,inv,1e-3,tmax): continue
        data=int(u[n,7]); c=data&7
        if c>0:
            first=data>>3
            for k in range(c):
                if hit_tri(o,d,tris[first+k],1e-3,tmax): return cnt,True
        else:
            l=n+1; r=root+int(u[l,3]) if False else None
            # children: left = n+1, right = skip target of left (local skip relative to blas start of this geometry)
            r=int(u[l,3])+base
            if order=="ref": stack.append(r); stack.append(l)
            elif order=="mirror": stack.append(l); stack.append(r)
            else:  # near first by box centre along d
                cl=(blas[l,0:3]+blas[l,4:7])@d; cr=(blas[r,0:3]+blas[r,4:7])@d
                if cl<=cr: stack.append(r); stack.append(l)
                else: stack.append(l); stack.append(r)
    return cnt,False
base=root
res={k:[0,0] for k in ("ref","mirror","near")}
occl=0; N=int(sys.argv[2]) if len(sys.argv)>2 else 600
for i in range(N):
    ti=rng.integers(0,len(tris)); o=tri_point(tris[ti])
    li=lights[rng.integers(0,len(lights))]; p=tri_point(tris[li[1]])
    d=p-o; dist=np.linalg.norm(d); d=d/dist; d=np.where(d==0,1e-9,d)
    o2=o+d*1e-3
    h=None
    for k in res:
        c,hit=traverse(o2,d,dist-2e-3,k); res[k][0]+=c; res[k][1]+=hit
    occl+=res["ref"][1]>0 and 0
print({k:(v[0]/N, v[1]/N) for k,v in res.items()})
