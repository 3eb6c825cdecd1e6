#!/usr/bin/env python3
"""CPU estimate for an LDS-staged search: the cell box (+1 margin) spanned by 64 Morton-consecutive
queries, the target points inside it, and how many waves would fit given LDS limits."""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from point_cloud_registration_amd.synthetic import street, perturbed_scan, harness_scan
def spread(v):
    v = v.astype(np.uint64) & np.uint64(0x1fffff)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v
def analyse(target, scan, h, name):
    lo = scan.min(0); ext = (scan.max(0)-lo).max(); sc = 2097151.0/ext
    q = np.clip((scan-lo)*sc, 0, 2097151).astype(np.uint64)
    key = spread(q[:,0]) | (spread(q[:,1])<<np.uint64(1)) | (spread(q[:,2])<<np.uint64(2))
    s = scan[np.argsort(key, kind='stable')]
    tlo = target.min(0)
    dims = (np.floor((target.max(0)-tlo)/h)+2).astype(int)
    tc = np.floor((target-tlo)/h).astype(int)
    cid = (tc[:,2]*dims[1]+tc[:,1])*dims[0]+tc[:,0]
    cnt = np.bincount(cid, minlength=int(np.prod(dims))).reshape(dims[2],dims[1],dims[0])
    # integral image for box sums
    I = np.zeros((dims[2]+1,dims[1]+1,dims[0]+1), np.int64); I[1:,1:,1:] = cnt.cumsum(0).cumsum(1).cumsum(2)
    c = np.floor((s-tlo)/h).astype(int); c = np.clip(c, 0, dims-1)
    n = (len(s)//64)*64; c = c[:n].reshape(-1,64,3)
    mn = np.maximum(c.min(1)-1,0); mx = np.minimum(c.max(1)+1, dims-1)
    b = mx-mn+1
    x0,y0,z0 = mn[:,0],mn[:,1],mn[:,2]; x1,y1,z1 = mx[:,0]+1,mx[:,1]+1,mx[:,2]+1
    tot = (I[z1,y1,x1]-I[z0,y1,x1]-I[z1,y0,x1]-I[z1,y1,x0]+I[z0,y0,x1]+I[z0,y1,x0]+I[z1,y0,x0]-I[z0,y0,x0])
    rows = b[:,1]*b[:,2]
    print(name, "waves", len(b), "median box", np.median(b,0), "median rows", np.median(rows), "median pts", np.median(tot))
    for (BX,ROWS,CAP) in ((16,32,256),(16,32,384),(16,64,384),(32,64,512),(16,64,512)):
        fit = (b[:,0]<=BX)&(rows<=ROWS)&(tot<=CAP)
        print(f"   BX{BX} ROWS{ROWS} CAP{CAP}: fit {fit.mean()*100:.1f}%  (bx ok {np.mean(b[:,0]<=BX)*100:.1f} rows ok {np.mean(rows<=ROWS)*100:.1f} cap ok {np.mean(tot<=CAP)*100:.1f})")
target = street(1_060_000, seed=0)
scan,_ = perturbed_scan(target, None)
analyse(target, scan, 0.405, "b01 full scan")
scan2,_ = perturbed_scan(target, 100_000)
analyse(target, scan2, 0.405, "100k scan")
