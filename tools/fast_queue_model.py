#!/usr/bin/env python3
"""What per-WORKGROUP instead of per-wave candidate queues would save in k_fast_strip: a numpy model of the kernel's stage A
(v_sad_u8 rejection of 4-pixel groups), stage B (four antipodal pairs) and the exact-score stage on the oracle's pyramid levels of three bench frames,
cut into the kernel's strips (<= 7 cells of a cell row) and row bands (4 waves).  Prints, per wave: stage-A iterations, stage-B iterations (64 groups each)
and score iterations (128 pixels each) as they are (ceil per wave), with the queues balanced over the workgroup (ceil per strip) and the fractional ideal.
Round 5: B 3.31 -> 2.94, scores 2.10 -> 1.72 iterations per wave = about 92 of 1224 vector instructions per wave.
usage: python tools/fast_queue_model.py        (CPU only: oracle + numpy, ~1 min)"""
import math
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from orb_slam3_amd import synth
from oracle import oracle_binding as ob
W,H=752,480
frames=synth.make_frames(7, 3, W, H)
tot=dict(A=0,B_now=0,B_bal=0,S_now=0,S_bal=0,waves=0,B_ideal=0.0,S_ideal=0.0, S1_now=0)
for fi in range(3):
    oe=ob.OracleExtractor(1000,1.2,8,20,7)
    oe.extract(frames[fi])
    for lvl in range(8):
        w,h=oe.level_size(lvl)
        pad=oe.level_padded(lvl).astype(np.int32)   # padded by 19
        E=19
        img=pad[E:E+h,E:E+w]
        minBX=16; minBY=16; maxBX=w-16; maxBY=h-16
        width=maxBX-minBX; height=maxBY-minBY
        nCols=width//35; nRows=height//35
        wCell=math.ceil(width/nCols); hCell=math.ceil(height/nRows)
        t=20
        # per pixel maps over whole image (valid region 3..)
        def sh(dx,dy):
            out=np.zeros_like(img); 
            ys=slice(max(0,-dy),h-max(0,dy)); xs=slice(max(0,-dx),w-max(0,dx))
            yd=slice(max(0,dy),h-max(0,-dy)) ; xd=slice(max(0,dx),w-max(0,-dx))
            out[ys,xs]=img[yd,xd]; return out
        c=img
        p0=sh(0,3);p8=sh(0,-3);p4=sh(3,0);p12=sh(-3,0);p2=sh(2,2);p10=sh(-2,-2);p6=sh(2,-2);p14=sh(-2,2)
        mb=np.minimum.reduce([np.maximum(p0,p8),np.maximum(p4,p12),np.maximum(p2,p10),np.maximum(p6,p14)])
        md=np.maximum.reduce([np.minimum(p0,p8),np.minimum(p4,p12),np.minimum(p2,p10),np.minimum(p6,p14)])
        passB=(mb>c+t)|(md<c-t)
        a0=np.abs(c-p0);a8=np.abs(c-p8);a4=np.abs(c-p4);a12=np.abs(c-p12)
        per_strip=max(1,256//wCell)
        for i in range(nRows):
            iniY=minBY+i*hCell
            if iniY>=maxBY-3: continue
            y0=iniY+3; ih=min(hCell, maxBY-3-y0) if iniY+hCell+6>maxBY else hCell
            ih=min(hCell, (min(iniY+hCell+6,maxBY)-3)-y0)
            if ih<=0: continue
            for j0 in range(0,nCols,per_strip):
                nc=min(per_strip,nCols-j0)
                x0=minBX+j0*wCell+3
                x1=min(minBX+(j0+nc)*wCell+3, maxBX-3)
                iw=x1-x0
                if iw<=0: continue
                G=(iw+3)//4
                BH=(ih+3)//4
                gns=[];qns=[]
                for wv in range(4):
                    ya=wv*BH; yb=min(ih,ya+BH)
                    if ya>=yb: gns.append(0);qns.append(0);continue
                    ys=slice(y0+ya,y0+yb)
                    gn=0;qn=0
                    # groups
                    xe=x0+4*G
                    def grp(a):
                        blk=a[ys,x0:min(xe,w)]
                        if blk.shape[1]<4*G: blk=np.pad(blk,((0,0),(0,4*G-blk.shape[1])))
                        return blk.reshape(blk.shape[0],G,4).sum(2)
                    sv=np.maximum(grp(a0),grp(a8)); shh=np.maximum(grp(a4),grp(a12))
                    keep=np.minimum(sv,shh)>t
                    gn=int(keep.sum())
                    pb=passB[ys,x0:x1]
                    kp=np.repeat(keep,4,axis=1)[:,:iw]
                    qn=int((pb&kp).sum())
                    gns.append(gn);qns.append(qn)
                    tot['A']+=math.ceil((yb-ya)/max(1,64//G))
                tot['waves']+=4
                tot['B_now']+=sum(math.ceil(g/64) for g in gns)
                tot['B_bal']+=math.ceil(sum(gns)/64)
                tot['B_ideal']+=sum(gns)/64
                tot['S_now']+=sum(math.ceil(q/128) for q in qns)
                tot['S_bal']+=math.ceil(sum(qns)/128)
                tot['S_ideal']+=sum(qns)/128
                tot['S1_now']+=sum(math.ceil(q/64) for q in qns)
print(tot)
wv=tot['waves']
for k in ('A','B_now','B_bal','B_ideal','S_now','S_bal','S_ideal','S1_now'): print(k, tot[k]/wv)
