"""Random-image sweep of the emulated extractor against the oracle (test infrastructure, CPU only).

    python tests/simt/build.py                       # builds tests/simt/build/liborbx_emul.so
    for p in 0 1 2 3 4 5; do python tests/simt/sweep_emul.py $p 6 1800 > /tmp/sweep_$p.log & done

Part `p` of `nparts` takes the images k = 300+p, 300+p+nparts, ... < 300+N: five image shapes, three canvas sizes,
five shape densities, four noise levels, nine crops, with and without the stereo overlap band. Every image goes through
the whole emulated device path (pyramid, FAST, quad-tree, orientation, blur, descriptors) and must give the oracle's
keypoints and descriptors bit for bit. About 5 s per image; the round-2 run (1800 images, 6 processes, 25 min) had no mismatch.
"""
import sys, time, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import orb_slam3_amd._lib as _lib
_lib.LIB_PATH = ROOT / 'tests/simt/build/liborbx_emul.so'
import orb_slam3_amd as osa
from orb_slam3_amd import synth
from oracle import oracle_binding as ob
part=int(sys.argv[1]); nparts=int(sys.argv[2]); N=int(sys.argv[3])
shapes=[(752,480,1000),(640,480,800),(1241,376,2000),(512,512,1000),(1024,768,1500)]
exs={}
bad=0; t0=time.time()
for k in range(300 + part, 300 + N, nparts):
    w,h,nf=shapes[k%len(shapes)]
    seed=500+k; size=(1280,1536,2048)[k%3] if max(w,h)<=1024 else 2560; ns=(200,600,1200,2400,4800)[(k//3)%5]
    rng=np.random.default_rng(seed)
    img=synth.frame_from_canvas(synth.make_canvas(seed,size=size,n_shapes=ns), k%9, w, h, 7000+k, sigma=float(rng.choice([0.0,1.5,3.0,6.0])))
    if (w,h,nf) not in exs: exs[(w,h,nf)]=(osa.ORBextractor(nf,1.2,8,20,7), ob.OracleExtractor(nf,1.2,8,20,7))
    ex,oex=exs[(w,h,nf)]
    lap=(0,1000) if k%2 else (0,0)
    mono,kps,desc=ex(img,None,lap); omono,okps,odesc=oex.extract(img,lap=lap)
    if not (mono==omono and kps.tobytes()==okps.tobytes() and np.array_equal(desc,odesc)):
        bad+=1; print('MISMATCH case',k,(w,h,nf),'seed',seed,'size',size,'shapes',ns,len(kps),len(okps),flush=True)
print('part',part,'done',len(range(300+part,300+N,nparts)),'images, mismatches',bad,round(time.time()-t0),'s',flush=True)
