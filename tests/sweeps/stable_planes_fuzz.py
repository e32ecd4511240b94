"""One-off sweep (build container only, needs /root/reference; not collected by pytest): the ORACLE's stable-plane passes against the REFERENCE'S text compiled live, at 960x540, on
tests/fuzz_cases.stable_planes_case(1000 + seed): random viewpoints, plane counts, vertex depths, settings, previous poses, 1-3 fill sub-samples — every plane buffer and all live
plane records after the build pass and after the fill passes, ray counts. Round 4: seeds 0..699, all equal.   usage: python tests/sweeps/stable_planes_fuzz.py FIRST LAST"""
import sys, time, numpy as np
import os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'tests','golden'))
from rtxpt_amd import scenes
from oracle import ptref
import fuzz_cases as fz
fz.W, fz.H = 960, 540
KEYS=("header","depth","motion_vectors","stable_radiance","throughput","spec_hit_t")
def live(frame):
    hd=frame["header"]; P=frame["planes"].reshape(-1,20); rows=[]
    for pl in range(3):
        ys,xs=np.nonzero(hd[pl]!=0xFFFFFFFF)
        rows.append(P[np.sort(scenes.stable_planes_address(xs.astype(np.int64),ys.astype(np.int64),pl,fz.W,fz.H))])
    return np.concatenate(rows)
bad=[]
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    t0=time.time()
    try:
        sc,camd,S,prm,lp16,prev_pose,sample,subs=fz.stable_planes_case(1000+seed)
        out=[]
        for ref in (False,True):
            b=ptref.Oracle(reference_integrator=True,settings=S,lp16=lp16,mode=1) if ref else ptref.Oracle(lp16=lp16)
            b.set_scene(sc); b.set_camera(camd); b.set_settings(S); b.resize(fz.W,fz.H)
            if prev_pose is not None: b.set_previous_pose(*prev_pose)
            frame=b.build_stable_planes(sample,prm); rb=b.counters()["extendRays"]
            snap={k:np.array(frame[k]).copy() for k in KEYS}; snap["live"]=live(frame)
            f=ptref.Oracle(reference_integrator=True,settings=S,lp16=lp16,mode=2) if ref else b
            if ref: f.set_scene(sc); f.set_camera(camd); f.set_settings(S); f.resize(fz.W,fz.H)
            c0=f.counters()
            for s in range(subs): f.fill_stable_planes(sample+s,prm,frame)
            c=f.counters()
            fin={k:np.array(frame[k]).copy() for k in KEYS}; fin["live"]=live(frame)
            out.append((snap,fin,rb,(c["extendRays"]-(0 if ref else rb),c["shadowRays"])))
            b.close(); 
            if ref: f.close()
        (s0,f0,r0,c0),(s1,f1,r1,c1)=out
        diffs=[k for k in s0 if not np.array_equal(s0[k].view(np.uint8),s1[k].view(np.uint8))]+["fill_"+k for k in f0 if not np.array_equal(f0[k].view(np.uint8),f1[k].view(np.uint8))]
        ok=not diffs and r0==r1 and c0==c1
        print("seed %d %s %s build rays %s/%s fill %s/%s subs %d lp16 %s %.0f s"%(seed,"ok" if ok else "MISMATCH",diffs,r0,r1,c0,c1,subs,lp16,time.time()-t0),flush=True)
        if not ok: bad.append(seed)
    except Exception as e:
        print("seed",seed,"ERROR",repr(e)[:300],flush=True)
print("mismatches:",bad)
