"""One-off sweep (build container only, needs /root/reference; not collected by pytest): the ORACLE against the REFERENCE'S integrator text compiled live, at 960x540, on seeded random
compositions of the pin scenes' features — Cornell / street / animated street; analytic sphere lights and their proxy meshes, excluded geometry, a mirrored instance, the material zoo,
spec-gloss materials, rotated environments, sun discs with any compression; random settings incl. NEEType 0 / 1 / 2 (NEE-AT with and without tile tables and feedback) and both lp builds.
Round 4: seeds 0..1299 at 960x540 and 2000..2119 at 1920x1080 (SWEEP_W, SWEEP_H), all equal (frames, ray counts, reservoirs).   usage: python tests/sweeps/reference_text_fuzz.py FIRST LAST"""
import sys, time, math, numpy as np
import os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes as ps
W,H=int(os.environ.get("SWEEP_W","960")),int(os.environ.get("SWEEP_H","540"))
def case(seed):
    rng=np.random.default_rng(0xC0FFEE+seed)
    base = rng.integers(0,3)
    if base==0: make=lambda: scenes.cornell_box("C2")
    elif base==1: make=lambda: scenes.bistro_like(scale=float(rng.uniform(0.005,0.03)), seed=scenes.SEED_BASE+700+seed, tex_size=64)
    else: make=lambda: scenes.bistro_like(scale=0.01, seed=scenes.SEED_BASE+900+seed, tex_size=64, animated=True)
    wraps=[]
    if base==0:
        if rng.integers(0,2): make=ps.with_sphere_lights(make); wraps.append("sphere_lights")
        if wraps and rng.integers(0,2): make=ps.with_light_proxy(make); wraps.append("proxy")
        if rng.integers(0,3)==0: make=ps.with_excluded_geometry(make); wraps.append("excluded")
        if rng.integers(0,3)==0: make=ps.with_mirrored_instance(make); wraps.append("mirrored")
    else:
        if rng.integers(0,2): make=ps.with_material_zoo(make); wraps.append("zoo")
    if rng.integers(0,3)==0: make=ps.with_spec_gloss(make); wraps.append("specgloss")
    r=rng.integers(0,4)
    if r==0: make=ps.with_rotated_environment(make); wraps.append("envrot")
    elif r==1: make=ps.with_sun_discs(make, compression=int(rng.integers(0,3))); wraps.append("sun")
    sc,cam=make()
    if base!=0:
        yaw,pitch=rng.uniform(0,2*math.pi), rng.uniform(-0.4,0.5)
        cam=dict(cam,pos=(float(rng.uniform(5,110)),float(rng.uniform(0.5,18.0)),float(rng.uniform(10.0,30.0))),direction=(math.cos(yaw)*math.cos(pitch),math.sin(pitch),math.sin(yaw)*math.cos(pitch)),fov_y=float(rng.uniform(0.5,1.4)))
    lp16=bool(rng.integers(0,2))
    S=scenes.default_settings(bounceCount=int(rng.integers(1,9)),diffuseBounceCount=int(rng.integers(1,9)),NEEType=int(rng.integers(0,3)),NEECandidateSamples=int(rng.integers(1,8)),NEEFullSamples=int(rng.choice([1,1,1,2,3])),
        enableRussianRoulette=int(rng.integers(0,2)),nestedDielectricsQuality=int(rng.integers(0,3)),fireflyFilterThreshold=float(rng.choice([0.0,0.5,2.5])),texLODBias=float(rng.uniform(-2,1)),
        enableLDSamplerForBSDF=int(rng.integers(0,2)),diffuseBrdf=int(rng.choice([0,2])),envMapDiffuseSampleMIPLevel=float(rng.choice([0.0,2.0])),useFp16Types=int(lp16))
    neeat=None
    if int(S["NEEType"])==2:
        fb=bool(rng.integers(0,2)) and int(S["NEEFullSamples"])==1
        neeat=dict(seed=int(rng.integers(1,99)),jitter=(int(rng.integers(0,8)),int(rng.integers(0,8))),ratio=float(rng.choice([0.3,0.65,0.95])),ssc=float(rng.choice([0.0,0.3,1e9])),feedback=fb,tables=bool(rng.integers(0,4)))
    return sc,cam,S,lp16,neeat,int(rng.integers(0,40)),int(rng.integers(1,4)),wraps
bad=[]
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    t0=time.time()
    try:
        sc,cam,S,lp16,neeat,first,n,wraps=case(seed)
        res=[]
        for ref in (False,True):
            o=ptref.Oracle(reference_integrator=True,settings=S,lp16=lp16) if ref else ptref.Oracle(lp16=lp16)
            o.set_scene(sc); o.set_camera(scenes.bridge_camera(W,H,**cam)); o.set_settings(S); o.resize(W,H); o.L.ptref_prepare(o.h)
            if neeat is not None:
                nl=len(o.lights()["lights"])
                tab=scenes.synthetic_local_light_tables(nl,W,H,seed=neeat["seed"],jitter=neeat["jitter"]) if (neeat["tables"] and nl) else None
                o.set_local_light_sampling(tab,jitter=neeat["jitter"],ratio=neeat["ratio"],ssc_threshold=neeat["ssc"],feedback=neeat["feedback"])
            o.render(first,n); c=o.counters(); fb=None
            if neeat is not None and neeat["feedback"]: fb=o.light_feedback(n-1)
            res.append((o.radiance().copy(),(c["extendRays"],c["shadowRays"]),fb)); o.close()
        (a,ra,fa),(b,rb,fbb)=res
        diff=int((a.view(np.uint32)!=b.view(np.uint32)).any(-1).sum())
        fbd=0 if fa is None else int((fa[0].view(np.uint32)!=fbb[0].view(np.uint32)).sum()+(fa[1]!=fbb[1]).sum())
        ok = diff==0 and ra==rb and fbd==0
        print("seed %d %s pixels %d rays %s/%s fb %d  [%s lp16=%s NEEType=%d nee=%s] %.0f s"%(seed,"ok" if ok else "MISMATCH",diff,ra,rb,fbd,",".join(wraps),lp16,int(S["NEEType"]),neeat,time.time()-t0),flush=True)
        if not ok: bad.append(seed)
    except Exception as e:
        print("seed",seed,"ERROR",repr(e)[:300],flush=True)
print("mismatches:",bad)
