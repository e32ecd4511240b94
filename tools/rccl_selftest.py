"""developer tool: drives pt_comm_init / pt_gather with WORLD ranks. On a 1-GPU box both ranks sit on device 0, which RCCL may refuse ("duplicate GPU");
on a multi-GPU node (rank r -> device r) this is the real path. usage: python tools/rccl_selftest.py [world] [same_device 0/1]"""
import os, sys, socket
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp


def worker(rank, world, same_device, idfile, q):
    try:
        import time
        import rtxpt_amd as pt
        from rtxpt_amd import scenes
        dev = 0 if same_device else rank
        if rank == 0:
            uid = pt.comm_unique_id()
            with open(idfile + ".tmp", "wb") as f:
                f.write(uid)
            os.rename(idfile + ".tmp", idfile)
        else:
            while not os.path.exists(idfile):
                time.sleep(0.05)
            uid = open(idfile, "rb").read()
        sc, cam = scenes.cornell_box("C2")
        w, h = 200, 136
        g = pt.PathTracer(device=dev, shard_rank=rank, shard_count=world)
        g.set_scene(sc); g.set_camera(scenes.bridge_camera(w, h, **cam)); g.set_settings(scenes.config_settings("C2")); g.resize(w, h)
        g.comm_init(uid, rank, world)
        g.render(0, 2); g.gather()
        img = g.radiance()
        if rank == 0:
            full = pt.PathTracer(device=dev); full.set_scene(sc); full.set_camera(scenes.bridge_camera(w, h, **cam)); full.set_settings(scenes.config_settings("C2")); full.resize(w, h); full.render(0, 2)
            q.put(("ok", bool(np.array_equal(img, full.radiance()))))
        g.comm_destroy()
    except Exception as e:
        q.put(("error rank %d" % rank, repr(e)))


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    same = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    idfile = "/tmp/mi355pt_comm_id_%d" % os.getpid()
    procs = [ctx.Process(target=worker, args=(r, world, same, idfile, q)) for r in range(world)]
    for p in procs: p.start()
    try:
        print("result:", q.get(timeout=120))
    except Exception as e:
        print("no result:", repr(e))
    for p in procs:
        p.join(20)
        if p.is_alive(): p.terminate()
    print("exit codes", [p.exitcode for p in procs])
