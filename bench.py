#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path (BASELINE.json metric: Mrays/s + ms/frame at 4K, 4 spp, 8 bounces, Bistro).

A "step" is one pt_render() of the whole frame: generate -> {extend, shade, shadow}* -> accumulate for 4 accumulated
samples of the 3840x2160 bistro-like scene (the real Bistro asset is unobtainable offline), plus, for
N > 1, the single gather of the packed radiance tiles to rank 0 (RCCL over xGMI). Scene, BVH, textures and light tables are
resident in HBM before the timed region; nothing is uploaded inside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line. `value` = (extend + shadow rays of all ranks) / wall time of the K timed steps (max over ranks).
The frame is fixed as N grows (pixel-tile sharding) => "scaling": "strong".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s achievable
# algorithmic bytes: extend ray 52 B fixed + BVH: 128 B per BVH8 node visit + 48 B per triangle test
B_EXTEND_FIXED, B_NODE, B_TRI, B_SHADE, B_SHADOW_FIXED = 52.0, 128.0, 48.0, 656.0, 80.0
def _newest_counters():
    """the newest committed end-of-round counter summary, profiles/rNNz_counters.json (tools/profile_round.sh): quoted in roofline{} — only if it was collected on the sources this library was built from"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]z_counters.json")))
    return os.path.basename(c[-1]) if c else "none"


COUNTERS_FILE = _newest_counters()
ISSUE_CEILING, ISSUE_CEILING_4CYCLE = 0.486, 0.248      # VALU instructions per SIMD and cycle on gfx950: in total / for the 4-cycle class (tools/valu_ceiling, profiles/r06a_valu_ceiling.txt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--scale", type=float, default=1.0, help="triangle-count scale of the bistro-like generator (1.0 = 2.8 M triangles)")
    ap.add_argument("--tex", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp32-lp-types", action="store_true", help="RTXPT_LP_TYPES_USE_16BIT_PRECISION 0 instead of the reference's default build (lp types in binary16)")
    ap.add_argument("--no-env-compression", action="store_true", help="sample the uncompressed RGBA16F environment cube (the reference on Vulkan) instead of the BC6H one (its D3D12 default)")
    ap.add_argument("--skip-roofline-steps", action="store_true", help="profiling runs (tools/profile_round.sh): only the timed steps, no extra serial / counter steps")
    ap.add_argument("--serial-kernels", action="store_true", help="run every step with PT_DEVICE_SERIAL_KERNELS semantics (for rocprofv3 kernel traces: launches never overlap)")
    ap.add_argument("--transport", choices=["rccl", "host"], default="rccl",
                    help="N > 1: 'rccl' = one rank per GPU, the frame gather is pt_gather over RCCL (the measurement). 'host' = REHEARSAL of the multi-rank code path on fewer GPUs than "
                         "ranks: the ranks share the visible devices (rank % device count), the process group is gloo, the tiles travel through pt_gather_host (the library's own gather "
                         "protocol over a host transport) — every line of the N > 1 branch runs, the timing is not a scaling number and the JSON line says so")
    args = ap.parse_args()

    import torch
    import rtxpt_amd as pt
    from rtxpt_amd import scenes, parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the library has no CPU path")
    host_transport = world > 1 and args.transport == "host"
    if host_transport:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    red_dev = "cpu" if host_transport else "cuda"      # where the small reductions (flags, times, ray counts) live: gloo reduces host tensors
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if host_transport:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    sc, cam = scenes.bistro_like(scale=args.scale, tex_size=args.tex)
    sc["env_cube_dim"] = 2048                    # EnvMapBaker's cube resolution for an image source (EnvMapBaker.cpp:374-375)
    sc["env_compression"] = 0 if args.no_env_compression else 1      # ... and its BC6U compression of that cube, on by default on D3D12 (EnvMapBaker.h:157,193): bake-time only
    S = scenes.default_settings(useFp16Types=0 if args.fp32_lp_types else 1)      # 8 bounces, NEE (emissive triangles + env quads), Russian roulette; lp types as the reference ships them
    W, H, SPP = args.width, args.height, args.spp
    camd = scenes.bridge_camera(W, H, **cam)
    g = pt.PathTracer(device=local_rank, shard_rank=rank, shard_count=world)
    g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(W, H)
    g.set_serial_kernels(args.serial_kernels)
    n_owned, packed_bytes = g.shard_info()
    # the frame gather is the library's own (pt_comm_init + pt_gather: un-padded ncclSend / ncclRecv on the library's stream); torch.distributed only
    # carries the 128-byte unique id. Should the communicator not come up on this node, the bench falls back to a torch.distributed gather of the packed
    # tiles and says so in the JSON line (`config.gather`) — every rank takes the same branch.
    gather_mode = "none (1 GPU)"
    counts = send = None
    host_frame = None
    if host_transport:
        import ctypes
        gather_mode = "REHEARSAL: pt_gather_host (the library's gather protocol: layout, packing, un-padded point-to-point transfers, unpacking) over a gloo transport; %d ranks on %d device(s)" % (world, torch.cuda.device_count())

        def _send(ptr, nbytes, peer):
            dist.send(torch.frombuffer((ctypes.c_char * nbytes).from_address(ptr), dtype=torch.uint8).clone(), dst=peer)

        def _recv(ptr, nbytes, peer):
            t = torch.empty(nbytes, dtype=torch.uint8); dist.recv(t, src=peer); ctypes.memmove(ptr, t.data_ptr(), nbytes)
    elif world > 1:
        ok = 1
        try:
            idt = torch.tensor(list(pt.comm_unique_id()) if rank == 0 else [0] * pt.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
            dist.broadcast(idt, 0)
            g.comm_init(bytes(idt.cpu().tolist()), rank, world)
        except Exception as e:      # noqa: BLE001
            ok = 0; err = repr(e)
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda"); dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            gather_mode = "pt_gather: RCCL ncclSend/ncclRecv of the un-padded tile buffers on the library's stream"
        else:
            gather_mode = "torch.distributed.gather of padded tile buffers (pt_comm_init failed on some rank%s)" % ((": " + err) if not ok else "")
            counts = [parallel.shard_pixels(W, H, r, world).size for r in range(world)]
            send = torch.empty((n_owned, 4), dtype=torch.float32, device="cuda")

    gather_ms = []

    def step():
        g.reset_accumulation()
        st = g.render(0, SPP)
        if host_transport:
            nonlocal host_frame
            tg = time.perf_counter()
            host_frame = g.radiance()      # this rank's accumulation buffer (its own tiles are valid), then the library's gather towards rank 0 through the host
            pt.gather_host(W, H, rank, world, host_frame, _send, _recv); gather_ms.append((time.perf_counter() - tg) * 1e3)
        elif world > 1:
            torch.cuda.synchronize(); tg = time.perf_counter()      # (pt_render has drained its streams; the gather is timed on its own: pack + ncclSend / ncclRecv + unpack on the library's stream)
            if send is None:
                g.gather(); torch.cuda.synchronize(); gather_ms.append((time.perf_counter() - tg) * 1e3)
            else:
                g.pack_shard(send.data_ptr(), packed_bytes)
                got = parallel.gather_packed(send, rank, world, dist, counts)
                if rank == 0:
                    for r in range(1, world):
                        buf = got[r].contiguous()
                        g.unpack_shard(buf.data_ptr(), buf.numel() * 4, r)
        return st

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    gather_ms.clear()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0
    rays_local = float(sum(s["extendRays"] + s["shadowRays"] for s in stats))
    tl = torch.tensor([elapsed, rays_local], dtype=torch.float64, device=red_dev)
    if world > 1:
        tmax = tl[0:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rsum = tl[1:2].clone(); dist.all_reduce(rsum, op=dist.ReduceOp.SUM)
        elapsed, rays_total = float(tmax.item()), float(rsum.item())
    else:
        rays_total = rays_local

    # roofline of the dominant kernel (k_extend). In the timed region pt_render pipelines four sub-frame batches on four streams, so launches of
    # different kernels overlap and a per-launch HIP-event duration there measures "k_extend while sharing the GPU". The launch duration the
    # roofline needs is therefore measured live right after the timed region, on the same context, with the overlap switched off
    # (pt_set_serial_kernels): ROOF_STEPS steps, HIP events on the library's stream around every launch. One more step with the in-kernel BVH
    # counters compiled in gives the mean node visits / triangle tests per ray.
    if args.skip_roofline_steps:
        if rank == 0:
            print(json.dumps({"metric": "Mrays/s (profiling run)", "value": rays_total / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True}))
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        return
    ROOF_STEPS = 2
    g.set_serial_kernels(True)
    serial = []
    for _ in range(ROOF_STEPS):
        g.reset_accumulation(); serial.append(g.render(0, SPP))
    g.set_counters(True); g.reset_accumulation(); cst = g.render(0, SPP); g.set_counters(False)
    g.set_serial_kernels(args.serial_kernels)
    nodes_per_ext = cst["nodeVisitsExtend"] / max(1, cst["extendRays"]); tris_per_ext = cst["triTestsExtend"] / max(1, cst["extendRays"])
    nodes_per_sh = cst["nodeVisitsShadow"] / max(1, cst["shadowRays"]); tris_per_sh = cst["triTestsShadow"] / max(1, cst["shadowRays"])
    ext_ms = sum(s["extendKernelMs"] for s in serial); ext_launches = sum(s["extendLaunches"] for s in serial); ext_rays = sum(s["extendRays"] for s in serial)
    bytes_per_ext = B_EXTEND_FIXED + nodes_per_ext * B_NODE + tris_per_ext * B_TRI
    bytes_per_sh = B_SHADOW_FIXED + nodes_per_sh * B_NODE + tris_per_sh * B_TRI
    ext_gbs = ext_rays * bytes_per_ext / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else 0.0
    sh_ms = sum(s["shadowKernelMs"] for s in serial); shade_ms = sum(s["shadeKernelMs"] for s in serial)
    serial_ms = sum(s["gpuMilliseconds"] for s in serial) / ROOF_STEPS
    # whole frame, SURVEY.md 8(d): extend + shade x 656 B + shadow, against the wall time of the pipelined frame
    frame_bytes = cst["extendRays"] * bytes_per_ext + cst["hits"] * B_SHADE + cst["shadowRays"] * bytes_per_sh
    frame_gbs = frame_bytes / (elapsed / args.steps) / 1e9

    # What bounds the kernel is a counter question, and the counters cannot be read from inside this process: the committed summary of the same
    # workload (tools/profile_round.sh -> profiles/rNN_counters.json, rocprofv3 --pmc in separate passes) is quoted, null when the workload differs.
    traffic = hbm_counter_gbs = valu = l2 = None; counters_src = None; bound = "unknown (no counter summary for this workload)"
    cp = os.path.join(ROOT, "profiles", COUNTERS_FILE)
    digest = pt.kernel_source_digest()
    if os.path.exists(cp) and (W, H, SPP, args.scale, args.tex, world) == (3840, 2160, 4, 1.0, 1024, 1) and json.load(open(cp)).get("kernel_source_sha256") != digest:
        bound = "unknown (profiles/%s was collected on other kernel sources than the ones this library was built from: re-run tools/profile_round.sh)" % COUNTERS_FILE
    elif os.path.exists(cp) and (W, H, SPP, args.scale, args.tex, world) == (3840, 2160, 4, 1.0, 1024, 1):
        cj = json.load(open(cp))["groups"]["extend"]; counters_src = "profiles/" + COUNTERS_FILE + " (tools/profile_round.sh; kernel_source_sha256 matches the kernels of this run)"
        traffic = cj["hbm_bytes_per_launch"]; hbm_counter_gbs = cj["hbm_counter_gbs"]; l2 = cj["l2_hit_rate"]
        valu = {"busy": cj["valu_busy"], "lane_utilisation": cj["lane_utilisation"], "valu_instructions_per_vmem_read": cj["valu_per_vmem_read"],
                "wait_any_share_of_wave_cycles": cj["wait_any_share_of_wave_cycles"]}
        # What limits the loop (round 6, calibrated: tools/valu_ceiling -> profiles/r06a_valu_ceiling.txt). gfx950 issues at most 0.486 VALU instructions per SIMD and cycle in total
        # and 0.248 of the 4-cycle class (everything but two-operand add / mul / logic / mov); a lone wave issues one instruction per ~4.5 cycles. Filler instructions of either
        # class in the loop (+10 %) lengthen k_extend by 4-6 %: the loop is about half bound by the instructions a wave has to issue, half by its dependent loads.
        vps = cj.get("valu_instr_per_simd_cycle") or 0.0
        bound = "hbm" if (hbm_counter_gbs or 0) >= 0.5 * HBM_PEAK_GBS else ("issue" if vps >= 0.4 * ISSUE_CEILING else "latency")
        valu["share_of_total_issue_ceiling"] = vps / ISSUE_CEILING; valu["issue_ceilings"] = {"total": ISSUE_CEILING, "four_cycle_class": ISSUE_CEILING_4CYCLE}
        valu["instructions_per_simd_cycle"] = vps

    if rank == 0:
        info = g.scene_info(); bvh = g.bvh_info()
        out = {
            "metric": "Mrays/s at 4K 4spp 8-bounce bistro-like (extend + shadow rays / wall time of pt_render)",
            "value": rays_total / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3 bistro-like street canyon, %d triangles, 64 materials, 32 textures %d^2, %d emissive-triangle + env-quad lights, %dx%d, %d spp, 8 bounces, NEE 5 candidates + RR, lp types %s, environment cube %s; BVH builder=%s, built on %s, buildMs %.1f (host part %.1f), %d wide nodes"
                                   % (info["triangles"], args.tex, len(g.lights()["proxyCounters"]), W, H, SPP, "fp32" if args.fp32_lp_types else "binary16 (reference default)",
                                      "RGBA16F" if args.no_env_compression else "2048 BC6H (reference default on D3D12)", bvh["builderName"], bvh["builtOn"], bvh["buildMs"], bvh["hostMs"], bvh["numWideNodes"]),
                       "bvh": bvh,
                       "parallelism": "pixel-tile shard x%d + 1 gather" % world, "gather": gather_mode, "gather_ms_per_step_rank0": (sum(gather_ms) / len(gather_ms)) if gather_ms else None, "transport": args.transport if world > 1 else "none",
                       "rehearsal": bool(host_transport), "kernel_source_sha256": digest, "library_sha256": pt.library_digest(), "rays_per_step": rays_total / args.steps,
                       "extend_rays_per_step": sum(s["extendRays"] for s in stats) / args.steps * (world if world > 1 else 1), "paths_per_step": W * H * SPP,
                       "tail_kernel_launches_per_step": sum(s["tailLaunches"] for s in stats) / args.steps},
            # `frac` is the prescribed figure: algorithmic bytes / launch time / 8 TB/s. `bound` is what the counters say limits the kernel:
            # the BVH is served from L1/L2, HBM itself carries `hbm_counter_gbs`; the loop is bound by instruction issue and dependent-load latency together (`bound`: "issue").
            "roofline": {"bound": bound, "prescribed_bound": "hbm", "kernel": "k_extend", "achieved": ext_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ext_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": "quoted from the committed rocprofv3 --pmc summary (counters_source), not measured in this run", "hbm_counter_gbs": hbm_counter_gbs, "l2_hit_rate": l2, "valu": valu, "counters_source": counters_src,
                         "bound_evidence": "profiles/r06a_valu_ceiling.txt: measured issue ceilings (0.486 VALU instructions per SIMD and cycle in total, 0.248 for the 4-cycle class) and the filler probe (+10 % instructions of either class in the traversal loop = +4 ... 6 % k_extend time): 'issue' = about half bound by the instructions a wave issues, half by dependent-load latency; HBM itself at ~18 % of peak",
                         "whole_frame": {"algorithmic_bytes_per_step": frame_bytes, "achieved": frame_gbs, "frac": frame_gbs / HBM_PEAK_GBS,
                                         "terms": "extend rays x (52 + 128 nodes + 48 tris) + hits x 656 + shadow rays x (80 + 128 nodes + 48 tris), over the pipelined step time"},
                         "bytes_per_ray": bytes_per_ext, "node_visits_per_ray": nodes_per_ext, "tri_tests_per_ray": tris_per_ext,
                         "avg_launch_ms": ext_ms / max(1, ext_launches), "launches": ext_launches,
                         "kernel_ms_per_step": {"k_extend": ext_ms / ROOF_STEPS, "k_shade": shade_ms / ROOF_STEPS, "k_shadow": sh_ms / ROOF_STEPS},
                         "measured_on": "%d serial-kernel steps after the timed region (launches do not overlap, closest-hit and visibility rays in launches of their own; the timed region fuses them: k_trace_pair runs the same two loops); serial frame %.1f ms vs pipelined %.1f ms" % (ROOF_STEPS, serial_ms, elapsed / args.steps * 1e3),
                         "shadow_node_visits_per_ray": nodes_per_sh, "shadow_tri_tests_per_ray": tris_per_sh,
                         "leaf_visits_per_ray": cst["leafVisitsExtend"] / max(1, cst["extendRays"]),
                         "wave_iterations_per_ray": cst["waveItersExtend"] / max(1, cst["extendRays"]),
                         "phase_cycle_share": [c / max(1, sum(cst["extendPhaseCycles"])) for c in cst["extendPhaseCycles"]],
                         "cycles_per_wave_iteration": sum(cst["extendPhaseCycles"]) / max(1, cst["waveItersExtend"]),
                         "leaf_block_share": cst["leafBlocksExtend"] / max(1, cst["waveItersExtend"]),
                         "block_runs_per_wave_iteration": dict(zip(["refill", "chunk_load", "inner", "leaf", "alpha_test", "hit_reduce", "pop", "pop_trips"],
                                                                  [e / max(1, cst["waveItersExtend"]) for e in cst["extendEvents"]])),
                         "work_slots_per_quad_iteration": (cst["nodeVisitsExtend"] + cst["leafVisitsExtend"]) / max(1, 16 * cst["waveItersExtend"])},
            "build": g.build_stats(),
        }
        if world == 1 and not args.skip_roofline_steps:
            # beyond the headline (after the timed region, not part of `value`): the realtime mode's two path-tracing passes on the same scene and frame size, one sub-sample
            try:
                prm = scenes.stable_planes_params(W, H, scenes.view_projection(W, H, **cam), sub_samples=1)
                g.build_stable_planes(0, prm); b = g.build_stable_planes(1, prm)["stats"]; f = g.fill_stable_planes(1, prm)["stats"]
                out["realtime_passes"] = {"ok": True, "workload": "stable-plane build pass + one fill sub-sample, %dx%d, same scene (SURVEY.md 8f row N4)" % (W, H),
                                          "build_ms": b["gpuMilliseconds"], "build_rays": int(b["extendRays"]), "fill_ms": f["gpuMilliseconds"], "fill_rays": int(f["extendRays"]) + int(f["shadowRays"]),
                                          "fill_mrays_per_s": (int(f["extendRays"]) + int(f["shadowRays"])) / max(f["gpuMilliseconds"], 1e-9) / 1e3}
                # ... and the coupled frame (pt_realtime_frame with the light baker in the loop: UpdateBegin, build pass, UpdateEnd on the frame's depth + motion vectors, fill pass feeding the reservoirs)
                g.set_neeat(True)
                for fr in range(3): g.realtime_frame(fr, prm)
                t1 = time.perf_counter(); _, bs, fs = g.realtime_frame(3, prm); wall = (time.perf_counter() - t1) * 1e3
                g.set_neeat(False)
                out["realtime_passes"]["coupled_frame_with_neeat"] = {"build_ms": bs["gpuMilliseconds"], "fill_ms": fs["gpuMilliseconds"], "fill_rays": int(fs["extendRays"]) + int(fs["shadowRays"]),
                                                                     "note": "4th frame of a run (history and tile tables exist); baker passes run between the two on the device"}
            except Exception as e:      # the side leg never takes the bench line down, but it says so: "ok": false
                out["realtime_passes"] = {"ok": False, "error": str(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            # the oracle's block of the very frame the timed steps rendered doubles as the parity check of the benchmarked configuration
            out["cpu_baseline"], block, rect = cpu_baseline(sc, cam, S, W, H, SPP)
            g.reset_accumulation(); g.render(0, SPP)
            got = g.radiance()[rect[1]:rect[3], rect[0]:rect[2], :3]
            diff = int((got.view(np.uint32) != block.view(np.uint32)).any(-1).sum())
            out["parity"] = {"rel_l2": float(np.linalg.norm(got.astype(np.float64) - block) / max(np.linalg.norm(block.astype(np.float64)), 1e-30)),
                             "differing_pixels": diff, "pixels": int(block.shape[0] * block.shape[1]),
                             "block": "x %d..%d, y %d..%d of the %dx%d frame, %d spp (the cpu_baseline sample)" % (rect[0], rect[2], rect[1], rect[3], W, H, SPP),
                             "against": "oracle/ptref (pinned to the reference's integrator text, tests/test_oracle_refpin_integrator.py)", "tolerance": "bit-exact expected; north_star allows 1e-3 relative L2"}
        # ... and, for the default workload, the whole frame against the REFERENCE'S OWN integrator text: tests/golden/bench_frame_golden.npz holds the SHA-256 of this very frame as
        # PathTracer.hlsli & co. render it (tests/golden/make_bench_frame_golden.py, made in the build container), every 120th row and the ray counts. No oracle in this comparison.
        try:
            gp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "bench_frame_golden.npz")
            default_workload = (W, H, SPP) == (3840, 2160, 4) and args.scale == 1.0 and args.tex == 1024 and not args.no_env_compression and not args.fp32_lp_types
            if world == 1 and default_workload and os.path.exists(gp):
                import hashlib
                gold = np.load(gp)
                g.reset_accumulation(); st_ref = g.render(0, SPP); frame = g.radiance()
                rows = frame[::int(gold["row_step"][0])]
                out.setdefault("parity", {})["reference_text"] = {
                    "frame_sha256_equal": bool(np.array_equal(np.frombuffer(hashlib.sha256(np.ascontiguousarray(frame, np.float32).tobytes()).digest(), np.uint8), gold["sha256"])),
                    "differing_pixels_in_kept_rows": int((rows.view(np.uint32) != gold["rows"].view(np.uint32)).any(-1).sum()), "kept_rows": int(rows.shape[0]),
                    "ray_counts_equal": bool((int(st_ref["extendRays"]), int(st_ref["shadowRays"])) == tuple(int(v) for v in gold["rays"])),
                    "against": "the reference's integrator text (PathTracer.hlsli & co. compiled from the reference tree) rendering this frame: tests/golden/bench_frame_golden.npz"}
            if world > 1 and default_workload and os.path.exists(gp):      # N > 1: the frame rank 0 holds after the last timed step's gather (all ranks' tiles), against the same digest
                import hashlib
                gold = np.load(gp)
                frame = host_frame if host_transport else g.radiance()
                rows = frame[::int(gold["row_step"][0])]
                out.setdefault("parity", {})["reference_text_gathered_frame"] = {
                    "frame_sha256_equal": bool(np.array_equal(np.frombuffer(hashlib.sha256(np.ascontiguousarray(frame, np.float32).tobytes()).digest(), np.uint8), gold["sha256"])),
                    "differing_pixels_in_kept_rows": int((rows.view(np.uint32) != gold["rows"].view(np.uint32)).any(-1).sum()), "kept_rows": int(rows.shape[0]),
                    "ray_counts_equal": bool(int(round(rays_total / args.steps)) == int(gold["rays"][0]) + int(gold["rays"][1])),
                    "against": "tests/golden/bench_frame_golden.npz (the reference's integrator text rendering this frame), the %d ranks' tiles as gathered on rank 0" % world}
        except Exception as e:      # noqa: BLE001
            out.setdefault("parity", {})["reference_text"] = {"error": str(e)[:300]}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sc, cam, S, W, H, SPP):
    """The CPU oracle (a port: RTXPT has no CPU path) on a bounded sample of the SAME workload: a 1920x1080 centre block
    of the 4K frame, 4 accumulated samples, all host cores (OpenMP), about 10 s of CPU work on the 128-thread GPU-box host. Reported per ray so that it is resolution independent."""
    from oracle import ptref
    from rtxpt_amd import scenes
    o = ptref.Oracle(lp16=bool(int(S["useFp16Types"]))); o.set_scene(sc); o.set_camera(scenes.bridge_camera(W, H, **cam)); o.set_settings(S); o.resize(W, H)
    t0 = time.perf_counter(); o.L.ptref_prepare(o.h); prep = time.perf_counter() - t0
    bw, bh = min(W, 1920), min(H, 1080)
    x0, y0 = (W - bw) // 2, (H - bh) // 2
    rect = (x0, y0, x0 + bw, y0 + bh)
    t0 = time.perf_counter(); o.render(0, SPP, rect=rect); dt = time.perf_counter() - t0
    c = o.counters()
    rays = c["extendRays"] + c["shadowRays"]
    block = o.radiance()[y0:y0 + bh, x0:x0 + bw, :3].copy()
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": ptref.num_threads(), "kind": "port",
            "sample": "%dx%d centre block of the %dx%d frame, %d spp, %d rays in %.2f s (SAH BVH build + light bake %.1f s not included)" % (bw, bh, W, H, SPP, rays, dt, prep)}, block, rect


if __name__ == "__main__":
    main()
