#!/usr/bin/env python3
"""bench.py -- depth frames/sec of the integrate+raycast frame loop on MI355X.

Metric (BASELINE.json): depth frames/sec integrate+raycast @640x480, 4 cm voxel.
Workload (config.workload = cfg2): synthetic orbiting-sphere sequence S1,
640x480, 4 cm voxels, 500 k hash buckets (5 M entries), 1 M SDF blocks,
alloc + compactify + integrate + garbage-collect + raycast + normals per frame,
in the reference's order (render with the previous pose, then integrate;
DepthSensingCUDA/Source/DepthSensing.cpp:763,903).  A "step" is one frame.
Inputs (depth + colour of every frame) are generated on the device before the
timed region and stay resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1: launched by torch.distributed.run, one process per GPU, one independent
scene per GPU (orbit phase-shifted by 2*pi*rank/8); weak scaling; RCCL is used
only for the start/stop barriers and a MAX-reduce of the elapsed time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed
inside the timed region) and `cpu_baseline` (the CPU oracle timed on a bounded
sample of the same workload, rank 0 at N=1).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=900)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4"])
    p.add_argument("--offline", action="store_true", help="alloc until fixed point (blocking read-backs), as in parity runs")
    p.add_argument("--no-gc", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-frames", type=int, default=40, help="frames of the workload timed on the CPU oracle")
    p.add_argument("--stages", action="store_true", help="also print per-stage device times to stderr")
    p.add_argument("--scene", default=None, help="override the scene (S1, S2)")
    p.add_argument("--event-stride", type=int, default=8, help="HIP events around k_render on every n-th timed frame")
    return p.parse_args()


def dist_setup(n_gpus):
    """-> (rank, world, local_rank, dist or None)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if n_gpus <= 1 and world <= 1:
        return 0, 1, 0, None
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank, dist


def barrier(dist_mod):
    if dist_mod is not None:
        dist_mod.barrier()


def max_over_ranks(dist_mod, value, device):
    if dist_mod is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist_mod, value, device):
    if dist_mod is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist_mod.all_reduce(t, op=dist_mod.ReduceOp.SUM)
    return float(t.item())


class GpuWorkload:
    """one scene on one GPU: pre-generated frames + the reference frame loop"""

    def __init__(self, cfg_name, n_frames, rank, args):
        import torch
        from voxelhashing_amd import engine as E, synth, vhtypes as T
        self.torch, self.E, self.T, self.synth = torch, E, T, synth
        cfg = dict(synth.CONFIGS[cfg_name])
        if args.scene:
            cfg["scene"] = args.scene
        self.cfg = cfg
        self.hp, self.cp, self.rp = synth.config_params(cfg)
        # stage timers (HIP events around every stage) are switched on for a slice of the warm-up only;
        # the timed region records events around the dominant kernel (raycast) alone
        self.opt = T.make_scene_options(offline=args.offline, gc=not args.no_gc, starve=15, timings=False)
        spheres, inside, radius = synth.scene(cfg["scene"])
        self.n_frames = n_frames
        phase = 2.0 * math.pi * rank / 8.0
        self.poses = [synth.orbit_pose(k, 1000, radius, phase) for k in range(n_frames)]
        self.dev = torch.device("cuda", torch.cuda.current_device())
        H, W = self.cp.m_imageHeight, self.cp.m_imageWidth
        # inputs resident in HBM: torch owns the frame store, the engine reads raw pointers
        self.depth = torch.empty((n_frames, H, W), dtype=torch.float32, device=self.dev)
        self.color = torch.empty((n_frames, H, W, 4), dtype=torch.float32, device=self.dev)
        self.frames = []
        for k in range(n_frames):
            fr = E.DepthFrame(self.cp, depth_ptr=self.depth[k].data_ptr(), color_ptr=self.color[k].data_ptr())
            E.synth_frame(spheres, inside, self.poses[k], self.cp, out=fr)
            self.frames.append(fr)
        self.scene = E.CUDASceneRepHashSDF(self.hp, self.opt)
        self.ray = E.CUDARayCastSDF(self.rp)
        self.event_stride = max(int(getattr(args, 'event_stride', 8)), 1)
        # live HIP events around k_render in the timed region, on every event_stride-th frame (an event record idles
        # the queue for ~5 us, so bracketing every launch would cost ~10 % of the frame rate being measured)
        self.ray.setTiming(True, march_only=True, stride=self.event_stride)
        self.hd = self.scene.getHashData()
        torch.cuda.synchronize()

    def run(self, k0, k1):
        """frames k0..k1-1 of the sequence in the reference's order"""
        scene, ray, cp, hd = self.scene, self.ray, self.cp, self.hd
        for k in range(k0, k1):
            if k > 0:
                ray.render(hd, scene.getHashParams(), cp, self.poses[k - 1])
            scene.integrate(self.poses[k], self.frames[k], cp, None)

    def stage_timers(self, on):
        self.opt.s_timingsDetailledEnabled = 1 if on else 0
        self.scene.setOptions(self.opt)
        self.ray.setTiming(True, march_only=not on, stride=1 if on else self.event_stride)

    def timings(self):
        s = self.scene.getTimings()
        r = self.ray.getTimings()
        s.update(r)
        return s


def run_timed(workload, warmup, steps, dist_mod, device, sync=lambda: None, after_warmup=lambda: None, skip_warmup=False):
    """the measurement contract: `warmup` untimed frames, then exactly `steps` frames bracketed by a barrier
    and a device synchronisation on both sides; the elapsed time is the MAX over ranks and the unit count the
    SUM over ranks (weak scaling: every rank runs its own `steps` frames).  -> (seconds, total frames)"""
    if not skip_warmup:
        workload.run(0, warmup)
    sync()
    after_warmup()
    barrier(dist_mod)
    sync()
    t0 = time.perf_counter()
    workload.run(warmup, warmup + steps)
    sync()
    barrier(dist_mod)
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(dist_mod, elapsed, device)
    total = sum_over_ranks(dist_mod, float(steps), device)
    return elapsed, total


def stage_bytes(cfg_hp, cp, n_occ, stage):
    """ALGORITHMIC bytes per launch (E = 20 B entry payload, V = 8 B voxel)"""
    W, H = cp.m_imageWidth, cp.m_imageHeight
    E, V = 20, 8
    if stage == "raycast":
        return 52.0 * W * H
    if stage == "integrate":
        return n_occ * (E + 512 * V * 2) + 4.0 * W * H + 16.0 * W * H
    if stage == "alloc":
        return 4.0 * W * H
    if stage == "compactify":
        return 4.0 * cfg_hp.m_hashNumBuckets * 10 + 2.0 * E * n_occ
    if stage == "normals":
        return (16.0 * 5 + 16.0) * W * H
    raise KeyError(stage)


def cpu_baseline(cfg_name, n_frames, args):
    """the CPU oracle on the first n_frames of the same workload: one thread (the checker's build) and all cores
    (the same source built with OpenMP); `value` is the faster, all-core figure"""
    from oracle import oracle as O
    from voxelhashing_amd import synth, vhtypes as T
    cfg = dict(synth.CONFIGS[cfg_name])
    if args.scene:
        cfg["scene"] = args.scene
    hp, cp, rp = synth.config_params(cfg)
    opt = T.make_scene_options(offline=args.offline, gc=not args.no_gc, starve=15)
    spheres, inside, radius = synth.scene(cfg["scene"])
    poses = [synth.orbit_pose(k, 1000, radius, 0.0) for k in range(n_frames)]
    inputs = [O.synth_frame(spheres, inside, p, cp) for p in poses]

    def timed(omp):
        sc = O.OracleScene(hp, cp, rp, opt, omp=omp)
        t0 = time.perf_counter()
        for k in range(n_frames):
            if k > 0:
                sc.render(poses[k - 1])
            sc.integrate(poses[k], inputs[k][0], inputs[k][1])
        dt = time.perf_counter() - t0
        blocks = sc.hp.m_numOccupiedBlocks
        sc.close()
        return dt, blocks

    dt1, blocks1 = timed(False)
    # a GPU box gives one GPU's share of the host (16 cores): do not let OpenMP start a thread per host core
    # (set through the library: an OpenMP runtime that numpy or torch already started ignores the environment)
    O.lib(omp=True).vho_set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    cores = int(O.lib(omp=True).vho_num_threads())
    dtn, blocksn = timed(True)
    if blocks1 != blocksn:
        raise RuntimeError(f"all-core baseline diverged from the checker: {blocksn} vs {blocks1} blocks")
    return dict(value=n_frames / dtn, unit="frames/s", cores=cores, kind="port", single_thread=n_frames / dt1,
                sample=f"first {n_frames} frames of {cfg_name} ({cfg['scene']}): oracle/libvh_oracle_omp.so on {cores} threads "
                       f"{dtn:.1f} s; oracle/libvh_oracle.so on 1 thread {dt1:.1f} s")


def main():
    args = parse_args()
    rank, world, local_rank, dist_mod = dist_setup(args.gpus)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n_frames = args.warmup + args.steps
    wl = GpuWorkload(args.config, n_frames, rank, args)

    # warm-up, untimed.  Its last frames run with every stage timer on: per-stage device times of the steady state.
    n_stage = min(64, args.warmup)
    wl.run(0, args.warmup - n_stage)
    torch.cuda.synchronize()
    s0 = wl.timings()
    wl.stage_timers(True)
    wl.run(args.warmup - n_stage, args.warmup)
    torch.cuda.synchronize()
    s1 = wl.timings()
    wl.stage_timers(False)
    stage_us = {k[:-3]: 1e3 * (s1[k] - s0[k]) / max(n_stage, 1) for k in ("alloc_ms", "compactify_ms", "integrate_ms", "splat_ms", "raycast_ms", "normals_ms")}

    pre = {}

    def after_warmup():
        torch.cuda.synchronize()
        pre.update(wl.timings())  # the timed region starts from here

    elapsed, total_frames = run_timed(wl, args.warmup, args.steps, dist_mod, dev, sync=torch.cuda.synchronize,
                                      after_warmup=after_warmup, skip_warmup=True)
    value = total_frames / elapsed
    post = wl.timings()
    n_occ = wl.scene.getNumOccupiedBlocks()

    # dominant kernel: HIP events on the launch stream around k_render, every frame of the timed region
    launches = max(int(post["frames"] - pre["frames"]), 1)
    dominant = max(stage_us, key=stage_us.get) if n_stage else "raycast"
    if dominant != "raycast":
        print(f"note: dominant stage is {dominant}; live events cover raycast only", file=sys.stderr)
        dominant = "raycast"
    pair_us = 1e3 * (post["raycast_ms"] - pre["raycast_ms"]) / launches
    # An event pair reads a few microseconds with nothing between its two records (the records themselves): that
    # share, sampled with empty pairs while the warm-up timed every stage, is not the kernel's.
    overhead_us = 1e3 * wl.ray.getEventPairOverheadMs()
    dom_us = pair_us - overhead_us if 0.0 < overhead_us < 0.5 * pair_us else pair_us
    alg_bytes = stage_bytes(wl.hp, wl.cp, n_occ, dominant)
    achieved = alg_bytes / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
    traffic = None
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(prof):
        try:
            with open(prof) as f:
                tj = json.load(f)
            traffic = tj.get(args.config, {}).get(dominant)
        except Exception:
            traffic = None
    roofline = dict(bound="hbm", kernel="k_render (raycast)", achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 6), traffic=traffic,
                    algorithmic_bytes=alg_bytes, avg_launch_us=round(dom_us, 3), event_pair_us=round(pair_us, 3),
                    event_pair_overhead_us=round(overhead_us, 3), launches_timed=launches, event_stride=wl.event_stride,
                    stage_us_warmup={k: round(v, 3) for k, v in stage_us.items()}, blocks_in_frustum=n_occ)
    per_launch_us = stage_us

    result = None
    if rank == 0:
        cfg = wl.cfg
        result = {
            "metric": "depth frames/sec integrate+raycast @640x480, 4 cm voxel; HBM GB/s vs peak",
            "value": round(value, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {cfg['scene']} orbit, {cfg['width']}x{cfg['height']}, {cfg['params']} voxels, "
                            f"{cfg['num_buckets']} buckets, {cfg['num_sdf_blocks']} SDF blocks, "
                            f"alloc+compactify+integrate+{'gc+' if not args.no_gc else ''}raycast+normals per frame",
                "alloc_mode": "offline (fixed point)" if args.offline else "online (one pass per frame)",
                "streams": "one independent scene per GPU",
            },
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(args.config, args.cpu_frames, args)
            except Exception as e:  # the baseline must never sink the GPU result
                result["cpu_baseline"] = dict(value=None, unit="frames/s", cores=1, kind="port", sample=f"failed: {e}")
        else:
            result["cpu_baseline"] = None
        if args.stages:
            print(json.dumps(per_launch_us), file=sys.stderr)
        print(json.dumps(result), flush=True)
    if dist_mod is not None:
        dist_mod.barrier()
        dist_mod.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
