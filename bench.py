#!/usr/bin/env python3
"""bench.py -- depth frames/sec of the integrate+raycast frame loop on MI355X.

Metric (BASELINE.json): depth frames/sec integrate+raycast @640x480, 4 cm voxel.
Workload (config.workload = cfg2): synthetic orbiting-sphere sequence S1,
640x480, 4 cm voxels, 500 k hash buckets (5 M entries), 1 M SDF blocks,
alloc + compactify + integrate + garbage-collect + raycast + normals per frame,
in the reference's order (render with the previous pose, then integrate;
DepthSensingCUDA/Source/DepthSensing.cpp:763,903).  A "step" is one frame.
Inputs (depth + colour of every frame) are generated on the device before the
timed region and stay resident in HBM.  The frame loop is native
(vh_reconstruction_run, include/vh_api.h): one host call enqueues all frames.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU (started by torch.distributed.run, or by this script
itself when WORLD_SIZE is not set), one independent scene per GPU (orbit
phase-shifted by 2*pi*rank/8); weak scaling; RCCL is used only for the
start/stop barriers and a MAX / SUM reduce of two scalars.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed
inside the timed region, plus entries for the integrate kernel) and
`cpu_baseline` (the CPU oracle timed on a bounded sample of the same workload,
rank 0 at N=1).
"""
import argparse
import ctypes as C
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
METRIC = "depth frames/sec integrate+raycast @640x480, 4 cm voxel; HBM GB/s vs peak"


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=900)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4"])
    p.add_argument("--offline", action="store_true", help="alloc until fixed point (blocking read-backs), as in parity runs")
    p.add_argument("--no-gc", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-frames", type=int, default=40, help="frames of the workload timed on the CPU oracle")
    p.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-core CPU baseline (0: every CPU the process may run on)")
    p.add_argument("--stages", action="store_true", help="also print per-stage device times to stderr")
    p.add_argument("--scene", default=None, help="override the scene (S1, S2)")
    p.add_argument("--event-stride", type=int, default=0, help="HIP events around k_render on every n-th timed frame (0: steps/10, at most 8)")
    p.add_argument("--no-alloc-ahead", action="store_true", help="alloc + compactify on the main stream, behind the ray cast")
    p.add_argument("--no-streaming", action="store_true", help="cfg3 without its per-frame stream out / stream in")
    p.add_argument("--streaming-radius", type=float, default=0.0, help="cfg3: radius of the streaming sphere in metres (0: the reference's formula, DepthSensing.cpp:1340-1355 -- with scene S1 nothing ever leaves it)")
    p.add_argument("--streaming-pos-z", type=float, default=0.0, help="cfg3: centre of the streaming sphere in front of the camera (0: the reference's formula)")
    p.add_argument("--streaming-extent", type=float, default=0.0, help="cfg3: edge of a streaming chunk in metres (0: 1 m, zParametersDefault.txt)")
    p.add_argument("--frames-on-host", action="store_true", help="the main workload is fed from pinned host memory (as the host-fed leg): its value is then the PCIe-inclusive rate, not the contract's")
    p.add_argument("--frames-in-flight", type=int, default=16, help="frames the host may run ahead of the device (0: no bound)")
    p.add_argument("--preroll-seconds", type=float, default=0.3, help="untimed device pre-roll before the warm-up (clocks, code objects)")
    p.add_argument("--no-extra-legs", action="store_true", help="skip the dense-scene integrate leg, the host-fed leg and cfg1")
    p.add_argument("--python-loop", action="store_true", help="drive the frames from Python (two ctypes calls per frame) instead of the native loop")
    p.add_argument("--standin", action="store_true", help="TEST ONLY: a CPU stand-in workload (no GPU, no engine) to exercise the multi-rank harness")
    p.add_argument("--standin-fail", default="", help="TEST ONLY: 'R:before-init' or 'R:before-barrier' -- rank R of a --standin run exits with an error there")
    return p.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# ranks
# ---------------------------------------------------------------------------------------------------------------

def launch_ranks(args, argv):
    """--gpus N without a launcher: start N ranks of this script (fresh processes, before anything here touches the
    GPU), relay rank 0's JSON line.  Every child is watched: as soon as one exits with an error the others -- which
    would sit in a barrier or in the rendezvous until its time-out -- are terminated, and the launcher exits non-zero
    at once with the failing rank's last lines of stderr."""
    import socket
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, outs, errs = [], [], []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (files, not pipes: nobody has to drain them while the ranks run)
        outs.append(tempfile.TemporaryFile() if rank == 0 else None)
        errs.append(tempfile.TemporaryFile())
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=outs[rank] if rank == 0 else subprocess.DEVNULL, stderr=errs[rank]))

    def tail(f, n=30):
        f.seek(0)
        return "\n".join(f.read().decode(errors="replace").splitlines()[-n:])

    failed = None
    while failed is None:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
        elif all(c == 0 for c in codes):
            break
        else:
            time.sleep(0.05)
    if failed is not None:
        code = procs[failed].returncode
        for p in procs:  # our own children, by handle: first ask, then insist
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        sys.stderr.write(f"bench.py: rank {failed} of {args.gpus} exited with {code}; the other ranks were stopped.  Its last lines:\n{tail(errs[failed])}\n")
        sys.stderr.flush()
        raise SystemExit(code if isinstance(code, int) and 0 < code < 256 else 1)
    outs[0].seek(0)
    sys.stdout.write(outs[0].read().decode())
    sys.stdout.flush()
    for r, f in enumerate(errs):  # what the ranks had to say (warnings), rank by rank
        t = tail(f, 10)
        if t.strip():
            sys.stderr.write(f"[rank {r}] " + t.replace("\n", f"\n[rank {r}] ") + "\n")
    return 0


def dist_setup(n_gpus, force_cpu=False):
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
    backend = "nccl" if (torch.cuda.is_available() and not force_cpu) else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank, dist


def stay_near_the_gpu(device_index):
    """Run on the CPUs of the GPU's own NUMA node (what a deployment does with numactl): pinned frame buffers are then
    first-touched on that node, and host-fed frames cross one PCIe link instead of the socket interconnect and the link
    (measured on a two-socket box: the host-fed leg at a third of its rate when the buffers landed on the other node).
    Best effort: returns a description, or None where the topology cannot be read."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bus = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        node = open(f"/sys/bus/pci/devices/{bus}/numa_node").read().strip()
        cpus = set()
        for part in open(f"/sys/bus/pci/devices/{bus}/local_cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return f"NUMA node {node} of GPU {bus} ({len(cpus)} cpus)"
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


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


def gather_over_ranks(dist_mod, value, device):
    """-> the value of every rank, in rank order"""
    if dist_mod is None:
        return [value]
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist_mod.get_world_size())]
    dist_mod.all_gather(out, t)
    return [float(o.item()) for o in out]


PER_RANK = []  # frames/s of each rank's own stream in the last run_timed() (its own clock, up to its own synchronisation)


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
    own = time.perf_counter() - t0
    barrier(dist_mod)
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(dist_mod, elapsed, device)
    total = sum_over_ranks(dist_mod, float(steps), device)
    # a slow rank shows here, not only in the maximum (outside the timed region)
    PER_RANK[:] = [round(steps / t, 3) if t > 0 else 0.0 for t in gather_over_ranks(dist_mod, own, device)]
    return elapsed, total


class StandinWorkload:
    """TEST ONLY (--standin): a fixed CPU delay per frame, so that the rank launch, the barriers and the reductions of
    this script can be exercised on a machine without a GPU.  Nothing of the engine (or of the oracle) runs."""

    def __init__(self, rank):
        self.frames_done = 0
        self.rank = rank

    def run(self, k0, k1):
        for _ in range(k0, k1):
            time.sleep(0.002)
            self.frames_done += 1


# ---------------------------------------------------------------------------------------------------------------
# the GPU workload
# ---------------------------------------------------------------------------------------------------------------

class GpuWorkload:
    """one scene on one GPU: pre-generated frames + the reference frame loop (native: Reconstruction)"""

    def __init__(self, cfg_name, n_frames, rank, args, scene=None, frames_on_host=False, streaming=None):
        import numpy as np
        import torch
        from voxelhashing_amd import engine as E, synth, vhtypes as T
        self.torch, self.E, self.T, self.synth, self.np = torch, E, T, synth, np
        cfg = dict(synth.CONFIGS[cfg_name])
        if scene or args.scene:
            cfg["scene"] = scene or args.scene
        self.cfg = cfg
        self.cfg_name = cfg_name
        self.hp, self.cp, self.rp = synth.config_params(cfg)
        if args.streaming_extent > 0:
            self.hp.m_streamingVoxelExtents[:] = [args.streaming_extent] * 3
        self.streaming = bool(cfg.get("streaming")) and not args.no_streaming if streaming is None else streaming
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
        torch.cuda.synchronize()
        self.frames_on_host = frames_on_host
        if frames_on_host:
            # what a sensor delivers (RGBDSensor::getDepthFloat / getColorRGBX): float depth + RGBX bytes, pinned
            self.h_depth = torch.empty((n_frames, H, W), dtype=torch.float32).pin_memory()
            self.h_color = torch.empty((n_frames, H, W, 4), dtype=torch.uint8).pin_memory()
            self.h_depth.copy_(self.depth)
            self.h_color.copy_((self.color.clamp(0.0, 1.0) * 255.0).to(torch.uint8))
            torch.cuda.synchronize()
            del self.depth, self.color
            depth_ptrs = [self.h_depth[k].data_ptr() for k in range(n_frames)]
            color_ptrs = [self.h_color[k].data_ptr() for k in range(n_frames)]
        else:
            depth_ptrs = [f.depth_ptr for f in self.frames]
            color_ptrs = [f.color_ptr for f in self.frames]
        self.seq = E.Reconstruction.makeFrames(self.poses, depth_ptrs, color_ptrs)
        self.scene = E.CUDASceneRepHashSDF(self.hp, self.opt)
        self.ray = E.CUDARayCastSDF(self.rp)
        self.grid = None
        ropt = dict(s_allocAhead=0 if args.no_alloc_ahead else 1, s_framesOnHost=1 if frames_on_host else 0,
                    s_maxFramesInFlight=max(int(args.frames_in_flight), 0))
        if self.streaming:
            # DSC/DepthSensing.cpp:610-618 + :1340-1355: 1 m chunks, 257^3, the reference's 80 parts, worker thread on
            ext = [self.hp.m_streamingVoxelExtents[i] for i in range(3)]
            dims = [self.hp.m_streamingGridDimensions[i] for i in range(3)]
            mn = [self.hp.m_streamingMinGridPos[i] for i in range(3)]
            self.grid = E.CUDASceneRepChunkGrid(self.scene, ext, dims, mn, self.hp.m_streamingInitialChunkListSize, True, self.opt.s_streamingOutParts)
            pos, rad = synth.streaming_sphere(self.hp, self.cp)
            if args.streaming_radius > 0:
                rad = args.streaming_radius
            if args.streaming_pos_z > 0:
                pos = np.array([0.0, 0.0, args.streaming_pos_z], dtype=np.float32)
            self.streaming_sphere = (float(pos[2]), float(rad), float(ext[0]))
            ropt.update(s_streamingEnabled=1, s_streamingPos=pos, s_streamingRadius=rad)
        self.recon = E.Reconstruction(self.scene, self.ray, self.grid, self.cp, E.Reconstruction.defaultOptions(**ropt))
        self.python_loop = bool(args.python_loop)
        self.event_stride = 1
        self.hd = self.scene.getHashData()
        torch.cuda.synchronize()

    def set_event_stride(self, stride):
        # live HIP events around k_render in the timed region, on every stride-th frame (an event record idles the
        # queue for ~5 us: bracketing every launch of a long run would cost ~10 % of the frame rate being measured)
        self.event_stride = max(int(stride), 1)
        self.ray.setTiming(True, march_only=True, stride=self.event_stride)

    def run(self, k0, k1):
        """frames k0..k1-1 of the sequence in the reference's order"""
        if not self.python_loop:
            self.recon.run(self.seq, k0, k1 - k0)
            return
        scene, ray, cp, hd = self.scene, self.ray, self.cp, self.hd
        for k in range(k0, k1):
            if k > 0:
                ray.render(hd, scene.getHashParams(), cp, self.poses[k - 1])
            scene.integrate(self.poses[k], self.frames[k], cp, None)

    def restart(self):
        """back to an empty scene and frame 0 (after the pre-roll)"""
        self.recon.synchronize()
        if self.grid is not None:
            self.grid.reset()
        self.scene.reset()
        self.recon.reset()
        self.hd = self.scene.getHashData()

    def stage_timers(self, on):
        self.opt.s_timingsDetailledEnabled = 1 if on else 0
        self.scene.setOptions(self.opt)
        self.ray.setTiming(True, march_only=not on, stride=1 if on else self.event_stride)

    def timings(self):
        s = self.scene.getTimings()
        r = self.ray.getTimings()
        s.update(r)
        return s

    def stage_sample(self, k0, k1):
        """frames k0..k1-1 with every stage bracketed by HIP events -> per-stage device time in us per frame"""
        torch = self.torch
        self.recon.synchronize()
        s0 = self.timings()
        self.stage_timers(True)
        self.run(k0, k1)
        self.recon.synchronize()
        torch.cuda.synchronize()
        s1 = self.timings()
        self.stage_timers(False)
        n = max(k1 - k0, 1)
        return {k[:-3]: 1e3 * (s1[k] - s0[k]) / n for k in ("alloc_ms", "compactify_ms", "integrate_ms", "splat_ms", "raycast_ms", "normals_ms")}

    def close(self):
        self.recon.synchronize()
        self.recon.close()
        if self.grid is not None:
            self.grid.close()
        self.ray.close()
        self.scene.close()


def preroll(wl, seconds):
    """untimed: the sequence's first frames over and over until the device has been busy for `seconds` (clocks up, code
    objects loaded, first-touch of every buffer done); the scene is emptied again afterwards"""
    if seconds <= 0:
        return 0
    t0 = time.perf_counter()
    n = wl.n_frames if wl.frames_on_host else min(wl.n_frames, 64)  # (host-fed: every pinned page is transferred once)
    done = 0
    while time.perf_counter() - t0 < seconds:
        wl.run(0, n)
        wl.recon.synchronize()
        done += n
        if wl.streaming:  # frame 0 again on a scene whose far side has been streamed out would only churn the host grid
            wl.restart()
    wl.restart()
    return done


def stage_bytes(cfg_hp, cp, n_occ, stage, packed=True):
    """ALGORITHMIC bytes per launch (E = 20 B entry payload, V = 8 B voxel).  SURVEY.md section 8(d) prices the reference's
    integrate kernel at No*(E + 512*V*2) + 20*W*H: every block's entry, its voxels read and written, and the frame (4 B
    depth + 16 B colour per pixel).  Here the frame is read ONCE, by the alloc pass, which leaves it packed (8 B per
    pixel: what a voxel needs of its pixel); the pass over the voxels reads that.  The bytes go to the launch that
    moves them: 20*W*H read + 8*W*H written to alloc, 8*W*H (an upper bound: only the pixels under block footprints are
    read) to integrate.  packed=False: the un-fused / reference-sequence kernel, which reads the frame itself."""
    W, H = cp.m_imageWidth, cp.m_imageHeight
    E, V = 20, 8
    if stage == "raycast":
        return 52.0 * W * H
    if stage == "integrate":
        return n_occ * (E + 512 * V * 2) + (8.0 if packed else 20.0) * W * H
    if stage == "alloc":
        return (20.0 + 8.0 if packed else 4.0) * W * H
    if stage == "compactify":
        return 4.0 * cfg_hp.m_hashNumBuckets * 10 + 2.0 * E * n_occ
    if stage == "normals":
        return (16.0 * 5 + 16.0) * W * H
    raise KeyError(stage)


def valu_bound(kernel, launch_us, profiled_workload=True):
    """The ceiling that binds the ray caster: vector-instruction issue per SIMD.  What one SIMD needs per
    wave64 vector instruction was MEASURED on this machine (tools/valu_issue_probe.py -> profiles/r03_valu_issue.json: every
    SIMD holding 1 .. 8 waves, each a chain-free stream of one instruction): v_fma_f32 3.57 cycles at six waves per SIMD at the
    clock the device holds under that load (2.09 GHz) = 1.71 ns -- not the 2 cycles of the instruction's width, not the 4 of
    a lone wave's issue slot (a lone wave gets one every 6.1).  The instruction count is the committed profile's
    (SQ_INSTS_VALU per launch), not a measurement of this run -- the record says so; only the launch time beside it is live."""
    out = dict(bound="valu issue (not a contract field)", note="wave instructions x measured ns per instruction per SIMD / 1024 SIMDs; DESIGN.md section 6")
    ns = None
    try:
        for r in json.load(open(os.path.join(ROOT, "profiles", "r03_valu_issue.json")))["rows"]:
            if r["kind"] == "v_fma_f32" and r["waves_per_simd"] == 6:
                ns = float(r["ns_per_instr_per_simd"])
                out.update(ns_per_instruction_per_simd=ns, cycles_per_instruction_per_simd=r["cycles_per_instr_per_simd"], clock_mhz_under_load=r["clock_mhz_used"],
                           issue_cost_from="profiles/r03_valu_issue.json (v_fma_f32, six waves per SIMD; measured, not in this run)")
    except (OSError, KeyError, ValueError):
        pass
    if not profiled_workload or ns is None:  # the committed counters are cfg2's
        return out
    for path in ("r03_bench_pmc.csv", "r02_bench_pmc.csv"):
        try:
            import csv
            for row in csv.DictReader(open(os.path.join(ROOT, "profiles", path))):
                if row["kernel"].startswith(kernel):
                    n = float(row["SQ_INSTS_VALU"])
                    floor_us = n * ns / 1024.0 / 1e3
                    out.update(wave_instructions_per_launch=round(n), instructions_from=f"profiles/{path} (cfg2; not measured in this run)",
                               issue_floor_us=round(floor_us, 2), frac=round(floor_us / launch_us, 3) if launch_us > 0 else None)
                    return out
        except (OSError, KeyError, ValueError):
            pass
    return out


def integrate_roofline(kernel_us, n_occ, hp, cp, what):
    b = stage_bytes(hp, cp, n_occ, "integrate")
    gbs = b / (kernel_us * 1e-6) / 1e9 if kernel_us > 0 else 0.0
    W, H = cp.m_imageWidth, cp.m_imageHeight
    return dict(bound="hbm", kernel="k_integrate_fused (integrate + starve + GC)", workload=what, achieved=round(gbs, 3), peak=HBM_PEAK_GBS,
                unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 6), traffic=None, algorithmic_bytes=b,
                algorithmic_bytes_rule="No*(20 + 2*4096) + 8*W*H: entry, voxels read + written, the packed frame (upper bound); the raw frame's 20*W*H "
                                       "is the alloc pass's (it reads it and writes the packed one)",
                bytes_by_survey_8d_formula=n_occ * 8212 + 20.0 * W * H,
                avg_launch_us=round(kernel_us, 3), clock="the dispatch's own begin/end time stamps (hipExtLaunchKernel events): rocprofv3's clock",
                blocks_in_frustum=n_occ)


def cpu_model():
    """-> (model string, logical CPUs of the machine) from /proc/cpuinfo"""
    model, n = "unknown", 0
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                n += 1
                model = line.split(":", 1)[1].strip()
    except OSError:
        pass
    return model, n or (os.cpu_count() or 0)


def cpu_baseline(cfg_name, n_frames, args):
    """The CPU oracle on the first n_frames of the same workload, as BASELINE.md section 3 states it (CPU model and
    thread count in the record): the restatement built -O3 with OpenMP (oracle/libvh_oracle_omp.so -- a baseline
    only, never the checker; tests hold it to the checker's table byte for byte) on
      * one thread (`single_thread`), 16 threads (one GPU's share of an 8-GPU host) and every CPU this process may run on (the
        affinity mask after stay_near_the_gpu(), or --cpu-threads): `by_threads`;
      * `value`, `cores`: the best of them and the thread count it was measured on."""
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
    L = O.lib(omp=True)
    sc = O.OracleScene(hp, cp, rp, opt, omp=True)  # one scene for all three runs (4 GB of voxels to fault in)

    def timed(threads, frames=None):
        n_frames_ = n_frames if frames is None else frames
        # (set through the library: an OpenMP runtime that numpy or torch already started ignores the environment)
        L.vho_set_num_threads(max(1, int(threads)))
        used = int(L.vho_num_threads())
        sc.reset()
        # untimed: the first parallel regions after a change of the thread count run at a fraction of their speed
        # (the OpenMP runtime starting and placing its threads: 0.4 s each on the build machine)
        for k in range(min(2, n_frames_)):
            sc.integrate(poses[k], inputs[k][0], inputs[k][1])
            sc.render(poses[k])
        sc.reset()
        t0 = time.perf_counter()
        for k in range(n_frames_):
            if k > 0:
                sc.render(poses[k - 1])
            sc.integrate(poses[k], inputs[k][0], inputs[k][1])
        dt = time.perf_counter() - t0
        return dt, int(sc.hp.m_numOccupiedBlocks), used

    allowed = len(os.sched_getaffinity(0))
    want = args.cpu_threads if args.cpu_threads > 0 else allowed
    runs = []  # (threads used, frames, seconds)
    dt1, blocks1, _ = timed(1)
    runs.append((1, n_frames, dt1))
    dts, blockss, share = timed(min(16, allowed))
    runs.append((share, n_frames, dts))
    ok = blocks1 == blockss
    if want != share and want != 1:
        # every CPU the process may run on.  On a 128-thread NUMA node the restatement's short parallel loops scale
        # badly past a few dozen threads (measured: 27 x SLOWER on 128 threads than on 16), so this run is bounded to a
        # quarter of the frames, and `value` is the best configuration measured, with the thread count it was measured on
        n_all = max(2, n_frames // 4)
        dtn, _, cores = timed(want, n_all)
        runs.append((cores, n_all, dtn))
    sc.close()
    if not ok:
        raise RuntimeError(f"the CPU baseline's runs disagree: {blocks1} / {blockss} blocks")
    best = max(runs, key=lambda r: r[1] / r[2])
    model, logical = cpu_model()
    return dict(value=best[1] / best[2], unit="frames/s", cores=best[0], kind="port", cpu_model=model, cpus_of_the_machine=logical,
                cpus_allowed=allowed, build="gcc -O3 -fopenmp -ffp-contract=off (oracle/Makefile)",
                by_threads={str(t): round(f / dt, 3) for t, f, dt in runs}, single_thread=n_frames / dt1,
                sample=f"first {n_frames} frames of {cfg_name} ({cfg['scene']}), oracle/libvh_oracle_omp.so: " +
                       ", ".join(f"{t} thread{'s' if t > 1 else ''} {f} frames in {dt:.1f} s" for t, f, dt in runs) +
                       "; value = the best of them")


def cfg1_leg(args):
    """BASELINE.json configs[0]: ONE 640x480 frame, 4 cm voxels, 2^18 buckets, alloc until fixed point + compactify +
    integrate + GC, then the ray cast from the same pose: the HIP path and the CPU oracle (1 thread, all cores) side by side"""
    import torch
    from oracle import oracle as O
    from voxelhashing_amd import engine as E, synth, vhtypes as T
    hp, cp, rp = synth.config_params("cfg1")
    opt = T.make_scene_options(offline=True, gc=True, starve=15)
    spheres, inside, radius = synth.scene("S1")
    pose = synth.orbit_pose(0, 1000, radius, 0.0)
    scene, ray = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp)
    fr = E.synth_frame(spheres, inside, pose, cp)
    gpu = []
    for _ in range(3):  # the first pass loads code objects
        scene.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scene.integrate(pose, fr, cp, None)
        ray.render(scene.getHashData(), scene.getHashParams(), cp, pose)
        torch.cuda.synchronize()
        gpu.append(time.perf_counter() - t0)
    blocks = scene.getNumOccupiedBlocks()
    ray.close()
    scene.close()
    depth, color = O.synth_frame(spheres, inside, pose, cp)
    out = dict(workload="cfg1: one S1 frame, 640x480, P4, 2^18 buckets, offline alloc + compactify + integrate + gc + raycast",
               gpu_ms=round(1e3 * min(gpu), 3), blocks_in_frustum=int(blocks))
    allowed = len(os.sched_getaffinity(0))
    # (the -O3 baseline build; 16 threads: one GPU's share of the host -- on every CPU of a 128-thread node the short loops run slower)
    for threads, key in ((1, "cpu_1_thread_ms"), (args.cpu_threads if args.cpu_threads > 0 else min(16, allowed), "cpu_threads_ms")):
        O.lib(omp=True).vho_set_num_threads(threads)
        sc = O.OracleScene(hp, cp, rp, opt, omp=True)
        t0 = time.perf_counter()
        sc.integrate(pose, depth, color)
        sc.render(pose)
        dt = time.perf_counter() - t0
        if int(sc.hp.m_numOccupiedBlocks) != int(blocks):
            raise RuntimeError(f"cfg1: oracle {sc.hp.m_numOccupiedBlocks} blocks, HIP path {blocks}")
        sc.close()
        out[key] = round(1e3 * dt, 1)
    out["cpu_cores"] = int(O.lib(omp=True).vho_num_threads())
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    fail_rank, _, fail_where = args.standin_fail.partition(":")
    if args.standin and fail_where == "before-init" and int(os.environ.get("RANK", "0")) == int(fail_rank):
        raise SystemExit("bench.py --standin-fail: this rank gives up before the rendezvous")
    rank, world, local_rank, dist_mod = dist_setup(args.gpus, force_cpu=args.standin)
    import torch

    if args.standin:
        if fail_where == "before-barrier" and rank == int(fail_rank):
            sys.stderr.write("bench.py --standin-fail: this rank gives up before the first barrier\n")
            os._exit(3)  # (no clean-up: as a rank that crashed would)
        dev = torch.device("cpu")
        wl = StandinWorkload(rank)
        elapsed, total_frames = run_timed(wl, args.warmup, args.steps, dist_mod, dev)
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": round(total_frames / elapsed, 3), "unit": "frames/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 6),
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                              "data": "STANDIN: no GPU work, harness test only", "config": {"workload": "stand-in (sleep)"},
                              "per_rank_frames_per_s": list(PER_RANK), "roofline": None, "cpu_baseline": None}), flush=True)
        if dist_mod is not None:
            dist_mod.barrier()
            dist_mod.destroy_process_group()
        return 0

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cpu_affinity = stay_near_the_gpu(local_rank)

    n_frames = args.warmup + args.steps
    wl = GpuWorkload(args.config, n_frames, rank, args, frames_on_host=args.frames_on_host)
    # every pair of events idles the queue for a few microseconds: at least ten pairs behind avg_launch_us, no more than one frame in 32
    stride = args.event_stride if args.event_stride > 0 else max(1, min(32, args.steps // 10))
    wl.set_event_stride(stride)

    # untimed pre-roll (set-up): the device is busy with this workload's own kernels before anything is measured
    preroll_frames = preroll(wl, args.preroll_seconds)

    # warm-up, untimed.  Its last frames (never frame 0: the empty table) run with every stage timer on: per-stage
    # device times of the steady state.
    n_stage = min(64, max(args.warmup - 1, 0))
    wl.run(0, args.warmup - n_stage)
    stage_us = wl.stage_sample(args.warmup - n_stage, args.warmup) if n_stage else {}

    pre = {}
    st0 = {}

    def after_warmup():
        torch.cuda.synchronize()
        pre.update(wl.timings())  # the timed region starts from here
        st0.update(wl.recon.getStats())

    elapsed, total_frames = run_timed(wl, args.warmup, args.steps, dist_mod, dev, sync=torch.cuda.synchronize,
                                      after_warmup=after_warmup, skip_warmup=True)
    value = total_frames / elapsed
    post = wl.timings()
    st1 = wl.recon.getStats()
    n_occ = wl.scene.getNumOccupiedBlocks()
    host_enqueue_us = 1e6 * (st1["hostEnqueueSeconds"] - st0["hostEnqueueSeconds"]) / max(args.steps, 1)
    host_wait_us = 1e6 * (st1["hostWaitSeconds"] - st0["hostWaitSeconds"]) / max(args.steps, 1)

    # dominant kernel: HIP events on the launch stream around k_render, inside the timed region
    launches = max(int(post["frames"] - pre["frames"]), 1)
    dominant = max(stage_us, key=stage_us.get) if stage_us else "raycast"
    pair_us = 1e3 * (post["raycast_ms"] - pre["raycast_ms"]) / launches
    # The ray caster's launch (and computeNormals', and the fused integrate pass) are timed by the dispatch's own begin /
    # end time stamps (HIP events attached to the launch: hipExtLaunchKernel), the clock rocprofv3's kernel trace reads;
    # nothing is subtracted.  (An event PAIR around a launch reads 5-7 us more than the kernel: the records themselves;
    # what an empty pair reads is kept in the record as event_record_pair_us.)
    overhead_us = 1e3 * wl.ray.getEventPairOverheadMs()
    dom_us = pair_us
    W, H = wl.cp.m_imageWidth, wl.cp.m_imageHeight
    alg_bytes = stage_bytes(wl.hp, wl.cp, n_occ, "raycast")
    achieved = alg_bytes / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
    moved = 36.0 * W * H  # depth, depth4, colours: the normal map is left to computeNormals, which rewrites all of it
    roofline = dict(bound="hbm", kernel="k_render (raycast)", achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 6),
                    traffic=None, traffic_note="PMC counters are collected in separate rocprofv3 passes: profiles/README.md holds the measured HBM bytes of this kernel",
                    algorithmic_bytes=alg_bytes, algorithmic_bytes_rule="SURVEY.md 8(d): 52*W*H, the four output maps of renderKernel",
                    bytes_stored_by_launch=moved, frac_of_bytes_stored=round(moved / (dom_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 6) if dom_us > 0 else 0.0,
                    avg_launch_us=round(dom_us, 3), clock="the dispatch's own begin/end time stamps (HIP events attached to the launch, inside the timed region)",
                    event_record_pair_us=round(overhead_us, 3), launches_timed=launches, event_stride=wl.event_stride,
                    dominant_stage_of_warmup=dominant,
                    second_bound=valu_bound("k_render", dom_us, args.config == "cfg2" and wl.cfg["scene"] == "S1"),
                    stage_us_warmup={k: round(v, 3) for k, v in stage_us.items()}, stage_frames=n_stage, blocks_in_frustum=n_occ)
    rooflines = {}
    if stage_us.get("integrate", 0.0) > 0.0:
        kus = stage_us["integrate"]
        rooflines["integrate"] = integrate_roofline(kus, n_occ, wl.hp, wl.cp, f"{args.config} ({wl.cfg['scene']}), steady state of the warm-up")

    result = None
    if rank == 0:
        cfg = wl.cfg
        stream_note = ""
        if wl.streaming:
            stream_note = "stream out/in per frame + "
        result = {
            "metric": METRIC,
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
                            f"{stream_note}alloc+compactify+integrate+{'gc+' if not args.no_gc else ''}raycast+normals per frame",
                "alloc_mode": "offline (fixed point)" if args.offline else "online (one pass per frame)",
                "streams": "one independent scene per GPU",
                "frame_loop": "python (2 ctypes calls per frame)" if args.python_loop else "native (vh_reconstruction_run: one call for all frames)",
                "alloc_ahead": not args.no_alloc_ahead and not wl.streaming and not args.python_loop,
                "frames_in_flight": args.frames_in_flight,
                "preroll_frames": preroll_frames,
                "frames_on_host": bool(args.frames_on_host),
                "cpu_affinity": cpu_affinity,
            },
            "per_rank_frames_per_s": list(PER_RANK),
            "host_enqueue_us_per_frame": round(host_enqueue_us, 3),
            "host_wait_us_per_frame": round(host_wait_us, 3),
            # timed frames whose pass over the voxels rode in computeNormals' launch (two launches a frame instead of three)
            "frames_in_two_launches": int(st1.get("framesInTwoLaunches", 0) - st0.get("framesInTwoLaunches", 0)),
            "roofline": roofline,
            "rooflines": rooflines,
        }
        if wl.streaming:
            result["streaming"] = dict(sphere_centre_z=wl.streaming_sphere[0], sphere_radius=wl.streaming_sphere[1], chunk_extent=wl.streaming_sphere[2],
                                       parts=int(wl.opt.s_streamingOutParts), worker_thread=True,
                                       frames_without_a_streaming_step=int(st1.get("streamingStepsSkipped", 0) - st0.get("streamingStepsSkipped", 0)),
                                       frames_pipelined=int(st1.get("streamingFramesPipelined", 0) - st0.get("streamingFramesPipelined", 0)), blocks_out=int(st1["blocksStreamedOut"] - st0["blocksStreamedOut"]),
                                       blocks_in=int(st1["blocksStreamedIn"] - st0["blocksStreamedIn"]),
                                       blocks_per_second=round((st1["blocksStreamedOut"] - st0["blocksStreamedOut"] + st1["blocksStreamedIn"] - st0["blocksStreamedIn"]) / elapsed, 1))
    wl.close()
    del wl
    torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_extra_legs:
        # 1. the integrate kernel where it can fill the machine: camera inside a 3 m sphere (S2), 1 cm voxels
        try:
            n = 48
            dl = GpuWorkload("cfg3", n, 0, args, scene="S2", streaming=False)
            dl.set_event_stride(1)
            dl.run(0, n - 16)
            dus = dl.stage_sample(n - 16, n)
            docc = dl.scene.getNumOccupiedBlocks()
            kus = dus["integrate"]
            result["rooflines"]["integrate_dense"] = integrate_roofline(kus, docc, dl.hp, dl.cp, "cfg3 tables, scene S2 (every pixel valid), 1 cm voxels, no streaming; 16 frames")
            result["rooflines"]["integrate_dense"]["stage_us"] = {k: round(v, 3) for k, v in dus.items()}
            dl.close()
            del dl
            torch.cuda.empty_cache()
        except Exception as e:
            result["rooflines"]["integrate_dense"] = dict(failed=str(e))
        # 2. the same loop fed from the host: float depth + RGBX bytes in pinned memory, uploaded on a copy stream.
        # The rate of the link depends on the pinned allocation it reads from -- on this pool about every other allocation
        # of the leg's buffers copies at a third of the rate of the others, for as long as it lives and whatever thread or
        # NUMA node made it (tools/h2d_leg_probe.py, tools/h2d_streams_probe.py: not the stream, not the copy engine) --
        # so the leg does what a long-running feeder would do at start-up: allocate, measure, and allocate again (three
        # times at most) if the buffers turn out slow.  Every trial is in the record; value_with_upload is the median trial, value_with_upload_best the best.
        try:
            hw, hs = min(args.warmup, 20), min(args.steps, 300)
            trials = []
            best = None
            for trial in range(3):
                hl = GpuWorkload(args.config, hw + hs, 0, args, frames_on_host=True)
                hl.set_event_stride(1 << 30)
                # untimed passes over every frame first: the first transfers out of a freshly pinned page are several times
                # slower than the following ones (the main workload's pre-roll does the same for its frames)
                preroll(hl, max(args.preroll_seconds, 0.3))
                hl.run(0, hw)
                hl.recon.synchronize()
                torch.cuda.synchronize()
                h0 = hl.recon.getStats()
                t0 = time.perf_counter()
                hl.run(hw, hw + hs)
                hl.recon.synchronize()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                h1 = hl.recon.getStats()
                ups = max(h1["uploadsTimed"] - h0["uploadsTimed"], 1)
                one = dict(frames_per_s=round(hs / dt, 3), upload_us=round(1e3 * (h1["uploadMs"] - h0["uploadMs"]) / ups, 3), bytes_per_frame=int(h1["uploadBytes"]))
                trials.append(one)
                hl.close()
                del hl
                torch.cuda.empty_cache()
                if best is None or one["frames_per_s"] > best["frames_per_s"]:
                    best = one
                if one["frames_per_s"] >= 0.6 * value:
                    break
            ranked = sorted(trials, key=lambda t: t["frames_per_s"])
            med = ranked[(len(ranked) - 1) // 2]  # the median trial (the lower one of two)
            result["value_with_upload"] = med["frames_per_s"]
            result["value_with_upload_best"] = best["frames_per_s"]
            result["upload"] = dict(frames=hs, upload_us=med["upload_us"], bytes_per_frame=med["bytes_per_frame"],
                                    allocations_tried=[dict(frames_per_s=t["frames_per_s"], upload_us=t["upload_us"]) for t in trials],
                                    how="pinned host frames (float depth + RGBX colour), hipMemcpyAsync on two copy streams (one copy engine each) into a ring of "
                                        "four staging slots, colour converted on the device; uploads run beside the frame loop, which only waits for "
                                        "their events; upload_us spans both copies and the conversion, every 8th frame timed; value_with_upload is the MEDIAN "
                                        "of up to three allocations of the frame buffers (a slow one copies at a third of the rate: all listed; "
                                        "value_with_upload_best is the best)")
        except Exception as e:
            result["value_with_upload"] = None
            result["upload"] = dict(failed=str(e))

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(args.config, args.cpu_frames, args)
            except Exception as e:  # the baseline must never sink the GPU result
                result["cpu_baseline"] = dict(value=None, unit="frames/s", cores=1, kind="port", sample=f"failed: {e}")
            if not args.no_extra_legs:
                try:
                    result["cfg1"] = cfg1_leg(args)
                except Exception as e:
                    result["cfg1"] = dict(failed=str(e))
        else:
            result["cpu_baseline"] = None
        if args.stages:
            print(json.dumps(stage_us), file=sys.stderr)
        print(json.dumps(result), flush=True)
    if dist_mod is not None:
        dist_mod.barrier()
        dist_mod.destroy_process_group()
    return result


if __name__ == "__main__":
    r = main()
    sys.exit(r if isinstance(r, int) else 0)
