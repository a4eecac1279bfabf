#!/usr/bin/env python3
"""Measures the sensor pre-processing kernels on a large image so that the kernels, not the
launches, are what is timed: each kernel `--reps` times on a WxH frame resident in HBM, HIP-event time per launch and
the algorithmic bytes it moves (every input pixel read once, every output pixel written once) against the 8 TB/s roof.
One JSON line.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel view.

    python tools/bench_sensor.py [--width 3840 --height 2160 --reps 50]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    import torch
    from voxelhashing_amd import lib, vhtypes as T
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    L = lib.load()
    W, H = args.width, args.height
    n = W * H
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    depth = (1.0 + 2.0 * torch.rand((H, W), device=dev, generator=g)).float()
    depth[torch.rand((H, W), device=dev, generator=g) < 0.05] = float("-inf")
    rgbx = torch.randint(0, 256, (H, W, 4), device=dev, dtype=torch.uint8, generator=g)
    colf = torch.empty((H, W, 4), device=dev, dtype=torch.float32)
    out1 = torch.empty((H, W), device=dev, dtype=torch.float32)
    out4 = torch.empty((H, W, 4), device=dev, dtype=torch.float32)
    half1 = torch.empty((H // 2, W // 2), device=dev, dtype=torch.float32)
    half4 = torch.empty((H // 2, W // 2, 4), device=dev, dtype=torch.float32)
    cp = T.make_depth_camera_params(W, H)
    p = lambda t: C.c_void_p(t.data_ptr())
    lib.check(L.vh_convert_color_raw_to_float4(p(colf), p(rgbx), W, H, None))
    cases = [
        ("convert_color_raw_to_float4", lambda: L.vh_convert_color_raw_to_float4(p(colf), p(rgbx), W, H, None), 4 * n + 16 * n),
        ("resample_float_map (same size)", lambda: L.vh_resample_float_map(p(out1), W, H, p(depth), W, H, None), 4 * n + 4 * n),
        ("resample_float_map (to half size)", lambda: L.vh_resample_float_map(p(half1), W // 2, H // 2, p(depth), W, H, None), 4 * n + n),
        ("resample_float4_map (to half size)", lambda: L.vh_resample_float4_map(p(half4), W // 2, H // 2, p(colf), W, H, None), 16 * n + 4 * n),
        ("convert_color_to_intensity_float", lambda: L.vh_convert_color_to_intensity_float(p(out1), p(colf), W, H, None), 16 * n + 4 * n),
        ("convert_depth_float_to_camera_space_float4", lambda: L.vh_convert_depth_float_to_camera_space_float4(p(out4), p(depth), C.byref(cp), W, H, None), 4 * n + 16 * n),
        ("compute_normals", lambda: L.vh_compute_normals(p(colf), p(out4), W, H, None), 16 * n + 16 * n),
        ("gauss_filter_float_map (sigmaD 2: 9x9)", lambda: L.vh_gauss_filter_float_map(p(out1), p(depth), 2.0, 0.1, W, H, None), 4 * n + 4 * n),
        ("erode_depth_map (5: 11x11)", lambda: L.vh_erode_depth_map(p(out1), p(depth), 5, W, H, 0.05, 0.3, None), 4 * n + 4 * n),
    ]
    rows = []
    for name, fn, nbytes in cases:
        lib.check(fn())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / args.reps
        gbs = nbytes / (us * 1e-6) / 1e9
        rows.append(dict(kernel=name, us=round(us, 2), algorithmic_bytes=nbytes, GBps=round(gbs, 1), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 4)))
    print(json.dumps(dict(metric="sensor pre-processing kernels: HBM GB/s vs peak", image=[W, H], peak_GBps=HBM_PEAK_GBS, kernels=rows)))


if __name__ == "__main__":
    main()
