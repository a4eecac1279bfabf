#!/usr/bin/env python3
"""Plays `.sens` sequences through the frame loop the way the reference application does (parameter file, tracking
parameter file, optional mesh at the end) and prints one JSON line with the timing.

    python tools/replay.py --params zParameters.txt [--tracking zParametersTracking.txt] [--sens a.sens b.sens]
                           [--mesh scan.ply] [--max-frames N] [--record out.sens]

Without --sens the files named by s_binaryDumpSensorFile[i] in the parameter file are played."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--params", required=True)
    ap.add_argument("--tracking", default=None)
    ap.add_argument("--sens", nargs="*", default=None)
    ap.add_argument("--mesh", default=None)
    ap.add_argument("--record", default=None, help="write what was processed, with the poses used, to this .sens file")
    ap.add_argument("--max-frames", type=int, default=None)
    args = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU (there is no CPU fallback)")
    from voxelhashing_amd import reconstruction as R
    g = R.read_app_state(args.params)
    if args.record:
        g.s_recordData = 1
    t = R.read_tracking_state(args.tracking) if args.tracking else None
    rec = R.Reconstruction(g, t, args.sens or None)
    t0 = time.perf_counter()
    n = rec.run(args.max_frames)
    rec.scene.synchronize()
    dt = time.perf_counter() - t0
    out = dict(frames=n, seconds=round(dt, 3), frames_per_s=round(n / dt, 1) if dt > 0 else None, lost_frames=rec.lost_frames,
               blocks=rec.scene.getNumOccupiedBlocks(), heap_free=rec.scene.getHeapFreeCount(),
               pose_source="recorded trajectory" if g.s_binaryDumpSensorUseTrajectory and not g.s_binaryDumpSensorUseTrajectoryOnlyInit else "projective ICP")
    if args.record:
        out["recorded"] = rec.saveRecordedFramesToFile(args.record)
    if args.mesh:
        m = rec.extractIsoSurface(args.mesh)
        out["mesh"] = dict(file=args.mesh, vertices=int(len(m["vertices"])), faces=int(len(m["faces"])))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
