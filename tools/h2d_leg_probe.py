#!/usr/bin/env python3
"""The host-fed leg of bench.py several times in one process: does its rate depend on where the pinned frames land?"""
import argparse, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=320)
    ap.add_argument("--trials", type=int, default=4)
    a = ap.parse_args()
    import torch
    args = bench.parse_args([]) if hasattr(bench, "parse_args") else None
    print("cpus allowed:", sorted(os.sched_getaffinity(0)))
    try:
        p = torch.cuda.get_device_properties(0)
        bus = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        for f in ("numa_node", "local_cpulist"):
            print(f, open(f"/sys/bus/pci/devices/{bus}/{f}").read().strip())
    except Exception as e:
        print("no pci info:", e)
    try:
        print("numa nodes:", os.listdir("/sys/devices/system/node"))
    except Exception as e:
        print(e)
    for t in range(a.trials):
        hl = bench.GpuWorkload("cfg2", a.frames, 0, args, frames_on_host=True)
        hl.set_event_stride(1 << 30)
        bench.preroll(hl, 0.3)
        hl.run(0, 20)
        hl.recon.synchronize(); torch.cuda.synchronize()
        h0 = hl.recon.getStats(); t0 = time.perf_counter()
        hl.run(20, a.frames)
        hl.recon.synchronize(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0; h1 = hl.recon.getStats()
        ups = max(h1["uploadsTimed"] - h0["uploadsTimed"], 1)
        addr = hl.h_depth.data_ptr()
        node = "?"
        print(json.dumps(dict(trial=t, fps=round((a.frames - 20) / dt, 1), upload_us=round(1e3 * (h1["uploadMs"] - h0["uploadMs"]) / ups, 1), addr=hex(addr))))
        hl.close(); del hl
        torch.cuda.empty_cache()

main()
