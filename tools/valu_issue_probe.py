#!/usr/bin/env python3
"""How fast does one SIMD of this machine issue vector instructions?  (the ray caster's "vector
issue" ceiling assumed 4 cycles per wave64 instruction; the guide lists 2 cycles for a full machine.)

Every SIMD of the device runs N in {1, 2, 3, 4, 6, 8} waves, each a chain-free stream of 32 * iters instructions of one
kind (vh_debug_valu_probe).  Per wave the kernel stamps s_memrealtime (100 MHz) at both ends and the s_memtime ticks
in between.  Reported per (kind, N):
  ns_per_instr_simd = span of the launch (first start .. last end) / (N * 32 * iters): the time one SIMD needs per wave
                      instruction when N waves share it;
  cyc              = that in shader cycles, with the clock taken from the ratio s_memtime / s_memrealtime of the same
                      waves if s_memtime runs at the shader clock (printed), else from --mhz.
Writes a JSON record (default profiles/r03_valu_issue.json)."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20000)
    ap.add_argument("--mhz", type=float, default=2400.0, help="shader clock to convert with when s_memtime is not the shader clock")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_valu_issue.json"))
    a = ap.parse_args()
    from voxelhashing_amd import lib
    L = lib.load()
    cap = 8 * 4 * 512
    d = C.c_void_p()
    assert L.vh_malloc(C.byref(d), 16 * cap) == 0
    host = np.zeros((cap, 4), dtype=np.uint32)
    kinds = ["v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_mul_lo_u32"]
    rec = dict(iters=a.iters, instructions_per_wave=32 * a.iters, rows=[])
    for mode, kind in enumerate(kinds):
        for n in (1, 2, 3, 4, 6, 8):
            best = None
            for rep in range(3):
                nw = C.c_uint32(0)
                assert L.vh_debug_valu_probe(mode, n, a.iters, d, C.byref(nw), None) == 0
                assert L.vh_device_synchronize() == 0
                assert L.vh_memcpy_d2h(host.ctypes.data, d, 16 * nw.value, None) == 0
                st = host[: nw.value].astype(np.int64)
                t0, t1 = st[:, 0], st[:, 1]
                span = ((t1 - t0.min()) & 0xFFFFFFFF).max() * 10.0  # ns (100 MHz ticks)
                life = ((t1 - t0) & 0xFFFFFFFF) * 10.0
                clk_ratio = float(np.median(st[:, 2] / np.maximum(life, 1.0)))  # s_memtime ticks per ns
                # per SIMD (where the waves really sat: the dispatcher does not deal them evenly): instructions its waves
                # issued / the time between its first wave's start and its last wave's end
                w3 = st[:, 3]
                simd_key = ((w3 >> 20) & 0xF) * 4096 + ((w3 >> 13) & 0x7) * 512 + ((w3 >> 12) & 1) * 256 + ((w3 >> 8) & 0xF) * 16 + ((w3 >> 4) & 0x3)
                per_simd = []
                counts = []
                for key in np.unique(simd_key):
                    m = simd_key == key
                    busy = (((t1[m] - t0[m].min()) & 0xFFFFFFFF).max()) * 10.0
                    per_simd.append(busy / (int(m.sum()) * 32 * a.iters))
                    counts.append(int(m.sum()))
                per_simd, counts = np.array(per_simd), np.array(counts)
                exact = per_simd[counts == n] if (counts == n).any() else per_simd
                row = dict(kind=kind, waves_per_simd=n, waves=int(nw.value), span_us=round(span / 1e3, 2), median_wave_life_us=round(float(np.median(life)) / 1e3, 2),
                           simds_seen=int(len(counts)), waves_on_a_simd_min_max=[int(counts.min()), int(counts.max())], simds_with_exactly_n=int((counts == n).sum()),
                           ns_per_instr_per_simd=float(np.median(exact)),
                           ns_per_instr_simd=span / (n * 32 * a.iters), ns_per_instr_simd_median_wave=float(np.median(life)) / (n * 32 * a.iters),
                           memtime_ticks_per_ns=round(clk_ratio, 4))
                if best is None or row["span_us"] < best["span_us"]:
                    best = row
            mhz = 1e3 * best["memtime_ticks_per_ns"] if best["memtime_ticks_per_ns"] > 0.5 else a.mhz
            best["clock_mhz_used"] = round(mhz, 1)
            best["cycles_per_instr_per_simd"] = round(best["ns_per_instr_per_simd"] * mhz / 1e3, 3)  # THE figure: median over the SIMDs that held exactly n waves
            best["ns_per_instr_per_simd"] = round(best["ns_per_instr_per_simd"], 5)
            best["cycles_per_instr_simd"] = round(best["ns_per_instr_simd"] * mhz / 1e3, 3)
            best["cycles_per_instr_simd_median_wave"] = round(best["ns_per_instr_simd_median_wave"] * mhz / 1e3, 3)
            best["ns_per_instr_simd"] = round(best["ns_per_instr_simd"], 5)
            best["ns_per_instr_simd_median_wave"] = round(best["ns_per_instr_simd_median_wave"], 5)
            rec["rows"].append(best)
            print(json.dumps(best), flush=True)
    L.vh_free(d)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
