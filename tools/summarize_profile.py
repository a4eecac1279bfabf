#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of one evidence run into the summaries kept under profiles/.

    python tools/summarize_profile.py <run dir> <tag>

<run dir> holds trace/p_kernel_stats.csv (--kernel-trace --stats of the default bench command) and
pmc_FETCH_SIZE/, pmc_WRITE_SIZE/, pmc_TCC_HIT_sum_TCC_MISS_sum/ (one --pmc pass each, p_counter_collection.csv), plus
the bench lines bench*.json.  Writes profiles/<tag>_bench_kernel_stats.csv, profiles/<tag>_bench_pmc_hbm_l2.csv,
profiles/<tag>_bench_line*.json and refreshes profiles/traffic.json (k_render: (2*FETCH_SIZE + WRITE_SIZE) KB, FETCH
doubled for gfx950 as MI355X_MICROARCH.md prescribes)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(k_\w+|__amd_rocclr_\w+)(<\w+>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main():
    run, tag = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_TCC_HIT_sum_TCC_MISS_sum"):
        for r in csv.DictReader(open(os.path.join(run, d, "p_counter_collection.csv"))):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = lambda l: sum(l) / len(l) if l else 0.0
    rows = []
    for k in sorted(acc):
        a = acc[k]
        f, w, h, m = (avg(a.get(c, [])) for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"))
        rows.append((k, len(a.get("FETCH_SIZE", [])), round(f, 1), round(w, 1), round(h), round(m), round(h / (h + m), 3) if h + m else ""))
    with open(os.path.join(prof, f"{tag}_bench_pmc_hbm_l2.csv"), "w") as o:
        o.write("kernel,dispatches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,TCC_HIT_sum_avg,TCC_MISS_sum_avg,l2_hit_rate\n")
        for r in rows:
            o.write(",".join(map(str, r)) + "\n")
    shutil.copy(os.path.join(run, "trace", "p_kernel_stats.csv"), os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
    for f in glob.glob(os.path.join(run, "bench*.json")):
        n = os.path.basename(f)[len("bench"):]
        if n != "_under_profiler.json":
            shutil.copy(f, os.path.join(prof, f"{tag}_bench_line{n}"))
    rk = [r for r in rows if r[0].startswith("k_render")][0]
    tpath = os.path.join(prof, "traffic.json")
    t = json.load(open(tpath))
    t["cfg2"].update(raycast=int(round((2 * rk[2] + rk[3]) * 1024)), fetch_kb_raw=rk[2], write_kb=rk[3])
    t["cfg2"]["_note"] = (f"k_render per launch (profiles/{tag}_*): (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes; FETCH_SIZE doubled per "
                          "MI355X_MICROARCH.md (gfx950 reports half of wide reads; the 8-byte gathers of this kernel are uncalibrated, so this is "
                          "an upper bound); separate --pmc passes, see profiles/README.md")
    json.dump(t, open(tpath, "w"), indent=1)
    total = 0.0
    print("| kernel | calls | avg us | % | FETCH KB | WRITE KB | L2 hit |")
    pm = {r[0]: r for r in rows}
    for r in csv.DictReader(open(os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))):
        if float(r["Percentage"]) > 0.5:
            k = short(r["Name"])
            p = pm.get(k, ("", 0, "", "", "", "", ""))
            print(f"| {k} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} | {p[2]} | {p[3]} | {p[6]} |")
    for f in sorted(glob.glob(os.path.join(run, "bench*.json"))):
        d = json.load(open(f))
        print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("cpu_baseline"))


if __name__ == "__main__":
    main()
