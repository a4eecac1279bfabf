#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of one evidence run (tools/collect_profile.sh -> gpurun_out/final/) into the summaries
kept under profiles/.

    python tools/summarize_profile.py <run dir> <tag>

Writes profiles/<tag>_bench_kernel_stats.csv and <tag>_dense_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the
default bench command and of the dense-scene run), <tag>_bench_pmc.csv and <tag>_dense_pmc.csv (per-kernel averages of
the counters, one --pmc pass each), <tag>_bench_line*.json (the bench lines) and profiles/traffic.json (HBM bytes per
launch of the kernels the bench prices: 2 * FETCH_SIZE + WRITE_SIZE, FETCH doubled for gfx950 as MI355X_MICROARCH.md
prescribes for wide reads)."""
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


def pmc_table(run, prefix):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(run, prefix + "_*"))):
        f = os.path.join(d, "p_counter_collection.csv")
        if os.path.isdir(d) and os.path.exists(f):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in acc for c in acc[k]})
    rows = []
    for k in sorted(acc):
        n = max(len(v) for v in acc[k].values())
        if n < 5:
            continue
        row = {"kernel": k, "dispatches": n}
        for c in counters:
            v = acc[k].get(c, [])
            v = v[len(v) // 4:]  # the first quarter is warm-up
            row[c] = round(sum(v) / len(v), 1) if v else ""
        if row.get("TCC_HIT_sum") != "" and row.get("TCC_MISS_sum") != "":
            h, m = row["TCC_HIT_sum"], row["TCC_MISS_sum"]
            row["l2_hit_rate"] = round(h / (h + m), 3) if h + m else ""
        if row.get("FETCH_SIZE") != "" and row.get("WRITE_SIZE") != "":
            row["hbm_bytes_2xFETCH_plus_WRITE"] = int(round((2 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024))
        rows.append(row)
    return rows, counters


def write_table(path, rows, counters):
    cols = ["kernel", "dispatches"] + counters + ["l2_hit_rate", "hbm_bytes_2xFETCH_plus_WRITE"]
    with open(path, "w") as o:
        o.write(",".join(cols) + "\n")
        for r in rows:
            o.write(",".join(str(r.get(c, "")) for c in cols) + "\n")


def main():
    run, tag = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    traffic = {}
    for what, prefix, trace in (("bench", "pmc", "trace"), ("dense", "pmcdense", "trace_dense")):
        rows, counters = pmc_table(run, prefix)
        write_table(os.path.join(prof, f"{tag}_{what}_pmc.csv"), rows, counters)
        shutil.copy(os.path.join(run, trace, "p_kernel_stats.csv"), os.path.join(prof, f"{tag}_{what}_kernel_stats.csv"))
        stats = {short(r["Name"]): r for r in csv.DictReader(open(os.path.join(run, trace, "p_kernel_stats.csv")))}
        print(f"\n### {what}\n| kernel | calls | avg us | % | FETCH KB (raw) | WRITE KB | L2 hit | VALU insts | VALU busy (quad-cycles) |")
        pm = {r["kernel"]: r for r in rows}
        for k, r in stats.items():
            if float(r["Percentage"]) > 0.5:
                p = pm.get(k, {})
                print(f"| {k} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} | {p.get('FETCH_SIZE', '')} | {p.get('WRITE_SIZE', '')} | "
                      f"{p.get('l2_hit_rate', '')} | {p.get('SQ_INSTS_VALU', '')} | {p.get('SQ_ACTIVE_INST_VALU', '')} |")
        traffic[what] = {r["kernel"]: r.get("hbm_bytes_2xFETCH_plus_WRITE") for r in rows if r.get("hbm_bytes_2xFETCH_plus_WRITE")}
    # rocprofv3's stats average every launch of the process (the untimed pre-roll on the orbit's first frames included); what
    # bench.py times live is the timed region: the same kernels averaged over the last `steps` dispatches of the trace
    try:
        line = json.loads([l for l in open(os.path.join(run, "bench_under_profiler.json")) if l.startswith("{")][-1])
        steps = int(line["steps"])
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(run, "trace", "p_kernel_trace.csv"))):
            per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        with open(os.path.join(prof, f"{tag}_bench_kernel_stats_timed_region.csv"), "w") as o:
            o.write("kernel,launches_in_trace,avg_ns_all,avg_ns_last_%d_launches\n" % steps)
            print(f"\n### timed region (last {steps} launches of the trace)")
            for k in sorted(per):
                v = sorted(per[k])
                if len(v) >= steps:
                    d = [e - b for b, e in v]
                    o.write(f"{k},{len(d)},{sum(d) / len(d):.0f},{sum(d[-steps:]) / steps:.0f}\n")
                    print(f"{k}: all {sum(d) / len(d) / 1e3:.2f} us, timed region {sum(d[-steps:]) / steps / 1e3:.2f} us; live in that run: "
                          f"{line['roofline']['avg_launch_us'] if k.startswith('k_render') else ''}")
    except (OSError, KeyError, ValueError, IndexError) as e:
        print("no timed-region table:", e)
    traffic["_note"] = (f"HBM bytes per launch from profiles/{tag}_*_pmc.csv: (2*FETCH_SIZE + WRITE_SIZE)*1024; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                        "(gfx950 reports half of wide reads; gathers are uncalibrated, so this is an upper bound); separate --pmc passes.  bench.py does "
                        "not copy these numbers into its line: `traffic` there is null unless counters ran in that very run.")
    json.dump(traffic, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
    for f in glob.glob(os.path.join(run, "bench*.json")):
        n = os.path.basename(f)[len("bench"):]
        if n != "_under_profiler.json":
            lines = [l for l in open(f) if l.startswith("{")]
            if lines:
                open(os.path.join(prof, f"{tag}_bench_line{n}"), "w").write(lines[-1])
    print()
    for f in sorted(glob.glob(os.path.join(run, "bench*.json"))):
        lines = [l for l in open(f) if l.startswith("{")]
        if not lines:
            continue
        d = json.loads(lines[-1])
        print(os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d.get("value_with_upload"),
              (d.get("rooflines") or {}).get("integrate_dense", {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))


if __name__ == "__main__":
    main()
