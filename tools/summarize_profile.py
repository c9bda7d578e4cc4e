#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into the summary committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("chisel_hip::", "").replace("void ", "")[:70]


def main(out):
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
        rows = list(csv.DictReader(open(f)))
        for r in rows[:12]:
            print("  %-70s calls %7s  total %12s ns  avg %12s ns  pct %s" % (short(r.get("Name", "")), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
    # our own per-kernel average from the trace (skipping nothing: includes warm-up frames)
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        agg = defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print("== kernel trace durations:", os.path.relpath(f, out))
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
            v2 = sorted(v)
            print("  %-70s n %6d  avg %9.2f us  p50 %9.2f us  p90 %9.2f us" % (k, len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3, v2[int(len(v2) * 0.9)] / 1e3))
    for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
            agg = defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == ctr:
                    agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            print("== %s per dispatch (counter units: KiB per rocprofv3 definition):" % ctr, os.path.relpath(f, out))
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
                print("  %-70s n %6d  avg %12.1f  total %14.1f" % (k, len(v), sum(v) / len(v), sum(v)))
    # machine-readable digest next to the text: per kernel average duration and PMC bytes per dispatch
    digest = {"kernels": {}}
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            digest["kernels"].setdefault(short(r["Name"]), {})["avg_us"] = float(r["AverageNs"]) / 1e3
            digest["kernels"][short(r["Name"])]["calls"] = int(r["Calls"])
    for tag, ctr, key in (("pmc_fetch", "FETCH_SIZE", "fetch_kib"), ("pmc_write", "WRITE_SIZE", "write_kib")):
        for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
            agg = defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == ctr:
                    agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                digest["kernels"].setdefault(k, {})[key] = sum(v) / len(v)
    import json
    # which bench command this is a profile of (bench.py: pmc_traffic() takes a digest only for exactly that command)
    try:
        b = json.loads(open(os.path.join(out, "bench_plain.json")).read().strip().splitlines()[-1])
        digest["bench"] = {"steps": b["steps"], "warmup": b["warmup"], "frames_per_launch": b["roofline"]["frames_per_launch"],
                           "mesh_every": b["config"]["mesh_every"], "image": b["config"]["image"], "voxel_m": b["config"]["voxel_m"],
                           "chunk": b["config"]["chunk"], "color": b["config"]["color"]}
        digest["bench_line"] = {"value": b["value"], "roofline": {k: b["roofline"][k] for k in ("frac", "avg_kernel_us", "algorithmic_bytes_per_launch", "launches")}}
    except Exception as e:  # noqa
        print("no bench_plain.json line:", e)
    # the integration kernel's launches inside the timed windows only (every pass of bench.py runs W warm-up frames first; the
    # all-launch average above mixes those in): passes are equal-length runs of launches, the timed launches are each pass's tail
    try:
        per_pass = -(-(b["steps"] + b["warmup"]) // b["config"]["frames_per_call"]) if b["warmup"] % b["config"]["frames_per_call"] == 0 else None
        if per_pass is None:
            per_pass = -(-b["warmup"] // b["config"]["frames_per_call"]) + -(-b["steps"] // b["config"]["frames_per_call"])
        timed = -(-b["steps"] // b["config"]["frames_per_call"])
        for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if short(r["Kernel_Name"]).startswith("integrate_kernel")]
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
            if per_pass and len(d) % per_pass == 0:
                sel = [x for i, x in enumerate(d) if i % per_pass >= per_pass - timed]
                digest["integrate_timed_window"] = {"launches": len(sel), "avg_us": sum(sel) / len(sel) / 1e3, "passes": len(d) // per_pass,
                                                    "launches_per_pass": per_pass, "timed_launches_per_pass": timed}
                print("== integrate_kernel inside the timed windows only: %d launches (%d passes x last %d of %d), avg %.2f us"
                      % (len(sel), len(d) // per_pass, timed, per_pass, sum(sel) / len(sel) / 1e3))
    except Exception as e:  # noqa
        print("timed-window average not derived:", e)
    json.dump(digest, open(os.path.join(out, "digest.json"), "w"), indent=1, sort_keys=True)
    for name in ("bench_plain.json", "bench_trace.json"):
        p = os.path.join(out, name)
        if os.path.exists(p):
            print("== %s: %s" % (name, open(p).read().strip()[:3000]))


if __name__ == "__main__":
    main(sys.argv[1])
