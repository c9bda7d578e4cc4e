#!/usr/bin/env python3
"""GPU time of the steps of a wait-free sharded recompute from a rocprofv3 kernel trace of tools/sharded_host_time.py:
    python3 tools/sharded_kernel_time.py <rocprof output dir> <shards> <wait-free recomputes in the run>"""
import csv, glob, os, sys
from collections import defaultdict
out, shards, recomputes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = [("dirty ids", ("list_dirty_ids_kernel",)), ("plan", ("shell_jobs_kernel", "shell_items_kernel")), ("export + status", ("shell_export_kernel",)),
         ("import", ("shell_ensure_ghosts_fixed_kernel", "shell_import_fixed_kernel")),
         ("mesh", ("clear_dirty_kernel", "mesh_count_kernel_16", "mesh_triangle_kernel", "shell_abort_relist_kernel")), ("drop", ("shell_reset_boxes_kernel", "shell_remove_ghosts_kernel"))]
agg = defaultdict(list)
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("chisel_hip::", "").replace("void ", "").split("<")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
per = shards * recomputes  # launches of a once-per-recompute kernel
total = 0.0
print("GPU time per recompute of ONE shard (%d shards of the driver's stream on one GPU, %d wait-free recomputes, kernels only):" % (shards, recomputes))
for name, kernels in steps:
    t = 0.0
    parts = []
    for k in kernels:
        v = agg.get(k, [])
        # the blocking first recompute launched some of these kernels as well (one per shard): the average over all launches is used
        if v:
            t += sum(v) / len(v) / 1e3
            parts.append("%s %.1f (n %d)" % (k, sum(v) / len(v) / 1e3, len(v)))
    total += t
    print("  %-14s %7.1f us   %s" % (name, t, ", ".join(parts)))
print("  %-14s %7.1f us" % ("sum", total))
