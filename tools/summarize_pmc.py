#!/usr/bin/env python3
"""Average every PMC counter per kernel from rocprofv3 counter_collection CSVs below a directory."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("chisel_hip::", "").replace("void ", "")[:60]


def main(out):
    for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", os.path.relpath(f, out))
        for k, ctrs in agg.items():
            if "rocclr" in k:
                continue
            n = max(len(v) for v in ctrs.values())
            print("  %-60s dispatches %d" % (k, n))
            for c, v in sorted(ctrs.items()):
                print("      %-28s avg %16.1f" % (c, sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1])
