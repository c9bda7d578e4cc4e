#!/usr/bin/env python3
"""Condense `make -C cvids_amd/csrc resource-usage` output: one line per kernel (registers, scratch, occupancy)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else subprocess.run(["make", "-C", "cvids_amd/csrc", "resource-usage"], capture_output=True, text=True).stderr
cur = None
for line in txt.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
    elif cur is not None:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS"):
            print("%-70s sgpr %3s vgpr %3s scratch %4s occ %s lds %s" % (cur["name"][:70], cur.get("TotalSGPRs"), cur.get("VGPRs"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
            cur = None
