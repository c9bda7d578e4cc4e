#!/usr/bin/env python3
"""Regenerate tests/golden/ref_kat.json from the reference's own Eigen-free headers.

Runs only in the build container (needs /root/reference): builds oracle/_ref/ref_kat with
`make -C oracle ref` (g++ on the reference's DistVoxel/ColorVoxel/truncator/weighter/ColorImage
sources where they lie) and stores its output.  The fixture is data (inputs + reference outputs
as IEEE-754 bit patterns); no reference source text is stored.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if not os.path.isdir("/root/reference/OpenChisel/open_chisel/include"):
        print("reference tree absent; golden fixtures are committed, nothing to do")
        return 0
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out = subprocess.check_output([os.path.join(ROOT, "oracle/_ref/ref_kat")])
    data = json.loads(out)  # validates
    data["generator"] = "oracle/ref_kat/ref_kat_main.cpp built from /root/reference/OpenChisel/open_chisel headers"
    with open(os.path.join(ROOT, "tests/golden/ref_kat.json"), "w") as f:
        json.dump(data, f, separators=(",", ":"))
        f.write("\n")
    print("wrote tests/golden/ref_kat.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in data.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
