#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S dump: tools/isa_stats.py /tmp/chisel.s <mangled-prefix>"""
import sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(s) if l.startswith(sys.argv[2])][0]
end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
lines = [l.strip() for l in s[start + 1:end] if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
ins = [l for l in lines if not l.split()[0].endswith(':')]
print("instructions:", len(ins), dict(Counter(l.split()[0].split('_')[0] for l in ins)))
keys = ["v_div_scale", "v_rcp", "ds_read", "ds_write", "global_load", "global_store", "global_atomic", "flat_", "scratch_", "s_load", "s_barrier", "s_waitcnt", "s_cbranch", "v_readlane", "v_writelane"]
print({k: sum(k in l for l in ins) for k in keys})
for l in s[end:end + 80]:
    if any(k in l for k in (".vgpr_count", ".sgpr_count", "scratch", ".lds_size")):
        print(l.strip())
