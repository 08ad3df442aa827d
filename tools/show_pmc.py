#!/usr/bin/env python3
"""prints per-kernel FETCH_SIZE (x2, gfx950) / WRITE_SIZE medians in MB from gpurun_out/<tag>/{fetch,write}_<S>/ (tools/gpu_pmc.sh)"""
import collections, csv, glob, sys
tag = sys.argv[1]
for S in sys.argv[2:]:
    for nm, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        f = glob.glob("gpurun_out/%s/%s_%s/**/*counter_collection.csv" % (tag, d, S), recursive=True)
        if not f:
            print("missing", S, d); continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == nm and ("fdr::fft" in r["Kernel_Name"] or "normalize" in r["Kernel_Name"]):
                acc[r["Kernel_Name"][10:60]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v.sort(); m = v[len(v) // 2]
            print(S, nm, k, len(v), "MB %.1f" % (m * 1024 * (2 if nm == "FETCH_SIZE" else 1) / 1e6))
