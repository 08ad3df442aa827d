#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the small
tracked files under profiles/: the rocprofv3 kernel-stats summary of our kernels, the per-launch HBM
traffic from the PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced
reads on gfx950, after checking that factor on a known-size float4 copy; WRITE_SIZE as is; both in
KiB), and profiles/traffic.json, which bench.py reads to fill roofline.traffic.

    python tools/summarize_profiles.py r01 fast/half 4096
"""
import collections
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASS_OF = {
    "fft_rows4_fwd_packed_kernel": "A rows: pad+FFT (real->complex)",
    "fft_cols_panel_fused_kernel": "B' cols: FFT*W*IFFT",
    "fft_cols_panel_fused16_kernel": "B' cols: FFT*W*IFFT",
    "fft_cols_panel_fused_lean_kernel": "B' cols: FFT*W*IFFT",
    "fft_rows4_inv_packed_kernel": "C' rows: IFFT+real+minmax",
    "normalize_kernel": "E normalize+crop",
}


def counter(path, name):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def main():
    tag, mode, size = sys.argv[1], sys.argv[2], int(sys.argv[3])
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    # 1. kernel stats (only rows of this library, full precision kept)
    stats = list(csv.DictReader(open(glob.glob(os.path.join(src, "kt", "*kernel_stats.csv"))[0])))
    with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
        w.writeheader()
        for r in stats:
            if "fdr::" in r["Name"]:
                w.writerow(r)
    # 2. calibration on the float4 copy of 128 MiB (first 12 dispatches of membench are that size)
    calf = counter(glob.glob(os.path.join(src, "cal_fetch", "*counter_collection.csv"))[0], "FETCH_SIZE")
    calw = counter(glob.glob(os.path.join(src, "cal_write", "*counter_collection.csv"))[0], "WRITE_SIZE")
    cf = [v for k, v in calf.items() if "copy_f4" in k][0][0]
    cw = [v for k, v in calw.items() if "copy_f4" in k][0][0]
    known_kib = 128 * 1024
    fetch_factor = known_kib / cf
    write_factor = known_kib / cw
    # 3. per-launch traffic of our passes
    fe = counter(glob.glob(os.path.join(src, "fetch", "*counter_collection.csv"))[0], "FETCH_SIZE")
    wr = counter(glob.glob(os.path.join(src, "write", "*counter_collection.csv"))[0], "WRITE_SIZE")
    traffic = {}
    rows = []
    for kname, vals in fe.items():
        for key, pname in PASS_OF.items():
            lg = size.bit_length() - 1
            if key in kname and ("<%d>" % lg in kname or "<%d," % lg in kname or key == "normalize_kernel"):
                rd = statistics.median(vals) * round(fetch_factor) * 1024.0
                wv = [v for k, v in wr.items() if k == kname]
                wt = statistics.median(wv[0]) * 1024.0 if wv else 0.0
                traffic[pname] = rd + wt
                rows.append((pname, kname[:90], len(vals), statistics.median(vals), statistics.median(wv[0]) if wv else 0, rd, wt))
    with open(os.path.join(dst, "%s_hbm_traffic.csv" % tag), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["calibration", "copy_f4 128 MiB", "FETCH_SIZE_KiB", cf, "factor", fetch_factor, "WRITE_SIZE_KiB", cw, "factor", write_factor])
        w.writerow(["pass", "kernel", "dispatches", "FETCH_SIZE_median_KiB", "WRITE_SIZE_median_KiB", "read_bytes(corrected)", "write_bytes"])
        for r in rows:
            w.writerow(r)
    tj_path = os.path.join(dst, "traffic.json")
    tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
    tj["%s/%d" % (mode, size)] = traffic  # mode like "fast/half" or "parity/full"
    json.dump(tj, open(tj_path, "w"), indent=1, sort_keys=True)
    bl = os.path.join(src, "bench_line.json")
    if os.path.exists(bl):
        open(os.path.join(dst, "%s_bench_line.json" % tag), "w").write(open(bl).read())
    print("fetch factor %.3f write factor %.3f" % (fetch_factor, write_factor))
    for r in rows:
        print(r[0], "read %.1f MB write %.1f MB" % (r[5] / 1e6, r[6] / 1e6))


if __name__ == "__main__":
    main()
