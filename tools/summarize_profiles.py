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
    "fft_rows4_fwd_pers_kernel": "A rows: pad+FFT (real->complex)",
    "fft_rows4_inv_pers_kernel": "C' rows: IFFT+real+minmax",
    "fft_cols_panel_fused_kernel": "B' cols: FFT*W*IFFT",
    "fft_cols_panel_fused16_kernel": "B' cols: FFT*W*IFFT",
    "fft_rows4_inv_packed_kernel": "C' rows: IFFT+real+minmax",
    "normalize_kernel": "E normalize+crop",
}


def counter(path, name, grids=None):
    """Counter values per kernel name; when `grids` (a dict) is given, only the dispatches of each kernel's MOST FREQUENT
    launch grid are kept and that grid (threads) is recorded in it."""
    raw = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            raw[r["Kernel_Name"]][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    d = {}
    for k, by_grid in raw.items():
        if grids is None:
            d[k] = [v for vs in by_grid.values() for v in vs]
        else:
            g = max(by_grid, key=lambda gg: len(by_grid[gg]))
            d[k] = by_grid[g]
            grids[k] = g
    return collections.defaultdict(list, d)


def workgroup_sizes(path):
    """kernel name -> most frequent Workgroup_Size of its dispatches in a counter_collection.csv"""
    raw = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(path)):
        if r.get("Workgroup_Size"):
            raw[r["Kernel_Name"]][int(r["Workgroup_Size"])] += 1
    return {k: c.most_common(1)[0][0] for k, c in raw.items()}


# threads one IMAGE's launch of a kernel has (one-group row kernels: M/4 groups x the workgroup size the trace reports --
# 256 threads up to 4096 points, 512 for the 8192-point inverse kernel; column tiles: N/8 tiles x T)
def images_of(kname, grid, size, default, wg=256):
    lg = size.bit_length() - 1
    if "fft_rows4_fwd_packed_kernel<%d" % lg in kname or "fft_rows4_inv_packed_kernel<%d" % lg in kname:
        return max(1, round(grid / ((size // 4) * (wg or 256))))
    if "fft_cols_panel_fused16_kernel<%d>" % lg in kname:
        return max(1, round(grid / ((size // 8) * (size // 16))))
    return default


def main():
    """usage: python tools/summarize_profiles.py <tag>   (reads gpurun_out/prof_<tag>/, writes profiles/<tag>_*)"""
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    # calibration on the float4 copy of 128 MiB (first dispatches of membench are that size)
    calf = counter(glob.glob(os.path.join(src, "cal_fetch", "**", "*counter_collection.csv"), recursive=True)[0], "FETCH_SIZE")
    calw = counter(glob.glob(os.path.join(src, "cal_write", "**", "*counter_collection.csv"), recursive=True)[0], "WRITE_SIZE")
    cf = [v for k, v in calf.items() if "copy_f4" in k][0][0]
    cw = [v for k, v in calw.items() if "copy_f4" in k][0][0]
    known_kib = 128 * 1024
    fetch_factor, write_factor = known_kib / cf, known_kib / cw
    print("calibration: FETCH_SIZE factor %.3f, WRITE_SIZE factor %.3f" % (fetch_factor, write_factor))
    fp_path = os.path.join(src, "csrc_fingerprint.txt")  # written on the GPU box by collect_profiles.sh: the sources the counters saw
    fingerprint = open(fp_path).read().strip() if os.path.exists(fp_path) else None
    print("csrc fingerprint of the collection:", fingerprint)
    tj_path = os.path.join(dst, "traffic.json")
    tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
    for size in (4096, 8192):
        lg = size.bit_length() - 1
        kt = glob.glob(os.path.join(src, "kt_%d" % size, "**", "*kernel_stats.csv"), recursive=True)
        if kt:  # kernel stats (only rows of this library, full precision kept)
            stats = list(csv.DictReader(open(kt[0])))
            with open(os.path.join(dst, "%s_kernel_stats_%d.csv" % (tag, size)), "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
                w.writeheader()
                for r in stats:
                    if "fdr::" in r["Name"]:
                        w.writerow(r)
        tr = glob.glob(os.path.join(src, "kt_%d" % size, "**", "*kernel_trace.csv"), recursive=True)
        if tr:  # the same trace split by launch grid: a kernel's one-image and grouped launches are different things
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(tr[0])):
                if "fdr::" in r["Kernel_Name"]:
                    acc[(r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            with open(os.path.join(dst, "%s_kernel_stats_by_grid_%d.csv" % (tag, size)), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Name", "Grid_Size_X", "Grid_Size_Y", "Workgroup_Size_X", "Calls", "AverageNs", "MedianNs", "MinNs", "MaxNs"])
                for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
                    w.writerow(list(k) + [len(v), "%.1f" % (sum(v) / len(v)), "%.1f" % statistics.median(v), min(v), max(v)])
        fe_f = glob.glob(os.path.join(src, "fetch_%d" % size, "**", "*counter_collection.csv"), recursive=True)
        wr_f = glob.glob(os.path.join(src, "write_%d" % size, "**", "*counter_collection.csv"), recursive=True)
        if not (fe_f and wr_f):
            continue
        fgrids, wgrids = {}, {}
        fe, wr = counter(fe_f[0], "FETCH_SIZE", fgrids), counter(wr_f[0], "WRITE_SIZE", wgrids)
        wgs = workgroup_sizes(fe_f[0])
        traffic, rows = {}, []
        default_images = 4 if size <= 4096 else 2  # images per launch of bench.py's default grouping (what the PMC runs used)
        for kname, vals in fe.items():
            for key, pname in PASS_OF.items():
                if key.startswith("fft_rows4_inv_"):  # last template argument: 0 raw plane (C'), 1 min/max only (C1), 2 normalised (C2)
                    if ", 1>(" in kname: pname = "C1 rows: IFFT+minmax"
                    elif ", 2>(" in kname: pname = "C2 rows: IFFT+normalize+crop"
                if len(vals) >= 2 and key in kname and ("<%d>" % lg in kname or "<%d," % lg in kname or key == "normalize_kernel"):  # (single dispatches: the PSF's own passes)
                    rd = statistics.median(vals) * round(fetch_factor) * 1024.0
                    wv = [v for k, v in wr.items() if k == kname]
                    wt = statistics.median(wv[0]) * 1024.0 if wv else 0.0
                    images = images_of(kname, fgrids.get(kname, 0), size, default_images, wgs.get(kname))
                    traffic[pname] = {"per_launch": rd + wt, "images": images, "kernel": kname.split("(")[0].replace("void ", ""),
                                      "grid_threads": fgrids.get(kname, 0), "workgroup_threads": wgs.get(kname), "csrc": fingerprint,
                                      "collected": tag}
                    rows.append((pname, kname[:90], len(vals), statistics.median(vals), statistics.median(wv[0]) if wv else 0, rd, wt, images))
        with open(os.path.join(dst, "%s_hbm_traffic_%d.csv" % (tag, size)), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["calibration", "copy_f4 128 MiB", "FETCH_SIZE_KiB", cf, "factor", fetch_factor, "WRITE_SIZE_KiB", cw, "factor", write_factor])
            w.writerow(["(per LAUNCH; images per launch in the last column: the inverse row passes C1 / C2 go in chunks of 2 images at 4096^2)"])
            w.writerow(["pass", "kernel", "dispatches", "FETCH_SIZE_median_KiB", "WRITE_SIZE_median_KiB", "read_bytes(corrected)", "write_bytes", "images_per_launch"])
            for r in rows:
                w.writerow(r)
        tj["fast/half/%d" % size] = traffic
        for r in rows:
            print(size, r[0], "read %.1f MB write %.1f MB" % (r[5] / 1e6, r[6] / 1e6))
    json.dump(tj, open(tj_path, "w"), indent=1, sort_keys=True)
    kts = glob.glob(os.path.join(src, "kt_single_1024", "**", "*kernel_stats.csv"), recursive=True)
    if kts:  # ONE 1024^2 image per step: device durations of the four launches
        stats = list(csv.DictReader(open(kts[0])))
        with open(os.path.join(dst, "%s_kernel_stats_single_1024.csv" % tag), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
            w.writeheader()
            for r in stats:
                if "fdr::" in r["Name"]:
                    w.writerow(r)
    for name in ("single_image_passbench.log", "passbench_4096.log", "passbench_8192.log", "seam_bench.log", "rmw_bench.log", "passB_phase_stamps.log",
                 "passbench_parity_mode.log"):
        pth = os.path.join(src, name)
        if os.path.exists(pth):
            txt = open(pth).read().replace(ROOT + "/", "")
            import re
            txt = re.sub(r"/tmp/code/[^ ]*/repo/", "", txt)
            open(os.path.join(dst, "%s_%s" % (tag, name)), "w").write(txt)
    for name in ("bench_line.json", "bench_line_streams1_4096.json", "bench_line_streams1_8192.json", "two_rank_weak.log", "two_rank_strong.log",
                 "two_rank_bcast_filter.log", "five_rank_config5.log", "single_image_512.log", "single_image_1024.log", "single_image_2048.log",
                 "config5_one_gpu.log", "config2_size.log", "config4_size.log", "bench_raw_plane.log", "status.txt"):
        pth = os.path.join(src, name)
        if os.path.exists(pth):
            txt = open(pth).read()
            if name.endswith(".log"):  # keep the JSON line and the last lines of the launcher's output
                txt = "\n".join([l for l in txt.splitlines() if l.startswith("{") or "rror" in l][-6:]) + "\n"
            open(os.path.join(dst, "%s_%s" % (tag, name)), "w").write(txt)


if __name__ == "__main__":
    main()
