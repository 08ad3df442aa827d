"""Per-kernel medians of every counter collected by tools/gpu_issue_pmc.sh: argv[1] = output directory, argv[2] = prefix of
the pass directories of one configuration.  SQ cycle counters are also given as a share of SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import sys

out, pre = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/%s*/**/*counter_collection.csv" % (out, pre), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fdr::" not in k:
            continue
        name = k.replace("void fdr::", "").split("(")[0][:60] + " grid=" + r["Grid_Size"] + " wg=" + r.get("Workgroup_Size", "?")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in sorted(acc.items()):
    n = max(len(v) for v in d.values())
    if n < 3:
        continue
    med = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    wc = med.get("SQ_WAVE_CYCLES")
    print(name, "(launches seen: %d)" % n)
    for c in sorted(med):
        share = ""
        if wc and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" and ("CYCLES" in c or "WAIT" in c or "ACTIVE" in c or "LEVEL" in c or "FULL" in c or "CONFLICT" in c or "STALL" in c or c == "SQ_IFETCH"):
            share = "   %6.2f %% of SQ_WAVE_CYCLES" % (100.0 * med[c] / wc)
        print("   %-40s %14.6g%s" % (c, med[c], share))
