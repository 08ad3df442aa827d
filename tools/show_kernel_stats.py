#!/usr/bin/env python3
"""Prints the rocprofv3 --stats kernel table found under a directory: tools/show_kernel_stats.py <dir>"""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not fs:
    print("no kernel_stats.csv under", sys.argv[1]); sys.exit(1)
for r in csv.DictReader(open(fs[0])):
    print("%-84s calls %5s avg %8.1f us" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3))
