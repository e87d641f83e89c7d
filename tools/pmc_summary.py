#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, per counter, average over dispatches."""
import csv, sys, collections, glob
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"][:120], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print("%-122s %-24s n=%-3d avg=%.6g" % (k, c, len(v), sum(v) / len(v)))
