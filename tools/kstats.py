#!/usr/bin/env python3
"""Prints a rocprofv3 *_kernel_stats.csv compactly: name (shortened), calls, average us, share."""
import csv, signal, sys
signal.signal(signal.SIGPIPE, signal.SIG_DFL)        # `| head` is a normal way to read this
for path in sys.argv[1:]:
    print("==", path)
    for r in list(csv.DictReader(open(path)))[:14]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("ck::", "")
        n = n.split("(")[0] if not n.startswith("void at::") else n[:60]
        print("  %-78s calls %5s avg %10.1f us  %5s%%" % (n[:78], r["Calls"], float(r["AverageNs"]) / 1e3, r.get("Percentage", "?")[:5]))
