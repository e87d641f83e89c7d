#!/usr/bin/env python3
"""profiles/traffic.json from PMC summaries of tools/pmc_run.sh.
usage: make_traffic.py out.json  summary kernel-substring workload-key records  [summary kernel workload records ...]
One entry per workload, for its DOMINANT kernel (named in the entry).  FETCH_SIZE / WRITE_SIZE are in KiB per launch;
gfx950 reports half the bytes of wide streaming reads (/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so
FETCH_SIZE is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Both calibrations are for exactly
the access widths of these kernels' payload streams (16 B per lane); the table kernels of `uniq` (scattered 16-byte
accesses and atomics: uncalibrated widths) are not in the figure."""
import json, sys
out = sys.argv[1]
entries = []
args = sys.argv[2:]
for k in range(0, len(args), 4):
    path, kernel, workload, n = args[k], args[k + 1], args[k + 2], int(args[k + 3])
    vals = {}
    for line in open(path):
        if kernel not in line:
            continue
        parts = line.split()
        name = [p for p in parts if p.isupper() or "_" in p and p.upper() == p][-1]
        vals[name] = float(line.rsplit("avg=", 1)[1])
    fetch = vals["FETCH_SIZE"] * 1024 * 2
    write = vals["WRITE_SIZE"] * 1024
    entries.append({
        "workload": workload,
        "kernel": kernel,
        "scope": "the workload's dominant kernel, per launch",
        "source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, per launch)" % ("profiles/" + path.rsplit("/", 1)[-1]),
        "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write,
        "correction": "FETCH_SIZE x2 (gfx950 wide-read undercount), WRITE_SIZE exact",
        "insts_valu_per_record": vals["SQ_INSTS_VALU"] / n, "insts_salu_per_record": vals["SQ_INSTS_SALU"] / n,
        "insts_lds_per_record": vals["SQ_INSTS_LDS"] / n,
    })
json.dump({"entries": entries}, open(out, "w"), indent=1)
print(open(out).read())
