#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summary of tools/pmc_run.sh (arguments: summary file, kernel-name substring).
FETCH_SIZE / WRITE_SIZE are in KiB per launch; gfx950 reports half the bytes of wide streaming reads
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so FETCH_SIZE is doubled; WRITE_SIZE is exact."""
import json, sys
path, kernel = sys.argv[1], sys.argv[2]
vals = {}
for line in open(path):
    if kernel not in line:
        continue
    parts = line.split()
    name = [p for p in parts if p.isupper() or "_" in p and p.upper() == p][-1]
    vals[name] = float(line.rsplit("avg=", 1)[1])
n = 10_000_000
fetch = vals["FETCH_SIZE"] * 1024 * 2
write = vals["WRITE_SIZE"] * 1024
json.dump({
    "workload": "canonicalize 10000000 x 1000",
    "kernel": kernel,
    "source": "profiles/r02_pmc_counters.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, per launch)",
    "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write,
    "correction": "FETCH_SIZE x2 (gfx950 wide-read undercount), WRITE_SIZE exact",
    "insts_valu_per_record": vals["SQ_INSTS_VALU"] / n, "insts_salu_per_record": vals["SQ_INSTS_SALU"] / n,
    "insts_lds_per_record": vals["SQ_INSTS_LDS"] / n,
}, open(sys.argv[3], "w"), indent=1)
print(open(sys.argv[3]).read())
