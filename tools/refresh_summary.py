"""After tools/r04_refresh_profiles.sh (gpurun_out/refresh/): the eight bench lines next to the judged ones under profiles/, and the
kernel averages of two workloads -> profiles/r04_refresh_final_library.txt.  python tools/refresh_summary.py"""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", "refresh")
TAGS = ["canonicalize", "uniq", "mixed", "canonicalize_n1pct", "mixed_n1pct", "mixed_uniq", "uniq_hash_only", "mixed_uniq_n1pct"]
out = ["Round 4, the FINAL library (4-bit team, the tier kernel in two builds, stage A's walk over its non-empty segments only, the list",
       "entries' second flag bit) through tools/r04_refresh_profiles.sh on one MI355X box, next to the judged r04_* files -- taken",
       "earlier in the round on another box; those four changes touch only the stages behind the dominant kernels (canon_kernel<4> there,",
       "canon_kernel<4, false> / <4, true> here).  bench.py lines, single stream; frac = algorithmic bytes / time / 8 TB/s.", "",
       "%-22s %9s %8s %8s %15s   %s" % ("workload", "ms/step", "frac", "of copy", "PMC traffic B", "judged file: ms/step, frac")]
for tag in TAGS:
    d = json.load(open(os.path.join(R, "r04_bench_%s.json" % tag)))
    j = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_%s.json" % tag)))
    r = d["roofline"]
    out.append("%-22s %9.3f %8.4f %8.3f %15.0f   %.3f, %.4f" % (tag, d["ms_per_step"], r["frac"], r.get("frac_of_copy") or 0, r["traffic"] or 0,
                                                                 j["ms_per_step"], j["roofline"]["frac"]))
out.append("")
for tag in ("mixed_n1pct", "canonicalize_n1pct", "canonicalize"):
    out.append("kernel averages of `%s` (rocprofv3 --kernel-trace --stats, same run; kernels with >= 10 calls):" % tag)
    for r in csv.DictReader(open(os.path.join(R, "r04_kernel_stats_%s.csv" % tag))):
        n = r["Name"].replace("(anonymous namespace)::", "")
        if any(k in n for k in ("at::native", "rocprim", "synth", "rocclr")) or int(r["Calls"]) < 10:
            continue
        out.append("  %-100s calls %4s avg %9.1f us" % (n[:100], r["Calls"], float(r["AverageNs"]) / 1e3))
    out.append("")
open(os.path.join(ROOT, "profiles", "r04_refresh_final_library.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:16]))
