import csv,sys
for v in sys.argv[1:]:
    row={}
    for r in csv.DictReader(open("gpurun_out/pv_%s_kernel_stats.csv"%v)):
        n=r["Name"].replace("(anonymous namespace)::","")
        if "at::native" in n or "rocprim" in n: continue
        key=n.split("(")[0][:44]
        row[key]=row.get(key,0)+float(r["TotalDurationNs"])/30/1e3
    print("%-8s "%v+" ".join("%s=%.0f"%(k.replace("void ","")[:26],val) for k,val in sorted(row.items(), key=lambda x:-x[1]) if val>8 and "synth" not in k))
