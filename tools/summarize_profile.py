#!/usr/bin/env python3
"""Turns one tools/profile_bench.sh output directory into the files kept under profiles/:
   <prefix>_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats table
   <prefix>_trace_pmc.csv           mean per trace-kernel launch of every collected counter
   <prefix>_trace_hbm_bytes.json    HBM bytes per launch (FETCH_SIZE and WRITE_SIZE as counted: see the note in the file)
usage: tools/summarize_profile.py <profile dir> <prefix, e.g. profiles/r01> "<workload label>" [kernel source id]
"""
import collections, csv, glob, json, os, shutil, sys

src, prefix, label = sys.argv[1], sys.argv[2], sys.argv[3]
shutil.copy(os.path.join(src, "stats", "run_kernel_stats.csv"), prefix + "_bench_kernel_stats.csv")
tot, cnt, kname = collections.defaultdict(float), collections.defaultdict(set), None
for f in glob.glob(os.path.join(src, "pmc*", "run_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if "trace_kernel" not in row["Kernel_Name"]:
            continue
        kname = row["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[row["Counter_Name"]].add(row["Dispatch_Id"])
mean = {k: tot[k] / max(1, len(cnt[k])) for k in tot}
with open(prefix + "_trace_pmc.csv", "w") as o:
    o.write("counter,mean_per_launch\n")
    for k in sorted(mean):
        o.write("%s,%g\n" % (k, mean[k]))
fetch = mean.get("FETCH_SIZE", 0.0) * 1024
write = mean.get("WRITE_SIZE", 0.0) * 1024
json.dump({"workload": label, "kernel": kname, "kernel_source_id": sys.argv[4] if len(sys.argv) > 4 else None,
           "bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
           "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate runs of `python3 bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline --valu-steps 0`, mean over the trace launches, both as counted (KB x 1024).  The factor 2 "
                   "that gfx950 needs for FETCH_SIZE on wide coalesced streams does not apply to this kernel's reads: nearly "
                   "all of them are the camera-ray records, 32 contiguous bytes per lane at a 2 KiB lane stride, whose byte "
                   "count is known (32 B x samples traced) and equals the raw counter within 2 %; WRITE_SIZE likewise equals "
                   "the known 32 B x samples of queue records plus the plane stores"},
          open(prefix + "_trace_hbm_bytes.json", "w"), indent=1)
for k in sorted(mean):
    print("%-32s %.5g" % (k, mean[k]))
