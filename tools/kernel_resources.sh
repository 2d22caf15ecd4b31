#!/bin/bash
# Per-kernel register, spill and occupancy table of one .hip file (compiler remarks, no GPU needed).
#   tools/kernel_resources.sh ray-tracing_amd/csrc/rt_trace.hip
# (with the extra flags ray-tracing_amd/csrc/FLAGS lists for the file, as the Makefile builds it)
EXTRA=$(sed -n "s/^$(basename "$1") //p" "$(dirname "$0")/../ray-tracing_amd/csrc/FLAGS")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC $EXTRA -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except Exception: pass
        cur = {"name": name[:60]}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print("%-60s %5s %5s %5s %6s %6s %4s" % ("kernel", "SGPR", "VGPR", "AGPR", "sSpill", "vSpill", "occ"))
for r in rows:
    print("%-60s %5s %5s %5s %6s %6s %4s" % (r["name"], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("Occupancy [waves/SIMD]")))
'
