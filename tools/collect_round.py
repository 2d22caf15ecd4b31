#!/usr/bin/env python3
"""Turns one tools/profile_round.sh output directory into the files kept under profiles/ (all carry the kernel source id
they were measured on):  tools/collect_round.py <outdir> <prefix, e.g. profiles/r03>"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

src, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as ge

kid = ge.kernel_source_id()


def last_json(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no JSON line in " + path)


def text(name, out, header):
    body = open(os.path.join(src, name)).read()
    with open(out, "w") as o:
        o.write("# %s\n# kernel source id %s\n" % (header, kid))
        o.write(body)


def pmc_means(d, out):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "pmc*", "run_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if "trace_kernel" not in row["Kernel_Name"]:
                continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[row["Counter_Name"]].add(row["Dispatch_Id"])
    with open(out, "w") as o:
        o.write("counter,mean_per_launch\n")
        for k in sorted(tot):
            o.write("%s,%g\n" % (k, tot[k] / max(1, len(cnt[k]))))


# the benchmark itself
line = last_json(os.path.join(src, "bench_line.txt"))
assert line["config"]["kernel_source_id"] == kid, "the tree changed since the run"
subprocess.check_call([sys.executable, os.path.join(root, "tools", "summarize_profile.py"), os.path.join(src, "bench"), prefix,
                       "cover 1920x1080 1000spp", kid], stdout=subprocess.DEVNULL)
json.dump(line, open(prefix + "_bench_line.json", "w"), indent=1)
# nested shells and 100 000 spheres
shutil.copy(os.path.join(src, "nested", "stats", "run_kernel_stats.csv"), prefix + "_nested_kernel_stats.csv")
pmc_means(os.path.join(src, "nested"), prefix + "_nested_trace_pmc.csv")
json.dump(last_json(os.path.join(src, "nested_bench_line.txt")), open(prefix + "_nested_bench_line.json", "w"), indent=1)
shutil.copy(os.path.join(src, "k100", "stats", "run_kernel_stats.csv"), prefix + "_100k_kernel_stats.csv")
json.dump(last_json(os.path.join(src, "k100_bench_line.txt")), open(prefix + "_100k_bench_line.json", "w"), indent=1)
# wavefront kernels
shutil.copy(os.path.join(src, "wf_cover", "stats", "run_kernel_stats.csv"), prefix + "_wavefront_kernel_stats.csv")
shutil.copy(os.path.join(src, "wf_nested", "stats", "run_kernel_stats.csv"), prefix + "_wavefront_nested_kernel_stats.csv")
with open(prefix + "_wavefront.txt", "w") as o:
    o.write("# wavefront kernels (RT_OPT_KERNEL 1), tools/wf_run.py under rocprofv3 --kernel-trace --stats; kernel source id %s\n" % kid)
    for name, label in (("wavefront_cover.txt", "cover 1200x675 x 128 spp, depth 50"), ("wavefront_nested.txt", "400 nested shells 1200x675 x 16 spp, depth 100")):
        o.write("## %s\n" % label)
        o.write("".join(l for l in open(os.path.join(src, name)) if l.startswith("wavefront:")))
    o.write("## the persistent kernel on the nested shells at the same size (tools/exp_libs.py: checksum, ns per wave trip, Msamples*px/s)\n")
    o.write(open(os.path.join(src, "persistent_nested_small.txt")).read())
# tables
text("configs.txt", prefix + "_configs.txt", "tools/exp_configs.py: every configuration on one MI355X, automatic options")
with open(prefix + "_phase_profile.txt", "w") as o:
    o.write("# tools/phase_profile.py (stamped kernel, wave-cycles per trip of the main loop): cover at 1200x675x64, nested shells at 1920x1080x16 (depth 100)\n# kernel source id %s\n" % kid)
    o.write(open(os.path.join(src, "phase_cover.txt")).read())
    o.write(open(os.path.join(src, "phase_nested.txt")).read())
text("progressive.txt", prefix + "_progressive_tiles.txt", "tools/exp_progressive.py: the reference's tile flow, one blocking rt_render_tile per 64x64 tile")
text("prune_selftest.txt", prefix + "_prune_selftest.txt", "tools/exp_prune_selftest.py 4 2048 8")
text("rank_share.txt", prefix + "_rank_share.txt", "tools/exp_rank_share.py: one rank's share of the benchmark frame; last number of a label: 0 = lanes share planes before the counter runs out, 1 = only after")
text("narrow.txt", prefix + "_narrow.txt", "tools/exp_narrow.py: the staged matrix kind with its sphere filter at half width for the rays that can touch the box of the scene's small spheres (RT_OPT_NARROW 1) against full width (0)")
text("narrow_tiled.txt", prefix + "_narrow_tiled.txt", "tools/exp_narrow_tiled.py: tiles read in place by the waves whose rays can touch them (RT_OPT_NARROW 1, path 5) against all tiles streamed through LDS (0, path 4)")
text("box_selftest.txt", prefix + "_box_selftest.txt", "tools/exp_box_selftest.py 4 2048 8: rays the box tests leave out against the exact hit test, with the library's padding (2^0) and with less")
text("tile_box_margin.txt", prefix + "_tile_box_margin.txt", "tools/exp_tile_box_margin.py: rt_selftest_tile_boxes (tiles read in place), 8 seeds, at the shipped allowances (radius growth 64u D^2) and at a half, a quarter, an eighth of them (the bound on the roundings of the reference asks for 17u)")
print("collected under", prefix, "kernel source id", kid)
