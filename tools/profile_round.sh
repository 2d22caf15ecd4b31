#!/bin/bash
# Everything the tables under profiles/ are made from, in one call on the GPU box:
#   tools/profile_round.sh <outdir>          (then: tools/collect_round.py <outdir> profiles/r0N)
# Every step runs under its own time limit and the script stops at the first failure.
set -u
OUT=$1
mkdir -p "$OUT"
export TMPDIR=/tmp
step() { echo "== $1 ($(date +%T))"; }
run() { name=$1; lim=$2; shift 2; step "$name"; timeout -k 10 "$lim" "$@" > "$OUT/$name.txt" 2> "$OUT/$name.err" || { echo "step $name failed"; tail -5 "$OUT/$name.err"; exit 1; }; }

run bench_line 400 python3 bench.py
step profile_bench; tools/profile_bench.sh "$OUT/bench" > "$OUT/profile_bench.log" 2>&1 || { echo "profile_bench failed"; tail -5 "$OUT/profile_bench.log"; exit 1; }
step profile_nested; tools/profile_trace.sh "$OUT/nested" --scene nested_glass --width 1920 --height 1080 --spp 64 --depth 100 > "$OUT/profile_nested.log" 2>&1 || { echo "profile_nested failed"; tail -5 "$OUT/profile_nested.log"; exit 1; }
run nested_bench_line 200 python3 bench.py --scene nested_glass --spp 64 --depth 100 --no-cpu-baseline --valu-steps 0 --steps 3 --warmup 1
step stats_100k; timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT/k100/stats" -o run --output-format csv -- ./ray-tracing_amd/lib/rt_render --scene random_spheres --width 1920 --height 1080 --spp 64 --depth 50 > "$OUT/stats_100k.log" 2>&1 || { echo "stats_100k failed"; tail -5 "$OUT/stats_100k.log"; exit 1; }
run k100_bench_line 300 python3 bench.py --scene random_spheres --spp 64 --depth 50 --no-cpu-baseline --valu-steps 0 --steps 2 --warmup 1
step wavefront_cover; timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT/wf_cover/stats" -o run --output-format csv -- python3 tools/wf_run.py > "$OUT/wavefront_cover.txt" 2>&1 || { echo "wavefront_cover failed"; tail -5 "$OUT/wavefront_cover.txt"; exit 1; }
step wavefront_nested; SCENE=nested_glass DEPTH=100 SPP=16 timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT/wf_nested/stats" -o run --output-format csv -- python3 tools/wf_run.py > "$OUT/wavefront_nested.txt" 2>&1 || { echo "wavefront_nested failed"; tail -5 "$OUT/wavefront_nested.txt"; exit 1; }
run configs 300 python3 tools/exp_configs.py
SIZE=1200x675x64 run phase_cover 200 python3 tools/phase_profile.py cover
SIZE=1920x1080x16 run phase_nested 200 python3 tools/phase_profile.py nested_glass
run progressive 200 python3 tools/exp_progressive.py
run prune_selftest 400 python3 tools/exp_prune_selftest.py 4 2048 8
run rank_share 400 python3 tools/exp_rank_share.py -1,-1,-1,-1,0 -1,-1,-1,-1,1
run persistent_nested_small 200 env SCENE=nested_glass DEPTH=100 SIZE=1200x675 python3 tools/exp_libs.py 16 1 ray-tracing_amd/lib/librt_amd.so
run narrow 300 python3 tools/exp_narrow.py
run narrow_tiled 400 python3 tools/exp_narrow_tiled.py 2000 5000 20000 100000
run box_selftest 600 python3 tools/exp_box_selftest.py 4 2048 8
run tile_box_margin 600 python3 tools/exp_tile_box_margin.py
echo "== done ($(date +%T))"
