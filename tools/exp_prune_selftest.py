"""rt_selftest_prune at length on the dense test scenes, the nested shells and the cover scene: pairs checked, how many the
single-sphere test and the bounding balls strike, and how many of those the reference would have taken (must be 0).
Usage: exp_prune_selftest.py [seeds] [blocks] [rays per thread]"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import __graft_entry__ as ge
from test_gpu_parity import _dense_scene

rt = ge.load_package()
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
rays = int(sys.argv[3]) if len(sys.argv) > 3 else 8
r = rt.MetalRenderer(0)
bad = 0
for kind in ("nested_glass", "cover", "drifting_shells", "clutter", "far_shells", "shells_and_strays", "rising_shells"):
    if kind in ("nested_glass", "cover"):
        arr, n = rt.scene(kind)
    else:
        objs, _, _ = _dense_scene(rt, kind)
        n = len(objs)
        arr = rt.object_array(objs)
    r.setScene(arr, n)
    tot = {}
    for seed in range(1, seeds + 1):
        for k, v in r.selftestPrune(seed=seed, blocks=blocks, rays_per_thread=rays).items():
            tot[k] = tot.get(k, 0) + v
    print("%-18s pairs %.3g  with a root %.1f %%  struck by the single test %.1f %%  by the balls %.1f %%  wrong %d / %d"
          % (kind, tot["pairs"], 100.0 * tot["reference"] / tot["pairs"], 100.0 * tot["single"] / tot["pairs"],
             100.0 * tot["balls"] / tot["pairs"], tot["single_wrong"], tot["balls_wrong"]), flush=True)
    bad += tot["single_wrong"] + tot["balls_wrong"]
sys.exit(1 if bad else 0)
