"""Census of the ray segments of a scene (CPU, through the oracle's exported pieces): which spatial blocks of 64 spheres
could a segment's ray touch at all?  Answers whether culling blocks per ray before the sphere filter could pay.

  python tools/exp_ray_census.py [scene] [width height spp]

For every path of a coarse pixel grid the segments are walked with rto_world_hit / rto_scatter; for every segment the ray
(t > 0.001, unbounded) is tested against the bounding box of every block of 64 spheres (spheres sorted along a Morton
curve of their centres; spheres much larger than the median radius form blocks of their own, which every ray visits).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from conftest import Oracle  # noqa: E402

rt = ge.load_package()
ora = Oracle().lib
name = sys.argv[1] if len(sys.argv) > 1 else "cover"
W, H, SPP = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (1920, 1080, 2)
STEP = int(os.environ.get("STEP", "24"))
DEPTH = int(os.environ.get("DEPTH", "50"))
objs, n = rt.scene(name)
u = rt.config_uniforms(name, W, H, SPP, DEPTH, n)

sph = np.array([[o.s.center.x, o.s.center.y, o.s.center.z, o.s.radius] for o in objs[:n] if o.type == 0], np.float64)
med = np.median(sph[:, 3])
big = sph[:, 3] > 4 * med
small = sph[~big]


def morton(p, lo, hi):
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-9) * 1023).astype(np.int64), 0, 1023)
    code = np.zeros(len(p), np.int64)
    for b in range(10):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return code


order = np.argsort(morton(small[:, :3], small[:, :3].min(0), small[:, :3].max(0)), kind="stable")
small = small[order]
BLOCK = int(os.environ.get("BLOCK", "64"))
blocks = [small[i:i + BLOCK] for i in range(0, len(small), BLOCK)]
box_lo = np.array([(b[:, :3] - b[:, 3:4]).min(0) for b in blocks])
box_hi = np.array([(b[:, :3] + b[:, 3:4]).max(0) for b in blocks])
print("%s: %d spheres, %d large (always visited), %d blocks of <= %d small ones" % (name, len(sph), big.sum(), len(blocks), BLOCK))


def boxes_touched(o, d):
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0 = (box_lo - o) * inv
        t1 = (box_hi - o) * inv
    tn = np.where(np.isnan(t0), -np.inf, np.minimum(t0, t1)).max(1)
    tf = np.where(np.isnan(t1), np.inf, np.maximum(t0, t1)).min(1)
    return (tn <= tf) & (tf > 0.001)


f3 = C.c_float * 3
rays = []  # (touched mask, depth)
out8, out10 = (C.c_float * 8)(), (C.c_float * 10)()
for py in range(STEP // 2, H, STEP):
    for px in range(STEP // 2, W, STEP):
        for k in range(SPP):
            d, o, seed = f3(), f3(), (C.c_float * 2)()
            ora.rto_get_ray(C.byref(u), px, py, k, d, o, seed)
            for depth in range(DEPTH):
                rays.append((boxes_touched(np.array(o[:], np.float64), np.array(d[:], np.float64)), depth))
                idx = ora.rto_world_hit(objs, n, d, o, out8)
                if idx < 0:
                    break
                if not ora.rto_scatter(C.byref(objs[idx].mat), d, o, out8, seed, out10):
                    break
                d, o = f3(*out10[4:7]), f3(*out10[7:10])

masks = np.array([m for m, _ in rays])
depths = np.array([dp for _, dp in rays])
print("segments: %d; blocks touched per segment: mean %.2f of %d; segments touching none: %.1f %%" %
      (len(rays), masks.sum(1).mean(), len(blocks), 100.0 * (masks.sum(1) == 0).mean()))
for lo, hi in ((0, 1), (1, 2), (2, 5), (5, 50)):
    sel = (depths >= lo) & (depths < hi)
    if sel.any():
        print("  depth %2d..%2d: %6d segments, %.2f blocks, none %.1f %%" % (lo, hi - 1, sel.sum(), masks[sel].sum(1).mean(), 100.0 * (masks[sel].sum(1) == 0).mean()))
# waves of 64 segments drawn at random (the lanes of a wave are at unrelated depths of unrelated paths)
rng = np.random.default_rng(1)
T = 4000
need_any, per_block_lanes, half_tiles, full_tiles, skipped = [], [], 0, 0, 0
for _ in range(T):
    sel = masks[rng.integers(0, len(masks), 64)]
    need_any.append((sel.sum(1) > 0).sum())
    lanes = sel.sum(0)
    per_block_lanes.append(lanes)
    skipped += (lanes == 0).sum()
    half_tiles += ((lanes > 0) & (lanes <= 32)).sum()
    full_tiles += (lanes > 32).sum()
per_block_lanes = np.array(per_block_lanes)
print("random waves: lanes needing any block: mean %.1f, <= 32 in %.1f %% of the waves" % (np.mean(need_any), 100.0 * (np.array(need_any) <= 32).mean()))
print("  per block and wave: lanes needing it mean %.1f; no lane %.1f %%, 1..32 lanes %.1f %%, more %.1f %%" %
      (per_block_lanes.mean(), 100.0 * skipped / per_block_lanes.size, 100.0 * half_tiles / per_block_lanes.size, 100.0 * full_tiles / per_block_lanes.size))
