"""Random scenes x every sphere-fetch kind, filter and kernel, with and without the box tests that leave rays out of the
filter (RT_OPT_NARROW): all images must be the same bits as the scalar-cache vector path.   fuzz_kinds.py [first_seed end_seed]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
rt = ge.load_package()
r = rt.MetalRenderer(0)
seeds = range(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 24)
f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
bad = 0
for seed in seeds:
    rng = np.random.default_rng(seed)
    nsph = int(rng.choice([1, 3, 40, 65, 300, 900, 1100, 2000, 3500]))
    extent = float(10.0 ** rng.uniform(-1, 2.5))
    offset = float(rng.choice([0.0, 0.0, 100.0, -3000.0]))
    objs = []
    for i in range(nsph):
        c = offset + extent * (rng.random(3) - 0.5) * np.array([1.0, rng.choice([0.02, 1.0]), 1.0])
        rad = extent * 10.0 ** rng.uniform(-3.0, -0.8)
        k = rng.random()
        m = rt.SwiftMaterial.lambertian(*rng.random(3)) if k < 0.5 else (rt.SwiftMaterial.metal(*(0.5 + 0.5 * rng.random(3)), 0.5 * rng.random()) if k < 0.75 else rt.SwiftMaterial.dielectric(1.5))
        objs.append(rt.makeSphere(tuple(float(v) for v in c), float(rad), m))
    if rng.random() < 0.5:
        objs.insert(int(rng.integers(0, len(objs) + 1)), rt.makeSphere((offset, offset - 100.0 * extent, offset), 100.0 * extent - 0.4 * extent, rt.SwiftMaterial.lambertian(0.5, 0.5, 0.5)))
    for _ in range(int(rng.choice([0, 0, 2, 5, 12]))):   # a few spheres much larger than the rest, anywhere in the order
        c = offset + extent * (rng.random(3) - 0.5)
        objs.insert(int(rng.integers(0, len(objs) + 1)), rt.makeSphere(tuple(float(v) for v in c), float(extent * rng.uniform(0.05, 0.3)), rt.SwiftMaterial.dielectric(1.5)))
    if rng.random() < 0.3 and len(objs) > 2:
        objs.append(rt.Object.from_buffer_copy(objs[int(rng.integers(0, len(objs)))]))   # a duplicate: equal distances
    n = len(objs)
    arr = rt.object_array(objs)
    w, h, spp = 64, 40, 3
    u = rt.config_uniforms("cover", w, h, spp, 12, n)
    centre = np.full(3, offset)
    rt.host_lib().rt_camera_setup(C.byref(u), f3(centre + extent * np.array([1.1, 0.5, 0.8])), f3(centre), f3([0, 1, 0]), 45.0, 0.2, float(1.4 * extent), w, h)
    r.setupTexture(w, h)
    imgs = {}
    for fetch, filt, kern, narrow in ((1, 2, 0, 1), (0, 0, 0, 1), (0, 0, 0, 0), (0, 3, 0, 1), (0, 3, 0, 0), (0, 2, 0, 1), (0, 1, 0, 1), (2, 0, 0, 1), (3, 0, 0, 1),
                                      (3, 0, 0, 0), (3, 0, 1, 1), (0, 0, 1, 1), (0, 3, 1, 1)):
        r.set_option(rt.RT_OPT_SCENE_FETCH, fetch); r.set_option(rt.RT_OPT_FILTER, filt); r.set_option(rt.RT_OPT_KERNEL, kern)
        r.set_option(rt.RT_OPT_NARROW, narrow)
        imgs[(fetch, filt, kern, narrow)] = (r.renderImage(u, arr, n, chunk=0).copy(), r.lastStats().scene_path)
    ref = imgs[(1, 2, 0, 1)][0].view(np.uint32)
    line = "seed %2d n=%4d extent %8.3g offset %6g:" % (seed, n, extent, offset)
    for key, (img, path) in imgs.items():
        same = np.array_equal(img.view(np.uint32), ref)
        bad += not same
        line += " %s->%d%s" % ("".join(map(str, key)), path, "" if same else "!!DIFF")
    print(line, flush=True)
r.set_option(rt.RT_OPT_SCENE_FETCH, 0); r.set_option(rt.RT_OPT_FILTER, 0); r.set_option(rt.RT_OPT_KERNEL, 0); r.set_option(rt.RT_OPT_NARROW, 1)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
