"""Regenerates tests/golden/oracle_vectors.npz from the CPU oracle and the C host driver.

These vectors are produced by THIS repository's oracle (oracle/rt_oracle.c), not by the reference: the
reference's sources need <metal_stdlib>, which this image lacks, and the reference holds no vectors of
its own.  They pin the oracle and the host driver against accidental change (compiler flags, libm,
refactoring) and let the GPU tests run against fixed inputs.  The reference-derived anchors are in
survey_kat.json.

Run from the repository root:  python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as ge  # noqa: E402
from conftest import Oracle  # noqa: E402

rt = ge.load_package()
ora = Oracle()
L = ora.lib
out = {}

# pcg_hash
xs = np.concatenate([np.arange(1024, dtype=np.uint32), np.array([0xdeadbeef, 0xffffffff, 0x80000000], np.uint32)])
out["pcg_in"] = xs
out["pcg_out"] = np.array([L.rto_pcg_hash(int(x)) for x in xs], np.uint32)

# random_double streams
seeds = np.array([[0.25, 0.75], [0.0, 0.0], [float.fromhex("0x1.fffffep-1"), float.fromhex("0x1.fffffep-1")], [0.5, 0.125]], np.float32)
streams, finals = [], []
for s in seeds:
    st = (C.c_float * 2)(*s)
    streams.append([L.rto_random_double(st) for _ in range(256)])
    finals.append([st[0], st[1]])
out["rd_seeds"], out["rd_streams"], out["rd_final"] = seeds, np.array(streams, np.float32), np.array(finals, np.float32)

# random vectors
for kind, name in enumerate(["square", "disk", "unit"]):
    st = (C.c_float * 2)(0.3, 0.9)
    vecs, states = [], []
    for _ in range(64):
        v = (C.c_float * 3)()
        L.rto_random_vec(kind, st, v)
        vecs.append(list(v))
        states.append([st[0], st[1]])
    out["rv_%s" % name], out["rv_%s_state" % name] = np.array(vecs, np.float32), np.array(states, np.float32)

# per-sample seeds (uint -> float rounding above 2^24 at large resolutions)
combos = [(x, y, p, r) for (x, y) in [(0, 0), (399, 224), (1919, 1079), (3839, 2159)] for p in (0, 1, 999) for r in (0, 1, 999)]
ss = []
for x, y, p, r in combos:
    o = (C.c_float * 2)()
    L.rto_sample_seed(x, y, p, r, o)
    ss.append([o[0], o[1]])
out["seed_combos"], out["seed_values"] = np.array(combos, np.int32), np.array(ss, np.float32)


def uniforms_bytes(u):
    return np.frombuffer(bytes(u), np.uint8).copy()


def objects_bytes(objs, n):
    return np.frombuffer(bytes(objs), np.uint8)[: 160 * n].copy()


# host driver outputs
for name, w, h in [("three_spheres", 400, 225), ("cover", 1920, 1080), ("cornell", 1000, 800), ("nested_glass", 640, 360)]:
    objs, n = rt.scene(name)
    u = rt.config_uniforms(name, w, h, 16, 50, n)
    out["uniforms_%s" % name] = uniforms_bytes(u)
    if name != "cover":
        out["objects_%s" % name] = objects_bytes(objs, n) if n < 64 else objects_bytes(objs, 8)
objs, n = rt.scene("cover")
ob = objects_bytes(objs, n)
out["cover_count"] = np.array([n], np.int32)
out["cover_first8"] = ob[: 160 * 8]
out["cover_fnv"] = np.array([int(np.bitwise_xor.reduce(np.frombuffer(ob.tobytes(), np.uint32) * np.arange(1, len(ob) // 4 + 1, dtype=np.uint32)))], np.uint32)

# hitObject / scatter vectors on the three-sphere scene
objs, n = rt.scene("three_spheres")
u = rt.config_uniforms("three_spheres", 400, 225, 4, 50, n)
rays, hits, scat = [], [], []
rng = np.random.RandomState(7)
for k in range(96):
    px, py = int(rng.randint(0, 400)), int(rng.randint(0, 225))
    d, o, sd = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 2)()
    L.rto_get_ray(C.byref(u), px, py, k % 4, d, o, sd)
    h8 = (C.c_float * 8)()
    idx = L.rto_world_hit(objs, n, d, o, h8)
    rays.append([px, py, k % 4] + list(d) + list(o) + [sd[0], sd[1]])
    hits.append([idx] + (list(h8) if idx >= 0 else [0.0] * 8))
    if idx >= 0:
        so = (C.c_float * 10)()
        ok = L.rto_scatter(C.byref(objs[idx].mat), d, o, h8, sd, so)
        scat.append([ok] + list(so) + [sd[0], sd[1]])
    else:
        scat.append([0] + [0.0] * 12)
out["hv_rays"], out["hv_hits"], out["hv_scatter"] = np.array(rays, np.float32), np.array(hits, np.float32), np.array(scat, np.float32)

# hand-made hitSphere cases: front, inside, tangent-ish, behind, zero-length direction, non-unit direction, beyond tmax
sph = rt.makeSphere((0, 0, -3), 1.0, rt.SwiftMaterial.lambertian(0.5, 0.5, 0.5))
cases = [((0, 0, -1), (0, 0, 0), 0.001, 3.4e38), ((0, 0, -1), (0, 0, -3), 0.001, 3.4e38), ((1.0, 0, -3), (0, 0, 0), 0.001, 3.4e38),
         ((0, 0, 1), (0, 0, 0), 0.001, 3.4e38), ((0, 0, -1e-4), (0, 0, 0), 0.001, 3.4e38), ((0, 0, -7.5), (0.1, 0.2, 0), 0.001, 3.4e38),
         ((0, 0, -1), (0, 0, 0), 0.001, 1.5), ((0.33333334, 0, -1), (0, 0, 0), 0.001, 3.4e38)]
hs = []
for d, o, tmin, tmax in cases:
    h8 = (C.c_float * 8)()
    ok = L.rto_hit_object(C.byref(sph), (C.c_float * 3)(*d), (C.c_float * 3)(*o), tmin, tmax, h8)
    hs.append(list(d) + list(o) + [tmin, tmax, ok] + (list(h8) if ok else [0.0] * 8))
out["hs_cases"] = np.array(hs, np.float32)

# first-hit buffers and small images
for name, w, h, spp, depth in [("three_spheres", 100, 57, 8, 50), ("cover", 96, 54, 4, 50), ("cornell", 50, 40, 4, 10)]:
    objs, n = rt.scene(name)
    u = rt.config_uniforms(name, w, h, spp, depth, n)
    t4, ids = ora.first_hit(u, objs, w, h)
    out["fh_%s_t" % name] = t4
    out["fh_%s_id" % name] = ids.astype(np.int16)
    out["img_%s" % name] = ora.image(u, objs, w, h)
    out["img_%s_cfg" % name] = np.array([w, h, spp, depth], np.int32)

np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
print("wrote", os.path.join(HERE, "oracle_vectors.npz"), os.path.getsize(os.path.join(HERE, "oracle_vectors.npz")), "bytes")
