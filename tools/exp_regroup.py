"""Scatter by every lane for its own path against scatter regrouped by material over the workgroup (RT_OPT_REGROUP 1 / 2):
bit-identity against the oracle on a small frame, checksums and timings on the benchmark frame, phase stamps."""
import ctypes as C
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

rt = ge.load_package()
ora = ge.load_oracle()
ora.rto_dispatch.restype = None
ora.rto_dispatch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
r = rt.MetalRenderer(0)
scene = os.environ.get("SCENE", "cover")
depth = int(os.environ.get("DEPTH", "50"))
objs, n = rt.scene(scene)

w, h, spp = 240, 135, 8
u = rt.config_uniforms(scene, w, h, spp, depth, n)
r.setupTexture(w, h)
tex = np.zeros((h, w, 4), np.float32)
tex[..., 3] = 1.0
ora.rto_dispatch(tex.ctypes.data, w, h, C.byref(u), objs, 0, 1, None)
for mode in (1, 2):
    r.set_option(rt.RT_OPT_REGROUP, mode)
    img = r.renderImage(u, objs, n, chunk=0)
    same = np.array_equal(img.view(np.uint32), tex.view(np.uint32))
    print("regroup %d: %dx%dx%d against the oracle: %s" % (mode, w, h, spp, "bit-identical" if same else "DIFFERENT"), flush=True)
    if not same:
        sys.exit(1)

w, h = 1920, 1080
spp = int(os.environ.get("SPP", "256"))
u = rt.config_uniforms(scene, w, h, spp, depth, n)
r.setupTexture(w, h)
r.setScene(objs, n)
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
for mode in (1, 2, 1, 2):
    r.set_option(rt.RT_OPT_REGROUP, mode)
    best = 1e30
    for rep in range(2):
        r.accumClear(); r.render(u, None, 0, 0, spp); r.synchronize()
        st = r.lastStats()
        best = min(best, st.kernel_ms)
    acc = np.empty((h, w, 4), np.float32)
    hip.hipMemcpy(acc.ctypes.data, r.accumPtr(), acc.nbytes, 2)
    print("regroup %d: %dx%dx%d  %8.1f ms  %7.1f Msamples*px/s  crc %08x  trips %d" % (
        mode, w, h, spp, best, w * h * spp / best * 1e-3, zlib.crc32(acc.tobytes()), st.wave_trips), flush=True)

names = ["regen", "filter", "resolve", "quads+record", "shade"]
spp = 64
u = rt.config_uniforms(scene, w, h, spp, depth, n)
for mode in (1, 2):
    r.set_option(rt.RT_OPT_REGROUP, mode)
    r.set_option(rt.RT_OPT_KERNEL, 2)
    r.accumClear(); r.render(u, None, 0, 0, spp); r.synchronize()
    st = r.lastStats()
    pc = list(st.phase_cycles)
    trips = max(1, pc[5])
    print("regroup %d stamped: %.1f ms, wave trips %d" % (mode, st.kernel_ms, trips))
    for i, nm in enumerate(names):
        print("  %-13s %8.0f wave-cycles/trip" % (nm, pc[i] / trips))
    r.set_option(rt.RT_OPT_KERNEL, 0)
r.close()
