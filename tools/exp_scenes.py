"""Msamples*px/s of the configurations' scene classes at reduced sample counts, for A/B runs of library builds (the library
comes from RT_AMD_LIB): exp_scenes.py [scene ...]"""
import os
import sys
import zlib
import ctypes as C

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

rt = ge.load_package()
r = rt.MetalRenderer(0)
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
cases = [("cover", 1920, 1080, 128, 50), ("nested_glass", 1920, 1080, 64, 100), ("random_spheres", 1920, 1080, 8, 50),
         ("cornell", 800, 600, 256, 30), ("three_spheres", 400, 225, 100, 50)]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for scene, w, h, spp, depth in cases:
    objs, n = rt.scene(scene)
    r.setupTexture(w, h)
    r.setScene(objs, n)
    u = rt.config_uniforms(scene, w, h, spp, depth, n)
    best = 1e30
    for rep in range(3 if n < 10000 else 1):
        r.accumClear(); r.render(u, None, 0, 0, spp); r.synchronize()
        best = min(best, r.lastStats().kernel_ms)
    acc = np.empty((h, w, 4), np.float32)
    hip.hipMemcpy(acc.ctypes.data, r.accumPtr(), acc.nbytes, 2)
    print("%-15s %4dx%-4d x %3d: %9.2f ms  %8.1f Msamples*px/s  crc %08x" % (scene, w, h, spp, best, w * h * spp / best * 1e-3, zlib.crc32(acc.tobytes())), flush=True)
r.close()
