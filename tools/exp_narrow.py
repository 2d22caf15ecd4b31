"""The staged matrix kind with its filter at half width (RT_OPT_NARROW 1, the default) against full width (0): same build,
same checksums.  exp_narrow.py [scene ...]"""
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
cases = [("cover", 1920, 1080, 128, 50, {}), ("cover", 1200, 675, 64, 50, {}), ("random_spheres", 1200, 675, 32, 50, {"n": 600, "extent": 12.0}),
         ("random_spheres", 1200, 675, 32, 50, {"n": 1000, "extent": 15.0})]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for scene, w, h, spp, depth, kw in cases:
    objs, n = rt.scene(scene, **kw)
    r.setupTexture(w, h)
    r.setScene(objs, n)
    u = rt.config_uniforms(scene, w, h, spp, depth, n)
    for narrow in (0, 1, 0, 1):
        r.set_option(rt.RT_OPT_NARROW, narrow)
        best = 1e30
        for rep in range(2):
            r.accumClear(); r.render(u, None, 0, 0, spp); r.synchronize()
            best = min(best, r.lastStats().kernel_ms)
        st = r.lastStats()
        acc = np.empty((h, w, 4), np.float32)
        hip.hipMemcpy(acc.ctypes.data, r.accumPtr(), acc.nbytes, 2)
        print("%-15s %5d spheres %4dx%-4d x %3d narrow %d: %9.2f ms  %8.1f Msamples*px/s  crc %08x  path %d  blocks/trip %.2f" %
              (scene, n, w, h, spp, narrow, best, w * h * spp / best * 1e-3, zlib.crc32(acc.tobytes()), st.scene_path,
               st.matrix_blocks / max(1, st.wave_trips)), flush=True)
r.close()
