"""The tiled matrix kind with its per-tile box test (RT_OPT_NARROW 1, the default) against without (0): same build, same
checksums.  exp_narrow_tiled.py [spheres ...]"""
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
sizes = [int(a) for a in sys.argv[1:]] or [5000, 100000]
for n_req in sizes:
    w, h, spp = (1200, 675, 8) if n_req < 50000 else (1920, 1080, 4)
    objs, n = rt.scene("random_spheres", n=n_req) if n_req >= 50000 else rt.scene("random_spheres", n=n_req, extent=float(np.sqrt(n_req / 4.0)))
    r.setupTexture(w, h)
    r.setScene(objs, n)
    u = rt.config_uniforms("random_spheres", w, h, spp, 50, n)
    for narrow in (0, 1):
        r.set_option(rt.RT_OPT_NARROW, narrow)
        best = 1e30
        for rep in range(2):
            r.accumClear(); r.render(u, None, 0, 0, spp); r.synchronize()
            best = min(best, r.lastStats().kernel_ms)
        st = r.lastStats()
        acc = np.empty((h, w, 4), np.float32)
        hip.hipMemcpy(acc.ctypes.data, r.accumPtr(), acc.nbytes, 2)
        print("random_spheres %6d spheres %4dx%-4d x %2d narrow %d: %9.2f ms  %8.2f Msamples*px/s  crc %08x  path %d  blocks/trip %.2f" %
              (n, w, h, spp, narrow, best, w * h * spp / best * 1e-3, zlib.crc32(acc.tobytes()), st.scene_path,
               st.matrix_blocks / max(1, st.wave_trips)), flush=True)
r.close()
