"""One rank's share of the benchmark frame at N = 1, 2, 4, 8 ranks (1000, 500, 250, 125 samples per pixel of the cover scene
at 1920x1080): Msamples*px/s of the trace launch with the automatic schedule and with fixed work-item sizes (chunk) and
samples per partial-sum plane (plane; 0 = one plane per item, i.e. no hand-over of planes at the end of the launch)."""
import os
import sys
import zlib
import ctypes as C

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

rt = ge.load_package()
r = rt.MetalRenderer(0)
w, h = 1920, 1080
objs, n = rt.scene("cover")
r.setupTexture(w, h)
r.setScene(objs, n)
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
configs = [(-1, -1), (-1, 0), (64, 8), (64, 0), (32, 8), (64, 4), (64, 16)]
if len(sys.argv) > 1:
    configs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for spp in [int(v) for v in os.environ.get('SPP', '1000,500,250,125').split(',')]:
    u = rt.config_uniforms("cover", w, h, 1000, 50, n)
    row = []
    for cfg in configs:
        chunk, plane = cfg[0], cfg[1]
        r.set_option(rt.RT_OPT_CHUNK, chunk)
        r.set_option(rt.RT_OPT_PLANE_SAMPLES, plane)
        # optional third and fourth numbers: shorter items at the end of the launch (item size, samples per pixel in them)
        r.set_option(rt.RT_OPT_TAIL_CHUNK, cfg[2] if len(cfg) > 2 else -1)
        r.set_option(rt.RT_OPT_TAIL_SAMPLES, cfg[3] if len(cfg) > 3 else -1)
        r.set_option(rt.RT_OPT_SHARE, cfg[4] if len(cfg) > 4 else 1)  # fifth number: 0 = planes change lanes before the counter is exhausted too
        best, st = 1e30, None
        for rep in range(2):
            r.accumClear(); r.render(u, None, 0, 0, spp); r.synchronize()
            s = r.lastStats()
            if s.kernel_ms < best:
                best, st = s.kernel_ms, s
        row.append("%s(%d x%d+%d): %6.1f" % (",".join(str(v) for v in cfg), st.chunk, st.chunk_count, st.tail_chunk, w * h * spp / best * 1e-3))
    print("%4d spp per rank:  " % spp + "   ".join(row), flush=True)
r.close()
