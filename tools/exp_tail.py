"""Shorter work items at the end of a launch (RT_OPT_TAIL_CHUNK / RT_OPT_TAIL_SAMPLES) on top of the hand-over of planes:
one rank's share of the benchmark frame at 8 and 4 ranks (125 and 250 spp of the cover scene at 1920x1080)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

rt = ge.load_package()
r = rt.MetalRenderer(0)
w, h = 1920, 1080
objs, n = rt.scene("cover")
r.setupTexture(w, h)
r.setScene(objs, n)
u = rt.config_uniforms("cover", w, h, 1000, 50, n)
for spp in (125, 250):
    for chunk, small, tail, plane in ((-1, -1, -1, -1), (64, 0, 0, 4), (64, 32, 32, 4), (64, 32, 64, 4), (64, 16, 16, 4), (64, 16, 32, 4), (64, 8, 16, 4), (64, 32, 32, 2)):
        r.set_option(rt.RT_OPT_CHUNK, chunk)
        r.set_option(rt.RT_OPT_TAIL_CHUNK, small)
        r.set_option(rt.RT_OPT_TAIL_SAMPLES, tail)
        r.set_option(rt.RT_OPT_PLANE_SAMPLES, plane)
        best, st = 1e30, None
        for rep in range(3):
            r.accumClear(); r.render(u, None, 0, 0, spp); r.synchronize()
            s = r.lastStats()
            if s.kernel_ms < best:
                best, st = s.kernel_ms, s
        print("%4d spp  items %3d, last %3d samples in items of %2d, planes of %d  (%d x%d+%d): %7.1f Msamples*px/s  %.1f ms" % (
            spp, chunk, tail, small, plane, st.chunk, st.chunk_count, st.tail_chunk, w * h * spp / best * 1e-3, best), flush=True)
r.close()
