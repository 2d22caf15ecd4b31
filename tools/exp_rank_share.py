"""One rank's share of the benchmark frame at N = 1, 2, 4, 8 ranks (1000, 500, 250, 125 samples per pixel of the cover scene
at 1920x1080) with the automatic work-item size and with fixed ones: Msamples*px/s of the trace launch."""
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
for spp in (1000, 500, 250, 125):
    u = rt.config_uniforms("cover", w, h, 1000, 50, n)
    row = []
    for chunk in (-1, 64, 32, 16, 8):
        r.set_option(rt.RT_OPT_CHUNK, chunk)
        best, st = 1e30, None
        for rep in range(2):
            r.accumClear(); r.render(u, None, 0, 0, spp); r.synchronize()
            s = r.lastStats()
            if s.kernel_ms < best:
                best, st = s.kernel_ms, s
        row.append("chunk %3d(%2d): %7.1f" % (chunk, st.chunk, w * h * spp / best * 1e-3))
    print("%4d spp per rank:  " % spp + "   ".join(row), flush=True)
r.close()
