"""Launch time of the trace kernel against samples per pixel (cover 1920x1080, automatic schedule and fixed item sizes):
the intercept of the fit is what a launch costs whatever its length -- start-up and the drain of the last items."""
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
configs = [(-1, -1)]
if len(sys.argv) > 1:
    configs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for chunk, plane in configs:
    r.set_option(rt.RT_OPT_CHUNK, chunk)
    r.set_option(rt.RT_OPT_PLANE_SAMPLES, plane)
    pts = []
    for spp in (8, 16, 32, 64, 128, 256, 512):
        u = rt.config_uniforms("cover", w, h, 1000, 50, n)
        best, st = 1e30, None
        for rep in range(3):
            r.accumClear(); r.render(u, None, 0, 0, spp); r.synchronize()
            s = r.lastStats()
            if s.kernel_ms < best:
                best, st = s.kernel_ms, s
        pts.append((spp, best))
        print("chunk %d plane %d: %4d spp  %8.3f ms  (%d x%d)  %7.1f Msamples*px/s" % (chunk, plane, spp, best, st.chunk, st.chunk_count, w * h * spp / best * 1e-3), flush=True)
    (x0, y0), (x1, y1) = pts[-2], pts[-1]
    b = (y1 - y0) / (x1 - x0)
    print("  slope %.4f ms per sample of a pixel; intercepts:" % b, "  ".join("%d: %.2f" % (x, y - b * x) for x, y in pts))
r.close()
