"""Work-item size (RT_OPT_CHUNK) against frame size and sample count on the cover scene, planes automatic: the
measurements behind the automatic item size of rt_render."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

rt = ge.load_package()
r = rt.MetalRenderer(0)
objs, n = rt.scene("cover")
for w, h, spp in ((1920, 1080, 64), (1920, 1080, 16), (1200, 675, 64), (640, 360, 64), (640, 360, 256), (320, 180, 64)):
    r.setupTexture(w, h)
    r.setScene(objs, n)
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    row = []
    for chunk in [int(v) for v in os.environ.get('CHUNKS', '-1,4,8,12,16,24,32,48,64').split(',')]:
        if chunk > spp:
            continue
        r.set_option(rt.RT_OPT_CHUNK, chunk)
        best, st = 1e30, None
        for rep in range(3):
            r.accumClear(); r.render(u, None, 0, 0, spp); r.synchronize()
            s = r.lastStats()
            if s.kernel_ms < best:
                best, st = s.kernel_ms, s
        row.append("%d(%d x%d+%d): %6.1f" % (chunk, st.chunk, st.chunk_count, st.tail_chunk, w * h * spp / best * 1e-3))
    print("%dx%d x %d:  " % (w, h, spp) + "  ".join(row), flush=True)
r.close()
