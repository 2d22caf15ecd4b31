"""Automatic work items against fixed sizes on the other scene classes (three spheres, Cornell box, nested shells,
100 000 spheres): the automatic size is fitted on the cover scene (tools/exp_chunk.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

rt = ge.load_package()
r = rt.MetalRenderer(0)
cases = [("three_spheres", {}, 400, 225, 100, 50, (-1, 16, 32, 48, 64)), ("cornell", {}, 800, 600, 256, 30, (-1, 16, 32, 64)),
         ("nested_glass", {}, 1920, 1080, 64, 100, (-1, 16, 32, 64)), ("random_spheres", {}, 1920, 1080, 64, 50, (-1, 2, 8))]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, kw, w, h, spp, depth, chunks in cases:
    objs, n = rt.scene(name, **kw)
    r.setupTexture(w, h)
    r.setScene(objs, n)
    u = rt.config_uniforms(name, w, h, spp, depth, n)
    row = []
    for chunk in chunks:
        r.set_option(rt.RT_OPT_CHUNK, chunk)
        best, st = 1e30, None
        for rep in range(1 if name == "random_spheres" else 3):
            r.accumClear(); r.render(u, None, 0, 0, spp); r.synchronize()
            s = r.lastStats()
            if s.kernel_ms < best:
                best, st = s.kernel_ms, s
        row.append("%d(%d x%d+%d): %7.1f" % (chunk, st.chunk, st.chunk_count, st.tail_chunk, w * h * spp / best * 1e-3))
        print("%s %dx%d x %d:  " % (name, w, h, spp) + row[-1], flush=True)
r.close()
