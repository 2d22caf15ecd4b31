"""Every configuration of the benchmark table on one GPU through rt_render with automatic options: time per frame,
Msamples*px/s, TFLOP/s counted at 23 flops per ray-object test, the sphere filter path and the segments per sample."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

rt = ge.load_package()
r = rt.MetalRenderer(0)
cases = [("three_spheres", {}, 400, 225, 100, 50), ("cover", {}, 1200, 675, 500, 50), ("cover", {}, 1920, 1080, 1000, 50),
         ("cover", {}, 3840, 2160, 1000, 50), ("nested_glass", {}, 1920, 1080, 64, 100),
         ("random_spheres", {}, 1920, 1080, 64, 50), ("cornell", {}, 800, 600, 256, 30)]
paths = {0: "vector, LDS-staged", 1: "vector, scalar cache", 2: "vector, streamed", 3: "matrix cores, staged", 4: "matrix cores, streamed tiles", 5: "matrix cores, tiles in place"}
for name, kw, w, h, spp, depth in cases:
    objs, n = rt.scene(name, **kw)
    r.setupTexture(w, h)
    r.setScene(objs, n)
    u = rt.config_uniforms(name, w, h, spp, depth, n)
    best, st = 1e30, None
    for rep in range(1 if w * h * spp > 4e9 or n > 10000 else 2):
        r.accumClear(); r.render(u, None, 0, 0, spp); r.resolve(spp); r.synchronize()
        s = r.lastStats()
        if s.kernel_ms < best:
            best, st = s.kernel_ms, s
    print("%-14s %6d objects %4dx%-4d %4d spp depth %3d: %9.1f ms  %8.1f Msamples*px/s  %6.1f TFLOP/s  %-28s %.2f seg/sample"
          % (name, n, w, h, spp, depth, best, w * h * spp / best * 1e-3, 23 * st.tests / best * 1e-9, paths[st.scene_path],
             st.segments / st.samples), flush=True)
r.close()
