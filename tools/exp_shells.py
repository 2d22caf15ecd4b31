"""Nested shells beyond the staged matrix kind (1000, 1500, 3000 shells at 1080p x 8 spp, depth 100): which kind the library picks
by itself, and the kinds one can ask for."""
import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
rt = ge.load_package()
r = rt.MetalRenderer(0)
w, h, spp = 1920, 1080, 8
r.setupTexture(w, h)
for shells in (1000, 1500, 3000):
    objs, n = rt.scene("nested_glass", shells=shells)
    r.setScene(objs, n)
    u = rt.config_uniforms("nested_glass", w, h, spp, 100, n)
    for label, opts in (("auto", {}), ("vector filter", {rt.RT_OPT_FILTER: 2}), ("scalar cache", {rt.RT_OPT_SCENE_FETCH: 1})):
        for k, v in opts.items():
            r.set_option(k, v)
        try:
            best = 1e30
            for rep in range(2):
                r.accumClear(); r.render(u, None, 0, 0, spp); r.resolve(spp); r.synchronize()
                best = min(best, r.lastStats().kernel_ms)
            print("%5d shells %-14s path %d: %8.1f ms %7.1f Msamples*px/s" % (shells, label, r.lastStats().scene_path, best, w * h * spp / best * 1e-3), flush=True)
        finally:
            r.set_option(rt.RT_OPT_FILTER, 0); r.set_option(rt.RT_OPT_SCENE_FETCH, 0)
