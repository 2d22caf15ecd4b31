"""One rank's share of the benchmark frame at 8 and 4 ranks (125 and 250 samples per pixel of the cover scene at 1920x1080):
does handing out the last samples of every pixel in shorter work items (RT_OPT_TAIL_CHUNK, RT_OPT_TAIL_SAMPLES) pay once
the launch is short?  Msamples*px/s of the trace launch."""
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
for spp in (125, 250):
    u = rt.config_uniforms("cover", w, h, 1000, 50, n)
    for chunk in (-1, 32):
        for tail_chunk, tail_samples in ((-1, -1), (16, 16), (16, 32), (8, 16), (32, 32), (8, 8)):
            r.set_option(rt.RT_OPT_CHUNK, chunk)
            r.set_option(rt.RT_OPT_TAIL_CHUNK, tail_chunk)
            r.set_option(rt.RT_OPT_TAIL_SAMPLES, tail_samples)
            best = 1e30
            for rep in range(2):
                r.accumClear(); r.render(u, None, 0, 0, spp); r.synchronize()
                best = min(best, r.lastStats().kernel_ms)
            print("%4d spp  chunk %3d  tail items of %2d samples over the last %2d: %7.2f ms  %7.1f Msamples*px/s"
                  % (spp, chunk, tail_chunk, tail_samples, best, w * h * spp / best * 1e-3), flush=True)
r.close()
