"""Pass B forms (RT_OPT_RESOLVE 1 pooled, 2 per lane, 0 automatic) on the cover scene and the nested shells: time and a
checksum of the linear sums (the three forms must agree bit for bit)."""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

rt = ge.load_package()
r = rt.MetalRenderer(0)
cases = [("cover", 1920, 1080, 64, 50), ("nested_glass", 1920, 1080, 16, 100), ("three_spheres", 1200, 675, 64, 50)]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, w, h, spp, depth in cases:
    objs, n = rt.scene(name)
    r.setupTexture(w, h)
    r.setScene(objs, n)
    u = rt.config_uniforms(name, w, h, spp, depth, n)
    for mode in (1, 2, 0):
        r.set_option(rt.RT_OPT_RESOLVE, mode)
        best = 1e30
        for rep in range(2):
            r.accumClear(); r.render(u, None, 0, 0, spp); r.resolve(spp); r.synchronize()
            best = min(best, r.lastStats().kernel_ms)
        img = r.readTexture()
        print("%-14s resolve %d: %8.1f ms  %7.1f Msamples*px/s  crc %08x" % (name, mode, best, w * h * spp / best * 1e-3, zlib.crc32(img.tobytes())), flush=True)
r.close()
