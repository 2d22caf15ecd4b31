"""Matrix-core filter against the vector filters: self-test margins, bit-identity and time on the cover scene."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
rt = ge.load_package()
r = rt.MetalRenderer(0)
for sl in (-16, -18, -20, -22, -24):
    print("selftest slack 2^%d:" % sl, r.selftestMatrixFilter(seed=7, blocks=512, rounds=32, slack_log2=sl), flush=True)
w, h, spp = 1200, 675, 128
objs, n = rt.scene("cover")
r.setupTexture(w, h)
r.setScene(objs, n)
u = rt.config_uniforms("cover", w, h, spp, 50, n)
imgs = {}
for rep in range(2):
    for filt in (2, 0):
        r.set_option(rt.RT_OPT_FILTER, filt)
        r.accumClear(); r.render(u, None, 0, 0, spp); r.resolve(spp); r.synchronize()
        st = r.lastStats()
        imgs[filt] = r.readTexture().copy()
        print("filter %d: %.1f ms, grid %d x %d, %.2f seg/sample, %.1f Msamples/s" % (filt, st.kernel_ms, st.grid_blocks, st.block_threads, st.segments / st.samples, st.samples / st.kernel_ms * 1e-3), flush=True)
print("bit-identical:", np.array_equal(imgs[0].view(np.uint32), imgs[2].view(np.uint32)), "max abs diff", float(np.abs(imgs[0] - imgs[2]).max()))
