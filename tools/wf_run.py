"""One wavefront-formulation render of the cover scene (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
rt = ge.load_package()
w, h, spp = 1200, 675, int(os.environ.get("SPP", "128"))
r = rt.MetalRenderer(0)
objs, n = rt.scene(os.environ.get("SCENE", "cover"))
r.setupTexture(w, h)
r.setScene(objs, n)
r.set_option(rt.RT_OPT_KERNEL, 1)
if os.environ.get("CHUNK"):
    r.set_option(rt.RT_OPT_CHUNK, int(os.environ["CHUNK"]))
if os.environ.get("SLOTS"):
    r.set_option(rt.RT_OPT_WAVEFRONT_SLOTS, int(os.environ["SLOTS"]))
u = rt.config_uniforms(os.environ.get("SCENE", "cover"), w, h, spp, int(os.environ.get("DEPTH", "50")), n)
import time
for rep in range(2):
    t0 = time.perf_counter()
    r.accumClear(); r.render(u, None, 0, 0, spp); r.resolve(spp); r.synchronize()
    dt = time.perf_counter() - t0
    st = r.lastStats()
    print("wavefront: %.1f ms wall, %.1f ms kernels, %.1f Msamples/s, path %d" % (dt * 1e3, st.kernel_ms, st.samples / dt * 1e-6, st.scene_path), flush=True)
