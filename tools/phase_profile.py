"""Per-phase cycle stamps of the persistent kernel (RT_OPT_KERNEL = 2) for the three scene classes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
rt = ge.load_package()
r = rt.MetalRenderer(0)
cases = [("cover", {}, 1200, 675, 64, 50), ("nested_glass", {}, 640, 360, 16, 100),
         ("random_spheres", {"n": int(os.environ.get("NSPH", "20000"))}, 640, 360, 8, 50)]
if os.environ.get("SIZE"):
    w_, h_, spp_ = (int(v) for v in os.environ["SIZE"].split("x"))
    cases = [(c[0], c[1], w_, h_, spp_, c[5]) for c in cases]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
names = ["regen", "filter", "resolve", "quads+record", "shade"]
for name, kw, w, h, spp, depth in cases:
    objs, n = rt.scene(name, **kw)
    r.setupTexture(w, h)
    r.setScene(objs, n)
    r.set_option(rt.RT_OPT_SCENE_FETCH, int(os.environ.get('FETCH', '0')))
    r.set_option(rt.RT_OPT_CHUNK, int(os.environ.get('CHUNK', '-1')))
    r.set_option(rt.RT_OPT_TAIL_CHUNK, int(os.environ.get('TAIL', '-1')))
    r.set_option(rt.RT_OPT_RESOLVE, int(os.environ.get('RESOLVE', '0')))
    u = rt.config_uniforms(name, w, h, spp, depth, n)
    for kernel in (0, 2):
        r.set_option(rt.RT_OPT_KERNEL, kernel)
        r.accumClear(); r.render(u, None, 0, 0, spp); r.resolve(spp); r.synchronize()
        st = r.lastStats()
        print("%s n=%d kernel %d path %d: %.1f ms, %.2f seg/sample, %.1f TFLOP/s" % (
            name, n, kernel, st.scene_path, st.kernel_ms, st.segments / st.samples, 23 * st.tests / st.kernel_ms * 1e-9), flush=True)
    pc = list(st.phase_cycles)
    trips = max(1, pc[5])
    tot = sum(pc[:5])
    print("  ideal wave trips (segments / 64) %d" % (st.segments // 64))
    print("  wave trips %d, resolve-loop trips/trip %.2f, candidates/segment %.2f" % (trips, pc[6] / trips, pc[7] / max(1, st.segments)))
    for i, nm in enumerate(names):
        print("  %-13s %8.0f wave-cycles/trip  %5.1f %%" % (nm, pc[i] / trips, 100.0 * pc[i] / tot))
    if pc[6]:
        print("  resolve wave-cycles per resolve-loop trip: %.0f" % (pc[2] / pc[6]))
