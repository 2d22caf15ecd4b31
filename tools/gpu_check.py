"""First-contact GPU script: parity against the oracle on small cases, then timings of the kernel variants."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

rt = ge.load_package()
ora = ge.load_oracle()
ora.rto_dispatch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]


def oracle_image(u, objs, w, h):
    tex = np.zeros((h, w, 4), np.float32)
    tex[..., 3] = 1
    st = (C.c_uint64 * 5)()
    ora.rto_dispatch(tex.ctypes.data, w, h, C.byref(u), objs, 0, 1, st)
    return tex, st[0]


def compare(name, a, b):
    same = (a.view(np.uint32) == b.view(np.uint32)).all(axis=-1)
    mse = float(np.mean((a[..., :3].astype(np.float64) - b[..., :3]) ** 2))
    psnr = 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)
    print("%-40s identical pixels %d/%d  max|diff| %.3g  PSNR %.1f dB" % (name, same.sum(), same.size, np.abs(a - b).max(), psnr), flush=True)
    return same.all()


def main():
    r = rt.MetalRenderer(0)
    cases = [("three_spheres", 400, 225, 100, 50), ("cover", 240, 135, 16, 50), ("cornell", 200, 160, 8, 10)]
    for name, w, h, spp, depth in cases:
        objs, n = rt.scene(name)
        u = rt.config_uniforms(name, w, h, spp, depth, n)
        t0 = time.time()
        ref, seg = oracle_image(u, objs, w, h)
        t1 = time.time()
        print("oracle %s %dx%d %dspp: %.2fs, %.2f segments/sample" % (name, w, h, spp, t1 - t0, seg / (w * h * spp)), flush=True)
        r.setupTexture(w, h)
        for fetch in (0, 1):
            r.set_option(rt.RT_OPT_SCENE_FETCH, fetch)
            for chunk in (0, 8):
                img = r.renderImage(u, objs, n, chunk=chunk)
                st = r.lastStats()
                compare("%s fetch=%d chunk=%d (%.1f ms, seg/s %.2f)" % (name, fetch, chunk, st.kernel_ms, st.segments / (w * h * spp)), img, ref)
    # timings
    name, w, h, spp, depth = "cover", 1200, 675, 128, 50
    objs, n = rt.scene(name)
    u = rt.config_uniforms(name, w, h, spp, depth, n)
    r.setupTexture(w, h)
    r.setScene(objs, n)
    for fetch in (0, 1):
        r.set_option(rt.RT_OPT_SCENE_FETCH, fetch)
        for bpc in (0, 2, 4, 8):
            r.set_option(rt.RT_OPT_BLOCKS_PER_CU, bpc)
            for chunk in (32, 8):
                r.set_option(rt.RT_OPT_CHUNK, chunk)
                r.accumClear()
                r.render(u, None, 0, 0, spp)
                r.synchronize()
                st = r.lastStats()
                tf = 23.0 * st.tests / (st.kernel_ms * 1e-3) * 1e-12
                print("cover %dx%d %dspp fetch=%d blocks/CU=%d(grid %d) chunk=%d: %.1f ms  %.1f Msamples/s  %.2f seg/sample  %.1f TFLOP/s alg (%.1f%% of 157.3)"
                      % (w, h, spp, fetch, bpc, st.grid_blocks, chunk, st.kernel_ms, w * h * spp / st.kernel_ms * 1e-3, st.segments / (w * h * spp), tf, tf / 1.573), flush=True)
    r.close()


if __name__ == "__main__":
    main()
