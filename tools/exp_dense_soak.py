"""Soak of the pruned exact pass: the dense test scenes of tests/test_gpu_parity.py at a larger size, image checksums of the
pooled exact pass (1, resolves every candidate) against the per-lane form with pruning (2) and the automatic choice (0).
Usage: exp_dense_soak.py [width height spp]"""
import ctypes as C
import os
import sys
import zlib

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import __graft_entry__ as ge
from test_gpu_parity import _dense_scene

rt = ge.load_package()
w, h, spp = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (640, 360, 16)
r = rt.MetalRenderer(0)
r.setupTexture(w, h)
bad = 0
for kind in ("drifting_shells", "clutter", "far_shells", "shells_and_strays"):
    objs, cam_from, cam_at = _dense_scene(rt, kind)
    n = len(objs)
    arr = rt.object_array(objs)
    u = rt.config_uniforms("nested_glass", w, h, spp, 60, n)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    rt.host_lib().rt_camera_setup(C.byref(u), f3(cam_from), f3(cam_at), f3([0, 1, 0]), 40.0, 0.0, 1.0, w, h)
    crcs = {}
    for mode in (1, 2, 0):
        r.set_option(rt.RT_OPT_RESOLVE, mode)
        img = r.renderImage(u, arr, n)
        crcs[mode] = zlib.crc32(img.tobytes())
        print("%-18s resolve %d: %8.1f ms  %7.1f Msamples*px/s  crc %08x" % (kind, mode, r.lastStats().kernel_ms,
                                                                             w * h * spp / r.lastStats().kernel_ms * 1e-3, crcs[mode]), flush=True)
    bad += crcs[1] != crcs[2] or crcs[1] != crcs[0]
r.set_option(rt.RT_OPT_RESOLVE, 0)
sys.exit(1 if bad else 0)
