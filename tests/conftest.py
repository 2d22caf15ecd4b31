import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """The host-side package (ctypes bindings over librt_amd.so / librt_host.so)."""
    return ge.load_package()


class Oracle:
    """ctypes view of oracle/librt_oracle.so -- the CPU restatement used only as the checker."""

    def __init__(self):
        lib = ge.load_oracle()
        vp, ip, u32 = C.c_void_p, C.c_int, C.c_uint32
        lib.rto_pcg_hash.restype = u32
        lib.rto_pcg_hash.argtypes = [u32]
        lib.rto_random_double.restype = C.c_float
        lib.rto_random_double.argtypes = [vp]
        lib.rto_random_vec.restype = None
        lib.rto_random_vec.argtypes = [ip, vp, vp]
        lib.rto_sample_seed.restype = None
        lib.rto_sample_seed.argtypes = [u32, u32, ip, ip, vp]
        lib.rto_get_ray.restype = None
        lib.rto_get_ray.argtypes = [vp, u32, u32, ip, vp, vp, vp]
        lib.rto_hit_object.argtypes = [vp, vp, vp, C.c_float, C.c_float, vp]
        lib.rto_world_hit.argtypes = [vp, ip, vp, vp, vp]
        lib.rto_scatter.argtypes = [vp, vp, vp, vp, vp, vp]
        lib.rto_sample.argtypes = [vp, vp, u32, u32, ip, vp, vp]
        lib.rto_accumulate.restype = None
        lib.rto_accumulate.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp]
        lib.rto_dispatch.restype = None
        lib.rto_dispatch.argtypes = [vp, ip, ip, vp, vp, ip, ip, vp]
        lib.rto_first_hit.restype = None
        lib.rto_first_hit.argtypes = [vp, vp, ip, ip, vp, vp]
        lib.rto_resolve.restype = None
        lib.rto_resolve.argtypes = [vp, ip, ip, vp]
        lib.rto_present_bgra8.restype = None
        lib.rto_present_bgra8.argtypes = [vp, ip, ip, vp, ip, ip]
        lib.rto_pow5_sweep.restype = C.c_uint64
        lib.rto_pow5_sweep.argtypes = [u32, u32, vp, vp]
        self.lib = lib

    def image(self, u, objs, w, h, tex=None, reference_grid=0, stats=None):
        """One dispatch of the kernel over the tile in u onto tex (fresh (0,0,0,1) target when None)."""
        if tex is None:
            tex = np.zeros((h, w, 4), np.float32)
            tex[..., 3] = 1.0
        st = (C.c_uint64 * 5)()
        self.lib.rto_dispatch(tex.ctypes.data, w, h, C.byref(u), objs, reference_grid, 1, st)
        if stats is not None:
            stats[:] = list(st)
        return tex

    def accumulate(self, u, objs, w, h, ray_begin, ray_end):
        sums = np.zeros((h, w, 4), np.float32)
        self.lib.rto_accumulate(C.byref(u), objs, w, h, ray_begin, ray_end, sums.ctypes.data, None)
        return sums

    def resolve(self, sums, spp):
        tex = np.empty_like(sums)
        self.lib.rto_resolve(np.ascontiguousarray(sums).ctypes.data, sums.shape[0] * sums.shape[1], spp, tex.ctypes.data)
        return tex

    def first_hit(self, u, objs, w, h):
        out4 = np.empty((h, w, 4), np.float32)
        ids = np.empty((h, w), np.int32)
        self.lib.rto_first_hit(C.byref(u), objs, w, h, out4.ctypes.data, ids.ctypes.data)
        return out4, ids

    def present(self, tex, out_w, out_h):
        out = np.empty((out_h, out_w, 4), np.uint8)
        t = np.ascontiguousarray(tex, np.float32)
        self.lib.rto_present_bgra8(t.ctypes.data, t.shape[1], t.shape[0], out.ctypes.data, out_w, out_h)
        return out


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def renderer(rt):
    """One GPU context shared by the GPU tests (a fresh one per test would re-pay HIP start-up)."""
    r = rt.MetalRenderer(device=0)
    yield r
    r.close()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def psnr(a, b):
    mse = float(np.mean((a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)) ** 2))
    return 99.0 if mse == 0 else 10.0 * np.log10(1.0 / mse)
