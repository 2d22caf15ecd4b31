"""Runs the matrix-filter self-test three times with the same seed on a given build of the library and prints the six
counters of each run: identical lines = deterministic.  Usage: exp_nofence.py <path to librt_amd.so> [...]"""
import ctypes as C
import sys

for path in sys.argv[1:]:
    lib = C.CDLL(path)
    vp, ip = C.c_void_p, C.c_int
    lib.rt_create.argtypes = [C.POINTER(vp), ip, vp]
    lib.rt_selftest_matrix_filter.argtypes = [vp, C.c_uint32, ip, ip, ip, vp]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_destroy.restype = None
    ctx = vp()
    assert lib.rt_create(C.byref(ctx), 0, None) == 0
    print(path)
    for rep in range(3):
        out = (C.c_uint64 * 8)()
        rc = lib.rt_selftest_matrix_filter(ctx, 12345, 4096, 64, -16, out)
        print("  rc %d:" % rc, list(out)[:6], flush=True)
    lib.rt_destroy(ctx)
