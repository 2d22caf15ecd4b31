"""A/B/... of several builds of the library on one workload (batch path, default options): alternates the builds, `reps`
times each, one process per measurement, and prints a checksum of the linear sums so that builds that must agree can be
seen to.  Usage: exp_libs.py <spp> <reps> <lib> [<lib> ...]   (SCENE, SIZE=WxH, DEPTH, BEGIN = first sample index, FILTER = RT_OPT_FILTER, from the environment)"""
import os
import subprocess
import sys

spp, reps, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = os.environ.get("SCENE", "cover")
size = os.environ.get("SIZE", "1920x1080")
depth = os.environ.get("DEPTH", "50")
begin = int(os.environ.get("BEGIN", "0"))
filt = int(os.environ.get("FILTER", "0"))   # RT_OPT_FILTER
code = ("import sys, zlib, ctypes as C, numpy as np; sys.path.insert(0, %r); import __graft_entry__ as ge; rt = ge.load_package(); r = rt.MetalRenderer(0);"
        "objs, n = rt.scene(%r); w, h = (int(v) for v in %r.split('x')); spp = %s; r.setupTexture(w, h); r.set_option(rt.RT_OPT_FILTER, %d); r.setScene(objs, n);"
        "u = rt.config_uniforms(%r, w, h, spp, %s, n)\n"
        "best = 1e30\n"
        "for i in range(3):\n"
        "    r.accumClear(); r.render(u, None, 0, %d, %d + spp); r.synchronize(); st = r.lastStats(); best = min(best, st.kernel_ms)\n"
        "hip = C.CDLL('libamdhip64.so'); hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]\n"
        "acc = np.empty((h, w, 4), np.float32); hip.hipMemcpy(acc.ctypes.data, r.accumPtr(), acc.nbytes, 2)\n"
        "print('%%08x %%.3f %%.1f' %% (zlib.crc32(acc.tobytes()), best * 1e6 / max(1, st.wave_trips) * 1024, w * h * spp / best * 1e-3))") % (root, scene, size, spp, filt, scene, depth, begin, begin)
res = {lib: [] for lib in libs}
crc = {}
for i in range(reps):
    for lib in libs:
        env = dict(os.environ, RT_AMD_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            sys.exit(lib + ": " + out.stderr[-2000:])
        c, t, v = out.stdout.split()[-3:]
        crc[lib] = c + "  %s ns per wave trip and SIMD" % t
        res[lib].append(float(v))
        print(lib, c, t, v, flush=True)
for lib in libs:
    print("%-32s crc %s  best %.1f  mean %.1f Msamples*px/s" % (lib, crc[lib], max(res[lib]), sum(res[lib]) / len(res[lib])))
