"""A/B of two builds of the library on the cover scene (batch path): alternates the two builds, `reps` times each, one process
per measurement.  Usage: exp_ab.py <lib A> <lib B> [spp] [reps]"""
import os
import subprocess
import sys

a, b = sys.argv[1], sys.argv[2]
spp = sys.argv[3] if len(sys.argv) > 3 else "256"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as ge; rt = ge.load_package(); r = rt.MetalRenderer(0);"
        "objs, n = rt.scene('cover'); w, h, spp = 1920, 1080, %s; r.setupTexture(w, h); r.setScene(objs, n);"
        "u = rt.config_uniforms('cover', w, h, spp, 50, n)\n"
        "best = 1e30\n"
        "for i in range(3):\n"
        "    r.accumClear(); r.render(u, None, 0, 0, spp); r.resolve(spp); r.synchronize(); best = min(best, r.lastStats().kernel_ms)\n"
        "print('%%.1f' %% (w * h * spp / best * 1e-3))") % (root, spp)
res = {a: [], b: []}
for i in range(reps):
    for lib in (a, b):
        env = dict(os.environ, RT_AMD_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            sys.exit(out.stderr)
        res[lib].append(float(out.stdout.split()[-1]))
        print(lib, res[lib][-1], flush=True)
for lib in (a, b):
    print("%-40s best %.1f  mean %.1f Msamples*px/s" % (lib, max(res[lib]), sum(res[lib]) / len(res[lib])))
