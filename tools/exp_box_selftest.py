"""rt_selftest_box at length on the cover scene and the layer scenes of the tests, with the library's padding and with
less: rays, rays the box test leaves out, pairs checked, wrongly left out.   exp_box_selftest.py [seeds blocks rays_per_thread]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as ge

rt = ge.load_package()
from test_gpu_parity import _layer_scene

seeds, blocks, rpt = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 2048, 8)
r = rt.MetalRenderer(0)
for kind in ("cover", "cover_like", "wide_in_the_middle", "two_rounds", "thick_cloud"):
    if kind == "cover":
        arr, n = rt.scene(kind)
    else:
        objs = _layer_scene(rt, kind)[0]
        n = len(objs)
        arr = rt.object_array(objs)
    r.setScene(arr, n)
    for padding in (0, -3, -6, -10, -40):
        tot = {}
        for seed in range(1, seeds + 1):
            for k, v in r.selftestBox(seed=seed, blocks=blocks, rays_per_thread=rpt, padding_log2=padding).items():
                tot[k] = tot.get(k, 0) + v
        print("%-20s padding 2^%-3d rays %.3g left out %.3g pairs %.3g roots %.3g wrong %d" %
              (kind, padding, tot["rays"], tot["left_out"], tot["pairs"], tot["roots"], tot["wrong"]), flush=True)
for n, extent in ((5000, 20.0), (20000, 22.0), (100000, None)):
    arr, n = rt.scene("random_spheres", n=n, extent=extent) if extent else rt.scene("random_spheres")
    r.setScene(arr, n)
    for padding in (0, -3, -6, -10, -40):
        tot = {}
        for seed in range(1, seeds + 1):
            for k, v in r.selftestTileBoxes(seed=seed, blocks=max(1, blocks // (8 if n < 50000 else 64)), rays_per_thread=2, padding_log2=padding).items():
                tot[k] = tot.get(k, 0) + v
        print("tiles of %6d spheres padding 2^%-3d rays %.3g (ray, tile) left out %.3g pairs %.3g roots %.3g wrong %d" %
              (n, padding, tot["rays"], tot["left_out"], tot["pairs"], tot["roots"], tot["wrong"]), flush=True)
r.close()
