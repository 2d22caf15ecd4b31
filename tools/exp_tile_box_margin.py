"""rt_selftest_tile_boxes at the shipped allowances and at a half, a quarter and an eighth of them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
rt = ge.load_package()
r = rt.MetalRenderer(0)
for n, extent in ((20000, 22.0), (5000, 20.0), (100000, None)):
    arr, n = rt.scene("random_spheres", n=n, extent=extent) if extent else rt.scene("random_spheres")
    r.setScene(arr, n)
    for padding in (0, -1, -2, -3):
        tot = {}
        for seed in range(1, 9):
            for k, v in r.selftestTileBoxes(seed=seed, blocks=512 if n < 50000 else 64, rays_per_thread=2, padding_log2=padding).items():
                tot[k] = tot.get(k, 0) + v
        print("tiles of %6d spheres padding 2^%-3d rays %.3g left out %.3g pairs %.3g wrong %d" % (n, padding, tot["rays"], tot["left_out"], tot["pairs"], tot["wrong"]), flush=True)
r.close()
