"""GPU: the HIP path, called through the C-ABI, against the CPU oracle and the committed vectors.

Bar: bit-identical float images and first-hit buffers (the task's stated tolerance -- exact object ids, 1e-5
relative on t and normal, PSNR >= 45 dB on converged images -- is met with equality).
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import bits, psnr

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "oracle_vectors.npz"))


def strict(r, rt, chunk=0, fetch=0):
    r.set_option(rt.RT_OPT_SCENE_FETCH, fetch)
    r.set_option(rt.RT_OPT_FILTER, 0)
    r.set_option(rt.RT_OPT_CHUNK, chunk)
    r.set_option(rt.RT_OPT_TAIL_CHUNK, -1)
    r.set_option(rt.RT_OPT_TAIL_SAMPLES, -1)
    r.set_option(rt.RT_OPT_BLOCKS_PER_CU, 0)
    r.set_option(rt.RT_OPT_RESOLVE, 0)
    r.set_option(rt.RT_OPT_NARROW, 1)


def test_committed_images_bit_exact(rt, renderer):
    """No oracle call: GPU output against the vectors in tests/golden/."""
    strict(renderer, rt)
    for name in ("three_spheres", "cover", "cornell"):
        w, h, spp, depth = [int(v) for v in GOLD["img_%s_cfg" % name]]
        objs, n = rt.scene(name)
        u = rt.config_uniforms(name, w, h, spp, depth, n)
        renderer.setupTexture(w, h)
        img = renderer.renderImage(u, objs, n)
        assert np.array_equal(bits(img), bits(GOLD["img_%s" % name])), name
        t4, ids = renderer.firstHit(u, objs, n)
        assert np.array_equal(ids, GOLD["fh_%s_id" % name].astype(np.int32)), name
        assert np.array_equal(bits(t4), bits(GOLD["fh_%s_t" % name])), name


@pytest.mark.parametrize("name,w,h", [("three_spheres", 400, 225), ("cover", 400, 225), ("cornell", 250, 200)])
def test_first_hit_buffers(rt, oracle, renderer, name, w, h):
    """Exact object id; t and normal within 1e-5 relative (they are in fact bit-identical)."""
    strict(renderer, rt)
    objs, n = rt.scene(name)
    u = rt.config_uniforms(name, w, h, 1, 50, n)
    renderer.setupTexture(w, h)
    t4, ids = renderer.firstHit(u, objs, n)
    want4, want_ids = oracle.first_hit(u, objs, w, h)
    assert np.array_equal(ids, want_ids)
    np.testing.assert_allclose(t4, want4, rtol=1e-5, atol=0)
    assert np.array_equal(bits(t4), bits(want4))
    assert (ids >= 0).any() and len(np.unique(ids)) >= 3


@pytest.mark.parametrize("fetch", [0, 1])
def test_three_sphere_full_config_bit_exact(rt, oracle, renderer, fetch):
    """The 400x225, 100 spp, depth 50 three-sphere configuration in full."""
    strict(renderer, rt, fetch=fetch)
    objs, n = rt.scene("three_spheres")
    u = rt.config_uniforms("three_spheres", 400, 225, 100, 50, n)
    renderer.setupTexture(400, 225)
    img = renderer.renderImage(u, objs, n)
    stats = [0] * 5
    want = oracle.image(u, objs, 400, 225, stats=stats)
    assert np.array_equal(bits(img), bits(want))
    st = renderer.lastStats()
    assert st.segments == stats[0] and st.tests == stats[1] and st.samples == 400 * 225 * 100


@pytest.mark.parametrize("name,w,h,spp,depth", [("cover", 240, 135, 32, 50), ("cornell", 200, 160, 16, 10),
                                                 ("nested_glass", 160, 90, 4, 100)])
def test_images_bit_exact(rt, oracle, renderer, name, w, h, spp, depth):
    strict(renderer, rt)
    objs, n = rt.scene(name)
    u = rt.config_uniforms(name, w, h, spp, depth, n)
    renderer.setupTexture(w, h)
    img = renderer.renderImage(u, objs, n)
    want = oracle.image(u, objs, w, h)
    assert np.array_equal(bits(img), bits(want))


@pytest.mark.parametrize("fetch,narrow,path", [(0, 1, 5), (0, 0, 4), (1, 1, 1), (2, 1, 2)])
def test_scene_larger_than_lds_bit_exact(rt, oracle, renderer, fetch, narrow, path):
    """5000 spheres exceed the LDS staging limit: by default every wave reads the lifted tiles its rays can touch in place
    (matrix-core filter at half width, spheres visited in spatial order); with RT_OPT_NARROW 0 the workgroups stream all
    tiles through LDS; on request the records come through the scalar cache or per-wave tiles (vector filter, object
    order).  All give the oracle's bits."""
    strict(renderer, rt, fetch=fetch)
    renderer.set_option(rt.RT_OPT_NARROW, narrow)
    objs, n = rt.scene("random_spheres", n=5000, extent=20.0)
    u = rt.config_uniforms("random_spheres", 96, 54, 2, 50, n)
    renderer.setupTexture(96, 54)
    try:
        img = renderer.renderImage(u, objs, n)
        assert renderer.lastStats().scene_path == path
    finally:
        strict(renderer, rt)
    assert np.array_equal(bits(img), bits(oracle.image(u, objs, 96, 54)))


def test_tiled_matrix_filter_awkward_spheres_bit_exact(rt, oracle, renderer):
    """The tiled kind sorts spheres in space, so equal hit distances are settled by an explicit rule instead of by the
    visiting order.  3000 random spheres plus: exact duplicates of three of them appended at the end (the earlier
    object must win although it is visited later), concentric shells sharing a centre, radius 0, negative radius, NaN
    radius, NaN centre, a huge enclosing glass sphere and a far-away sphere -- against the oracle, and the scalar-cache
    kind on the same scene."""
    strict(renderer, rt)
    w, h, spp = 96, 54, 3
    base, n0 = rt.scene("random_spheres", n=3000, extent=12.0)
    glass = rt.SwiftMaterial.dielectric(1.5)
    red = rt.SwiftMaterial.lambertian(0.9, 0.1, 0.1)
    extra = [rt.Object.from_buffer_copy(base[i]) for i in (7, 1500, 2999)]
    for e in extra:
        e.mat = red   # a different material: picking the wrong twin changes the image
    extra.append(rt.makeSphere((0.5, 0.3, 0.5), 0.3, glass))
    extra.append(rt.makeSphere((0.5, 0.3, 0.5), 0.2, red))
    extra.append(rt.makeSphere((0.5, 0.3, 0.5), 0.3, red))
    extra.append(rt.makeSphere((1.0, 0.2, 1.0), 0.0, red))
    extra.append(rt.makeSphere((2.0, 0.5, -1.0), -0.5, glass))
    extra.append(rt.makeSphere((3.0, 0.5, 1.0), float("nan"), red))
    extra.append(rt.makeSphere((float("nan"), 0.5, 1.0), 0.3, red))
    extra.append(rt.makeSphere((0.0, 0.0, 0.0), 300.0, glass))
    extra.append(rt.makeSphere((5000.0, 100.0, -3000.0), 50.0, red))
    n = n0 + len(extra)
    objs = (rt.Object * n)(*([base[i] for i in range(n0)] + extra))
    u = rt.config_uniforms("random_spheres", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    want = oracle.image(u, objs, w, h)
    for fetch, narrow, path in ((0, 1, 5), (3, 1, 5), (0, 0, 4), (3, 0, 4), (1, 1, 1)):
        strict(renderer, rt, fetch=fetch)
        renderer.set_option(rt.RT_OPT_NARROW, narrow)
        try:
            img = renderer.renderImage(u, objs, n)
            assert renderer.lastStats().scene_path == path
        finally:
            strict(renderer, rt)
        assert np.array_equal(bits(img), bits(want)), (fetch, narrow)


@pytest.mark.parametrize("eye,fov", [((2600.0, 400.0, 600.0), 0.6), ((3.0, 0.15, 2.0), 70.0), ((0.0, 60.0, 0.1), 25.0)])
def test_tiles_read_in_place_from_far_near_and_above(rt, oracle, renderer, eye, fov):
    """The boxes of the tiled kind that reads its tiles in place take their allowances from each ray's distance: a
    camera thousands of radii away (narrow view, so that the rays still find the spheres), one between the spheres, and
    one looking straight down -- against the oracle, and against the tiles streamed through LDS."""
    strict(renderer, rt)
    w, h, spp = 96, 54, 3
    objs, n = rt.scene("random_spheres", n=4000, extent=16.0)
    u = rt.config_uniforms("random_spheres", w, h, spp, 50, n)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    rt.host_lib().rt_camera_setup(C.byref(u), f3(eye), f3((0.0, 0.2, 0.0)), f3([0, 1, 0]), fov, 0.0, 10.0, w, h)
    renderer.setupTexture(w, h)
    want = oracle.image(u, objs, w, h)
    assert len(np.unique(bits(want).reshape(-1, 4), axis=0)) > 50
    try:
        for narrow, path in ((1, 5), (0, 4)):
            renderer.set_option(rt.RT_OPT_NARROW, narrow)
            img = renderer.renderImage(u, objs, n)
            assert renderer.lastStats().scene_path == path
            assert np.array_equal(bits(img), bits(want)), narrow
    finally:
        renderer.set_option(rt.RT_OPT_NARROW, 1)


def test_tiled_and_vector_filters_agree_on_100k_spheres(rt, renderer):
    """The 100 000-sphere configuration at 640x360 x 4 spp, beyond what the oracle checks in seconds: lifted tiles read
    in place (default, twice: run-to-run determinism) and streamed through LDS against the vector filter through the
    scalar cache, bit for bit."""
    strict(renderer, rt, chunk=2)
    w, h, spp = 640, 360, 4
    objs, n = rt.scene("random_spheres")
    u = rt.config_uniforms("random_spheres", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    renderer.setScene(objs, n)
    imgs = []
    for fetch, narrow, path in ((0, 1, 5), (0, 1, 5), (1, 1, 1), (0, 0, 4)):
        strict(renderer, rt, chunk=2, fetch=fetch)
        renderer.set_option(rt.RT_OPT_NARROW, narrow)
        try:
            renderer.accumClear()
            renderer.render(u, None, 0, 0, spp)
            renderer.resolve(spp)
            imgs.append(renderer.readTexture().copy())
            assert renderer.lastStats().scene_path == path
        finally:
            strict(renderer, rt)
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))
    assert np.array_equal(bits(imgs[0]), bits(imgs[2]))
    assert np.array_equal(bits(imgs[0]), bits(imgs[3]))


def test_chunked_work_items_sum_in_chunk_order(rt, oracle, renderer):
    """With RT_OPT_CHUNK=c each pixel's samples are summed per chunk, then chunks are added in order."""
    strict(renderer, rt, chunk=5)
    w, h, spp = 120, 68, 16
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    img = renderer.renderImage(u, objs, n, chunk=5)
    acc = np.zeros((h, w, 4), np.float32)
    for b in range(0, spp, 5):
        acc = acc + oracle.accumulate(u, objs, w, h, b, min(b + 5, spp))
    assert np.array_equal(bits(img), bits(oracle.resolve(acc, spp)))
    # and the result is within float-sum reordering distance of the reference's single sequential sum
    np.testing.assert_allclose(img, oracle.image(u, objs, w, h), atol=2e-6)


def test_item_schedules_sum_in_the_reported_order(rt, oracle, renderer):
    """Whatever the item schedule -- one-sample items, long items followed by a tail of short ones, the automatic
    choice -- each reported range is summed sample by sample and the range sums are added in order."""
    w, h, spp = 96, 54, 23
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    for chunk, tail_chunk, tail_samples in ((1, -1, -1), (7, 3, 8), (6, 2, -1), (-1, -1, -1), (-1, 1, 5)):
        strict(renderer, rt, chunk=chunk)
        renderer.set_option(rt.RT_OPT_TAIL_CHUNK, tail_chunk)
        renderer.set_option(rt.RT_OPT_TAIL_SAMPLES, tail_samples)
        img = renderer.renderImage(u, objs, n)
        ranges = renderer.lastStats().chunk_ranges(0, spp)
        assert ranges[0][0] == 0 and ranges[-1][1] == spp and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        if chunk == 1:
            assert len(ranges) == spp
        if 0 < tail_chunk < chunk:
            assert ranges[-2][1] - ranges[-2][0] == tail_chunk and ranges[0][1] - ranges[0][0] > tail_chunk
        acc = np.zeros((h, w, 4), np.float32)
        for b, e in ranges:
            acc = acc + oracle.accumulate(u, objs, w, h, b, e)
        assert np.array_equal(bits(img), bits(oracle.resolve(acc, spp))), (chunk, tail_chunk, tail_samples)
    strict(renderer, rt)


@pytest.mark.parametrize("w,h,spp,chunk,plane", [(120, 68, 128, 32, 4), (96, 54, 100, 64, 8), (200, 120, 48, 16, 2)])
def test_planes_taken_over_at_the_end_of_a_launch_bit_exact(rt, oracle, renderer, w, h, spp, chunk, plane):
    """Frames so small that the work counter is exhausted almost at once: most of the launch is lanes without work taking
    planes over from the lanes of their wave that have samples left (RT_OPT_PLANE_SAMPLES).  Whole image against the
    oracle: every reported range summed sample by sample, the range sums added in order."""
    strict(renderer, rt, chunk=chunk)
    renderer.set_option(rt.RT_OPT_PLANE_SAMPLES, plane)
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    try:
        img = renderer.renderImage(u, objs, n)
        st = renderer.lastStats()
    finally:
        renderer.set_option(rt.RT_OPT_PLANE_SAMPLES, -1)
    ranges = st.chunk_ranges(0, spp)
    assert st.tail_chunk == plane and len(ranges) > spp // chunk   # finer planes at the end
    assert ranges[0][0] == 0 and ranges[-1][1] == spp and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    sums = np.zeros((h, w, 4), np.float32)
    for b, e in ranges:
        sums = sums + oracle.accumulate(u, objs, w, h, b, e)
    assert np.array_equal(bits(img), bits(oracle.resolve(sums, spp)))


def test_lanes_sharing_items_before_the_counter_runs_out_bit_exact(rt, oracle, renderer):
    """A frame with many more work items than lanes, so that the waves spend the last item or so per lane sharing their
    items' planes and drawing new items in between (RT_OPT_SHARE 0): the whole frame equals the one in which planes
    change lanes only once the counter is exhausted (RT_OPT_SHARE 1), bit for bit, and scattered pixels equal the
    oracle's sums over the reported ranges."""
    strict(renderer, rt, chunk=16)
    w, h, spp = 960, 540, 64
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    imgs = []
    try:
        renderer.set_option(rt.RT_OPT_PLANE_SAMPLES, 4)
        for share in (0, 1):
            renderer.set_option(rt.RT_OPT_SHARE, share)
            imgs.append(renderer.renderImage(u, objs, n))
            st = renderer.lastStats()
    finally:
        renderer.set_option(rt.RT_OPT_SHARE, 1)
        renderer.set_option(rt.RT_OPT_PLANE_SAMPLES, -1)
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))
    ranges = st.chunk_ranges(0, spp)
    assert st.tail_chunk == 4 and ranges[-1][1] == spp
    rng = np.random.RandomState(11)
    for _ in range(24):
        x, y = int(rng.randint(0, w)), int(rng.randint(0, h))
        total = np.zeros(4, np.float32)
        for b, e in ranges:
            acc = np.zeros(4, np.float32)
            for k in range(b, e):
                c = (C.c_float * 4)()
                if oracle.lib.rto_sample(C.byref(u), objs, x, y, k, c, None):
                    acc = acc + np.array(list(c), np.float32)
            total = total + acc
        want = oracle.resolve(total.reshape(1, 1, 4), spp)[0, 0]
        assert np.array_equal(bits(imgs[0][y, x]), bits(want)), (x, y)


def test_sample_split_equals_one_call(rt, renderer):
    """Two calls over [0,s/2) and [s/2,s) into one buffer == one call with chunk s/2 (same add order)."""
    strict(renderer, rt, chunk=8)
    w, h, spp = 320, 180, 16
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    one = renderer.renderImage(u, objs, n, chunk=8)
    renderer.setScene(objs, n)
    renderer.accumClear()
    renderer.render(u, None, 0, 0, 8)
    renderer.render(u, None, 0, 8, 16)
    renderer.resolve(spp)
    renderer.synchronize()
    two = renderer.readTexture()
    assert np.array_equal(bits(one), bits(two))


def test_tile_row_split_equals_one_call(rt, renderer):
    """The fallback partition: rows of 64-pixel tiles rendered by separate calls (ranks) into one buffer give the
    bits of one call over the whole image -- a pixel's sum does not depend on which call traced it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rt_partition", os.path.join(HERE, "..", "ray-tracing_amd", "partition.py"))
    part = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(part)
    strict(renderer, rt, chunk=4)
    w, h, spp = 200, 200, 8
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    one = renderer.renderImage(u, objs, n, chunk=4)
    renderer.setScene(objs, n)
    renderer.accumClear()
    rows = []
    for rank in range(3):
        y0, y1 = part.tile_rows(rank, 3, h)
        rows.append((y0, y1))
        if y1 > y0:
            u.tileY, u.tileHeight = y0, y1 - y0
            renderer.render(u, None, 0, 0, spp)
    assert rows[0][0] == 0 and rows[-1][1] == h and all(rows[i][1] == rows[i + 1][0] for i in range(2))
    renderer.resolve(spp)
    renderer.synchronize()
    assert np.array_equal(bits(one), bits(renderer.readTexture()))


def test_full_size_cover_spot_pixels(rt, oracle, renderer):
    """1200x675 cover configuration: 160 scattered pixels' linear sums against the oracle, bit for bit,
    plus run-to-run determinism of the whole frame."""
    strict(renderer, rt)
    w, h, spp = 1200, 675, 24
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    img1 = renderer.renderImage(u, objs, n)
    img2 = renderer.renderImage(u, objs, n, chunk=0)
    assert np.array_equal(bits(img1), bits(img2))
    rng = np.random.RandomState(3)
    for _ in range(160):
        x, y = int(rng.randint(0, w)), int(rng.randint(0, h))
        acc = np.zeros(4, np.float32)
        for k in range(spp):
            c = (C.c_float * 4)()
            if oracle.lib.rto_sample(C.byref(u), objs, x, y, k, c, None):
                acc = acc + np.array(list(c), np.float32)
        want = oracle.resolve(acc.reshape(1, 1, 4), spp)[0, 0]
        assert np.array_equal(bits(img1[y, x]), bits(want)), (x, y)
    assert np.isfinite(img1).all() and img1.min() >= 0 and img1.max() <= 1 and (img1[..., 3] == 1).all()


def test_cover_1200x675_at_its_own_500_spp_spot_pixels(rt, oracle, renderer):
    """The 1200x675 cover configuration at the sample count it is specified with (500 spp, depth 50, default options;
    rayIndex enters the seed, Shaders.metal:121-126): 32 scattered pixels, each chunk's sequential sum, the chunks added
    in order, mean and gamma recomputed by the oracle, bit for bit."""
    for opt in (rt.RT_OPT_SCENE_FETCH, rt.RT_OPT_FILTER, rt.RT_OPT_BLOCKS_PER_CU, rt.RT_OPT_RESOLVE):
        renderer.set_option(opt, 0)
    for opt in (rt.RT_OPT_CHUNK, rt.RT_OPT_TAIL_CHUNK, rt.RT_OPT_TAIL_SAMPLES):
        renderer.set_option(opt, -1)
    w, h, spp = 1200, 675, 500
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    renderer.setScene(objs, n)
    renderer.accumClear()
    renderer.render(u, None, 0, 0, spp)
    renderer.resolve(spp)
    img = renderer.readTexture()
    st = renderer.lastStats()
    assert st.scene_path == 3 and st.samples == w * h * spp
    ranges = st.chunk_ranges(0, spp)
    assert ranges[0][0] == 0 and ranges[-1][1] == spp and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    rng = np.random.RandomState(29)
    pixels = [(int(rng.randint(0, w)), int(rng.randint(0, h))) for _ in range(32)]
    seen = set()
    for (x, y), acc in zip(pixels, oracle_spot_pixels(oracle, u, objs, pixels, ranges)):
        want = oracle.resolve(acc.reshape(1, 1, 4), spp)[0, 0]
        assert np.array_equal(bits(img[y, x]), bits(want)), (x, y)
        seen.add(tuple(want))
    assert len(seen) > 16


def test_bench_configuration_spot_pixels(rt, oracle, renderer):
    """The configuration bench.py times -- cover scene, 1920x1080, 1000 spp, depth 50, default options (matrix-core
    filter, automatic work items) -- at 32 scattered pixels: each chunk's sequential sum, the chunks added in order,
    mean, gamma, all recomputed by the oracle and compared bit for bit."""
    for opt in (rt.RT_OPT_SCENE_FETCH, rt.RT_OPT_FILTER, rt.RT_OPT_BLOCKS_PER_CU):
        renderer.set_option(opt, 0)
    for opt in (rt.RT_OPT_CHUNK, rt.RT_OPT_TAIL_CHUNK, rt.RT_OPT_TAIL_SAMPLES):
        renderer.set_option(opt, -1)
    w, h, spp = 1920, 1080, 1000
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    renderer.setScene(objs, n)
    renderer.accumClear()
    renderer.render(u, None, 0, 0, spp)
    renderer.resolve(spp)
    img = renderer.readTexture()
    st = renderer.lastStats()
    assert st.scene_path == 3 and st.samples == w * h * spp
    ranges = st.chunk_ranges(0, spp)
    assert ranges[0][0] == 0 and ranges[-1][1] == spp and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    rng = np.random.RandomState(11)
    for _ in range(32):
        x, y = int(rng.randint(0, w)), int(rng.randint(0, h))
        acc = np.zeros(4, np.float32)
        for b, e in ranges:
            part = np.zeros(4, np.float32)
            for k in range(b, e):
                c = (C.c_float * 4)()
                if oracle.lib.rto_sample(C.byref(u), objs, x, y, k, c, None):
                    part = part + np.array(list(c), np.float32)
            acc = acc + part
        want = oracle.resolve(acc.reshape(1, 1, 4), spp)[0, 0]
        assert np.array_equal(bits(img[y, x]), bits(want)), (x, y)


def oracle_spot_pixels(oracle, u, objs, pixels, ranges, workers=8):
    """Linear sums of the given pixels as the kernel forms them: each range summed sample by sample, the range sums
    added in order.  The oracle's per-sample call releases the GIL, so the pixels are spread over a few threads."""
    from concurrent.futures import ThreadPoolExecutor

    def one(xy):
        x, y = xy
        acc = np.zeros(4, np.float32)
        for b, e in ranges:
            part = np.zeros(4, np.float32)
            for k in range(b, e):
                c = (C.c_float * 4)()
                if oracle.lib.rto_sample(C.byref(u), objs, x, y, k, c, None):
                    part = part + np.array(list(c), np.float32)
            acc = acc + part
        return acc

    with ThreadPoolExecutor(workers) as pool:
        return list(pool.map(one, pixels))


def test_100k_sphere_configuration_spot_pixels(rt, oracle, renderer):
    """The 100 000-sphere configuration at its own size -- 1920x1080, 64 spp, depth 50, default options (lifted tiles
    read in place by the waves whose rays can touch them, matrix-core filter at half width, spheres visited in spatial
    order) -- at 24 scattered pixels against the
    oracle's brute-force loop over all objects (Shaders.metal:13-32), bit for bit."""
    for opt in (rt.RT_OPT_SCENE_FETCH, rt.RT_OPT_FILTER, rt.RT_OPT_BLOCKS_PER_CU):
        renderer.set_option(opt, 0)
    renderer.set_option(rt.RT_OPT_NARROW, 1)
    for opt in (rt.RT_OPT_CHUNK, rt.RT_OPT_TAIL_CHUNK, rt.RT_OPT_TAIL_SAMPLES):
        renderer.set_option(opt, -1)
    w, h, spp = 1920, 1080, 64
    objs, n = rt.scene("random_spheres")
    assert n >= 100000
    u = rt.config_uniforms("random_spheres", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    renderer.setScene(objs, n)
    renderer.accumClear()
    renderer.render(u, None, 0, 0, spp)
    renderer.resolve(spp)
    img = renderer.readTexture()
    st = renderer.lastStats()
    assert st.scene_path == 5 and st.samples == w * h * spp and st.tests == st.segments * n
    ranges = st.chunk_ranges(0, spp)
    assert ranges[0][0] == 0 and ranges[-1][1] == spp and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    rng = np.random.RandomState(5)
    pixels = [(int(rng.randint(0, w)), int(rng.randint(0, h))) for _ in range(24)]
    sums = oracle_spot_pixels(oracle, u, objs, pixels, ranges)
    seen = set()
    for (x, y), acc in zip(pixels, sums):
        want = oracle.resolve(acc.reshape(1, 1, 4), spp)[0, 0]
        assert np.array_equal(bits(img[y, x]), bits(want)), (x, y)
        seen.add(tuple(want))
    assert len(seen) > 12  # the pixels see different things: not all sky


def test_nested_shell_configuration_spot_pixels(rt, oracle, renderer):
    """The 400 nested glass shells at their own size -- 1920x1080, 64 spp, depth 100, default options (staged matrix
    kind, the kernels that strike candidates before the exact pass) -- at 24 pixels, most of them on the ball, against
    the oracle's brute-force loop, bit for bit."""
    for opt in (rt.RT_OPT_SCENE_FETCH, rt.RT_OPT_FILTER, rt.RT_OPT_BLOCKS_PER_CU, rt.RT_OPT_RESOLVE):
        renderer.set_option(opt, 0)
    for opt in (rt.RT_OPT_CHUNK, rt.RT_OPT_TAIL_CHUNK, rt.RT_OPT_TAIL_SAMPLES):
        renderer.set_option(opt, -1)
    w, h, spp = 1920, 1080, 64
    objs, n = rt.scene("nested_glass")
    u = rt.config_uniforms("nested_glass", w, h, spp, 100, n)
    renderer.setupTexture(w, h)
    renderer.setScene(objs, n)
    renderer.accumClear()
    renderer.render(u, None, 0, 0, spp)
    renderer.resolve(spp)
    img = renderer.readTexture()
    st = renderer.lastStats()
    assert st.scene_path == 3 and st.samples == w * h * spp and st.tests == st.segments * n
    assert st.segments > 15 * st.samples   # the long paths of the configuration are there
    ranges = st.chunk_ranges(0, spp)
    assert ranges[0][0] == 0 and ranges[-1][1] == spp and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    rng = np.random.RandomState(23)
    pixels = [(int(rng.randint(0, w)), int(rng.randint(0, h))) for _ in range(6)]
    pixels += [(int(rng.randint(w // 2 - 260, w // 2 + 260)), int(rng.randint(h // 2 - 260, h // 2 + 260))) for _ in range(18)]
    sums = oracle_spot_pixels(oracle, u, objs, pixels, ranges)
    seen = set()
    for (x, y), acc in zip(pixels, sums):
        want = oracle.resolve(acc.reshape(1, 1, 4), spp)[0, 0]
        assert np.array_equal(bits(img[y, x]), bits(want)), (x, y)
        seen.add(tuple(want))
    assert len(seen) > 12


def test_4k_cover_per_rank_share_spot_pixels(rt, oracle, renderer):
    """The 3840x2160 x 1000 spp cover configuration as one rank of eight traces it: all pixels, the sample indices
    ray_range(rank, 8, 1000), default options, for the first and the last rank; 16 scattered pixels' linear sums each
    against the oracle, bit for bit (the seed expression of Shaders.metal:121-126 exceeds 2^24 here)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rt_partition", os.path.join(HERE, "..", "ray-tracing_amd", "partition.py"))
    part = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(part)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for opt in (rt.RT_OPT_SCENE_FETCH, rt.RT_OPT_FILTER, rt.RT_OPT_BLOCKS_PER_CU):
        renderer.set_option(opt, 0)
    for opt in (rt.RT_OPT_CHUNK, rt.RT_OPT_TAIL_CHUNK, rt.RT_OPT_TAIL_SAMPLES):
        renderer.set_option(opt, -1)
    w, h, spp = 3840, 2160, 1000
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    assert u.defocusAngle > 0
    renderer.setupTexture(w, h)
    renderer.setScene(objs, n)
    rng = np.random.RandomState(17)
    for rank in (0, 7):
        begin, end = part.ray_range(rank, 8, spp)
        assert end - begin == 125
        renderer.accumClear()
        renderer.render(u, None, 0, begin, end)
        st = renderer.lastStats()
        assert st.scene_path == 3 and st.samples == w * h * 125
        sums = np.empty((h, w, 4), np.float32)
        renderer.synchronize()
        assert hip.hipMemcpy(sums.ctypes.data, renderer.accumPtr(), sums.nbytes, 2) == 0
        ranges = st.chunk_ranges(begin, end)
        assert ranges[0][0] == begin and ranges[-1][1] == end
        pixels = [(int(rng.randint(0, w)), int(rng.randint(0, h))) for _ in range(14)] + [(w - 1, h - 1), (0, h - 1)]
        for (x, y), acc in zip(pixels, oracle_spot_pixels(oracle, u, objs, pixels, ranges)):
            assert np.array_equal(bits(sums[y, x]), bits(acc)), (rank, x, y)
    renderer.setupTexture(64, 64)  # give the 4K buffers back


def test_bench_rccl_calls_on_one_rank(rt, tmp_path):
    """The RCCL leg of bench.py -- process group on the nccl backend with the rank's device, the communicator's first
    all-reduce, the reduce of the accumulation tensor on the script's stream, barrier, MAX of the times, tear-down --
    executed for real as a one-rank job (RT_BENCH_FORCE_DIST=1), which is all a box with one GPU allows; the reduced
    sums must equal a plain render of the same frame."""
    import subprocess
    import json
    root = os.path.dirname(HERE)
    w, h, spp = 320, 180, 16
    dump = str(tmp_path / "one_rank.npz")
    port = 29700 + (os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--width", str(w), "--height", str(h), "--spp", str(spp), "--backend", "nccl", "--dump", dump,
           "--no-cpu-baseline", "--valu-steps", "0"]
    env = dict(os.environ, OMP_NUM_THREADS="2", RT_BENCH_FORCE_DIST="1")
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=170)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    got = np.load(dump)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    r = rt.MetalRenderer(device=0)
    try:
        objs, n = rt.scene("cover")
        u = rt.config_uniforms("cover", w, h, spp, 50, n)
        r.setupTexture(w, h)
        r.setScene(objs, n)
        r.accumClear()
        r.render(u, None, 0, 0, spp)
        r.synchronize()
        want = np.empty((h, w, 4), np.float32)
        assert hip.hipMemcpy(want.ctypes.data, r.accumPtr(), want.nbytes, 2) == 0
    finally:
        r.close()
    assert want.any() and np.array_equal(bits(got["accum"]), bits(want))


def test_bench_two_ranks_real_path(rt, tmp_path):
    """bench.py's own multi-rank step -- per-rank sample range, trace, reduce to rank 0, resolve -- run as two
    processes (one GPU each when two are visible, else both on GPU 0 with the gloo backend), against the same two
    ranges traced here by one process and added in the same order: rank 0's reduced sums and image, bit for bit."""
    import subprocess
    import torch
    root = os.path.dirname(HERE)
    w, h, spp = 320, 180, 24
    dump = str(tmp_path / "two_ranks.npz")
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    port = 29600 + (os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--width", str(w), "--height", str(h), "--spp", str(spp), "--backend", backend, "--dump", dump]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=170)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    import json
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["samples_per_rank"] == spp // 2 and line["value"] > 0
    got = np.load(dump)

    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    r = rt.MetalRenderer(device=0)
    try:
        objs, n = rt.scene("cover")
        u = rt.config_uniforms("cover", w, h, spp, 50, n)
        r.setupTexture(w, h)
        r.setScene(objs, n)
        parts = []
        for begin, end in ((0, spp // 2), (spp // 2, spp)):
            r.accumClear()
            r.render(u, None, 0, begin, end)
            r.synchronize()
            a = np.empty((h, w, 4), np.float32)
            assert hip.hipMemcpy(a.ctypes.data, r.accumPtr(), a.nbytes, 2) == 0
            parts.append(a)
        st = r.lastStats()
        assert (st.chunk, st.chunk_count) == (int(got["chunk"]), int(got["chunk_count"]))
    finally:
        r.close()
    want = parts[0] + parts[1]
    assert want.any() and np.array_equal(bits(got["accum"]), bits(want))
    from conftest import Oracle
    assert np.array_equal(bits(got["image"]), bits(Oracle().resolve(want, spp)))


def test_4k_seed_rounding_spot_pixels(rt, oracle, renderer):
    """At 3840x2160 the integer seed expression exceeds 2^24 and is rounded on conversion to float."""
    strict(renderer, rt)
    w, h, spp = 3840, 2160, 2
    objs, n = rt.scene("three_spheres")
    u = rt.config_uniforms("three_spheres", w, h, spp, 50, n)
    u.currentSample = 999
    renderer.setupTexture(w, h)
    renderer.setScene(objs, n)
    renderer.accumClear()
    renderer.render(u, None, 0, 0, spp)
    renderer.resolve(spp)
    renderer.synchronize()
    img = renderer.readTexture()
    for x, y in [(3839, 2159), (3838, 2159), (0, 2159), (3839, 0), (2047, 1301), (1919, 1079)]:
        acc = np.zeros(4, np.float32)
        for k in range(spp):
            c = (C.c_float * 4)()
            if oracle.lib.rto_sample(C.byref(u), objs, x, y, k, c, None):
                acc = acc + np.array(list(c), np.float32)
        assert np.array_equal(bits(img[y, x]), bits(oracle.resolve(acc.reshape(1, 1, 4), spp)[0, 0])), (x, y)


def test_progressive_tiles_match_reference_semantics(rt, oracle, renderer):
    """The reference's own flow: 64x64 tiles, several passes, gamma-encoded running mean, and the yellow
    pixels its 8x8 threadgroups paint for edge tiles whose size is not a multiple of 8."""
    strict(renderer, rt)
    w, h, passes = 100, 70, 3
    objs, n = rt.scene("cornell")
    renderer.setupTexture(w, h)
    rt.initRayTracing(renderer, w, h)
    renderer.uniforms.sampleCount = 4
    renderer.uniforms.maxRayDepth = 6
    renderer.objects = [objs[i] for i in range(n)]
    renderer.maxIterations = passes
    renderer.startProgressiveRender()
    steps = 0
    while renderer.renderTile():
        steps += 1
    assert steps == passes * (len(renderer.tiles) + 1) - 1 and len(renderer.tiles) == 4
    got = renderer.readTexture()
    tex = np.zeros((h, w, 4), np.float32)
    tex[..., 3] = 1
    for p in range(passes):
        for t in renderer.tiles:
            u = rt.Uniforms.from_buffer_copy(renderer.uniforms)
            u.currentSample, u.totalSamples, u.objCount = p, passes, n
            u.tileX, u.tileY, u.tileWidth, u.tileHeight = t.x, t.y, t.width, t.height
            oracle.image(u, objs, w, h, tex=tex, reference_grid=1)
    assert np.array_equal(bits(got), bits(tex))
    yellow = np.array([1, 1, 0, 1], np.float32)
    assert (got[..., :] == yellow).all(axis=-1).any()  # 36x6 edge tiles: threads 36..39 / 6..7 paint yellow
    # present pass
    assert np.array_equal(renderer.present(), oracle.present(got, w, h))
    assert np.array_equal(renderer.present(2 * w, 2 * h), oracle.present(got, 2 * w, 2 * h))


def test_current_sample_above_10000_paints_yellow(rt, oracle, renderer):
    strict(renderer, rt)
    w, h = 64, 48
    objs, n = rt.scene("three_spheres")
    renderer.setupTexture(w, h)
    u = rt.config_uniforms("three_spheres", w, h, 2, 5, n)
    u.currentSample = 10001
    u.tileX, u.tileY, u.tileWidth, u.tileHeight = 16, 8, 20, 12
    rc = rt.amd_lib().rt_render_tile(renderer._ctx, C.byref(u), objs, n)
    assert rc == 1
    got = renderer.readTexture()
    want = np.zeros((h, w, 4), np.float32)
    want[..., 3] = 1
    oracle.image(u, objs, w, h, tex=want, reference_grid=1)
    assert np.array_equal(bits(got), bits(want))
    assert (got[0:16, 0:24] == np.array([1, 1, 0, 1], np.float32)).all()


def test_degenerate_inputs(rt, oracle, renderer):
    strict(renderer, rt)
    w, h = 40, 24
    objs, n = rt.scene("three_spheres")
    renderer.setupTexture(w, h)
    # empty scene: every ray misses -> sky
    u = rt.config_uniforms("three_spheres", w, h, 3, 5, 0)
    img = renderer.renderImage(u, objs, n)
    assert np.array_equal(bits(img), bits(oracle.image(u, objs, w, h)))
    # zero samples per pass: 0/0 -> NaN -> scrubbed to (0,0,0,1), on both the first and a later pass
    u = rt.config_uniforms("three_spheres", w, h, 0, 5, n)
    tex = np.zeros((h, w, 4), np.float32)
    tex[..., 3] = 1
    for p in (0, 1):
        u.currentSample = p
        assert rt.amd_lib().rt_render_tile(renderer._ctx, C.byref(u), objs, n) == 0
        oracle.image(u, objs, w, h, tex=tex)
        assert np.array_equal(bits(renderer.readTexture()), bits(tex))
    # depth 0: color_ray's loop never runs -> black with alpha 0 -> stored as (0,0,0,0)
    u = rt.config_uniforms("three_spheres", w, h, 2, 0, n)
    img = renderer.renderImage(u, objs, n)
    assert np.array_equal(bits(img), bits(oracle.image(u, objs, w, h)))
    # unknown material type absorbs; unknown object type never hits
    objs2 = (rt.Object * n)(*[objs[i] for i in range(n)])
    objs2[1].mat.type = 7
    objs2[4].type = 5
    u = rt.config_uniforms("three_spheres", w, h, 4, 20, n)
    img = renderer.renderImage(u, objs2, n)
    assert np.array_equal(bits(img), bits(oracle.image(u, objs2, w, h)))


def test_contexts_give_their_device_memory_back(rt):
    """setupMetal / teardown in a loop (ContentView.swift:328-356 runs once per app; a host that re-creates renderers must
    not leak): every buffer a context allocated -- target, accumulation, planes, queues, scene in all its forms, the
    tiled kind's tiles, staging -- is freed by rt_destroy, so free device memory is the same after every round."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        free, total = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    seen = []
    for rep in range(4):
        r = rt.MetalRenderer(device=0)
        try:
            for name, kw in (("cover", {}), ("random_spheres", {"n": 3000}), ("cornell", {})):
                objs, n = rt.scene(name, **kw)
                w, h = 96, 54
                u = rt.config_uniforms(name, w, h, 2, 8, n)
                r.setupTexture(w, h)
                r.renderImage(u, objs, n)
                r.firstHit(u, objs, n)
                r.present(w, h)
                u.tileWidth, u.tileHeight = 64, 54
                assert r._lib.rt_render_tile(r._ctx, C.byref(u), objs, n) in (0, 1)
        finally:
            r.close()
        seen.append(free_bytes())
    assert seen[1] == seen[2] == seen[3], seen  # (the first round also pays for HIP's own one-time allocations)


def test_error_codes(rt):
    r = rt.MetalRenderer(0)
    objs, n = rt.scene("three_spheres")
    u = rt.config_uniforms("three_spheres", 32, 32, 1, 5, n)
    lib = rt.amd_lib()
    assert lib.rt_render_tile(r._ctx, C.byref(u), objs, n) == -4           # no target yet
    r.setupTexture(32, 32)
    assert lib.rt_render(r._ctx, C.byref(u), None, 0, 0, 1, None) == -5      # no resident scene
    u.tileWidth = 0
    assert lib.rt_render_tile(r._ctx, C.byref(u), objs, n) == -1
    u.tileWidth = 64
    assert lib.rt_render(r._ctx, C.byref(u), objs, n, 0, 1, None) == -1      # batch tile must lie inside
    assert lib.rt_set_target(r._ctx, 0, 5) == -1
    assert lib.rt_set_option(r._ctx, 99, 0) == -1
    assert b"rt" in lib.rt_last_error(r._ctx)
    r.close()


def test_sphere_filter_never_misses_a_reference_candidate(rt, renderer):
    """The fused-multiply-add filter in front of the exact hit test may over-report, never under-report:
    1.3e8 pseudo-random grazing ray/sphere pairs over 20 octaves of scale, three seeds."""
    for seed in (1, 2, 3):
        r = renderer.selftestFilter(seed=seed, blocks=2048, pairs_per_thread=256)
        assert r["pairs"] == 2048 * 256 * 256
        assert r["reference"] > r["pairs"] // 4            # the generator really produces grazing hits
        assert r["general_missed"] == 0 and r["compact_missed"] == 0, r
        assert r["general"] >= r["reference"] and r["compact"] >= r["reference"]
        # many generated pairs graze within the filters' 2^-17 / 2^-16 slack on purpose, so both over-report here;
        # the general filter must still be the tighter of the two
        assert r["general"] <= r["compact"] < 2 * r["reference"]


def test_matrix_filter_never_misses_a_reference_candidate(rt, renderer):
    """The same guarantee for the matrix-core filter: scene frames over 24 octaves of scale with and without a
    translation, 64 x 64 ray/sphere pairs per round with the diagonal pairs grazing.  No miss at the slack the
    library renders with (2^-16), nor at a sixteenth of it."""
    for seed, slack in ((1, -16), (2, -16), (3, -20)):
        r = renderer.selftestMatrixFilter(seed=seed, blocks=1024, rounds=48, slack_log2=slack)
        assert r["pairs"] > 1024 * 4 * 48 * 64 * 64 // 4     # most pairs are inside the filter's range
        assert r["reference"] > r["pairs"] // 200            # the generator really produces hits
        assert r["missed"] == 0, r
        assert r["reference"] <= r["matrix"] < 2 * r["reference"]
        assert r["unflagged"] == 0
    # the same seed twice gives the same counts, reference side included (a build that lets vector instructions
    # mingle with the matrix instructions drifts here by a few hundred pairs from run to run)
    a = renderer.selftestMatrixFilter(seed=1, blocks=1024, rounds=48, slack_log2=-16)
    b = renderer.selftestMatrixFilter(seed=1, blocks=1024, rounds=48, slack_log2=-16)
    assert a == b


def test_tiled_filter_never_misses_a_reference_candidate(rt, renderer):
    """The form the tiled kind lifts its rays in -- frame local to the spheres, ray origins up to 2^20 radii away,
    the ray stood on its point next to the frame's centre, a slack term for what still grows with distance (the
    reference's own rounding and the displacement of the computed line).  No miss at the slack the library renders
    with, nor at a quarter of it (the first handful in 10^9 pairs appears at an eighth); the margin is thinner than the
    staged form's by design, because at these distances the reference's rounding error is the larger part of the
    slack."""
    for seed, slack in ((1, -16), (2, -16), (3, -17), (4, -18)):
        r = renderer.selftestTiledFilter(seed=seed, blocks=1024, rounds=48, slack_log2=slack)
        assert r["pairs"] > 1024 * 4 * 48 * 64 * 64 // 4
        assert r["reference"] > r["pairs"] // 200
        assert r["missed"] == 0, r
        assert r["unflagged"] == 0
    a = renderer.selftestTiledFilter(seed=1, blocks=1024, rounds=48, slack_log2=-16)
    b = renderer.selftestTiledFilter(seed=1, blocks=1024, rounds=48, slack_log2=-16)
    assert a == b


@pytest.mark.parametrize("name,kw,w,h,spp,depth,shift", [
    ("three_spheres", {}, 160, 90, 8, 50, 0.0),                       # one block, ground sphere outside the FP16 range
    ("nested_glass", {}, 96, 54, 4, 100, 0.0),                        # concentric shells, 100-bounce paths
    ("random_spheres", dict(n=700, extent=20.0), 96, 54, 4, 50, 0.0),  # 11 blocks, two filter rounds
    ("random_spheres", dict(n=300, extent=20.0), 96, 54, 4, 50, 1000.0)])  # far from the origin: the frame translation
def test_matrix_filter_scenes_bit_exact(rt, oracle, renderer, name, kw, w, h, spp, depth, shift):
    """RT_OPT_FILTER = 3 puts every staged scene the lifted form covers on the matrix cores."""
    strict(renderer, rt)
    objs, n = rt.scene(name, **kw)
    u = rt.config_uniforms(name, w, h, spp, depth, n)
    if shift:
        for i in range(n):
            objs[i].s.center.x += shift
            objs[i].s.center.z -= shift
        for v in (u.cameraCenter, u.pixelOrigin):
            v.x += shift
            v.z -= shift
    renderer.setupTexture(w, h)
    renderer.set_option(rt.RT_OPT_FILTER, 3)
    try:
        img = renderer.renderImage(u, objs, n)
        st = renderer.lastStats()
    finally:
        renderer.set_option(rt.RT_OPT_FILTER, 0)
    assert st.scene_path == 3
    assert np.array_equal(bits(img), bits(oracle.image(u, objs, w, h)))


def test_matrix_and_vector_filters_agree_at_scale(rt, renderer):
    """Larger than the oracle can check in seconds: 1200x675 at 96 spp (1.5e9 closest-hit searches), matrix-core
    filter twice and vector filters once -- all three images must be the same bits."""
    strict(renderer, rt, chunk=16)
    w, h, spp = 1200, 675, 96
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    renderer.setScene(objs, n)
    imgs = []
    for filt in (0, 0, 2):
        renderer.set_option(rt.RT_OPT_FILTER, filt)
        try:
            renderer.accumClear()
            renderer.render(u, None, 0, 0, spp)
            renderer.resolve(spp)
            imgs.append(renderer.readTexture().copy())
            assert renderer.lastStats().scene_path == (3 if filt == 0 else 0)
        finally:
            renderer.set_option(rt.RT_OPT_FILTER, 0)
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))
    assert np.array_equal(bits(imgs[0]), bits(imgs[2]))


def test_matrix_filter_awkward_spheres_bit_exact(rt, oracle, renderer):
    """Spheres the lifted FP16 form cannot represent or that stress its rules, mixed into the cover scene: an exact
    duplicate of an earlier sphere (equal distances: the lower index must win), radius 0, negative radius, NaN
    radius, NaN centre, a huge enclosing sphere, a tiny sphere, one centred on the camera, one far outside the
    frame, and an unknown object type.  Every filter choice must give the oracle's bits."""
    strict(renderer, rt)
    w, h, spp = 120, 68, 4
    base, n0 = rt.scene("cover")
    extra = []
    dup = rt.Object.from_buffer_copy(base[40])
    extra.append(dup)
    glass = rt.SwiftMaterial.dielectric(1.5)
    grey = rt.SwiftMaterial.lambertian(0.5, 0.5, 0.5)
    extra.append(rt.makeSphere((1.0, 0.2, 1.0), 0.0, grey))
    extra.append(rt.makeSphere((2.0, 0.5, -1.0), -0.5, glass))
    extra.append(rt.makeSphere((3.0, 0.5, 1.0), float("nan"), grey))
    extra.append(rt.makeSphere((float("nan"), 0.5, 1.0), 0.3, grey))
    extra.append(rt.makeSphere((0.0, 0.0, 0.0), 400.0, glass))
    extra.append(rt.makeSphere((5.0, 0.3, 1.5), 1e-4, grey))
    extra.append(rt.makeSphere((13.0, 2.0, 3.0), 0.05, glass))
    extra.append(rt.makeSphere((5000.0, 100.0, -3000.0), 50.0, grey))
    odd = rt.makeSphere((-2.0, 0.4, 2.0), 0.4, grey)
    odd.type = 9
    extra.append(odd)
    n = n0 + len(extra)
    objs = (rt.Object * n)(*([base[i] for i in range(n0)] + extra))
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    want = oracle.image(u, objs, w, h)
    for filt, path in ((3, 3), (0, 3), (2, 0), (1, 0)):
        renderer.set_option(rt.RT_OPT_FILTER, filt)
        try:
            img = renderer.renderImage(u, objs, n)
            assert renderer.lastStats().scene_path == path
        finally:
            renderer.set_option(rt.RT_OPT_FILTER, 0)
        assert np.array_equal(bits(img), bits(want)), filt
    t4, ids = renderer.firstHit(u, objs, n)
    want4, want_ids = oracle.first_hit(u, objs, w, h)
    assert np.array_equal(ids, want_ids) and np.array_equal(bits(t4), bits(want4))


@pytest.mark.parametrize("seed,extent,offset", [(1, 1.0, 0.0), (2, 20.0, 0.0), (3, 300.0, 0.0), (4, 20.0, 1000.0),
                                                (5, 0.01, 0.0), (6, 3000.0, -7000.0)])
def test_random_scenes_matrix_equals_vector(rt, renderer, seed, extent, offset):
    """Random scenes over five decades of scale, at and far from the origin: the matrix-core filter and the
    vector filters must give the same bits (the vector filters are checked against the oracle elsewhere)."""
    strict(renderer, rt)
    rng = np.random.default_rng(seed)
    nsph = int(rng.integers(50, 400))
    objs = []
    for i in range(nsph):
        c = offset + extent * (rng.random(3) - 0.5)
        r = extent * 10.0 ** rng.uniform(-2.5, -1.0)
        kind = rng.random()
        if kind < 0.5:
            m = rt.SwiftMaterial.lambertian(*rng.random(3))
        elif kind < 0.75:
            m = rt.SwiftMaterial.metal(*(0.5 + 0.5 * rng.random(3)), 0.5 * rng.random())
        else:
            m = rt.SwiftMaterial.dielectric(1.5)
        objs.append(rt.makeSphere(tuple(float(v) for v in c), float(r), m))
    arr = rt.object_array(objs)
    w, h, spp = 96, 64, 4
    u = rt.config_uniforms("cover", w, h, spp, 30, nsph)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    centre = np.full(3, offset)
    rt.host_lib().rt_camera_setup(C.byref(u), f3(centre + extent * np.array([1.2, 0.4, 0.9])), f3(centre), f3([0, 1, 0]), 40.0,
                                  0.3, float(1.5 * extent), w, h)
    renderer.setupTexture(w, h)
    imgs = {}
    for filt in (3, 2):
        renderer.set_option(rt.RT_OPT_FILTER, filt)
        try:
            imgs[filt] = renderer.renderImage(u, arr, nsph)
            path = renderer.lastStats().scene_path
        finally:
            renderer.set_option(rt.RT_OPT_FILTER, 0)
        if filt == 3:
            assert path == 3
    assert np.array_equal(bits(imgs[3]), bits(imgs[2]))
    assert len(np.unique(bits(imgs[3]).reshape(-1, 4), axis=0)) > 50   # the camera really sees the spheres
    renderer.set_option(rt.RT_OPT_SCENE_FETCH, 3)   # the same scene as one lifted tile in its own frame, read in place or staged
    try:
        for narrow, path in ((1, 5), (0, 4)):
            renderer.set_option(rt.RT_OPT_NARROW, narrow)
            tiled = renderer.renderImage(u, arr, nsph)
            assert renderer.lastStats().scene_path == path
            assert np.array_equal(bits(tiled), bits(imgs[2])), narrow
    finally:
        renderer.set_option(rt.RT_OPT_SCENE_FETCH, 0)
        renderer.set_option(rt.RT_OPT_NARROW, 1)


def _layer_scene(rt, case):
    """Scenes for the half-width sphere filter: many small spheres in a layer, a few large ones and a ground sphere.
    Returns (objects, camera position, look-at, whether the half-width path must have run)."""
    rng = np.random.default_rng({"cover_like": 11, "wide_in_the_middle": 12, "camera_in_the_layer": 13, "far_camera": 14,
                                 "too_many_wide": 15, "two_rounds": 16, "thick_cloud": 17}[case])
    n_small = 700 if case == "two_rounds" else 300
    extent = 12.0

    def material():
        kind = rng.random()
        if kind < 0.6:
            return rt.SwiftMaterial.lambertian(*rng.random(3))
        if kind < 0.8:
            return rt.SwiftMaterial.metal(*(0.5 + 0.5 * rng.random(3)), 0.3 * rng.random())
        return rt.SwiftMaterial.dielectric(1.5)

    small = []
    for i in range(n_small):
        r = float(rng.uniform(0.08, 0.3))
        y = r if case != "thick_cloud" else float(rng.uniform(0.0, 6.0))
        small.append(rt.makeSphere((float(rng.uniform(-extent, extent)), y, float(rng.uniform(-extent, extent))), r, material()))
    ground = rt.makeSphere((0.0, -1000.0, 0.0), 1000.0, rt.SwiftMaterial.lambertian(0.5, 0.5, 0.5))
    large = [rt.makeSphere((0.0, 1.0, 0.0), 1.0, rt.SwiftMaterial.dielectric(1.5)),
             rt.makeSphere((-4.0, 1.0, 0.0), 1.0, rt.SwiftMaterial.lambertian(0.4, 0.2, 0.1)),
             rt.makeSphere((4.0, 1.0, 0.0), 1.0, rt.SwiftMaterial.metal(0.7, 0.6, 0.5, 0.0)),
             rt.makeSphere((0.0, 2.5, -5.0), 2.5, rt.SwiftMaterial.dielectric(1.3))]
    eye, look, narrow = (13.0, 2.0, 3.0), (0.0, 0.0, 0.0), True
    if case in ("cover_like", "thick_cloud"):
        objs = [ground] + small + large[:3]
        narrow = case == "cover_like"     # a cloud six units thick: most rays touch its box
    elif case == "wide_in_the_middle":    # the large spheres and the ground in three words in the middle of the slots
        objs = small[:70] + large[:2] + small[70:200] + [ground] + small[200:290] + large[2:] + small[290:]
    elif case == "camera_in_the_layer":   # rays that run along the top of the box, a camera between the small spheres
        objs = [ground] + small + large[:3]
        eye, look = (9.0, 0.35, 2.0), (-3.0, 0.3, -1.0)
    elif case == "far_camera":            # origins beyond what the box test speaks for, and beyond the lifted form's range
        objs = [ground] + small + large[:3]
        eye = (2600.0, 400.0, 600.0)
    elif case == "too_many_wide":         # nine spheres outside the box: the scene does without the half-width path
        more = [rt.makeSphere((float(6 * np.cos(a)), 1.5, float(6 * np.sin(a))), 1.5, material()) for a in np.arange(5) * 1.2]
        objs = [ground] + small + large + more
        narrow = False
    else:                                 # two resolve rounds of 512 slots, large spheres in both
        objs = small[:100] + large[:2] + small[100:600] + [ground] + large[2:] + small[600:]
    return objs, eye, look, narrow


@pytest.mark.parametrize("case", ["cover_like", "wide_in_the_middle", "camera_in_the_layer", "far_camera", "too_many_wide", "two_rounds",
                                  "thick_cloud"])
def test_half_width_filter_bit_exact(rt, oracle, renderer, case):
    """RT_OPT_NARROW: rays that cannot touch the box around the scene's small spheres are left out of the matrix-core
    filter, the others share one ray tile.  With it (the default) and without, the image is the oracle's."""
    strict(renderer, rt)
    objs, eye, look, narrow = _layer_scene(rt, case)
    n = len(objs)
    arr = rt.object_array(objs)
    w, h, spp = 96, 54, 4
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    rt.host_lib().rt_camera_setup(C.byref(u), f3(eye), f3(look), f3([0, 1, 0]), 25.0, 0.4, 10.0, w, h)
    renderer.setupTexture(w, h)
    want = oracle.image(u, arr, w, h)
    per_trip = {}
    try:
        for opt in (1, 0):
            renderer.set_option(rt.RT_OPT_NARROW, opt)
            img = renderer.renderImage(u, arr, n)
            st = renderer.lastStats()
            assert st.scene_path == 3
            per_trip[opt] = st.matrix_blocks / st.wave_trips
            assert np.array_equal(bits(img), bits(want)), (case, opt)
    finally:
        renderer.set_option(rt.RT_OPT_NARROW, 1)
    assert len(np.unique(bits(want).reshape(-1, 4), axis=0)) > 50
    if narrow and case != "far_camera":
        assert per_trip[1] < 0.8 * per_trip[0], per_trip   # the half-width path did run
    if not narrow and case == "too_many_wide":
        assert per_trip[1] == per_trip[0], per_trip


def test_half_width_filter_at_scale(rt, renderer):
    """Larger than the oracle checks in seconds: the cover scene at 1200x675 x 64 samples with and without the half-width
    path, and the wavefront kernels with it: the same bits (work items of 16 samples, so that all three add up alike)."""
    strict(renderer, rt, chunk=16)
    w, h, spp = 1200, 675, 64
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    renderer.setScene(objs, n)
    imgs, per_trip = [], []
    try:
        for opt, kernel in ((1, 0), (0, 0), (1, 1)):
            renderer.set_option(rt.RT_OPT_NARROW, opt)
            renderer.set_option(rt.RT_OPT_KERNEL, kernel)
            renderer.accumClear()
            renderer.render(u, None, 0, 0, spp)
            renderer.resolve(spp)
            imgs.append(renderer.readTexture().copy())
            st = renderer.lastStats()
            per_trip.append(st.matrix_blocks / st.wave_trips)
    finally:
        renderer.set_option(rt.RT_OPT_NARROW, 1)
        renderer.set_option(rt.RT_OPT_KERNEL, 0)
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))
    assert np.array_equal(bits(imgs[0]), bits(imgs[2]))
    assert per_trip[0] < 0.6 * per_trip[1] and per_trip[1] == 8.0, per_trip


@pytest.mark.parametrize("kind", ["cover", "cover_like", "wide_in_the_middle", "two_rounds", "thick_cloud"])
def test_box_test_leaves_out_no_ray_the_reference_would_hit_with(rt, renderer, kind):
    """rt_selftest_box on the scene's own spheres and box: rays from on or next to sphere surfaces and box faces, tangent,
    aimed at silhouettes or running along a face.  No ray the box test leaves out may have an accepted root on a sphere
    the box stands for -- with the padding the library renders with, and with a quarter of it."""
    if kind == "cover":
        arr, n = rt.scene(kind)
    else:
        objs = _layer_scene(rt, kind)[0]
        n = len(objs)
        arr = rt.object_array(objs)
    renderer.setScene(arr, n)
    for padding in (0, -2):
        tot = {"rays": 0, "left_out": 0, "pairs": 0, "roots": 0}
        for seed in (1, 2, 3):
            r = renderer.selftestBox(seed=seed, blocks=512, rays_per_thread=4, padding_log2=padding)
            assert r["wrong"] == 0, (padding, r)
            for k in tot:
                tot[k] += r[k]
        assert tot["rays"] > 1e6 and tot["roots"] > tot["rays"]        # the rays do hit spheres ...
        assert tot["left_out"] > 0.1 * tot["rays"], tot                # ... and the test is not vacuous
        assert tot["pairs"] > 100 * tot["left_out"]


@pytest.mark.parametrize("n,extent", [(5000, 20.0), (20000, 22.0)])
def test_tile_boxes_leave_out_no_ray_the_reference_would_hit_with(rt, renderer, n, extent):
    """rt_selftest_tile_boxes: the boxes of the tiles, of the tile groups and of the tiles' blocks of 64 slots of the kind
    that reads its tiles in place, rays from on a sphere to 2^12 radii away.  No (ray, tile) or (ray, block) pair the boxes
    leave out may hold an accepted root -- with the allowances the library renders with (a radius grown to
    sqrt(r^2 + 64u D^2)), and with half of them (32u; the bound on the reference's roundings asks for 17u, and with a
    quarter, 16u, rays from the ground sphere, thousands of radii away, do show a handful of cases in 10^10 pairs)."""
    arr, n = rt.scene("random_spheres", n=n, extent=extent)
    renderer.setScene(arr, n)
    for padding in (0, -1):
        tot = {"rays": 0, "left_out": 0, "pairs": 0, "roots": 0, "blocks_left_out": 0}
        for seed in (1, 2):
            r = renderer.selftestTileBoxes(seed=seed, blocks=256 if n < 10000 else 64, rays_per_thread=2, padding_log2=padding)
            assert r["wrong"] == 0, (padding, r)
            for k in tot:
                tot[k] += r[k]
        assert tot["rays"] > 3e4 and tot["roots"] > tot["rays"]
        assert tot["left_out"] > 3 * tot["rays"], tot      # most tiles are left out for most rays
        assert tot["blocks_left_out"] > 7 * tot["left_out"], tot   # ... and of the others, some of their blocks


def test_default_path_of_the_cover_scene_is_the_matrix_filter(rt, renderer):
    strict(renderer, rt)
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", 64, 36, 2, 50, n)
    renderer.setupTexture(64, 36)
    renderer.renderImage(u, objs, n)
    assert renderer.lastStats().scene_path == 3


@pytest.mark.parametrize("opt,val", [("RT_OPT_FILTER", 1), ("RT_OPT_FILTER", 2), ("RT_OPT_SCENE_FETCH", 1),
                                     ("RT_OPT_SCENE_FETCH", 2), ("RT_OPT_SCENE_FETCH", 3), ("RT_OPT_KERNEL", 1)])
def test_kernel_variants_bit_exact(rt, oracle, renderer, opt, val):
    """General-only filter, vector-only filters, scalar-cache fetch and streamed fetch give the same bits as the
    defaults."""
    strict(renderer, rt)
    w, h, spp = 240, 135, 16
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    want = oracle.image(u, objs, w, h)
    renderer.set_option(getattr(rt, opt), val)
    try:
        img = renderer.renderImage(u, objs, n)
    finally:
        renderer.set_option(getattr(rt, opt), 0)
    assert np.array_equal(bits(img), bits(want))


@pytest.mark.parametrize("name,w,h,spp,depth", [("nested_glass", 160, 90, 4, 100), ("cover", 240, 135, 16, 50)])
def test_exact_pass_forms_bit_exact(rt, oracle, renderer, name, w, h, spp, depth):
    """The exact hit test over the filter's candidates pooled over the wave (1), lane by lane with four candidates in
    flight (2) and chosen per wave (0: lane by lane where candidate words are dense, as in the nested shells) all give
    the oracle's bits."""
    strict(renderer, rt)
    objs, n = rt.scene(name)
    u = rt.config_uniforms(name, w, h, spp, depth, n)
    renderer.setupTexture(w, h)
    want = oracle.image(u, objs, w, h)
    for mode in (1, 2, 0):
        renderer.set_option(rt.RT_OPT_RESOLVE, mode)
        try:
            img = renderer.renderImage(u, objs, n)
        finally:
            renderer.set_option(rt.RT_OPT_RESOLVE, 0)
        assert np.array_equal(bits(img), bits(want)), mode


@pytest.mark.parametrize("name,w,h,spp,depth,chunk", [("cover", 240, 135, 16, 50, 0), ("cover", 320, 180, 40, 50, -1),
                                                        ("shells_in_cornell", 160, 120, 8, 20, 0)])
def test_scatter_regrouped_over_the_workgroup_bit_exact(rt, oracle, renderer, name, w, h, spp, depth, chunk):
    """RT_OPT_REGROUP 2: the material scatter of a path is evaluated by whichever thread of the workgroup its request is
    sorted to (one material per wave), and the waves of a workgroup take every trip together -- same operations on
    the same operands, so the oracle's bits; with more items than lanes, with chunked items, and with quads, emission,
    absorbing and unknown materials in the scene."""
    strict(renderer, rt, chunk=chunk)
    if name == "shells_in_cornell":
        base, nb = rt.scene("cornell")
        extra = [rt.makeSphere((200.0 + 40.0 * i, 120.0 + 30.0 * (i % 3), 200.0 + 25.0 * i), 45.0,
                               [rt.SwiftMaterial.metal(0.8, 0.8, 0.9, 0.3), rt.SwiftMaterial.dielectric(1.5),
                                rt.SwiftMaterial.lambertian(0.3, 0.7, 0.4)][i % 3]) for i in range(60)]
        odd = rt.makeSphere((278.0, 60.0, 278.0), 50.0, rt.SwiftMaterial.lambertian(0.5, 0.5, 0.5))
        odd.mat.type = 7   # no such material: absorbs (Lib.metal:203-213)
        objs_l = [base[i] for i in range(nb)] + extra + [odd]
        objs, n = rt.object_array(objs_l), len(objs_l)
        u = rt.config_uniforms("cornell", w, h, spp, depth, n)
    else:
        objs, n = rt.scene(name)
        u = rt.config_uniforms(name, w, h, spp, depth, n)
    renderer.setupTexture(w, h)
    want = oracle.image(u, objs, w, h) if chunk == 0 else None
    renderer.set_option(rt.RT_OPT_FILTER, 3)
    try:
        plain = renderer.renderImage(u, objs, n)
        assert renderer.lastStats().scene_path == 3
        renderer.set_option(rt.RT_OPT_REGROUP, 2)
        img = renderer.renderImage(u, objs, n)
        st = renderer.lastStats()
        assert st.scene_path == 3 and st.samples == w * h * spp
    finally:
        renderer.set_option(rt.RT_OPT_REGROUP, 0)
        renderer.set_option(rt.RT_OPT_FILTER, 0)
    assert np.array_equal(bits(img), bits(plain))
    if want is not None:
        assert np.array_equal(bits(img), bits(want))


def test_lane_by_lane_exact_pass_on_overlapping_spheres(rt, oracle, renderer):
    """Spheres that overlap pairwise without being duplicates (a ray through the circle in which two of them meet gets
    the same distance from both, possibly from numerators a few ulps apart: the lower object index must win,
    Shaders.metal:17-22), diffuse, so that bounced rays have unnormalised directions (Lib.metal:158-164) -- resolved
    lane by lane (RT_OPT_RESOLVE 2), pooled (1) and automatically, against the oracle, images and first hits."""
    rng = np.random.RandomState(41)
    objs_l = []
    for i in range(120):
        c = rng.uniform(-0.6, 0.6, 3) + np.array([0.0, 0.0, -1.0])   # around the point the three-sphere camera looks at
        rad = float(rng.uniform(0.12, 0.25))
        mat = rt.SwiftMaterial.lambertian(*rng.uniform(0.2, 0.9, 3)) if i % 3 else rt.SwiftMaterial.dielectric(1.5)
        objs_l.append(rt.makeSphere(tuple(float(v) for v in c), rad, mat))
        # a partner that cuts it in a circle facing the camera
        c2 = c + rng.uniform(-0.08, 0.08, 3) + float(rng.uniform(0.02, 0.08)) * np.array([-2.0, 2.0, 2.0]) / 12.0 ** 0.5
        objs_l.append(rt.makeSphere(tuple(float(v) for v in c2), rad * float(rng.uniform(0.8, 1.1)), rt.SwiftMaterial.metal(0.8, 0.7, 0.6, 0.1)))
    objs, n = rt.object_array(objs_l), len(objs_l)
    w, h, spp = 200, 150, 8
    u = rt.config_uniforms("three_spheres", w, h, spp, 12, n)
    renderer.setupTexture(w, h)
    want = oracle.image(u, objs, w, h)
    want4, want_ids = oracle.first_hit(u, objs, w, h)
    assert len(np.unique(want_ids)) > 40
    try:
        for mode in (2, 1, 0):
            strict(renderer, rt)
            renderer.set_option(rt.RT_OPT_FILTER, 3)   # the staged matrix kind whatever the sphere count
            renderer.set_option(rt.RT_OPT_RESOLVE, mode)
            img = renderer.renderImage(u, objs, n)
            assert np.array_equal(bits(img), bits(want)), mode
            t4, ids = renderer.firstHit(u, objs, n)
            assert np.array_equal(ids, want_ids) and np.array_equal(bits(t4), bits(want4)), mode
    finally:
        strict(renderer, rt)


def _dense_scene(rt, kind):
    """Scenes whose rays cross many spheres, built to stress the pruning of candidates by bounding balls and by single
    tests: (objects, camera from, camera at)."""
    rng = np.random.default_rng(11)
    glass, air = rt.SwiftMaterial.dielectric(1.5), rt.SwiftMaterial.dielectric(1.0 / 1.5)
    grey = rt.SwiftMaterial.lambertian(0.6, 0.6, 0.6)
    objs = [rt.makeSphere((0.0, -100.5, -1.0), 100.0, rt.SwiftMaterial.lambertian(0.8, 0.8, 0.0))]
    if kind == "drifting_shells":
        # nested but not concentric, every shell internally tangent to the one around it at one point, every 25th
        # sphere twice (equal distances: the lower slot must win), words of 32 not in radius order
        shells = []
        c, r = np.array([0.0, 0.0, -1.0]), 0.5
        for k in range(220):
            shells.append((c.copy(), r, glass if k % 2 == 0 else air))
            if k % 25 == 7:
                shells.append((c.copy(), r, grey))
            shrink = 0.002
            c = c + np.array([shrink, 0.0, 0.0])    # tangent at +x
            r = r - shrink
        order = np.arange(len(shells))
        for b in range(0, len(order) - 16, 48):
            rng.shuffle(order[b:b + 16])
        for i in order:
            objs.append(rt.makeSphere(tuple(float(v) for v in shells[i][0]), float(shells[i][1]), shells[i][2]))
        return objs, (0.0, 0.25, 1.0), (0.0, 0.0, -1.0)
    if kind == "clutter":
        # overlapping spheres without any order, seen from just outside
        mats = [glass, glass, rt.SwiftMaterial.metal(0.8, 0.8, 0.9, 0.05), rt.SwiftMaterial.lambertian(0.7, 0.3, 0.3)]
        for i in range(300):
            c = np.array([0.0, 0.3, -1.0]) + 1.5 * (rng.random(3) - 0.5)
            objs.append(rt.makeSphere(tuple(float(v) for v in c), float(rng.uniform(0.15, 0.5)), mats[i % 4]))
        return objs, (0.3, 0.5, 1.2), (0.0, 0.3, -1.0)
    if kind == "far_shells":
        # concentric shells far from the origin of the coordinates, seen from a distance
        centre = np.array([1000.0, 500.0, -2000.0])
        objs = [rt.makeSphere(tuple(centre + np.array([0.0, -5050.0, 0.0])), 5000.0, grey)]
        for k in range(300):
            objs.append(rt.makeSphere(tuple(float(v) for v in centre), 50.0 * (1.0 - k / 300.0), glass if k % 2 == 0 else air))
        return objs, tuple(centre + np.array([0.0, 40.0, 300.0])), tuple(centre)
    if kind == "shells_and_strays":
        # shells with a sphere around everything, a sphere that is not a number and stray small spheres in between
        objs.append(rt.makeSphere((0.0, 0.0, -1.0), 30.0, glass))
        for k in range(260):
            objs.append(rt.makeSphere((0.0, 0.0, -1.0), 0.5 * (1.0 - k / 260.0), glass if k % 2 == 0 else air))
            if k % 37 == 5:
                objs.append(rt.makeSphere((float("nan") if k == 5 else 0.3, 0.2, -0.7), 0.05, grey))
            if k % 41 == 9:
                objs.append(rt.makeSphere((0.0, 0.0, -1.0), 0.5 * (1.0 - k / 260.0), grey))   # coincides with a shell
        return objs, (0.0, 0.25, 1.0), (0.0, 0.0, -1.0)
    if kind == "rising_shells":
        # shells about one centre with radii that rise along the slots, every 30th twice (equal roots: the lower slot
        # must win), a second set about another centre with falling radii beginning in the middle of a word, the ground
        # last: runs of both directions, a word that two runs share, an exempt sphere in the last word
        objs = []
        for k in range(200):
            objs.append(rt.makeSphere((0.0, 0.0, -1.0), 0.5 * (k + 1) / 200.0, glass if k % 2 == 0 else air))
            if k % 30 == 11:
                objs.append(rt.makeSphere((0.0, 0.0, -1.0), 0.5 * (k + 1) / 200.0, grey))
        for k in range(120):
            objs.append(rt.makeSphere((0.9, 0.1, -1.3), 0.3 * (1.0 - k / 120.0), glass if k % 2 == 0 else air))
        objs.append(rt.makeSphere((0.0, -100.5, -1.0), 100.0, rt.SwiftMaterial.lambertian(0.8, 0.8, 0.0)))
        return objs, (0.4, 0.25, 1.0), (0.3, 0.0, -1.1)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["drifting_shells", "clutter", "far_shells", "shells_and_strays", "rising_shells"])
def test_pruned_exact_pass_on_dense_scenes(rt, oracle, renderer, kind):
    """Kernels for scenes whose rays cross many spheres strike candidates that cannot win before the exact pass (whole
    words and granules by their bounding balls, single spheres by a test with margins).  Their images must have the
    bits of the pooled exact pass, which resolves every candidate, and of the oracle."""
    strict(renderer, rt)
    objs, cam_from, cam_at = _dense_scene(rt, kind)
    n = len(objs)
    arr = rt.object_array(objs)
    w, h, spp = 96, 54, 4
    u = rt.config_uniforms("nested_glass", w, h, spp, 60, n)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    rt.host_lib().rt_camera_setup(C.byref(u), f3(cam_from), f3(cam_at), f3([0, 1, 0]), 40.0, 0.0, 1.0, w, h)
    renderer.setupTexture(w, h)
    imgs = {}
    for mode in (1, 2, 0):
        renderer.set_option(rt.RT_OPT_RESOLVE, mode)
        try:
            imgs[mode] = renderer.renderImage(u, arr, n).copy()
        finally:
            renderer.set_option(rt.RT_OPT_RESOLVE, 0)
    assert len(np.unique(bits(imgs[1]).reshape(-1, 4), axis=0)) > 50   # the camera really sees the spheres
    assert np.array_equal(bits(imgs[2]), bits(imgs[1]))
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))
    assert np.array_equal(bits(imgs[1]), bits(oracle.image(u, arr, w, h)))


@pytest.mark.parametrize("kind", ["nested_glass", "drifting_shells", "clutter", "far_shells", "shells_and_strays", "rising_shells", "cover"])
def test_prune_strikes_nothing_the_reference_would_take(rt, renderer, kind):
    """rt_selftest_prune on the scene's own spheres and bounding balls: rays from on or next to surfaces, tangent or
    aimed at silhouettes, with bounds equal to other spheres' roots or a few ulps off them, every sphere against the
    single-sphere test and the balls of its word and granule.  Nothing may be struck whose reference root is not above
    the bound."""
    if kind in ("nested_glass", "cover"):
        arr, n = rt.scene(kind)
    else:
        objs, _, _ = _dense_scene(rt, kind)
        n = len(objs)
        arr = rt.object_array(objs)
    renderer.setScene(arr, n)
    tot = {"pairs": 0, "single": 0, "balls": 0, "reference": 0}
    for seed in (1, 2, 3):
        r = renderer.selftestPrune(seed=seed, blocks=512, rays_per_thread=4)
        assert r["single_wrong"] == 0 and r["balls_wrong"] == 0, r
        for k in tot:
            tot[k] += r[k]
    assert tot["pairs"] > 1e8 and tot["reference"] > 0.002 * tot["pairs"]
    assert tot["single"] > 0.2 * tot["pairs"]                      # the test is not vacuous ...
    if kind not in ("clutter", "cover"):
        assert tot["balls"] > 0.2 * tot["pairs"]                   # ... and among shells the balls do most of the work


@pytest.mark.parametrize("shells,path", [(1000, 3), (1500, None)])
def test_pruned_exact_pass_over_several_rounds(rt, oracle, renderer, shells, path):
    """1000 and 1500 nested shells: the candidate words of a search come in rounds of 512 slots, the bound carries
    over from round to round and the last hit slot may lie in another round than the one being pruned (staged
    matrix kind; beyond its 1024 spheres whatever kind the library picks, and the LDS-staged vector kind on request)."""
    strict(renderer, rt)
    objs, n = rt.scene("nested_glass", shells=shells)
    w, h, spp = 96, 54, 2
    u = rt.config_uniforms("nested_glass", w, h, spp, 60, n)
    renderer.setupTexture(w, h)
    want = oracle.image(u, objs, w, h)
    for mode in (0, 2):
        renderer.set_option(rt.RT_OPT_RESOLVE, mode)
        try:
            img = renderer.renderImage(u, objs, n)
            if path is not None:
                assert renderer.lastStats().scene_path == path
        finally:
            renderer.set_option(rt.RT_OPT_RESOLVE, 0)
        assert np.array_equal(bits(img), bits(want)), mode
    if shells == 1500:
        renderer.set_option(rt.RT_OPT_FILTER, 2)   # vector filter, spheres staged in LDS in object order
        renderer.set_option(rt.RT_OPT_RESOLVE, 2)
        try:
            img = renderer.renderImage(u, objs, n)
            assert renderer.lastStats().scene_path == 0
        finally:
            strict(renderer, rt)
        assert np.array_equal(bits(img), bits(want))


@pytest.mark.parametrize("case", ["three_spheres", "cornell", "odd_600", "scalar_cache_1500", "streamed_600", "lds_vector_600"])
def test_pruned_exact_pass_on_ordinary_scenes(rt, oracle, renderer, case):
    """RT_OPT_RESOLVE 2 puts every round of every scene through the bound, the bounding balls and the single-sphere prune
    (scenes the automatic choice would never send there): a handful of objects, quads, spheres that are not numbers or
    have radius 0, a negative radius, duplicates, an enclosing sphere; every kind that keeps the spheres in object order
    (staged matrix, LDS-staged vector, scalar cache, per-wave streamed tiles)."""
    fetch = {"scalar_cache_1500": 1, "streamed_600": 2}.get(case, 0)
    strict(renderer, rt, fetch=fetch)
    w, h, spp, depth = 96, 54, 4, 30
    if case in ("three_spheres", "cornell"):
        objs, n = rt.scene(case)
        u = rt.config_uniforms(case, w, h, spp, depth, n)
    else:
        n0 = 1480 if case == "scalar_cache_1500" else 580
        base, _ = rt.scene("random_spheres", n=n0 - 1, extent=6.0)
        glass = rt.SwiftMaterial.dielectric(1.5)
        red = rt.SwiftMaterial.lambertian(0.9, 0.1, 0.1)
        extra = [rt.Object.from_buffer_copy(base[i]) for i in (7, 300, n0 - 1)]
        for e in extra:
            e.mat = red
        extra += [rt.makeSphere((0.5, 0.3, 0.5), 0.3, glass), rt.makeSphere((0.5, 0.3, 0.5), 0.2, red),
                  rt.makeSphere((0.5, 0.3, 0.5), 0.3, red), rt.makeSphere((1.0, 0.2, 1.0), 0.0, red),
                  rt.makeSphere((2.0, 0.5, -1.0), -0.5, glass), rt.makeSphere((3.0, 0.5, 1.0), float("nan"), red),
                  rt.makeSphere((float("nan"), 0.5, 1.0), 0.3, red), rt.makeSphere((0.0, 0.0, 0.0), 300.0, glass),
                  rt.makeSphere((5000.0, 100.0, -3000.0), 50.0, red)]
        n = n0 + len(extra)
        objs = (rt.Object * n)(*([base[i] for i in range(n0)] + extra))
        u = rt.config_uniforms("random_spheres", w, h, spp, depth, n)
    renderer.setupTexture(w, h)
    renderer.set_option(rt.RT_OPT_RESOLVE, 2)
    if case == "lds_vector_600":
        renderer.set_option(rt.RT_OPT_FILTER, 2)
    try:
        img = renderer.renderImage(u, objs, n)
        path = renderer.lastStats().scene_path
    finally:
        strict(renderer, rt)
    assert path == {"scalar_cache_1500": 1, "streamed_600": 2, "lds_vector_600": 0, "odd_600": 3}.get(case, path)
    assert np.array_equal(bits(img), bits(oracle.image(u, objs, w, h)))


@pytest.mark.parametrize("name,w,h,spp,depth,slots", [("cornell", 120, 96, 8, 10, 1 << 20), ("nested_glass", 96, 54, 4, 100, 2048),
                                                       ("three_spheres", 400, 225, 20, 50, 4096)])
def test_wavefront_kernels_bit_exact(rt, oracle, renderer, name, w, h, spp, depth, slots):
    """generate / extend / shade kernels with compacted queues: same bits as the oracle, also when the pool is much
    smaller than the number of work items (slots are reused) and with quads, emission and 100-bounce paths."""
    strict(renderer, rt, chunk=3)
    objs, n = rt.scene(name)
    u = rt.config_uniforms(name, w, h, spp, depth, n)
    renderer.setupTexture(w, h)
    renderer.set_option(rt.RT_OPT_KERNEL, 1)
    renderer.set_option(rt.RT_OPT_WAVEFRONT_SLOTS, slots)
    try:
        img = renderer.renderImage(u, objs, n, chunk=3)
        st = renderer.lastStats()
    finally:
        renderer.set_option(rt.RT_OPT_KERNEL, 0)
        renderer.set_option(rt.RT_OPT_WAVEFRONT_SLOTS, 1 << 20)
    acc = np.zeros((h, w, 4), np.float32)
    for b in range(0, spp, 3):
        acc = acc + oracle.accumulate(u, objs, w, h, b, min(b + 3, spp))
    assert np.array_equal(bits(img), bits(oracle.resolve(acc, spp)))
    stats = [0] * 5
    oracle.image(u, objs, w, h, stats=stats)
    assert st.segments == stats[0]


def test_wavefront_kernels_with_automatic_items(rt, oracle, renderer):
    """The wavefront kernels keep one partial sum per work item: with automatic items (which the persistent kernel cuts
    into finer planes at the end of a launch) the reported schedule must be the one they use."""
    strict(renderer, rt, chunk=-1)
    w, h, spp = 320, 180, 40
    objs, n = rt.scene("cover")
    u = rt.config_uniforms("cover", w, h, spp, 50, n)
    renderer.setupTexture(w, h)
    renderer.set_option(rt.RT_OPT_KERNEL, 1)
    try:
        img = renderer.renderImage(u, objs, n)
        st = renderer.lastStats()
    finally:
        renderer.set_option(rt.RT_OPT_KERNEL, 0)
        strict(renderer, rt)
    ranges = st.chunk_ranges(0, spp)
    assert len(ranges) > 1 and ranges[0][0] == 0 and ranges[-1][1] == spp
    acc = np.zeros((h, w, 4), np.float32)
    for b, e in ranges:
        acc = acc + oracle.accumulate(u, objs, w, h, b, e)
    assert np.array_equal(bits(img), bits(oracle.resolve(acc, spp)))


def test_staged_matrix_kind_at_its_largest_scene(rt, renderer):
    """1024 spheres is the most the staged matrix-core kind takes: lifted and strict records plus the per-wave candidate
    words, list and minima fill the 160 KB of LDS.  Same bits as the scalar-cache vector path."""
    strict(renderer, rt)
    objs, n = rt.scene("random_spheres", n=1023, extent=8.0)
    assert n == 1024
    u = rt.config_uniforms("random_spheres", 160, 90, 4, 50, n)
    renderer.setupTexture(160, 90)
    renderer.setScene(objs, n)
    imgs = []
    for fetch, filt, path in ((1, 2, 1), (0, 0, 3)):
        strict(renderer, rt, fetch=fetch)
        renderer.set_option(rt.RT_OPT_FILTER, filt)
        try:
            imgs.append(renderer.renderImage(u, objs, n).copy())
            assert renderer.lastStats().scene_path == path
        finally:
            strict(renderer, rt)
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))


def test_plain_c_host_driver_end_to_end(rt, oracle, tmp_path):
    """The command-line driver (plain C over the C-ABI, no Python or torch in the process): batch render and the
    reference's progressive tile flow, each written as a PPM that must equal, byte for byte, the PPM of the oracle's
    image of the same configuration."""
    exe = os.path.join(os.path.dirname(HERE), "ray-tracing_amd", "lib", "rt_render")
    assert os.path.exists(exe), "make builds ray-tracing_amd/lib/rt_render"
    import subprocess
    w, h, spp = 64, 36, 6
    objs, n = rt.scene("three_spheres")
    u = rt.config_uniforms("three_spheres", w, h, spp, 50, n)
    want = str(tmp_path / "want.ppm")
    img = np.ascontiguousarray(oracle.image(u, objs, w, h))
    assert rt.host_lib().rt_write_ppm(want.encode(), img.ctypes.data, w, h) == 0
    got = str(tmp_path / "got.ppm")
    base = [exe, "--scene", "three_spheres", "--width", str(w), "--height", str(h), "--spp", str(spp), "--out", got]
    subprocess.run(base + ["--chunk", "0"], check=True, timeout=120, stdout=subprocess.DEVNULL)
    assert open(got, "rb").read() == open(want, "rb").read()
    # progressive: 2 passes of spp samples through 64x64 tiles = the oracle's two dispatches onto one texture
    tex = oracle.image(u, objs, w, h)
    u2 = rt.Uniforms.from_buffer_copy(u)
    u2.currentSample = 1
    tex = oracle.image(u2, objs, w, h, tex=tex)
    assert rt.host_lib().rt_write_ppm(want.encode(), np.ascontiguousarray(tex).ctypes.data, w, h) == 0
    subprocess.run(base + ["--progressive", "2"], check=True, timeout=120, stdout=subprocess.DEVNULL)
    assert open(got, "rb").read() == open(want, "rb").read()


def test_random_scenes_every_fetch_kind_filter_and_kernel_agree():
    """tools/fuzz_kinds.py on 12 seeds: scenes of 1 to 3500 spheres over three decades of radius and extent, at and far
    from the origin, flat or cubic, with and without a ground sphere and a duplicated object, rendered through every
    sphere-fetch kind, filter choice and kernel formulation -- all must give the bits of the scalar-cache vector path
    (which other tests pin to the oracle)."""
    import subprocess
    tool = os.path.join(os.path.dirname(HERE), "tools", "fuzz_kinds.py")
    out = subprocess.run([sys.executable, tool, "0", "12"], capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]

