"""GPU: rt_render_noalloc, the entry point that allocates nothing, against rt_render on the same inputs.

Called through the C-ABI (ctypes).  Bar: the linear accumulation buffers of the two entry points are bit-identical --
the same kernels run with the launch's scratch (counters, camera-ray queues, partial-sum planes) in the library's
static device array instead of in buffers the context allocates.  Two calls in a row, the second with a smaller
target, scene and sample count, so that whatever the first call left in the shared array is in the way of the second.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def hip_runtime():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    return hip


def accum_of(r, hip, w, h):
    """The context's linear accumulation buffer (float4 per pixel) copied to the host."""
    r.synchronize()
    out = np.empty((h, w, 4), np.float32)
    assert hip.hipMemcpy(out.ctypes.data, r.accumPtr(), out.nbytes, 2) == 0  # hipMemcpyDeviceToHost
    return out


def free_bytes(hip):
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def both_ways(rt, r, hip, name, w, h, spp, depth, chunk, ray_begin=0, tile=None):
    """Render with rt_render, then with rt_render_noalloc; returns the two accumulation buffers and statistics."""
    objs, n = rt.scene(name)
    u = rt.config_uniforms(name, w, h, spp, depth, n)
    if tile is not None:
        u.tileX, u.tileY, u.tileWidth, u.tileHeight = tile
    r.setupTexture(w, h)
    r.setScene(objs, n)
    r.set_option(rt.RT_OPT_CHUNK, chunk)
    r.accumClear()
    r.render(u, None, 0, ray_begin, spp)
    want = accum_of(r, hip, w, h)
    st_want = r.lastStats()
    r.resolve(spp)
    want_img = r.readTexture()
    # the planes, queues and counters rt_render needed exist by now; what follows must not change the device's free memory
    before = free_bytes(hip)
    r.accumClear()
    r.renderNoAlloc(u, ray_begin, spp)
    got = accum_of(r, hip, w, h)
    st_got = r.lastStats()
    r.resolve(spp)
    got_img = r.readTexture()
    assert free_bytes(hip) == before, "rt_render_noalloc changed the amount of free device memory"
    return want, got, want_img, got_img, st_want, st_got


def test_noalloc_entry_point_matches_rt_render_twice_in_a_row(rt):
    hip = hip_runtime()
    r = rt.MetalRenderer(device=0)
    try:
        # first call: cover scene (matrix-core filter), 8 planes of 640x360, more items than the cut grid has lanes
        want, got, wi, gi, sw, sg = both_ways(rt, r, hip, "cover", 640, 360, 48, 50, chunk=-1)
        assert len(sw.chunk_ranges(0, 48)) > 1 and sg.chunk_ranges(0, 48) == sw.chunk_ranges(0, 48)   # several planes, the same schedule
        assert sg.grid_blocks * sg.block_threads <= 32768 < sw.grid_blocks * sw.block_threads
        assert sg.segments == sw.segments and sg.samples == sw.samples == 640 * 360 * 48
        assert want.any() and np.array_equal(bits(got), bits(want))
        assert np.array_equal(bits(gi), bits(wi))
        # second call, smaller in every respect: fewer pixels, a tile inside the target, fewer samples from a non-zero
        # first index, five objects instead of 487 (vector filter)
        want, got, wi, gi, sw, sg = both_ways(rt, r, hip, "three_spheres", 200, 120, 10, 50, chunk=4, ray_begin=3,
                                              tile=(8, 16, 150, 90))
        assert sg.segments == sw.segments and sg.samples == sw.samples == 150 * 90 * 7
        assert want.any() and np.array_equal(bits(got), bits(want))
        assert not want[:16].any() and not got[:16].any()  # rows above the tile stay untouched
        assert np.array_equal(bits(gi), bits(wi))
        # third call: nested shells, i.e. the kernels that prune candidates with the scene's bounding balls
        want, got, wi, gi, sw, sg = both_ways(rt, r, hip, "nested_glass", 160, 90, 4, 60, chunk=-1)
        assert sg.segments == sw.segments and sg.samples == sw.samples == 160 * 90 * 4
        assert want.any() and np.array_equal(bits(got), bits(want))
        assert np.array_equal(bits(gi), bits(wi))
    finally:
        r.close()


def test_noalloc_entry_point_limits(rt):
    """Sizes beyond the static scratch are refused with a clear error; objects cannot be passed; a frame that fits with
    one plane renders identically with the grid cut to the queue part of the scratch."""
    hip = hip_runtime()
    r = rt.MetalRenderer(device=0)
    try:
        lib, ctx = r._lib, r._ctx
        objs, n = rt.scene("three_spheres")
        u = rt.config_uniforms("three_spheres", 1920, 1080, 64, 50, n)
        r.setupTexture(1920, 1080)
        assert lib.rt_render_noalloc(ctx, C.byref(u), 0, 64, None) == -5          # no resident scene
        r.setScene(objs, n)
        r.set_option(rt.RT_OPT_CHUNK, 8)                                          # 8 planes x 33 MB > 64 MiB
        assert lib.rt_render_noalloc(ctx, C.byref(u), 0, 64, None) == -6
        msg = lib.rt_last_error(ctx).decode()
        assert "rt_render_noalloc" in msg and "RT_OPT_CHUNK" in msg and str(rt.RT_NOALLOC_PLANE_BYTES) in msg
        r.set_option(rt.RT_OPT_KERNEL, 1)
        r.set_option(rt.RT_OPT_CHUNK, 0)
        assert lib.rt_render_noalloc(ctx, C.byref(u), 0, 64, None) == -1          # the wavefront pool would allocate
        r.set_option(rt.RT_OPT_KERNEL, 0)
        want, got, wi, gi, sw, sg = both_ways(rt, r, hip, "three_spheres", 1920, 1080, 6, 50, chunk=0)
        assert sg.chunk_count == 1 and sg.grid_blocks * sg.block_threads <= 32768
        assert np.array_equal(bits(got), bits(want)) and np.array_equal(bits(gi), bits(wi))
    finally:
        r.close()


def test_noalloc_entry_point_under_graph_capture(rt):
    """The caller the entry point names first: rt_accum_clear + rt_render_noalloc + rt_resolve captured into a HIP graph
    on the context's stream (global capture mode, which refuses allocating calls), the graph launched twice, buffers
    dirtied in between; sums and image bit-identical to rt_render outside a graph."""
    hip = hip_runtime()
    vp = C.c_void_p
    hip.hipStreamCreate.argtypes = [C.POINTER(vp)]
    hip.hipStreamDestroy.argtypes = [vp]
    hip.hipStreamBeginCapture.argtypes = [vp, C.c_int]
    hip.hipStreamEndCapture.argtypes = [vp, C.POINTER(vp)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(vp), vp, vp, vp, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [vp, vp]
    hip.hipGraphExecDestroy.argtypes = [vp]
    hip.hipGraphDestroy.argtypes = [vp]
    hip.hipMemsetAsync.argtypes = [vp, C.c_int, C.c_size_t, vp]
    stream = vp()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    r = rt.MetalRenderer(device=0, stream=stream.value)
    graph, gexec = vp(), vp()
    try:
        w, h, spp = 320, 180, 24
        objs, n = rt.scene("cover")
        u = rt.config_uniforms("cover", w, h, spp, 50, n)
        r.setupTexture(w, h)
        r.setScene(objs, n)
        r.set_option(rt.RT_OPT_CHUNK, -1)
        r.accumClear()
        r.render(u, None, 0, 0, spp)
        want = accum_of(r, hip, w, h)
        r.resolve(spp)
        want_img = r.readTexture()
        segments = r.lastStats().segments
        r.synchronize()
        assert hip.hipStreamBeginCapture(stream, 0) == 0  # hipStreamCaptureModeGlobal
        try:
            r.accumClear()
            r.renderNoAlloc(u, 0, spp)
            r.resolve(spp)
        finally:
            rc = hip.hipStreamEndCapture(stream, C.byref(graph))
        assert rc == 0 and graph.value
        assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
        for rep in range(2):
            # whatever the last run left must not matter (dirtied on the context's stream: ordered with the graph)
            assert hip.hipMemsetAsync(r.accumPtr(), 0x7F, w * h * 16, stream) == 0
            r.clearTexture()
            r.synchronize()
            assert hip.hipGraphLaunch(gexec, stream) == 0
            got = accum_of(r, hip, w, h)
            assert np.array_equal(bits(got), bits(want)), rep
            assert np.array_equal(bits(r.readTexture()), bits(want_img)), rep
        st = r.lastStats()   # counters of the graph's launch; no timing events were recorded under capture
        assert st.segments == segments and st.kernel_ms == 0.0
    finally:
        if gexec.value:
            hip.hipGraphExecDestroy(gexec)
        if graph.value:
            hip.hipGraphDestroy(graph)
        r.close()
        hip.hipStreamDestroy(stream)
