"""Throughput of the reference's own call shape: one blocking rt_render_tile per 64x64 tile (ContentView.swift:628-709),
cover scene at 1920x1080 (510 tiles), one pass.  Measured twice: the same Object[] re-sent on every call (the library
recognises it and keeps the compacted scene), and one byte of it changed between calls (every call rebuilds the scene:
compaction, FP16 lift, uploads -- what every call did before the comparison was added).  NARROW=0/1: RT_OPT_NARROW."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

rt = ge.load_package()
W, H, spp = 1920, 1080, int(os.environ.get("SPP", "32"))
objs, n = rt.scene("cover")
r = rt.MetalRenderer(0)
r.setupTexture(W, H)
r.generateTiles()
if os.environ.get("NARROW"):
    r.set_option(rt.RT_OPT_NARROW, int(os.environ["NARROW"]))
u = rt.config_uniforms("cover", W, H, spp, 50, n)
lib, ctx = r._lib, r._ctx
raw = (C.c_ubyte * (160 * n)).from_buffer(objs)
for label, touch in (("same objects re-sent", False), ("objects changed before every call", True)):
    r.clearTexture()
    for rep in range(2):
        t0 = time.perf_counter()
        for i, t in enumerate(r.tiles):
            u.tileX, u.tileY, u.tileWidth, u.tileHeight = t.x, t.y, t.width, t.height
            u.currentSample = 0
            if touch:
                raw[159] = (i + rep) & 0xFF  # a padding byte of the first Object: the scene is the same, the bytes are not
            rc = lib.rt_render_tile(ctx, C.byref(u), objs, n)
            assert rc in (0, 1), rc
        dt = time.perf_counter() - t0
    print("%-36s %d tiles in %7.1f ms = %6.3f ms per call, %7.1f Msamples*px/s at %d spp per pass"
          % (label, len(r.tiles), dt * 1e3, dt * 1e3 / len(r.tiles), W * H * spp / dt * 1e-6, spp), flush=True)
r.close()
