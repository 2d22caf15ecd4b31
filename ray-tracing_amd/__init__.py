"""Host-side Python mirror of the reference's renderer interface over the C-ABI of librt_amd.so.

The directory name has a hyphen, so load this package with ``importlib`` (see ``load_package`` in
``__graft_entry__.py``) under the module name ``ray_tracing_amd``.

The class and method names follow the reference's Swift host (reference:
"Ray Tracer/ContentView.swift"): ``MetalRenderer.setupTexture / clearTexture /
runComputeShaderForTile / startProgressiveRender / renderTile``, ``makeSphere``, ``makeQuad``,
``SwiftMaterial``, ``initRayTracing``.  Everything that computes runs in the HIP library; this file
only marshals arguments with ctypes.  There is no CPU fallback: if the library is missing or no GPU is
present the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

# ---- ABI structs (include/rt_abi.h; reference: "Ray Tracer/Definitions.h":13-78) -------------------


class Float2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("_pad", C.c_float)]


class Float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Uniforms(C.Structure):
    _fields_ = [
        ("color", Float4), ("time", C.c_float), ("_p0", C.c_float * 3),
        ("pixelDeltaX", Float3), ("pixelDeltaY", Float3), ("pixelOrigin", Float3),
        ("pixelSampleScale", C.c_float), ("_p1", C.c_float * 3),
        ("cameraCenter", Float3), ("viewportSize", Float2), ("_p2", C.c_float * 2),
        ("defocusDiskU", Float3), ("defocusDiskV", Float3),
        ("defocusAngle", C.c_float), ("objCount", C.c_int), ("lightCount", C.c_int), ("sampleCount", C.c_int),
        ("maxRayDepth", C.c_int), ("currentSample", C.c_int), ("totalSamples", C.c_int),
        ("tileX", C.c_int), ("tileY", C.c_int), ("tileWidth", C.c_int), ("tileHeight", C.c_int), ("_p3", C.c_int),
        ("globalIllumation", Float4),
    ]


class Sphere(C.Structure):
    _fields_ = [("center", Float3), ("radius", C.c_float), ("_p", C.c_float * 3)]


class Quad(C.Structure):
    _fields_ = [("origin", Float3), ("extentX", Float3), ("extentY", Float3)]


class MeshMaterial(C.Structure):
    _fields_ = [("type", C.c_int), ("emission", C.c_float), ("_p0", C.c_float * 2), ("albedo", Float4),
                ("emission_color", Float4), ("reflection_fuzz", C.c_float), ("refraction_index", C.c_float),
                ("_p1", C.c_float * 2)]


class Object(C.Structure):
    _fields_ = [("type", C.c_int), ("_p0", C.c_int * 3), ("s", Sphere), ("q", Quad), ("mat", MeshMaterial)]


assert C.sizeof(Uniforms) == 224 and Uniforms.globalIllumation.offset == 208 and Uniforms.tileX.offset == 188
assert C.sizeof(Sphere) == 32 and C.sizeof(Quad) == 48 and C.sizeof(MeshMaterial) == 64 and C.sizeof(Object) == 160
assert Object.s.offset == 16 and Object.q.offset == 48 and Object.mat.offset == 96

LAMBIERTIAN, REFLECTEE, DIELECTRIC = 0, 1, 2
TYPE_SPHERE, TYPE_QUAD = 0, 1


class TileInfo(C.Structure):
    """ContentView.swift:315-321 (x, y, width, height of one tile)."""
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("width", C.c_int), ("height", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("segments", C.c_uint64), ("tests", C.c_uint64), ("samples", C.c_uint64), ("kernel_ms", C.c_float),
                ("grid_blocks", C.c_int), ("block_threads", C.c_int), ("scene_path", C.c_int),
                ("chunk", C.c_int), ("chunk_count", C.c_int), ("tail_chunk", C.c_int), ("phase_cycles", C.c_uint64 * 8),
                ("wave_trips", C.c_uint64), ("matrix_blocks", C.c_uint64)]

    def chunk_ranges(self, ray_begin, ray_end):
        """The sample ranges the last rt_render launch summed separately, in the order their sums are added."""
        out, r = [], ray_begin
        for i in range(self.chunk_count):
            if r >= ray_end:
                break
            out.append((r, min(r + self.chunk, ray_end)))
            r += self.chunk
        while r < ray_end:
            out.append((r, min(r + self.tail_chunk, ray_end)))
            r += self.tail_chunk
        return out


RT_OPT_CHUNK, RT_OPT_KERNEL, RT_OPT_SCENE_FETCH, RT_OPT_BLOCKS_PER_CU, RT_OPT_FILTER, RT_OPT_WAVEFRONT_SLOTS = 1, 3, 4, 5, 6, 7
RT_OPT_TAIL_CHUNK, RT_OPT_TAIL_SAMPLES, RT_OPT_RESOLVE, RT_OPT_REGROUP, RT_OPT_PLANE_SAMPLES, RT_OPT_SHARE = 8, 9, 10, 11, 12, 13
RT_OPT_NARROW = 14
RT_NOALLOC_PLANE_BYTES = 64 << 20  # include/rt_amd.h

# Every symbol include/rt_amd.h declares.
AMD_SYMBOLS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_set_option", "rt_get_option", "rt_set_target", "rt_clear",
    "rt_write_target", "rt_read_target", "rt_render_tile", "rt_present_bgra8", "rt_set_scene", "rt_render",
    "rt_render_noalloc",
    "rt_accum_clear", "rt_accum_ptr", "rt_resolve", "rt_synchronize", "rt_first_hit", "rt_last_stats", "rt_trace_times",
    "rt_selftest_filter", "rt_selftest_matrix_filter", "rt_selftest_tiled_filter", "rt_selftest_prune", "rt_selftest_box", "rt_selftest_tile_boxes", "rt_group_bounds", "rt_filter_box",
]
HOST_SYMBOLS = [
    "rt_material_default", "rt_material_lambertian", "rt_material_metal", "rt_material_dielectric", "rt_material_light",
    "rt_make_sphere", "rt_make_quad", "rt_camera_setup", "rt_uniforms_reference_defaults", "rt_scene_cornell",
    "rt_scene_three_spheres", "rt_scene_cover", "rt_scene_nested_glass", "rt_scene_random_spheres", "rt_config_uniforms",
    "rt_generate_tiles", "rt_write_ppm", "rt_write_pfm",
]

_amd = None
_host = None


def amd_lib():
    """librt_amd.so (HIP kernels + C-ABI).  Raises OSError if it has not been built."""
    global _amd
    if _amd is None:
        # RT_AMD_LIB: another build of the same library, for A/B measurements (tools/)
        lib = C.CDLL(os.environ.get("RT_AMD_LIB") or os.path.join(LIB_DIR, "librt_amd.so"))
        vp, ip, fp = C.c_void_p, C.c_int, C.POINTER(C.c_float)
        lib.rt_create.argtypes = [C.POINTER(vp), ip, vp]
        lib.rt_destroy.argtypes = [vp]
        lib.rt_destroy.restype = None
        lib.rt_last_error.argtypes = [vp]
        lib.rt_last_error.restype = C.c_char_p
        lib.rt_set_option.argtypes = [vp, ip, ip]
        lib.rt_get_option.argtypes = [vp, ip, C.POINTER(ip)]
        lib.rt_set_target.argtypes = [vp, ip, ip]
        lib.rt_clear.argtypes = [vp]
        lib.rt_write_target.argtypes = [vp, vp]
        lib.rt_read_target.argtypes = [vp, vp]
        lib.rt_render_tile.argtypes = [vp, C.POINTER(Uniforms), vp, ip]
        lib.rt_present_bgra8.argtypes = [vp, vp, ip, ip]
        lib.rt_set_scene.argtypes = [vp, vp, ip]
        lib.rt_render.argtypes = [vp, C.POINTER(Uniforms), vp, ip, ip, ip, vp]
        lib.rt_render_noalloc.argtypes = [vp, C.POINTER(Uniforms), ip, ip, vp]
        lib.rt_accum_clear.argtypes = [vp, vp]
        lib.rt_accum_ptr.argtypes = [vp, C.POINTER(vp)]
        lib.rt_resolve.argtypes = [vp, vp, ip]
        lib.rt_synchronize.argtypes = [vp]
        lib.rt_first_hit.argtypes = [vp, C.POINTER(Uniforms), vp, ip, vp, vp]
        lib.rt_last_stats.argtypes = [vp, C.POINTER(Stats)]
        lib.rt_trace_times.argtypes = [vp, vp, ip, C.POINTER(ip)]
        lib.rt_selftest_filter.argtypes = [vp, C.c_uint32, ip, ip, vp]
        lib.rt_selftest_matrix_filter.argtypes = [vp, C.c_uint32, ip, ip, ip, vp]
        lib.rt_selftest_tiled_filter.argtypes = [vp, C.c_uint32, ip, ip, ip, vp]
        lib.rt_selftest_prune.argtypes = [vp, C.c_uint32, ip, ip, vp]
        lib.rt_selftest_box.argtypes = [vp, C.c_uint32, ip, ip, ip, vp]
        lib.rt_selftest_tile_boxes.argtypes = [vp, C.c_uint32, ip, ip, ip, vp]
        lib.rt_group_bounds.argtypes = [vp, ip, vp]
        _amd = lib
    return _amd


def host_lib():
    """librt_host.so (plain-C host driver: camera, scenes, tiles, image writers)."""
    global _host
    if _host is None:
        lib = C.CDLL(os.path.join(LIB_DIR, "librt_host.so"))
        f, ip = C.c_float, C.c_int
        f3 = C.POINTER(C.c_float)
        lib.rt_material_default.restype = MeshMaterial
        lib.rt_material_lambertian.restype = MeshMaterial
        lib.rt_material_lambertian.argtypes = [f, f, f]
        lib.rt_material_metal.restype = MeshMaterial
        lib.rt_material_metal.argtypes = [f, f, f, f]
        lib.rt_material_dielectric.restype = MeshMaterial
        lib.rt_material_dielectric.argtypes = [f]
        lib.rt_material_light.restype = MeshMaterial
        lib.rt_material_light.argtypes = [f, f, f, f]
        lib.rt_make_sphere.restype = Object
        lib.rt_make_sphere.argtypes = [f3, f, MeshMaterial, f]
        lib.rt_make_quad.restype = Object
        lib.rt_make_quad.argtypes = [f3, f3, f3, MeshMaterial, f]
        lib.rt_camera_setup.restype = None
        lib.rt_camera_setup.argtypes = [C.POINTER(Uniforms), f3, f3, f3, f, f, f, ip, ip]
        lib.rt_uniforms_reference_defaults.restype = None
        lib.rt_uniforms_reference_defaults.argtypes = [C.POINTER(Uniforms), ip, ip]
        lib.rt_scene_cornell.argtypes = [C.c_void_p, ip]
        lib.rt_scene_three_spheres.argtypes = [C.c_void_p, ip]
        lib.rt_scene_cover.argtypes = [C.c_void_p, ip, C.c_uint64]
        lib.rt_scene_nested_glass.argtypes = [C.c_void_p, ip, ip]
        lib.rt_scene_random_spheres.argtypes = [C.c_void_p, ip, ip, f, C.c_uint64]
        lib.rt_config_uniforms.argtypes = [C.c_char_p, C.POINTER(Uniforms), ip, ip, ip, ip]
        lib.rt_generate_tiles.argtypes = [ip, ip, ip, C.c_void_p, ip]
        lib.rt_write_ppm.argtypes = [C.c_char_p, C.c_void_p, ip, ip]
        lib.rt_write_pfm.argtypes = [C.c_char_p, C.c_void_p, ip, ip]
        _host = lib
    return _host


class RtError(RuntimeError):
    pass


# ---- scene / camera helpers (host driver) ------------------------------------------------------------

COVER_SEED = 20250711  # seed of the committed cover-scene layout


def _vec3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class SwiftMaterial:
    """ContentView.swift:15-45."""

    @staticmethod
    def lambertian(r, g, b):
        return host_lib().rt_material_lambertian(r, g, b)

    @staticmethod
    def metal(r, g, b, fuzz):
        return host_lib().rt_material_metal(r, g, b, fuzz)

    @staticmethod
    def dielectric(refraction_index):
        return host_lib().rt_material_dielectric(refraction_index)

    @staticmethod
    def light(strength, color=(1.0, 1.0, 1.0)):
        return host_lib().rt_material_light(strength, *color)


def makeSphere(center=(0, 0, 0), radius=0.0, material=None, scale=1.0):
    """ContentView.swift:47-49."""
    m = material if material is not None else host_lib().rt_material_default()
    return host_lib().rt_make_sphere(_vec3(center), radius, m, scale)


def makeQuad(origin=(0, 0, 0), extentX=(0, 0, 0), extentY=(0, 0, 0), material=None, scale=1.0):
    """ContentView.swift:51-53."""
    m = material if material is not None else host_lib().rt_material_default()
    return host_lib().rt_make_quad(_vec3(origin), _vec3(extentX), _vec3(extentY), m, scale)


def object_array(objs):
    arr = (Object * max(1, len(objs)))()
    for i, o in enumerate(objs):
        arr[i] = o
    return arr


def scene(name, **kw):
    """Named benchmark scene -> ctypes array of Object and its length."""
    h = host_lib()

    def build(fn, *args):
        n = fn(None, 0, *args)
        arr = (Object * max(1, n))()
        fn(arr, n, *args)
        return arr, n

    if name == "cornell":
        return build(h.rt_scene_cornell)
    if name == "three_spheres":
        return build(h.rt_scene_three_spheres)
    if name == "cover":
        return build(h.rt_scene_cover, kw.get("seed", COVER_SEED))
    if name == "nested_glass":
        return build(h.rt_scene_nested_glass, kw.get("shells", 400))
    if name == "random_spheres":
        return build(h.rt_scene_random_spheres, kw.get("n", 100000), kw.get("extent", 50.0), kw.get("seed", COVER_SEED))
    raise ValueError(name)


def config_uniforms(name, width, height, spp, depth, objCount):
    u = Uniforms()
    if host_lib().rt_config_uniforms(name.encode(), C.byref(u), width, height, spp, depth) != 0:
        raise ValueError(name)
    u.objCount = objCount
    return u


def group_bounds(spheres):
    """Bounding balls per word of 32 and granule of 8 sphere slots as the library builds them at scene upload (host
    only).  spheres: (n, 4) float32 (x, y, z, radius).  Returns (words, 5, 2, 4) float32: [word, 0] the word's pair
    (C, R2out), (R2in, exempt bits, 0, 0); [word, 1 + g] the pair of granule g."""
    import numpy as np
    sph = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 4)
    words = (len(sph) + 31) // 32
    out = np.zeros((max(words, 1), 5, 2, 4), np.float32)
    n = amd_lib().rt_group_bounds(sph.ctypes.data, len(sph), out.ctypes.data)
    if n < 0:
        raise RtError(n, "rt_group_bounds")
    return out[:n]


def filter_box(spheres):
    """The box of the half-width sphere filter as the library builds it at scene upload (host only).  spheres: (n, 4)
    float32 (x, y, z, radius).  Returns None when the scene does without, else a dict: mid, half (3 floats each), reach,
    and words = [(word index, tested bits, always bits), ...]."""
    import numpy as np
    sph = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 4)
    out = np.zeros(20, np.float32)
    lib = amd_lib()
    lib.rt_filter_box.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    r = lib.rt_filter_box(sph.ctypes.data, len(sph), out.ctypes.data)
    if r < 0:
        raise RtError(r, "rt_filter_box")
    if r == 0:
        return None
    bits = out.view(np.uint32)
    return {"mid": out[0:3].copy(), "half": out[3:6].copy(), "reach": float(out[6]),
            "words": [(int(bits[8 + 3 * k]), int(bits[9 + 3 * k]), int(bits[10 + 3 * k])) for k in range(int(out[7]))]}


def generate_tiles(width, height, tileSize=64):
    """ContentView.swift:380-401."""
    h = host_lib()
    n = h.rt_generate_tiles(width, height, tileSize, None, 0)
    arr = (TileInfo * n)()
    h.rt_generate_tiles(width, height, tileSize, arr, n)
    return list(arr)


# ---- the renderer -------------------------------------------------------------------------------------


class MetalRenderer:
    """Mirror of the reference's ``MetalRenderer`` (ContentView.swift:232-752) over librt_amd.so.

    ``objects`` is a Python list of ``Object`` and ``uniforms`` a ``Uniforms``; both are re-sent on every
    dispatch, as the reference does with setBytes (ContentView.swift:689-696).
    """

    tileSize = 64  # ContentView.swift:279

    def __init__(self, device=0, stream=None):
        """stream: None = the library creates a private non-blocking stream; else a non-zero hipStream_t handle
        (e.g. ``torch.cuda.Stream().cuda_stream``) that every call enqueues on.  Handle 0, the legacy default
        stream, is refused: rt_create reads a null handle as "create your own", and work the caller then enqueues on
        the default stream (a reduce of the accumulation buffer, say) would not be ordered against the library's."""
        self._lib = amd_lib()
        self._ctx = C.c_void_p()
        if stream is not None and int(stream) == 0:
            raise ValueError("stream handle 0 (the default stream) cannot be passed: create a stream of your own, "
                             "e.g. torch.cuda.Stream(), make it current and pass its handle; or pass None")
        rc = self._lib.rt_create(C.byref(self._ctx), device, C.c_void_p(int(stream)) if stream is not None else None)
        if rc != 0:
            raise RtError("rt_create failed with code %d (no GPU, or HIP runtime error)" % rc)
        self.objects = []
        self.uniforms = Uniforms()
        self.maxIterations = 400
        self.textureWidth = 0
        self.textureHeight = 0
        self.tiles = []
        self.currentSample = 0
        self.currentTileIndex = 0
        self.isRendering = False
        self.renderProgress = 0.0

    # -- plumbing --
    def close(self):
        if self._ctx:
            self._lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise RtError("%s (code %d)" % (self._lib.rt_last_error(self._ctx).decode(), rc))
        return rc

    def set_option(self, key, value):
        self._check(self._lib.rt_set_option(self._ctx, key, value))

    # -- reference interface --
    def setupTexture(self, width, height):
        """ContentView.swift:358-378."""
        self._check(self._lib.rt_set_target(self._ctx, width, height))
        self.textureWidth, self.textureHeight = width, height
        self.generateTiles()

    def generateTiles(self):
        self.tiles = generate_tiles(self.textureWidth, self.textureHeight, self.tileSize)

    def clearTexture(self):
        """ContentView.swift:661-664."""
        self._check(self._lib.rt_clear(self._ctx))

    def updateTexture(self, pixelData):
        """ContentView.swift:509-518: upload host pixels (H x W x 4 float32)."""
        a = np.ascontiguousarray(pixelData, dtype=np.float32)
        assert a.shape == (self.textureHeight, self.textureWidth, 4)
        self._check(self._lib.rt_write_target(self._ctx, a.ctypes.data))

    def runComputeShaderForTile(self, tile, sampleIndex):
        """ContentView.swift:666-709.  Returns 1 when the reference would have painted error pixels."""
        u = Uniforms.from_buffer_copy(self.uniforms)
        u.currentSample = int(sampleIndex)
        u.totalSamples = int(self.maxIterations)
        u.objCount = len(self.objects)
        u.tileX, u.tileY, u.tileWidth, u.tileHeight = tile.x, tile.y, tile.width, tile.height
        arr = object_array(self.objects)
        return self._check(self._lib.rt_render_tile(self._ctx, C.byref(u), arr, len(self.objects)))

    def startProgressiveRender(self):
        """ContentView.swift:600-620 (without the timer: call renderTile() until it returns False)."""
        if self.isRendering:
            return
        self.isRendering = True
        self.currentSample = 0
        self.currentTileIndex = 0
        self.renderProgress = 0.0
        self.clearTexture()

    def stopProgressiveRender(self):
        self.isRendering = False

    def renderTile(self):
        """ContentView.swift:628-659.  Returns False once rendering has stopped."""
        if not self.isRendering:
            return False
        if self.currentTileIndex >= len(self.tiles):
            self.currentSample += 1
            if self.maxIterations > 500:
                pass
            elif self.currentSample >= self.maxIterations:
                self.stopProgressiveRender()
                return False
            self.currentTileIndex = 0
            return True
        tile = self.tiles[self.currentTileIndex]
        self.runComputeShaderForTile(tile, self.currentSample)
        self.currentTileIndex += 1
        totalWork = len(self.tiles) * self.maxIterations
        completedWork = self.currentSample * len(self.tiles) + self.currentTileIndex
        self.renderProgress = completedWork / float(totalWork)
        return True

    def readTexture(self):
        """No reference counterpart: the target as an H x W x 4 float32 array."""
        out = np.empty((self.textureHeight, self.textureWidth, 4), np.float32)
        self._check(self._lib.rt_read_target(self._ctx, out.ctypes.data))
        return out

    def present(self, out_w=None, out_h=None):
        """draw(in:) + Present.metal: the drawable as an H x W x 4 uint8 array in B,G,R,A order."""
        out_w = out_w or self.textureWidth
        out_h = out_h or self.textureHeight
        out = np.empty((out_h, out_w, 4), np.uint8)
        self._check(self._lib.rt_present_bgra8(self._ctx, out.ctypes.data, out_w, out_h))
        return out

    # -- batch path --
    def setScene(self, objs, n):
        self._check(self._lib.rt_set_scene(self._ctx, objs, n))

    def render(self, u, objs=None, n=0, ray_begin=0, ray_end=None, accum_dev=None):
        ray_end = u.sampleCount if ray_end is None else ray_end
        self._check(self._lib.rt_render(self._ctx, C.byref(u), objs, n, ray_begin, ray_end,
                                        C.c_void_p(accum_dev) if accum_dev else None))

    def renderNoAlloc(self, u, ray_begin=0, ray_end=None, accum_dev=None):
        """rt_render on the resident scene with the library's static device scratch: allocates nothing."""
        ray_end = u.sampleCount if ray_end is None else ray_end
        self._check(self._lib.rt_render_noalloc(self._ctx, C.byref(u), ray_begin, ray_end,
                                                C.c_void_p(accum_dev) if accum_dev else None))

    def accumClear(self, accum_dev=None):
        self._check(self._lib.rt_accum_clear(self._ctx, C.c_void_p(accum_dev) if accum_dev else None))

    def accumPtr(self):
        p = C.c_void_p()
        self._check(self._lib.rt_accum_ptr(self._ctx, C.byref(p)))
        return p.value

    def resolve(self, spp, accum_dev=None):
        self._check(self._lib.rt_resolve(self._ctx, C.c_void_p(accum_dev) if accum_dev else None, spp))

    def synchronize(self):
        self._check(self._lib.rt_synchronize(self._ctx))

    def firstHit(self, u, objs, n):
        out4 = np.empty((self.textureHeight, self.textureWidth, 4), np.float32)
        ids = np.empty((self.textureHeight, self.textureWidth), np.int32)
        self._check(self._lib.rt_first_hit(self._ctx, C.byref(u), objs, n, out4.ctypes.data, ids.ctypes.data))
        return out4, ids

    def lastStats(self):
        s = Stats()
        self._check(self._lib.rt_last_stats(self._ctx, C.byref(s)))
        return s

    def traceTimes(self):
        """Device durations (ms) of the trace launches since the previous call."""
        buf = (C.c_float * 64)()
        cnt = C.c_int()
        self._check(self._lib.rt_trace_times(self._ctx, buf, 64, C.byref(cnt)))
        return list(buf[:cnt.value])

    def selftestFilter(self, seed=1, blocks=2048, pairs_per_thread=256):
        out = (C.c_uint64 * 6)()
        self._check(self._lib.rt_selftest_filter(self._ctx, seed, blocks, pairs_per_thread, out))
        return dict(zip(["pairs", "reference", "general", "compact", "general_missed", "compact_missed"], list(out)))

    def selftestMatrixFilter(self, seed=1, blocks=1024, rounds=64, slack_log2=-16):
        out = (C.c_uint64 * 6)()
        self._check(self._lib.rt_selftest_matrix_filter(self._ctx, seed, blocks, rounds, slack_log2, out))
        return dict(zip(["pairs", "reference", "matrix", "missed", "out_of_range", "unflagged"], list(out)))

    def selftestPrune(self, seed=1, blocks=256, rays_per_thread=4):
        """On the scene set with setScene."""
        out = (C.c_uint64 * 6)()
        self._check(self._lib.rt_selftest_prune(self._ctx, seed, blocks, rays_per_thread, out))
        return dict(zip(["pairs", "reference", "single", "balls", "single_wrong", "balls_wrong"], list(out)))

    def selftestBox(self, seed=1, blocks=256, rays_per_thread=4, padding_log2=0):
        """On the scene set with setScene."""
        out = (C.c_uint64 * 6)()
        self._check(self._lib.rt_selftest_box(self._ctx, seed, blocks, rays_per_thread, padding_log2, out))
        return dict(zip(["rays", "left_out", "pairs", "roots", "wrong"], list(out)))

    def selftestTileBoxes(self, seed=1, blocks=256, rays_per_thread=2, padding_log2=0):
        """On the scene set with setScene."""
        out = (C.c_uint64 * 6)()
        self._check(self._lib.rt_selftest_tile_boxes(self._ctx, seed, blocks, rays_per_thread, padding_log2, out))
        return dict(zip(["rays", "left_out", "pairs", "roots", "wrong", "blocks_left_out"], list(out)))

    def selftestTiledFilter(self, seed=1, blocks=1024, rounds=64, slack_log2=-16):
        out = (C.c_uint64 * 6)()
        self._check(self._lib.rt_selftest_tiled_filter(self._ctx, seed, blocks, rounds, slack_log2, out))
        return dict(zip(["pairs", "reference", "matrix", "missed", "out_of_range", "unflagged"], list(out)))

    def renderImage(self, u, objs, n, chunk=None):
        """Whole image in one go: clear accumulation, trace u.sampleCount samples, resolve, read back."""
        if chunk is not None:
            self.set_option(RT_OPT_CHUNK, chunk)
        self.accumClear()
        self.render(u, objs, n, 0, u.sampleCount)
        self.resolve(u.sampleCount)
        self.synchronize()
        return self.readTexture()


def initRayTracing(renderer, width, height):
    """ContentView.swift:168-213: the reference's default camera and sampling settings."""
    host_lib().rt_uniforms_reference_defaults(C.byref(renderer.uniforms), width, height)
    renderer.maxIterations = 400
