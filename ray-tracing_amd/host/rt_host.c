/*
 * rt_host.c -- plain-C host driver: object builders, camera -> Uniforms, scenes,
 * tile list, image writers.  See rt_host.h.  No GPU code here.
 */
#include "rt_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- small float3 helpers (host side only) ------------------------------ */
typedef struct { float x, y, z; } h3;
static h3 h3v(float x, float y, float z) { h3 r = {x, y, z}; return r; }
static h3 h3p(const float p[3]) { return h3v(p[0], p[1], p[2]); }
static h3 h3add(h3 a, h3 b) { return h3v(a.x + b.x, a.y + b.y, a.z + b.z); }
static h3 h3sub(h3 a, h3 b) { return h3v(a.x - b.x, a.y - b.y, a.z - b.z); }
static h3 h3scale(h3 a, float s) { return h3v(a.x * s, a.y * s, a.z * s); }
static h3 h3div(h3 a, float s) { return h3v(a.x / s, a.y / s, a.z / s); }
static float h3dot(h3 a, h3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static h3 h3cross(h3 a, h3 b) { return h3v(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static h3 h3normalize(h3 a) { return h3div(a, sqrtf(h3dot(a, a))); }
static rt_float3 to3(h3 a) { rt_float3 r; memset(&r, 0, sizeof r); r.x = a.x; r.y = a.y; r.z = a.z; return r; }
static rt_float4 to4(float x, float y, float z, float w) { rt_float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

/* ---- materials / objects ------------------------------------------------- */
struct MeshMaterial rt_material_default(void) {
    struct MeshMaterial m;
    memset(&m, 0, sizeof m);
    m.type = LAMBIERTIAN;
    m.emission = 0.0f;
    m.albedo = to4(1, 1, 1, 1);
    m.emission_color = to4(0, 0, 0, 1);
    m.reflection_fuzz = 0.0f;
    m.refraction_index = 0.0f;
    return m;
}
struct MeshMaterial rt_material_lambertian(float r, float g, float b) {
    struct MeshMaterial m = rt_material_default();
    m.albedo = to4(r, g, b, 1.0f);
    return m;
}
struct MeshMaterial rt_material_metal(float r, float g, float b, float fuzz) {
    struct MeshMaterial m = rt_material_default();
    m.type = REFLECTEE;
    m.albedo = to4(r, g, b, 1.0f);
    m.reflection_fuzz = fuzz;
    return m;
}
struct MeshMaterial rt_material_dielectric(float refraction_index) {
    struct MeshMaterial m = rt_material_default();
    m.type = DIELECTRIC;
    m.refraction_index = refraction_index;
    return m;
}
struct MeshMaterial rt_material_light(float strength, float r, float g, float b) {
    struct MeshMaterial m = rt_material_default();
    m.emission = strength;
    m.emission_color = to4(r, g, b, 1.0f);
    return m;
}

struct Object rt_make_sphere(const float center[3], float radius, struct MeshMaterial m, float scale) {
    struct Object o;
    memset(&o, 0, sizeof o);
    o.type = TYPE_SPHERE;
    o.s.center = to3(h3scale(h3p(center), scale));
    o.s.radius = radius * scale;
    o.mat = m;
    return o;
}
struct Object rt_make_quad(const float origin[3], const float extentX[3], const float extentY[3],
                           struct MeshMaterial m, float scale) {
    struct Object o;
    memset(&o, 0, sizeof o);
    o.type = TYPE_QUAD;
    o.q.origin = to3(h3scale(h3p(origin), scale));
    o.q.extentX = to3(h3scale(h3p(extentX), scale));
    o.q.extentY = to3(h3scale(h3p(extentY), scale));
    o.mat = m;
    return o;
}

/* ---- camera: ContentView.swift:168-213 ----------------------------------- */
/* Swift's Float.pi is pi rounded toward zero, 0x1.921fb4p+1 (not the nearest float 0x1.921fb6p+1). */
static float from_degrees(float degrees) { return 0x1.921fb4p+1f * degrees / 180.0f; } /* :156-160 */

void rt_camera_setup(struct Uniforms *un, const float lookFrom_[3], const float lookAt_[3], const float vUp_[3],
                     float vfov, float defocusAngle, float focusDistance, int width, int height) {
    h3 lookFrom = h3p(lookFrom_), lookAt = h3p(lookAt_), vUp = h3p(vUp_);
    h3 cameraCenter = lookFrom;

    float theta = from_degrees(vfov);
    float h = tanf(theta / 2.0f);

    h3 w = h3normalize(h3sub(lookFrom, lookAt));
    h3 u = h3normalize(h3cross(vUp, w));
    h3 v = h3cross(w, u);

    float viewportHeight = 2.0f * h * focusDistance;
    /* Float(geometry.size.width / geometry.size.height): the quotient is formed in double (CGFloat). */
    float viewportWidth = viewportHeight * (float)((double)width / (double)height);

    h3 viewportU = h3scale(u, viewportWidth);
    h3 viewportV = h3scale(h3v(-v.x, -v.y, -v.z), viewportHeight);

    h3 pixelDeltaX = h3div(viewportU, (float)width);
    h3 pixelDeltaY = h3div(viewportV, (float)height);

    h3 viewportUpperLeft =
        h3sub(h3sub(h3sub(cameraCenter, h3scale(w, focusDistance)), h3div(viewportU, 2.0f)), h3div(viewportV, 2.0f));

    float defocus_radius = focusDistance * tanf(from_degrees(defocusAngle / 2.0f));
    un->defocusDiskU = to3(h3scale(u, defocus_radius));
    un->defocusDiskV = to3(h3scale(v, defocus_radius));
    un->defocusAngle = defocusAngle;

    h3 firstPixelPos = h3add(viewportUpperLeft, h3scale(h3add(pixelDeltaX, pixelDeltaY), 0.5f));

    un->pixelOrigin = to3(firstPixelPos);
    un->pixelDeltaX = to3(pixelDeltaX);
    un->pixelDeltaY = to3(pixelDeltaY);
    un->cameraCenter = to3(cameraCenter);
    un->viewportSize.x = (float)width;
    un->viewportSize.y = (float)height;
}

void rt_uniforms_reference_defaults(struct Uniforms *u, int width, int height) {
    const float from[3] = {278, 278, -800}, at[3] = {278, 278, 0}, up[3] = {0, 1, 0};
    memset(u, 0, sizeof *u);
    rt_camera_setup(u, from, at, up, 40.0f, 0.0f, 1.0f, width, height);
    u->sampleCount = 32;                                   /* ContentView.swift:210 */
    u->maxRayDepth = 10;                                   /* :211 */
    u->pixelSampleScale = 1.0f / (float)u->sampleCount;    /* :212 */
    u->totalSamples = 400;                                 /* maxIterations, :208 */
    u->globalIllumation = to4(0.1f, 0.1f, 0.1f, 1.0f);     /* :94 */
    u->tileX = 0; u->tileY = 0; u->tileWidth = width; u->tileHeight = height;
}

/* ---- scene RNG: splitmix64 -> float in [0,1) with 24 bits ----------------- */
typedef struct { uint64_t s; } scene_rng;
static uint64_t splitmix64(scene_rng *r) {
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float rnd(scene_rng *r) { return (float)(splitmix64(r) >> 40) * (1.0f / 16777216.0f); }

static int emit(struct Object *out, int cap, int n, struct Object o) {
    if (out && n < cap) out[n] = o;
    return n + 1;
}

int rt_scene_cornell(struct Object *out, int cap) {
    /* ContentView.swift:72-92 */
    int n = 0;
    struct MeshMaterial red = rt_material_lambertian(0.65f, 0.05f, 0.05f);
    red.reflection_fuzz = 0.8f; red.refraction_index = 0.75f;
    struct MeshMaterial white = rt_material_lambertian(0.73f, 0.73f, 0.73f);
    struct MeshMaterial green = rt_material_lambertian(0.12f, 0.45f, 0.15f);
    green.reflection_fuzz = 0.8f; green.refraction_index = 0.75f;
    struct MeshMaterial light = rt_material_default();
    light.emission = 15.0f; light.emission_color = to4(1, 1, 1, 1);
    const float o0[3] = {0, 0, 0}, o555x[3] = {555, 0, 0}, o555y[3] = {0, 555, 0}, o555z[3] = {0, 0, 555};
    const float ey[3] = {0, 555, 0}, ez[3] = {0, 0, 555}, ex[3] = {555, 0, 0};
    n = emit(out, cap, n, rt_make_quad(o0, ey, ez, red, 1.0f));
    n = emit(out, cap, n, rt_make_quad(o555x, ey, ez, green, 1.0f));
    n = emit(out, cap, n, rt_make_quad(o0, ex, ez, white, 1.0f));
    n = emit(out, cap, n, rt_make_quad(o555y, ex, ez, white, 1.0f));
    n = emit(out, cap, n, rt_make_quad(o555z, ex, ey, white, 1.0f));
    const float lo[3] = {213, 554, 227}, lx[3] = {130, 0, 0}, lz[3] = {0, 0, 105};
    n = emit(out, cap, n, rt_make_quad(lo, lx, lz, light, 1.0f));
    struct MeshMaterial glass = rt_material_dielectric(1.5f);
    const float bc[3] = {278, 90, 278};
    n = emit(out, cap, n, rt_make_sphere(bc, 90.0f, glass, 1.0f));
    return n;
}

int rt_scene_three_spheres(struct Object *out, int cap) {
    int n = 0;
    const float g[3] = {0, -100.5f, -1}, c[3] = {0, 0, -1.2f}, l[3] = {-1, 0, -1}, r[3] = {1, 0, -1};
    n = emit(out, cap, n, rt_make_sphere(g, 100.0f, rt_material_lambertian(0.8f, 0.8f, 0.0f), 1.0f));
    n = emit(out, cap, n, rt_make_sphere(c, 0.5f, rt_material_lambertian(0.1f, 0.2f, 0.5f), 1.0f));
    n = emit(out, cap, n, rt_make_sphere(l, 0.5f, rt_material_dielectric(1.5f), 1.0f));
    n = emit(out, cap, n, rt_make_sphere(l, 0.4f, rt_material_dielectric(1.0f / 1.5f), 1.0f));
    n = emit(out, cap, n, rt_make_sphere(r, 0.5f, rt_material_metal(0.8f, 0.6f, 0.2f, 1.0f), 1.0f));
    return n;
}

/* One small sphere of the cover-scene material mix at `center`. Draw order: m was drawn by the
 * caller; then albedo r,g,b (two draws each for lambertian), or albedo r,g,b then fuzz for metal. */
static struct Object cover_small(scene_rng *rng, float m, const float center[3], float radius) {
    if (m < 0.8f) {
        float r0 = rnd(rng), r1 = rnd(rng), g0 = rnd(rng), g1 = rnd(rng), b0 = rnd(rng), b1 = rnd(rng);
        return rt_make_sphere(center, radius, rt_material_lambertian(r0 * r1, g0 * g1, b0 * b1), 1.0f);
    } else if (m < 0.95f) {
        float r0 = rnd(rng), g0 = rnd(rng), b0 = rnd(rng), f0 = rnd(rng);
        return rt_make_sphere(center, radius,
                              rt_material_metal(0.5f + 0.5f * r0, 0.5f + 0.5f * g0, 0.5f + 0.5f * b0, 0.5f * f0), 1.0f);
    }
    return rt_make_sphere(center, radius, rt_material_dielectric(1.5f), 1.0f);
}

int rt_scene_cover(struct Object *out, int cap, uint64_t seed) {
    scene_rng rng = {seed};
    int n = 0;
    const float g[3] = {0, -1000, 0};
    n = emit(out, cap, n, rt_make_sphere(g, 1000.0f, rt_material_lambertian(0.5f, 0.5f, 0.5f), 1.0f));
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            float m = rnd(&rng);
            float dx = rnd(&rng);
            float dz = rnd(&rng);
            float center[3] = {(float)a + 0.9f * dx, 0.2f, (float)b + 0.9f * dz};
            h3 d = h3sub(h3p(center), h3v(4.0f, 0.2f, 0.0f));
            if (sqrtf(h3dot(d, d)) > 0.9f) n = emit(out, cap, n, cover_small(&rng, m, center, 0.2f));
        }
    }
    const float c1[3] = {0, 1, 0}, c2[3] = {-4, 1, 0}, c3[3] = {4, 1, 0};
    n = emit(out, cap, n, rt_make_sphere(c1, 1.0f, rt_material_dielectric(1.5f), 1.0f));
    n = emit(out, cap, n, rt_make_sphere(c2, 1.0f, rt_material_lambertian(0.4f, 0.2f, 0.1f), 1.0f));
    n = emit(out, cap, n, rt_make_sphere(c3, 1.0f, rt_material_metal(0.7f, 0.6f, 0.5f, 0.0f), 1.0f));
    return n;
}

int rt_scene_nested_glass(struct Object *out, int cap, int shells) {
    int n = 0;
    const float g[3] = {0, -100.5f, -1}, c[3] = {0, 0, -1};
    n = emit(out, cap, n, rt_make_sphere(g, 100.0f, rt_material_lambertian(0.8f, 0.8f, 0.0f), 1.0f));
    for (int k = 0; k < shells; ++k) {
        float radius = 0.5f * (1.0f - (float)k / (float)shells);
        float ior = (k & 1) ? (1.0f / 1.5f) : 1.5f;
        n = emit(out, cap, n, rt_make_sphere(c, radius, rt_material_dielectric(ior), 1.0f));
    }
    return n;
}

int rt_scene_random_spheres(struct Object *out, int cap, int count, float extent, uint64_t seed) {
    scene_rng rng = {seed};
    int n = 0;
    const float g[3] = {0, -1000, 0};
    n = emit(out, cap, n, rt_make_sphere(g, 1000.0f, rt_material_lambertian(0.5f, 0.5f, 0.5f), 1.0f));
    for (int i = 0; i < count; ++i) {
        float m = rnd(&rng);
        float fx = rnd(&rng);
        float fz = rnd(&rng);
        float center[3] = {(2.0f * fx - 1.0f) * extent, 0.1f, (2.0f * fz - 1.0f) * extent};
        n = emit(out, cap, n, cover_small(&rng, m, center, 0.1f));
    }
    return n;
}

int rt_config_uniforms(const char *name, struct Uniforms *u, int width, int height, int spp, int depth) {
    const float up[3] = {0, 1, 0};
    memset(u, 0, sizeof *u);
    u->globalIllumation = to4(0.7f, 0.8f, 1.0f, 1.0f);
    if (strcmp(name, "three_spheres") == 0) {
        const float from[3] = {-2, 2, 1}, at[3] = {0, 0, -1};
        rt_camera_setup(u, from, at, up, 20.0f, 0.0f, 1.0f, width, height);
    } else if (strcmp(name, "cover") == 0) {
        const float from[3] = {13, 2, 3}, at[3] = {0, 0, 0};
        rt_camera_setup(u, from, at, up, 20.0f, 0.6f, 10.0f, width, height);
    } else if (strcmp(name, "nested_glass") == 0) {
        const float from[3] = {0, 0.25f, 1}, at[3] = {0, 0, -1};
        rt_camera_setup(u, from, at, up, 40.0f, 0.0f, 1.0f, width, height);
    } else if (strcmp(name, "random_spheres") == 0) {
        const float from[3] = {13, 2, 3}, at[3] = {0, 0, 0};
        rt_camera_setup(u, from, at, up, 40.0f, 0.0f, 10.0f, width, height);
    } else if (strcmp(name, "cornell") == 0) {
        rt_uniforms_reference_defaults(u, width, height);
    } else {
        return -1;
    }
    u->sampleCount = spp;
    u->maxRayDepth = depth;
    u->pixelSampleScale = 1.0f / (float)spp;
    u->currentSample = 0;
    u->totalSamples = 1;
    u->tileX = 0; u->tileY = 0; u->tileWidth = width; u->tileHeight = height;
    return 0;
}

/* ---- tiles: ContentView.swift:380-401 ------------------------------------ */
int rt_generate_tiles(int width, int height, int tileSize, struct rt_tile *out, int cap) {
    int tilesX = (width + tileSize - 1) / tileSize;
    int tilesY = (height + tileSize - 1) / tileSize;
    int n = 0;
    for (int y = 0; y < tilesY; ++y) {
        for (int x = 0; x < tilesX; ++x) {
            struct rt_tile t;
            t.x = x * tileSize;
            t.y = y * tileSize;
            t.width = tileSize < width - t.x ? tileSize : width - t.x;
            t.height = tileSize < height - t.y ? tileSize : height - t.y;
            if (out && n < cap) out[n] = t;
            ++n;
        }
    }
    return n;
}

/* ---- image writers -------------------------------------------------------- */
static unsigned char to_u8(float v) {
    if (!(v > 0.0f)) return 0;
    if (v > 1.0f) v = 1.0f;
    return (unsigned char)rintf(v * 255.0f);
}

int rt_write_ppm(const char *path, const float *rgba, int width, int height) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    unsigned char *row = (unsigned char *)malloc((size_t)width * 3);
    if (!row) { fclose(f); return -1; }
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const float *p = rgba + 4 * ((size_t)y * width + x);
            row[3 * x + 0] = to_u8(p[0]);
            row[3 * x + 1] = to_u8(p[1]);
            row[3 * x + 2] = to_u8(p[2]);
        }
        fwrite(row, 1, (size_t)width * 3, f);
    }
    free(row);
    fclose(f);
    return 0;
}

int rt_write_pfm(const char *path, const float *rgba, int width, int height) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
    for (int y = height - 1; y >= 0; --y)
        for (int x = 0; x < width; ++x) fwrite(rgba + 4 * ((size_t)y * width + x), sizeof(float), 3, f);
    fclose(f);
    return 0;
}
