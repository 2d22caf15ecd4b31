/*
 * rt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference's path-tracing compute kernel
 * (reference: "Ray Tracer/Shaders.metal" and "Ray Tracer/Lib.metal"), kept in
 * the reference's own shape: AoS `Object` array, one `Object` copied per
 * ray-object test, the same function boundaries, the same operation order.
 * It exists so that the HIP path can be checked against it (tests/,
 * __graft_entry__.smoke()) and so that bench.py can time it on the host
 * cores as the CPU baseline.  Nothing in the shipped library links it.
 *
 * PARITY STATUS: "parity unpinned" by the reference -- the reference holds no
 * test vectors for this path, and its sources need <metal_stdlib> and
 * <simd/simd.h>, which this image lacks, so they are not compiled here.  The
 * only anchors are known answers recorded when the reference's functions were
 * run as host C++ during the survey of the reference (pcg_hash and
 * random_double values, image statistics of the three-sphere scene); those
 * are checked in tests/test_oracle_kat.py.
 *
 * Semantics that matter (all arithmetic is IEEE binary32, no contraction):
 *  - every literal of the Metal sources is `float` (MSL has no double);
 *  - Metal built-ins are taken at their Metal Shading Language specification
 *    formulas: dot = x*x' + y*y' + z*z' (left to right), length = sqrt(dot),
 *    normalize = v / length(v), fract = min(x - floor(x), 0x1.fffffep-1f),
 *    clamp = min(max(x, lo), hi), reflect = I - 2*dot(N,I)*N,
 *    refract: k = 1 - eta*eta*(1 - dot(N,I)^2); k < 0 ? 0 : eta*I - (eta*dot(N,I) + sqrt(k))*N,
 *    cross the usual determinant form, pow = C powf;
 *  - texture writes outside the texture are dropped, as Metal does.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp (see oracle/Makefile).  Do not
 * build with -ffast-math or FMA contraction.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/rt_abi.h"

#define RTO_API __attribute__((visibility("default")))

typedef struct { float x, y; } f2;
typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;
typedef struct { uint32_t x, y; } u2;

/* ---- Metal built-ins at their specification formulas ------------------- */
static inline f3 v3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 ld3(rt_float3 a) { return v3(a.x, a.y, a.z); }
static inline f4 ld4(rt_float4 a) { f4 r = {a.x, a.y, a.z, a.w}; return r; }
static inline f3 add3(f3 a, f3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 neg3(f3 a) { return v3(-a.x, -a.y, -a.z); }
static inline f3 smul3(float s, f3 a) { return v3(s * a.x, s * a.y, s * a.z); }
static inline f3 sdiv3(f3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float length3(f3 a) { return sqrtf(dot3(a, a)); }
static inline f3 normalize3(f3 a) { return sdiv3(a, length3(a)); }
static inline f3 cross3(f3 a, f3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float fract1(float x) { return fminf(x - floorf(x), 0x1.fffffep-1f); }
static inline float clamp1(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline f3 reflect3(f3 i, f3 n) { return sub3(i, smul3(2.0f * dot3(n, i), n)); }
static inline f3 refract3(f3 i, f3 n, float eta) {
    float d = dot3(n, i);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return v3(0.0f, 0.0f, 0.0f);
    return sub3(smul3(eta, i), smul3(eta * d + sqrtf(k), n));
}
static inline f4 add4(f4 a, f4 b) { f4 r = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; return r; }
static inline f4 mul4(f4 a, f4 b) { f4 r = {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; return r; }
static inline f4 smul4(f4 a, float s) { f4 r = {a.x * s, a.y * s, a.z * s, a.w * s}; return r; }

/* ---- Lib.metal:12-57 ---------------------------------------------------- */
typedef struct { float min, max; } Interval;
static inline int interval_contains(Interval i, float x) { return i.min <= x && x <= i.max; }   /* Lib.metal:20-22 */
static inline int interval_surrounds(Interval i, float x) { return i.min < x && x < i.max; }    /* Lib.metal:24-26 */
static inline float interval_clamp(Interval i, float x) { return clamp1(x, i.min, i.max); }     /* Lib.metal:28-30 */

typedef struct {            /* Lib.metal:33-42; field order is {direction, origin} */
    f3 direction;
    f3 origin;
    Interval distance;
} Ray;
static inline f3 ray_at(const Ray *r, float t) { return add3(r->origin, smul3(t, r->direction)); }

typedef struct {            /* Lib.metal:44-57 */
    f3 point;
    f3 normal;
    float distance;
    int hit_front;
} HitInfo;
static inline void apply_normals(HitInfo *h, const Ray *r, f3 outward) {
    h->hit_front = dot3(r->direction, outward) < 0.0f;
    h->normal = h->hit_front ? outward : neg3(outward);
}

/* ---- RNG: Lib.metal:59-72 ---------------------------------------------- */
static inline uint32_t pcg_hash(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

static inline float random_double(f2 *seed) {
    u2 iseed = {(uint32_t)(seed->x * 65536.0f), (uint32_t)(seed->y * 65536.0f)};
    uint32_t hash = pcg_hash(iseed.x ^ pcg_hash(iseed.y));
    seed->x = fract1(seed->x + 0.618034f);
    seed->y = fract1(seed->y + 0.381966f);
    return (float)hash / 4294967296.0f;
}

/* Lib.metal:78-82 */
static inline f3 random_square(f2 *seed) {
    float x = random_double(seed) - 0.5f;
    float y = random_double(seed) - 0.5f;
    return v3(x, y, 0.0f);
}

/* Lib.metal:84-92 */
static inline f3 random_in_unit_disk(f2 *seed) {
    float x, y;
    do {
        x = 2.0f * random_double(seed) - 1.0f;
        y = 2.0f * random_double(seed) - 1.0f;
    } while (x * x + y * y >= 1.0f);
    return v3(x, y, 0.0f);
}

/* Lib.metal:94-97 */
static inline f3 defocusDiskSample(const Uniforms *u, f2 *seed) {
    f3 p = random_in_unit_disk(seed);
    return add3(add3(ld3(u->cameraCenter), smul3(p.x, ld3(u->defocusDiskU))), smul3(p.y, ld3(u->defocusDiskV)));
}

/* Lib.metal:99-106 */
static inline Ray getRay(int i, int j, const Uniforms *u, f2 *seed) {
    f3 offset = random_square(seed);
    f3 pixel_sample = add3(add3(ld3(u->pixelOrigin), smul3((float)i + offset.x, ld3(u->pixelDeltaX))),
                           smul3((float)j + offset.y, ld3(u->pixelDeltaY)));
    f3 ray_origin = (u->defocusAngle <= 0.0f) ? ld3(u->cameraCenter) : defocusDiskSample(u, seed);
    f3 ray_dir = normalize3(sub3(pixel_sample, ray_origin));
    Ray r;
    r.direction = ray_dir;
    r.origin = ray_origin;
    r.distance.min = 0.001f;
    r.distance.max = 1000.0f;
    return r;
}

/* Lib.metal:117-132 */
static inline f3 random_unit_vec(f2 *seed) {
    float x1, x2, w;
    do {
        x1 = 2.0f * random_double(seed) - 1.0f;
        x2 = 2.0f * random_double(seed) - 1.0f;
        w = x1 * x1 + x2 * x2;
    } while (w >= 1.0f);
    float multiplier = 2.0f * sqrtf(1.0f - w);
    return v3(x1 * multiplier, x2 * multiplier, 1.0f - 2.0f * w);
}

/* Lib.metal:144-150 */
static inline float linear_to_gamma(float linear) {
    if (linear > 0.0f) return sqrtf(linear);
    return 0.0f;
}

/* Lib.metal:152-155 */
static inline int isVecNearZero(f3 v) {
    float s = 1e-8f;
    return (fabsf(v.x) < s) && (fabsf(v.y) < s) && (fabsf(v.z) < s);
}

/* Lib.metal:157-167.  The brace initialiser fills Ray{direction, origin} in
 * declaration order, so direction = hit.point and origin = scatter_direction. */
static int lambertianScatter(const MeshMaterial *m, const Ray *r, const HitInfo *hit, f4 *attenuation,
                             Ray *scattered, f2 *seed) {
    (void)r;
    f3 scatter_direction = add3(hit->normal, random_unit_vec(seed));
    if (isVecNearZero(scatter_direction)) scatter_direction = hit->normal;
    scattered->direction = hit->point;
    scattered->origin = scatter_direction;
    scattered->distance.min = 0.001f;
    scattered->distance.max = 1000.0f;
    *attenuation = ld4(m->albedo);
    return 1;
}

/* Lib.metal:169-175 */
static int metalScatter(const MeshMaterial *m, const Ray *r, const HitInfo *hit, f4 *attenuation, Ray *scattered,
                        f2 *seed) {
    f3 reflected = reflect3(r->direction, hit->normal);
    reflected = add3(normalize3(reflected), smul3(m->reflection_fuzz, random_unit_vec(seed)));
    scattered->direction = normalize3(reflected);
    scattered->origin = hit->point;
    scattered->distance.min = 0.001f;
    scattered->distance.max = 1000.0f;
    *attenuation = ld4(m->albedo);
    return dot3(scattered->direction, hit->normal) > 0.0f;
}

/* Lib.metal:177-181 */
static inline float reflectance(float cosine, float refraction_index) {
    float r0 = (1.0f - refraction_index) / (1.0f + refraction_index);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powf(1.0f - cosine, 5.0f);
}

/* Lib.metal:183-201 */
static int dielectricScatter(const MeshMaterial *m, const Ray *r, const HitInfo *hit, f4 *attenuation,
                             Ray *scattered, f2 *seed) {
    f4 one = {1.0f, 1.0f, 1.0f, 1.0f};
    *attenuation = one;
    float ri = hit->hit_front ? (1.0f / m->refraction_index) : m->refraction_index;
    f3 unit_dir = normalize3(r->direction);
    float cos_theta = fminf(dot3(neg3(unit_dir), hit->normal), 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    int cannot_refract = ri * sin_theta > 1.0f;
    f3 direction;
    if (cannot_refract || reflectance(cos_theta, ri) > random_double(seed)) {
        direction = reflect3(unit_dir, hit->normal);
    } else {
        direction = refract3(unit_dir, hit->normal, ri);
    }
    scattered->direction = direction;
    scattered->origin = hit->point;
    scattered->distance.min = 0.001f;
    scattered->distance.max = 1000.0f;
    return 1;
}

/* Lib.metal:203-213 */
static int materialScatters(const MeshMaterial *m, const Ray *r, const HitInfo *hit, f4 *attenuation,
                            Ray *scattered, f2 *seed) {
    if (m->type == LAMBIERTIAN) return lambertianScatter(m, r, hit, attenuation, scattered, seed);
    else if (m->type == REFLECTEE) return metalScatter(m, r, hit, attenuation, scattered, seed);
    else if (m->type == DIELECTRIC) return dielectricScatter(m, r, hit, attenuation, scattered, seed);
    else return 0;
}

/* Lib.metal:215-246 */
static int hitSphere(Sphere s, const Ray *r, HitInfo *hit) {
    f3 center = ld3(s.center);
    f3 oc = sub3(r->origin, center);
    float a = dot3(r->direction, r->direction);
    float b = dot3(oc, r->direction);
    float c = dot3(oc, oc) - s.radius * s.radius;

    float discriminant = b * b - a * c;
    if (discriminant < 0.0f) return 0;
    if (fabsf(a) < 1e-6f) return 0;

    float sqrtd = sqrtf(discriminant);
    float root = (-b - sqrtd) / a;
    if (!interval_surrounds(r->distance, root)) {
        root = (-b + sqrtd) / a;
        if (!interval_surrounds(r->distance, root)) return 0;
    }
    hit->distance = root;
    hit->point = ray_at(r, root);
    f3 normals = sdiv3(sub3(hit->point, center), s.radius);
    apply_normals(hit, r, normals);
    return 1;
}

/* Lib.metal:248-280 */
static int hitPlane(Quad q, const Ray *r, HitInfo *hit) {
    f3 ex = ld3(q.extentX), ey = ld3(q.extentY), qo = ld3(q.origin);
    f3 n = cross3(ex, ey);
    f3 normal = normalize3(n);
    float denom = dot3(normal, r->direction);
    if (fabsf(denom) < 1e-8f) return 0;
    float t = dot3(normal, sub3(qo, r->origin)) / denom;
    if (!interval_contains(r->distance, t)) return 0;
    f3 inter = ray_at(r, t);
    f3 toPoint = sub3(inter, qo);
    float dotU = dot3(toPoint, normalize3(ex));
    float dotV = dot3(toPoint, normalize3(ey));
    float uLen = length3(ex);
    float vLen = length3(ey);
    if (dotU < 0.0f || dotU > uLen || dotV < 0.0f || dotV > vLen) return 0;
    hit->distance = t;
    hit->point = inter;
    apply_normals(hit, r, normal);
    return 1;
}

/* Lib.metal:283-291; the Object is passed by value as in the reference. */
static int hitObject(Object obj, const Ray *r, HitInfo *info) {
    if (obj.type == TYPE_SPHERE && hitSphere(obj.s, r, info)) return 1;
    else if (obj.type == TYPE_QUAD && hitPlane(obj.q, r, info)) return 1;
    else return 0;
}

typedef struct { uint64_t segments; uint64_t tests; uint64_t scatters[3]; } Stats;

/* Shaders.metal:13-32 */
static int world_hit(const Object *objs, int objectCount, Ray *r, HitInfo *hit, int *hitObjectIndex) {
    HitInfo temp_hit;
    int hit_anything = 0;
    float closest_so_far = r->distance.max;
    for (int i = 0; i < objectCount; ++i) {
        Object obj = objs[i];
        if (hitObject(obj, r, &temp_hit)) {
            if (temp_hit.distance < closest_so_far) {
                hit_anything = 1;
                closest_so_far = temp_hit.distance;
                *hit = temp_hit;
                *hitObjectIndex = i;
                r->distance.max = closest_so_far;
            }
        }
    }
    return hit_anything;
}

/* Shaders.metal:34-69 */
static f4 color_ray(const Object *objs, int objectCount, Ray r, f2 *seed, int maxDepth, f4 baseSkyColor, Stats *st) {
    f4 accumulatedColor = {1.0f, 1.0f, 1.0f, 1.0f};
    f4 finalColor = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int depth = 0; depth < maxDepth; ++depth) {
        r.distance.min = 0.001f;
        r.distance.max = 3.402823466e+38f; /* MAXFLOAT */
        HitInfo info;
        int hitObjectIndex = -1;
        if (st) { st->segments++; st->tests += (uint64_t)(objectCount > 0 ? objectCount : 0); }
        if (!world_hit(objs, objectCount, &r, &info, &hitObjectIndex)) {
            finalColor = add4(finalColor, mul4(accumulatedColor, baseSkyColor));
            break;
        }
        const Object obj = objs[hitObjectIndex];
        f4 colorFromEmission = {0.0f, 0.0f, 0.0f, 1.0f};
        if (obj.mat.emission > 0.0f) colorFromEmission = smul4(ld4(obj.mat.emission_color), obj.mat.emission);
        finalColor = add4(finalColor, mul4(accumulatedColor, colorFromEmission));
        Ray scattered;
        f4 attenuation;
        if (!materialScatters(&obj.mat, &r, &info, &attenuation, &scattered, seed)) break;
        if (st && obj.mat.type >= 0 && obj.mat.type < 3) st->scatters[obj.mat.type]++;
        accumulatedColor = mul4(accumulatedColor, attenuation);
        r = scattered;
    }
    return finalColor;
}

/* Texture: rgba32Float, row-major, 4 floats per texel; out-of-range writes are dropped. */
typedef struct { float *data; uint32_t width, height; } Texture;
static inline f4 tex_read(const Texture *t, uint32_t x, uint32_t y) {
    f4 r = {0.0f, 0.0f, 0.0f, 0.0f};
    if (x < t->width && y < t->height) memcpy(&r, t->data + 4 * ((size_t)y * t->width + x), sizeof r);
    return r;
}
static inline void tex_write(const Texture *t, f4 c, uint32_t x, uint32_t y) {
    if (x < t->width && y < t->height) memcpy(t->data + 4 * ((size_t)y * t->width + x), &c, sizeof c);
}

/* Shaders.metal:71-85 */
static void write_color(const Texture *texture, uint32_t px, uint32_t py, f4 color) {
    Interval i = {0.0f, 1.0f};
    float r = interval_clamp(i, isnan(color.x) ? 0.0f : color.x);
    float g = interval_clamp(i, isnan(color.y) ? 0.0f : color.y);
    float b = interval_clamp(i, isnan(color.z) ? 0.0f : color.z);
    float a = interval_clamp(i, isnan(color.w) ? 1.0f : color.w);
    r = linear_to_gamma(r);
    g = linear_to_gamma(g);
    b = linear_to_gamma(b);
    a = linear_to_gamma(a);
    f4 result = {r, g, b, a};
    tex_write(texture, result, px, py);
}

/* Shaders.metal:87-89 */
static void error_write(const Texture *texture, uint32_t x, uint32_t y) {
    f4 yellow = {1.0f, 1.0f, 0.0f, 1.0f};
    tex_write(texture, yellow, x, y);
}

/* Shaders.metal:121-126: the per-sample seed. */
static inline f2 sample_seed(uint32_t px, uint32_t py, int currentSample, int rayIndex) {
    f2 seed;
    seed.x = (float)(px * 1973u + py * 9277u + (uint32_t)(currentSample * 26699) + (uint32_t)(rayIndex * 7919)) / 65536.0f;
    seed.y = (float)(py * 4513u + px * 6389u + (uint32_t)(currentSample * 30103) + (uint32_t)(rayIndex * 8191)) / 65536.0f;
    seed.x = fract1(seed.x + 0.618034f);
    seed.y = fract1(seed.y + 0.381966f);
    return seed;
}

/* Shaders.metal:120-140, one trip of the sample loop.  Returns 0 when the ray
 * was skipped as invalid (Shaders.metal:130-132), else 1 with *out filled. */
static int one_sample(const Uniforms *u, const Object *objs, uint32_t px, uint32_t py, int rayIndex, f4 *out, Stats *st) {
    f2 seed = sample_seed(px, py, u->currentSample, rayIndex);
    Ray r = getRay((int)px, (int)py, u, &seed);
    if (length3(r.direction) < 1e-6f || isnan(r.direction.x) || isnan(r.direction.y) || isnan(r.direction.z)) return 0;
    int safeMaxDepth = u->maxRayDepth;
    f4 rayColor = color_ray(objs, u->objCount, r, &seed, safeMaxDepth, ld4(u->globalIllumation), st);
    rayColor.x = clamp1(rayColor.x, 0.0f, 10.0f);
    rayColor.y = clamp1(rayColor.y, 0.0f, 10.0f);
    rayColor.z = clamp1(rayColor.z, 0.0f, 10.0f);
    *out = rayColor;
    return 1;
}

/* Shaders.metal:91-161: one GPU thread. */
static void computeShader(const Texture *outputTexture, const Uniforms *uniforms, const Object *objs, uint32_t gidx,
                          uint32_t gidy, Stats *st) {
    uint32_t width = outputTexture->width;
    uint32_t height = outputTexture->height;
    uint32_t pixelPosX = (uint32_t)uniforms->tileX + gidx;
    uint32_t pixelPosY = (uint32_t)uniforms->tileY + gidy;

    if (gidx >= (uint32_t)uniforms->tileWidth || gidy >= (uint32_t)uniforms->tileHeight) {
        error_write(outputTexture, gidx, gidy);
        return;
    }
    if (pixelPosX >= width || pixelPosY >= height) {
        error_write(outputTexture, gidx, gidy);
        return;
    }
    if (uniforms->currentSample > 10000) {
        error_write(outputTexture, gidx, gidy);
        return;
    }

    const int raysPerPass = uniforms->sampleCount;
    f4 passAccumulator = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int rayIndex = 0; rayIndex < raysPerPass; rayIndex++) {
        f4 rayColor;
        if (!one_sample(uniforms, objs, pixelPosX, pixelPosY, rayIndex, &rayColor, st)) continue;
        passAccumulator = add4(passAccumulator, rayColor);
    }
    float n = (float)raysPerPass;
    passAccumulator.x /= n;
    passAccumulator.y /= n;
    passAccumulator.z /= n;
    passAccumulator.w /= n;

    f4 existingColor = tex_read(outputTexture, pixelPosX, pixelPosY);
    existingColor.x = existingColor.x * existingColor.x;
    existingColor.y = existingColor.y * existingColor.y;
    existingColor.z = existingColor.z * existingColor.z;

    f4 finalColor;
    if (uniforms->currentSample == 0) {
        finalColor = passAccumulator;
    } else {
        float totalRaysSoFar = (float)(uniforms->currentSample * raysPerPass + raysPerPass);
        float passWeight = (float)raysPerPass / totalRaysSoFar;
        finalColor.x = existingColor.x * (1.0f - passWeight) + passAccumulator.x * passWeight;
        finalColor.y = existingColor.y * (1.0f - passWeight) + passAccumulator.y * passWeight;
        finalColor.z = existingColor.z * (1.0f - passWeight) + passAccumulator.z * passWeight;
        finalColor.w = 1.0f;
    }
    write_color(outputTexture, pixelPosX, pixelPosY, finalColor);
}

/* ======================================================================== */
/* Exported test surface                                                     */
/* ======================================================================== */

#ifdef _OPENMP
#include <omp.h>
/* Number of threads the OpenMP loops below will use. */
RTO_API int rto_num_threads(void) { return omp_get_max_threads(); }
#else
RTO_API int rto_num_threads(void) { return 1; }
#endif

RTO_API uint32_t rto_pcg_hash(uint32_t x) { return pcg_hash(x); }

RTO_API float rto_random_double(float seed[2]) {
    f2 s = {seed[0], seed[1]};
    float r = random_double(&s);
    seed[0] = s.x; seed[1] = s.y;
    return r;
}

/* kind: 0 random_square, 1 random_in_unit_disk, 2 random_unit_vec */
RTO_API void rto_random_vec(int kind, float seed[2], float out[3]) {
    f2 s = {seed[0], seed[1]};
    f3 v = kind == 0 ? random_square(&s) : kind == 1 ? random_in_unit_disk(&s) : random_unit_vec(&s);
    seed[0] = s.x; seed[1] = s.y;
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}

RTO_API void rto_sample_seed(uint32_t px, uint32_t py, int currentSample, int rayIndex, float out[2]) {
    f2 s = sample_seed(px, py, currentSample, rayIndex);
    out[0] = s.x; out[1] = s.y;
}

/* Camera ray of sample (px, py, u->currentSample, rayIndex): dir[3], origin[3], seed after getRay. */
RTO_API void rto_get_ray(const Uniforms *u, uint32_t px, uint32_t py, int rayIndex, float dir[3], float origin[3],
                         float seed_out[2]) {
    f2 seed = sample_seed(px, py, u->currentSample, rayIndex);
    Ray r = getRay((int)px, (int)py, u, &seed);
    dir[0] = r.direction.x; dir[1] = r.direction.y; dir[2] = r.direction.z;
    origin[0] = r.origin.x; origin[1] = r.origin.y; origin[2] = r.origin.z;
    seed_out[0] = seed.x; seed_out[1] = seed.y;
}

/* hitObject on one object.  out = {t, px,py,pz, nx,ny,nz, hit_front}. Returns hit flag. */
RTO_API int rto_hit_object(const Object *obj, const float dir[3], const float origin[3], float tmin, float tmax,
                           float out[8]) {
    Ray r;
    r.direction = v3(dir[0], dir[1], dir[2]);
    r.origin = v3(origin[0], origin[1], origin[2]);
    r.distance.min = tmin; r.distance.max = tmax;
    HitInfo h;
    memset(&h, 0, sizeof h);
    int ok = hitObject(*obj, &r, &h);
    if (ok) {
        out[0] = h.distance; out[1] = h.point.x; out[2] = h.point.y; out[3] = h.point.z;
        out[4] = h.normal.x; out[5] = h.normal.y; out[6] = h.normal.z; out[7] = (float)h.hit_front;
    }
    return ok;
}

/* world_hit with distance = {0.001, MAXFLOAT}.  out as rto_hit_object; returns object index or -1. */
RTO_API int rto_world_hit(const Object *objs, int n, const float dir[3], const float origin[3], float out[8]) {
    Ray r;
    r.direction = v3(dir[0], dir[1], dir[2]);
    r.origin = v3(origin[0], origin[1], origin[2]);
    r.distance.min = 0.001f; r.distance.max = 3.402823466e+38f;
    HitInfo h;
    memset(&h, 0, sizeof h);
    int idx = -1;
    if (!world_hit(objs, n, &r, &h, &idx)) return -1;
    out[0] = h.distance; out[1] = h.point.x; out[2] = h.point.y; out[3] = h.point.z;
    out[4] = h.normal.x; out[5] = h.normal.y; out[6] = h.normal.z; out[7] = (float)h.hit_front;
    return idx;
}

/* materialScatters.  hit = {t, px,py,pz, nx,ny,nz, hit_front}; out = {att rgba, dir xyz, origin xyz}. */
RTO_API int rto_scatter(const MeshMaterial *m, const float dir[3], const float origin[3], const float hit[8],
                        float seed[2], float out[10]) {
    Ray r, sc;
    r.direction = v3(dir[0], dir[1], dir[2]);
    r.origin = v3(origin[0], origin[1], origin[2]);
    r.distance.min = 0.001f; r.distance.max = 3.402823466e+38f;
    HitInfo h;
    h.distance = hit[0]; h.point = v3(hit[1], hit[2], hit[3]); h.normal = v3(hit[4], hit[5], hit[6]);
    h.hit_front = hit[7] != 0.0f;
    f2 s = {seed[0], seed[1]};
    f4 att = {0, 0, 0, 0};
    memset(&sc, 0, sizeof sc);
    int ok = materialScatters(m, &r, &h, &att, &sc, &s);
    seed[0] = s.x; seed[1] = s.y;
    out[0] = att.x; out[1] = att.y; out[2] = att.z; out[3] = att.w;
    out[4] = sc.direction.x; out[5] = sc.direction.y; out[6] = sc.direction.z;
    out[7] = sc.origin.x; out[8] = sc.origin.y; out[9] = sc.origin.z;
    return ok;
}

/* One sample's radiance after the [0,10] clamp (Shaders.metal:121-137).
 * Returns 0 for a skipped (invalid) ray.  stats (optional) = {segments, tests}. */
RTO_API int rto_sample(const Uniforms *u, const Object *objs, uint32_t px, uint32_t py, int rayIndex, float out[4],
                       uint64_t *stats) {
    Stats st;
    memset(&st, 0, sizeof st);
    f4 c = {0, 0, 0, 0};
    int ok = one_sample(u, objs, px, py, rayIndex, &c, &st);
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
    if (stats) { stats[0] += st.segments; stats[1] += st.tests; }
    return ok;
}

/* Linear (pre-mean, pre-gamma) sum over rayIndex in [ray_begin, ray_end), sequential
 * float adds in rayIndex order, for every pixel of the w x h image.  sums: w*h*4 floats. */
RTO_API void rto_accumulate(const Uniforms *u, const Object *objs, int w, int h, int ray_begin, int ray_end,
                            float *sums, uint64_t *stats) {
    uint64_t seg = 0, tests = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : seg, tests)
    for (int y = 0; y < h; ++y) {
        Stats st;
        memset(&st, 0, sizeof st);
        for (int x = 0; x < w; ++x) {
            f4 acc = {0, 0, 0, 0};
            for (int k = ray_begin; k < ray_end; ++k) {
                f4 c;
                if (one_sample(u, objs, (uint32_t)x, (uint32_t)y, k, &c, &st)) acc = add4(acc, c);
            }
            memcpy(sums + 4 * ((size_t)y * w + x), &acc, sizeof acc);
        }
        seg += st.segments; tests += st.tests;
    }
    if (stats) { stats[0] += seg; stats[1] += tests; }
}

/* One Metal dispatch of computeShader over the tile in *u (Shaders.metal:91-161 driven as
 * ContentView.swift:698-705 does).  reference_grid != 0 launches ceil(tile/8)*8 threads per
 * axis, so out-of-tile threads run the yellow error_write branch like the reference's
 * 8x8 threadgroups do; 0 launches exactly tileWidth x tileHeight threads.
 * stride > 1 runs only threads with gid % stride == 0 (used for sub-sampled CPU timing).
 * stats (optional) accumulates {segments, ray-object tests, lambert, metal, dielectric scatters}. */
RTO_API void rto_dispatch(float *tex, int tex_w, int tex_h, const Uniforms *u, const Object *objs, int reference_grid,
                          int stride, uint64_t *stats) {
    Texture t = {tex, (uint32_t)tex_w, (uint32_t)tex_h};
    int gw = u->tileWidth, gh = u->tileHeight;
    if (reference_grid) { gw = (gw + 7) / 8 * 8; gh = (gh + 7) / 8 * 8; }
    if (stride < 1) stride = 1;
    uint64_t seg = 0, tests = 0, s0 = 0, s1 = 0, s2 = 0;
    /* Threads outside the tile only write yellow at their own gid, which no in-tile thread of a
     * tile at (0,0) touches and which, for other tiles, lies outside the tile: run them first so
     * that, as on the GPU where order is undefined, the deterministic part (in-tile pixels) wins
     * any overlap with error pixels of the same dispatch. */
    if (reference_grid) {
        for (int gy = 0; gy < gh; ++gy)
            for (int gx = 0; gx < gw; ++gx)
                if (gx >= u->tileWidth || gy >= u->tileHeight) computeShader(&t, u, objs, (uint32_t)gx, (uint32_t)gy, 0);
    }
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : seg, tests, s0, s1, s2)
    for (int gy = 0; gy < u->tileHeight; gy += stride) {
        Stats st;
        memset(&st, 0, sizeof st);
        for (int gx = 0; gx < u->tileWidth; gx += stride) computeShader(&t, u, objs, (uint32_t)gx, (uint32_t)gy, &st);
        seg += st.segments; tests += st.tests; s0 += st.scatters[0]; s1 += st.scatters[1]; s2 += st.scatters[2];
    }
    if (stats) { stats[0] += seg; stats[1] += tests; stats[2] += s0; stats[3] += s1; stats[4] += s2; }
}

/* First-hit buffer for sample (currentSample = u->currentSample, rayIndex = 0): per pixel
 * out5 = {t, nx, ny, nz} and id (object index or -1).  Uses getRay + world_hit exactly as
 * the first bounce of color_ray does. */
RTO_API void rto_first_hit(const Uniforms *u, const Object *objs, int w, int h, float *out4, int *ids) {
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            f2 seed = sample_seed((uint32_t)x, (uint32_t)y, u->currentSample, 0);
            Ray r = getRay(x, y, u, &seed);
            r.distance.min = 0.001f; r.distance.max = 3.402823466e+38f;
            HitInfo h_;
            memset(&h_, 0, sizeof h_);
            int idx = -1;
            size_t p = (size_t)y * w + x;
            if (world_hit(objs, u->objCount, &r, &h_, &idx)) {
                out4[4 * p + 0] = h_.distance; out4[4 * p + 1] = h_.normal.x;
                out4[4 * p + 2] = h_.normal.y; out4[4 * p + 3] = h_.normal.z;
                ids[p] = idx;
            } else {
                out4[4 * p + 0] = out4[4 * p + 1] = out4[4 * p + 2] = out4[4 * p + 3] = 0.0f;
                ids[p] = -1;
            }
        }
    }
}

/* write_color applied to mean = sums / spp (what computeShader stores for currentSample == 0). */
RTO_API void rto_resolve(const float *sums, int npix, int spp, float *tex) {
    float n = (float)spp;
    for (int p = 0; p < npix; ++p) {
        f4 c = {sums[4 * p] / n, sums[4 * p + 1] / n, sums[4 * p + 2] / n, sums[4 * p + 3] / n};
        Texture t = {tex + 4 * (size_t)p, 1, 1};
        write_color(&t, 0, 0, c);
    }
}

/* Present pass (Present.metal:16-44 + the bgra8Unorm drawable, ContentView.swift:222): nearest
 * sample of the texture at each drawable pixel centre, row 0 at the top, converted to unorm8
 * (clamp to [0,1], x*255 rounded to nearest even) in B,G,R,A byte order.  The float->unorm8 step is
 * fixed-function in Metal; its rounding is taken from the Metal specification's preferred rule. */
RTO_API void rto_present_bgra8(const float *tex, int tex_w, int tex_h, uint8_t *out, int out_w, int out_h) {
    for (int y = 0; y < out_h; ++y) {
        int ty = (int)(((int64_t)(2 * y + 1) * tex_h) / (2 * (int64_t)out_h));
        for (int x = 0; x < out_w; ++x) {
            int tx = (int)(((int64_t)(2 * x + 1) * tex_w) / (2 * (int64_t)out_w));
            const float *c = tex + 4 * ((size_t)ty * tex_w + tx);
            uint8_t *o = out + 4 * ((size_t)y * out_w + x);
            const int order[4] = {2, 1, 0, 3};
            for (int k = 0; k < 4; ++k) {
                float v = c[order[k]];
                v = isnan(v) ? 0.0f : fminf(fmaxf(v, 0.0f), 1.0f);
                o[k] = (uint8_t)rintf(v * 255.0f);
            }
        }
    }
}

/* pow(1 - cosine, 5) of Lib.metal:180 is C powf here; the HIP kernel forms the same power as a binary64 product
 * rounded once to binary32, (float)((x*x) * (x*x) * x) with x widened first.  This sweeps every binary32 in
 * [bits_lo, bits_hi] (bit patterns of non-negative floats) through both and returns how many results differ;
 * first_diff (optional) receives the bit pattern of the lowest differing argument, max_ulp (optional) the largest
 * difference in units of the last place. */
RTO_API uint64_t rto_pow5_sweep(uint32_t bits_lo, uint32_t bits_hi, uint32_t *first_diff, uint32_t *max_ulp) {
    uint64_t ndiff = 0;
    uint32_t first = 0xFFFFFFFFu, worst = 0;
#pragma omp parallel for schedule(static) reduction(+ : ndiff) reduction(min : first) reduction(max : worst)
    for (int64_t i = (int64_t)bits_lo; i <= (int64_t)bits_hi; ++i) {
        uint32_t b = (uint32_t)i;
        float x;
        memcpy(&x, &b, sizeof x);
        float a = powf(x, 5.0f);
        double p = (double)x, p2 = p * p;
        float g = (float)(p2 * p2 * p);
        uint32_t ab, gb;
        memcpy(&ab, &a, sizeof ab);
        memcpy(&gb, &g, sizeof gb);
        if (ab != gb) {
            ++ndiff;
            if (b < first) first = b;
            uint32_t d = ab > gb ? ab - gb : gb - ab;
            if (d > worst) worst = d;
        }
    }
    if (first_diff) *first_diff = first;
    if (max_ulp) *max_ulp = worst;
    return ndiff;
}
