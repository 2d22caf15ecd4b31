// rt_device.hpp -- device-side building blocks of the path tracer for gfx950.
//
// Arithmetic contract: every float operation below is a single IEEE binary32
// operation in the written order.  This file is compiled with -ffp-contract=off;
// division and square root are hipcc's correctly rounded forms.  The only fused
// multiply-adds in the library are the explicit __builtin_fmaf calls of the
// conservative sphere filters in rt_hit.hpp (and the matrix instructions of rt_lift.hpp), whose results never decide a hit.
// Metal built-ins are taken at their Metal Shading Language specification
// formulas (dot = x*x'+y*y'+z*z' left to right, normalize = v / sqrt(dot(v,v)),
// reflect = I - 2*dot(N,I)*N, refract with k = 1 - eta*eta*(1 - dot(N,I)^2),
// fract = min(x - floor(x), 0x1.fffffep-1f), clamp = min(max(x,lo),hi)).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__

namespace rt {

typedef float f4v __attribute__((ext_vector_type(4)));  // one compact sphere record, 16 bytes
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

RT_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV V4 v4(float x, float y, float z, float w) { V4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
RT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RT_DEV V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
RT_DEV V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV float length(V3 a) { return __builtin_sqrtf(dot(a, a)); }
RT_DEV V3 normalize(V3 a) { return a / length(a); }
RT_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// MSL fract = min(x - floor(x), 0x1.fffffep-1f).  v_fract_f32 returns exactly that: checked on gfx950 for every
// float with |x| <= 2^24, and the arguments here (seeds plus a constant) stay below 2^17.
RT_DEV float fract(float x) { return __builtin_amdgcn_fractf(x); }
RT_DEV float clampf(float x, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(x, lo), hi); }
RT_DEV V3 reflect(V3 i, V3 n) { return i - (2.0f * dot(n, i)) * n; }
RT_DEV V3 refract(V3 i, V3 n, float eta) {
    float d = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return v3(0.0f, 0.0f, 0.0f);
    return eta * i - (eta * d + __builtin_sqrtf(k)) * n;
}
RT_DEV V4 operator+(V4 a, V4 b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
RT_DEV V4 operator*(V4 a, V4 b) { return v4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

constexpr float kTMin = 0.001f;              // Shaders.metal:39
constexpr float kMaxFloat = 3.402823466e+38f; // MAXFLOAT

// ---- RNG (Lib.metal:59-72) -------------------------------------------------
RT_DEV uint32_t pcg_hash(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

struct Seed { float x, y; };

RT_DEV float random_double(Seed &seed) {
    uint32_t ix = (uint32_t)(seed.x * 65536.0f);
    uint32_t iy = (uint32_t)(seed.y * 65536.0f);
    uint32_t hash = pcg_hash(ix ^ pcg_hash(iy));
    seed.x = fract(seed.x + 0.618034f);
    seed.y = fract(seed.y + 0.381966f);
    return (float)hash / 4294967296.0f;
}

// Lib.metal:117-132
RT_DEV V3 random_unit_vec(Seed &seed) {
    float x1, x2, w;
    do {
        x1 = 2.0f * random_double(seed) - 1.0f;
        x2 = 2.0f * random_double(seed) - 1.0f;
        w = x1 * x1 + x2 * x2;
    } while (w >= 1.0f);
    float multiplier = 2.0f * __builtin_sqrtf(1.0f - w);
    return v3(x1 * multiplier, x2 * multiplier, 1.0f - 2.0f * w);
}

// Shaders.metal:121-126
RT_DEV Seed sample_seed(uint32_t px, uint32_t py, int currentSample, int rayIndex) {
    Seed s;
    s.x = (float)(px * 1973u + py * 9277u + (uint32_t)(currentSample * 26699) + (uint32_t)(rayIndex * 7919)) / 65536.0f;
    s.y = (float)(py * 4513u + px * 6389u + (uint32_t)(currentSample * 30103) + (uint32_t)(rayIndex * 8191)) / 65536.0f;
    s.x = fract(s.x + 0.618034f);
    s.y = fract(s.y + 0.381966f);
    return s;
}

// Camera part of Uniforms (Definitions.h:17-26), unpacked.
struct Camera {
    V3 pixelOrigin, pixelDeltaX, pixelDeltaY, cameraCenter, defocusDiskU, defocusDiskV;
    float defocusAngle;
};

// Lib.metal:99-106 with random_square (:78-82), defocusDiskSample (:94-97), random_in_unit_disk (:84-92).
RT_DEV void get_ray(int i, int j, const Camera &cam, Seed &seed, V3 &dir, V3 &origin) {
    float ox = random_double(seed) - 0.5f;
    float oy = random_double(seed) - 0.5f;
    V3 pixel_sample = cam.pixelOrigin + (((float)i + ox) * cam.pixelDeltaX) + (((float)j + oy) * cam.pixelDeltaY);
    V3 ray_origin;
    if (cam.defocusAngle <= 0.0f) {
        ray_origin = cam.cameraCenter;
    } else {
        float x, y;
        do {
            x = 2.0f * random_double(seed) - 1.0f;
            y = 2.0f * random_double(seed) - 1.0f;
        } while (x * x + y * y >= 1.0f);
        ray_origin = cam.cameraCenter + (x * cam.defocusDiskU) + (y * cam.defocusDiskV);
    }
    dir = normalize(pixel_sample - ray_origin);
    origin = ray_origin;
}

// ---- compact scene --------------------------------------------------------
// Per object, in the object's original index order.
struct MatDev {
    V4 albedo;     // DIELECTRIC (whose attenuation is 1, Lib.metal:184): x, y = ((1 - ri) / (1 + ri))^2 of Lib.metal:178-179
                   // for ri = 1 / refraction_index (front hit) and ri = refraction_index (back hit)
    V4 emis;       // emission > 0 ? emission_color * emission : (0,0,0,1)   (Shaders.metal:51-54)
    int type;      // LAMBIERTIAN 0 / REFLECTEE 1 / DIELECTRIC 2, anything else absorbs (Lib.metal:203-213)
    float fuzz;
    float ior;
    float inv_ior; // 1.0f / ior (Lib.metal:185)
};
// Per quad: geometry as given plus the object index.
struct QuadDev {
    V3 origin; int obj;
    V3 ex; float pad0;
    V3 ey; float pad1;
};

struct Hit {
    V3 point, normal;
    bool front;
};

// Lib.metal:53-56
RT_DEV void apply_normals(Hit &h, V3 dir, V3 outward) {
    h.front = dot(dir, outward) < 0.0f;
    h.normal = h.front ? outward : -outward;
}

// Division of root numerators by a = d.d, the same divisor for every sphere a ray is tested against.  The
// compiler's correctly rounded n / a is: scale both operands if either needs it (v_div_scale), reciprocal estimate,
// one refinement of the reciprocal, quotient estimate, two refinements of the quotient (v_div_fmas, v_div_fixup
// undoing the scaling).  For a in [2^-40, 2^40] and n in [a 2^-12, a 2^90) no operand is scaled and no fix-up
// applies, so the sequence is the five fused operations of quotient() on a reciprocal that depends on a alone and
// is refined once per ray.  Same operations, same roundings, same bits (tools/div_probe.hip: 0 differences from
// n / a on 4.3e9 random pairs over exactly these ranges).  The numerators that reach quotient() are in range:
// above nmin = a * 0.0009 (up to there the quotient is below kTMin = 0.001, which the interval test rejects, so
// they are not divided at all), and at most |b| + sqrt(disc) <= 2^65 for a finite discriminant, while a >= 1e-6
// puts a 2^90 above 2^70 (an infinite numerator never passes the n < nbest test in front of the division).  A wave
// with a ray whose a is outside [2^-40, 2^40] divides the ordinary way.
struct RootDiv {
    float a, r, nmin;
    bool slow;  // wave-uniform: a ray of this wave has a outside [2^-40, 2^40]; everybody divides the ordinary way
    RT_DEV float quotient(float n) const {
        if (slow) return n / a;
        return quotient_fast(n);
    }
    RT_DEV float quotient_fast(float n) const {  // callers have checked !slow
        float q = n * r;
        float e = __builtin_fmaf(-a, q, n);
        q = __builtin_fmaf(e, r, q);
        e = __builtin_fmaf(-a, q, n);
        return __builtin_fmaf(e, r, q);
    }
};
RT_DEV RootDiv root_div(float a, bool active) {
    RootDiv d;
    d.a = a;
    const float r0 = __builtin_amdgcn_rcpf(a);
    d.r = __builtin_fmaf(__builtin_fmaf(-a, r0, 1.0f), r0, r0);
    d.nmin = a * 0.0009f;
    d.slow = __builtin_amdgcn_ballot_w64(active && !(a >= 0x1p-40f && a <= 0x1p40f)) != 0;
    return d;
}

// Root selection of Lib.metal:230-238 against the interval (kTMin, tmax), for a > 0 (callers have excluded
// |a| < 1e-6, and a = d.d is never negative).  Returns true and the root.
// nbest is the numerator whose quotient by a is the current tmax (+inf while tmax is still MAXFLOAT).  Division
// by a positive a is monotonic, so a numerator <= a * 0.0009 gives a root < kTMin and a numerator >= nbest gives a
// root >= tmax: both fail the reference's strict interval test, and the division is skipped without changing
// any decision.  On success nbest is updated.
RT_DEV bool sphere_root(float disc, float b, const RootDiv &dv, float tmax, float &nbest, float &root_out) {
    if (disc < 0.0f) return false;
    float sqrtd = __builtin_sqrtf(disc);
    float n = -b - sqrtd;
    bool ok = false;
    float root = 0.0f;
    if (n > dv.nmin && n < nbest) {
        root = dv.quotient(n);
        ok = kTMin < root && root < tmax;
    }
    if (!ok) {
        n = -b + sqrtd;
        if (n > dv.nmin && n < nbest) {
            root = dv.quotient(n);
            ok = kTMin < root && root < tmax;
        }
    }
    if (!ok) return false;
    nbest = n;
    root_out = root;
    return true;
}

// The same selection for spheres visited out of object order (the tiled kind sorts them in space).  The reference
// visits objects in index order and keeps a later one only when strictly closer (Shaders.metal:21), i.e. it returns
// the smallest accepted root and, among equal roots, the lowest object index -- and the root a sphere offers does
// not depend on the order: if its first root is inside (kTMin, inf) but not below the current tmax, neither is its
// second.  So roots up to and including tmax are let through here (tie = true when equal), and the caller settles a
// tie by object index.  nskip is the numerator above which the quotient is certainly greater than tmax: nbest
// (1 + 2^-22), four ulps, where the correctly rounded quotients by the same a can no longer coincide.
RT_DEV bool sphere_root_any_order(float disc, float b, const RootDiv &dv, float tmax, float nskip, float &n_out, float &root_out, bool &tie) {
    if (disc < 0.0f) return false;
    float sqrtd = __builtin_sqrtf(disc);
    float n = -b - sqrtd;
    bool ok = false;
    float root = 0.0f;
    if (n > dv.nmin && n <= nskip) {
        root = dv.quotient(n);
        ok = kTMin < root && root <= tmax;
    }
    if (!ok) {
        n = -b + sqrtd;
        if (n > dv.nmin && n <= nskip) {
            root = dv.quotient(n);
            ok = kTMin < root && root <= tmax;
        }
    }
    if (!ok) return false;
    n_out = n;
    root_out = root;
    tie = root == tmax;
    return true;
}

// Lib.metal:248-280.  Interval is [kTMin, tmax] inclusive (Interval::contains).
RT_DEV bool hit_quad(const QuadDev &q, V3 o, V3 d, float tmax, float &t_out, V3 &point, V3 &normal_out) {
    V3 n = cross(q.ex, q.ey);
    V3 normal = normalize(n);
    float denom = dot(normal, d);
    if (__builtin_fabsf(denom) < 1e-8f) return false;
    float t = dot(normal, q.origin - o) / denom;
    if (!(kTMin <= t && t <= tmax)) return false;
    V3 inter = o + t * d;
    V3 toPoint = inter - q.origin;
    float dotU = dot(toPoint, normalize(q.ex));
    float dotV = dot(toPoint, normalize(q.ey));
    float uLen = length(q.ex);
    float vLen = length(q.ey);
    if (dotU < 0.0f || dotU > uLen || dotV < 0.0f || dotV > vLen) return false;
    t_out = t;
    point = inter;
    normal_out = normal;
    return true;
}

// pow(1 - cosine, 5) of Lib.metal:180, evaluated as a product in binary64 and rounded once:
// within 2^-50 relative of the exact power before the final rounding.
RT_DEV float pow5(float x) {
    double p = (double)x;
    double p2 = p * p;
    return (float)(p2 * p2 * p);
}

// Lib.metal:177-181.  r0sq = ((1 - ri) / (1 + ri))^2 depends on the material and on the side that was hit only; it is
// worked out when the scene is uploaded, in the same binary32 operations (MatDev::albedo of a dielectric holds it).
RT_DEV float reflectance(float cosine, float r0sq) {
    return r0sq + (1.0f - r0sq) * pow5(1.0f - cosine);
}

// materialScatters (Lib.metal:157-213).  On return true: attenuation, new ray (dir, origin).
// Lambertian keeps the reference's field order: the scattered ray's DIRECTION is the hit point and its
// ORIGIN is normal + random unit vector (Lib.metal:164 initialises Ray{direction, origin} positionally).
//
// The lanes of a wave hold all three materials at once, and every branch is paid for by the whole wave, so the
// operations the branches have in common are written once and run for the lanes of every kind that needs them:
//   one normalisation: of reflect(d, n) for metal (Lib.metal:170-171) and of d for glass (Lib.metal:187);
//   one first draw from the seed: the x1 of random_unit_vec's first turn for diffuse and metal (Lib.metal:120, the
//     first use of the seed in both, :158, :171) and the number glass compares its reflectance with (Lib.metal:196;
//     not drawn when the ray cannot refract: the reference's || does not evaluate its right side then).
// Every lane still performs exactly the operations of its own branch, in the reference's order.
RT_DEV bool scatter(const MatDev &m, V3 &dir, V3 &origin, const Hit &hit, V4 &attenuation, Seed &seed) {
    const bool diffuse = m.type == 0, metal = m.type == 1, glass = m.type == 2;
    if (!(diffuse || metal || glass)) return false;
    V3 u = dir;
    if (metal) u = reflect(dir, hit.normal);
    if (metal || glass) u = normalize(u);   // metal: normalize(reflected); glass: unit_direction
    float cos_theta = 0.0f, ri = 0.0f;
    bool cannot_refract = false;
    if (glass) {
        ri = hit.front ? m.inv_ior : m.ior;
        cos_theta = __builtin_fminf(dot(-u, hit.normal), 1.0f);
        float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
        cannot_refract = ri * sin_theta > 1.0f;
    }
    float first = 0.0f;
    if (!(glass && cannot_refract)) first = random_double(seed);
    if (glass) {
        attenuation = v4(1.0f, 1.0f, 1.0f, 1.0f);
        V3 direction;
        if (cannot_refract || reflectance(cos_theta, hit.front ? m.albedo.x : m.albedo.y) > first) {
            direction = reflect(u, hit.normal);
        } else {
            direction = refract(u, hit.normal, ri);
        }
        dir = direction;
        origin = hit.point;
        return true;
    }
    // Lib.metal:117-132, the first turn's first draw already made
    float x1 = 2.0f * first - 1.0f;
    float x2 = 2.0f * random_double(seed) - 1.0f;
    float w = x1 * x1 + x2 * x2;
    while (w >= 1.0f) {
        x1 = 2.0f * random_double(seed) - 1.0f;
        x2 = 2.0f * random_double(seed) - 1.0f;
        w = x1 * x1 + x2 * x2;
    }
    float multiplier = 2.0f * __builtin_sqrtf(1.0f - w);
    V3 unit = v3(x1 * multiplier, x2 * multiplier, 1.0f - 2.0f * w);
    attenuation = m.albedo;
    if (diffuse) {
        V3 scatter_direction = hit.normal + unit;
        const float s = 1e-8f;
        if ((__builtin_fabsf(scatter_direction.x) < s) && (__builtin_fabsf(scatter_direction.y) < s) &&
            (__builtin_fabsf(scatter_direction.z) < s))
            scatter_direction = hit.normal;
        dir = hit.point;
        origin = scatter_direction;
        return true;
    }
    V3 reflected = u + (m.fuzz * unit);
    dir = normalize(reflected);
    origin = hit.point;
    return dot(dir, hit.normal) > 0.0f;
}

// write_color (Shaders.metal:71-85) + linear_to_gamma (Lib.metal:144-150).
RT_DEV float gamma_channel(float c, float nan_value) {
    float v = clampf(__builtin_isnan(c) ? nan_value : c, 0.0f, 1.0f);
    return v > 0.0f ? __builtin_sqrtf(v) : 0.0f;
}
RT_DEV V4 write_color(V4 c) {
    return v4(gamma_channel(c.x, 0.0f), gamma_channel(c.y, 0.0f), gamma_channel(c.z, 0.0f), gamma_channel(c.w, 1.0f));
}

}  // namespace rt
