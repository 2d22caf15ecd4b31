/*
 * rt_abi.h -- host <-> device struct contract of the path tracer.
 *
 * These are the layouts of the reference's shared Swift/Metal header
 * (reference: "Ray Tracer/Definitions.h":13-78) restated so that they compile
 * without <simd/simd.h>.  A simd_float3 occupies 16 bytes with 16-byte
 * alignment (the 4th lane is padding); simd_float2 is 8 bytes, 8-aligned;
 * simd_float4 is 16 bytes, 16-aligned.  The names of the structs, of their
 * fields (including the reference's spelling "globalIllumation" and
 * "LAMBIERTIAN") and of the enum macros are kept so that Swift code written
 * against the reference's bridging header keeps compiling against this one.
 *
 * On Apple platforms, where <simd/simd.h> exists, the real simd types are
 * used and the layout is identical; everywhere else the rt_float{2,3,4}
 * stand-ins below are used.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>

#if defined(__APPLE__) && !defined(RT_ABI_NO_SIMD)
#include <simd/simd.h>
typedef simd_float2 rt_float2;
typedef simd_float3 rt_float3;
typedef simd_float4 rt_float4;
#else
#if defined(__cplusplus)
#define RT_ALIGNAS(n) alignas(n)
#else
#define RT_ALIGNAS(n) _Alignas(n)
#endif
typedef struct { RT_ALIGNAS(8) float x; float y; } rt_float2;
typedef struct { RT_ALIGNAS(16) float x; float y; float z; float _pad; } rt_float3;
typedef struct { RT_ALIGNAS(16) float x; float y; float z; float w; } rt_float4;
#endif

/* reference: Definitions.h:13-43 */
struct Uniforms {
    rt_float4 color;            /* never read by the kernel */
    float time;                 /* never read by the kernel */

    rt_float3 pixelDeltaX;
    rt_float3 pixelDeltaY;
    rt_float3 pixelOrigin;
    float pixelSampleScale;     /* never read by the kernel */

    rt_float3 cameraCenter;
    rt_float2 viewportSize;     /* never read by the kernel */
    rt_float3 defocusDiskU;
    rt_float3 defocusDiskV;
    float defocusAngle;

    int objCount;
    int lightCount;             /* never read by the kernel */
    int sampleCount;

    int maxRayDepth;
    int currentSample;
    int totalSamples;           /* never read by the kernel */

    /* Tile information */
    int tileX;
    int tileY;
    int tileWidth;
    int tileHeight;

    rt_float4 globalIllumation;
};

/* reference: Definitions.h:45-48 */
struct Sphere {
    rt_float3 center;
    float radius;
};

/* reference: Definitions.h:50-54 */
struct Quad {
    rt_float3 origin;
    rt_float3 extentX;
    rt_float3 extentY;
};

/* reference: Definitions.h:56-58 */
#define LAMBIERTIAN 0
#define REFLECTEE 1
#define DIELECTRIC 2

/* reference: Definitions.h:60-68 */
struct MeshMaterial {
    int type;
    float emission;
    rt_float4 albedo;
    rt_float4 emission_color;

    float reflection_fuzz;
    float refraction_index;
};

/* reference: Definitions.h:70-71 */
#define TYPE_SPHERE 0
#define TYPE_QUAD 1

/* reference: Definitions.h:73-78 */
struct Object {
    int type;
    struct Sphere s;            /* used when type == TYPE_SPHERE */
    struct Quad q;              /* used when type == TYPE_QUAD */
    struct MeshMaterial mat;
};

#ifndef __cplusplus
typedef struct Uniforms Uniforms;
typedef struct Sphere Sphere;
typedef struct Quad Quad;
typedef struct MeshMaterial MeshMaterial;
typedef struct Object Object;
#endif

/* Size/offset contract: what Metal and Swift see for Definitions.h. */
#if defined(__cplusplus)
#define RT_SA(c, m) static_assert(c, m)
#else
#define RT_SA(c, m) _Static_assert(c, m)
#endif
RT_SA(sizeof(rt_float2) == 8 && sizeof(rt_float3) == 16 && sizeof(rt_float4) == 16, "vector sizes");
RT_SA(sizeof(struct Uniforms) == 224, "Uniforms is 224 bytes");
RT_SA(offsetof(struct Uniforms, color) == 0, "color@0");
RT_SA(offsetof(struct Uniforms, time) == 16, "time@16");
RT_SA(offsetof(struct Uniforms, pixelDeltaX) == 32, "pixelDeltaX@32");
RT_SA(offsetof(struct Uniforms, pixelDeltaY) == 48, "pixelDeltaY@48");
RT_SA(offsetof(struct Uniforms, pixelOrigin) == 64, "pixelOrigin@64");
RT_SA(offsetof(struct Uniforms, pixelSampleScale) == 80, "pixelSampleScale@80");
RT_SA(offsetof(struct Uniforms, cameraCenter) == 96, "cameraCenter@96");
RT_SA(offsetof(struct Uniforms, viewportSize) == 112, "viewportSize@112");
RT_SA(offsetof(struct Uniforms, defocusDiskU) == 128, "defocusDiskU@128");
RT_SA(offsetof(struct Uniforms, defocusDiskV) == 144, "defocusDiskV@144");
RT_SA(offsetof(struct Uniforms, defocusAngle) == 160, "defocusAngle@160");
RT_SA(offsetof(struct Uniforms, objCount) == 164, "objCount@164");
RT_SA(offsetof(struct Uniforms, lightCount) == 168, "lightCount@168");
RT_SA(offsetof(struct Uniforms, sampleCount) == 172, "sampleCount@172");
RT_SA(offsetof(struct Uniforms, maxRayDepth) == 176, "maxRayDepth@176");
RT_SA(offsetof(struct Uniforms, currentSample) == 180, "currentSample@180");
RT_SA(offsetof(struct Uniforms, totalSamples) == 184, "totalSamples@184");
RT_SA(offsetof(struct Uniforms, tileX) == 188, "tileX@188");
RT_SA(offsetof(struct Uniforms, tileY) == 192, "tileY@192");
RT_SA(offsetof(struct Uniforms, tileWidth) == 196, "tileWidth@196");
RT_SA(offsetof(struct Uniforms, tileHeight) == 200, "tileHeight@200");
RT_SA(offsetof(struct Uniforms, globalIllumation) == 208, "globalIllumation@208");
RT_SA(sizeof(struct Sphere) == 32 && offsetof(struct Sphere, radius) == 16, "Sphere");
RT_SA(sizeof(struct Quad) == 48 && offsetof(struct Quad, extentX) == 16 &&
      offsetof(struct Quad, extentY) == 32, "Quad");
RT_SA(sizeof(struct MeshMaterial) == 64 && offsetof(struct MeshMaterial, emission) == 4 &&
      offsetof(struct MeshMaterial, albedo) == 16 &&
      offsetof(struct MeshMaterial, emission_color) == 32 &&
      offsetof(struct MeshMaterial, reflection_fuzz) == 48 &&
      offsetof(struct MeshMaterial, refraction_index) == 52, "MeshMaterial");
RT_SA(sizeof(struct Object) == 160 && offsetof(struct Object, s) == 16 &&
      offsetof(struct Object, q) == 48 && offsetof(struct Object, mat) == 96, "Object");

#endif /* RT_ABI_H */
