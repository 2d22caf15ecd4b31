/*
 * rt_host.h -- plain-C host driver of the path tracer: what the reference's
 * Swift host (reference: "Ray Tracer/ContentView.swift") does on the CPU side
 * of the Metal boundary, restated in C: material/object builders, the camera
 * set-up that fills `Uniforms`, the 64x64 tile list, the progressive
 * tile-by-tile render loop, plus the synthetic benchmark scenes and image
 * writers the reference has no equivalent of.
 *
 * Everything here is `float` arithmetic where the Swift code uses `Float`.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stdint.h>

#include "../../include/rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- materials and objects (ContentView.swift:15-53) -------------------- */

/* SwiftMaterial defaults: lambertian, emission 0, albedo (1,1,1,1), emission_color (0,0,0,1),
 * fuzz 0, refraction index 0 (ContentView.swift:15-21). */
struct MeshMaterial rt_material_default(void);
struct MeshMaterial rt_material_lambertian(float r, float g, float b);
struct MeshMaterial rt_material_metal(float r, float g, float b, float fuzz);
struct MeshMaterial rt_material_dielectric(float refraction_index);
/* SwiftMaterial.light(strength:color:) (ContentView.swift:42-44). */
struct MeshMaterial rt_material_light(float strength, float r, float g, float b);

/* makeSphere / makeQuad (ContentView.swift:47-53); `scale` multiplies positions and sizes. */
struct Object rt_make_sphere(const float center[3], float radius, struct MeshMaterial m, float scale);
struct Object rt_make_quad(const float origin[3], const float extentX[3], const float extentY[3],
                           struct MeshMaterial m, float scale);

/* ---- camera (ContentView.swift:168-213, initRayTracing) ----------------- */

/* Fills pixelOrigin, pixelDeltaX/Y, cameraCenter, viewportSize, defocusDiskU/V, defocusAngle of
 * *u from a look-from / look-at camera.  `width`/`height` are the view size the reference reads
 * from its GeometryProxy.  Other fields of *u are left untouched. */
void rt_camera_setup(struct Uniforms *u, const float lookFrom[3], const float lookAt[3], const float vUp[3],
                     float vfov_degrees, float defocusAngle_degrees, float focusDistance, int width, int height);

/* The reference's own defaults (ContentView.swift:169-212, 94): camera at (278,278,-800) looking at
 * (278,278,0), vfov 40, no defocus, sampleCount 32, maxRayDepth 10, sky (0.1,0.1,0.1,1). */
void rt_uniforms_reference_defaults(struct Uniforms *u, int width, int height);

/* ---- scenes -------------------------------------------------------------- */

/* Each builder writes at most `cap` objects to `out` and returns the number the scene has
 * (call with cap = 0 to size).  Builders that draw random numbers use a fixed documented
 * generator (rt_scene_rng_*), every draw sequenced explicitly. */

/* The app's default scene (ContentView.swift:72-92): Cornell box of five quads, one emissive
 * quad (strength 15), one glass ball. 7 objects. */
int rt_scene_cornell(struct Object *out, int cap);
/* "Ray Tracing in One Weekend" three-sphere scene with hollow glass: 5 objects. */
int rt_scene_three_spheres(struct Object *out, int cap);
/* "Ray Tracing in One Weekend" final cover scene: ground + grid of small spheres + 3 big. */
int rt_scene_cover(struct Object *out, int cap, uint64_t seed);
/* n concentric glass shells around (0,0,-1), alternating index 1.5 and 1/1.5, plus ground. */
int rt_scene_nested_glass(struct Object *out, int cap, int shells);
/* ground + n small random spheres in [-extent,extent]^2 on the ground plane, cover material mix. */
int rt_scene_random_spheres(struct Object *out, int cap, int n, float extent, uint64_t seed);

/* Camera + sky + depth for the named benchmark configurations.
 * name: "three_spheres" | "cover" | "nested_glass" | "random_spheres" | "cornell". */
int rt_config_uniforms(const char *name, struct Uniforms *u, int width, int height, int spp, int depth);

/* ---- tiles (ContentView.swift:380-401, generateTiles) ------------------- */
struct rt_tile { int x, y, width, height; };
/* Row-major list of tileSize x tileSize tiles covering width x height, edge tiles clipped.
 * Returns the tile count; writes at most cap. */
int rt_generate_tiles(int width, int height, int tileSize, struct rt_tile *out, int cap);

/* ---- image output (no reference equivalent) ----------------------------- */
/* rgba float (as stored in the render target, already gamma-encoded) -> binary PPM (P6). */
int rt_write_ppm(const char *path, const float *rgba, int width, int height);
/* rgba float -> PFM (PF, little endian, bottom-up rows), rgb only. */
int rt_write_pfm(const char *path, const float *rgba, int width, int height);

#ifdef __cplusplus
}
#endif
#endif
