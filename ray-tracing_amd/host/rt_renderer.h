/*
 * rt_renderer.h -- the reference's progressive tile loop in plain C over the C-ABI of librt_amd.so
 * (reference: "Ray Tracer/ContentView.swift":600-659: startProgressiveRender, renderTile,
 * stopProgressiveRender), without the 16 ms timer: the caller steps it.
 */
#ifndef RT_RENDERER_H
#define RT_RENDERER_H

#include "../../include/rt_amd.h"
#include "rt_host.h"

#ifdef __cplusplus
extern "C" {
#endif

struct rt_renderer {
    rt_ctx *ctx;
    struct Uniforms uniforms;     /* persistent uniforms; per-dispatch fields are set on a copy */
    const struct Object *objects; /* borrowed; re-sent on every dispatch */
    int objectCount;
    int maxIterations;            /* passes; > 500 means "never stop" in the reference (:631) */
    int tileSize;                 /* 64 (:279) */
    struct rt_tile *tiles;
    int tileCount;
    int currentSample;
    int currentTileIndex;
    int isRendering;
    float renderProgress;
};

/* setupTexture + generateTiles (:358-401). */
int rt_renderer_init(struct rt_renderer *r, rt_ctx *ctx, int width, int height);
void rt_renderer_free(struct rt_renderer *r);
/* startProgressiveRender (:600-620). */
int rt_renderer_start(struct rt_renderer *r);
/* renderTile (:628-659): one tile dispatch, or the end-of-pass bookkeeping.  Returns 1 while rendering,
 * 0 once stopped, negative RT_ERR_* on failure. */
int rt_renderer_step(struct rt_renderer *r);
/* stopProgressiveRender (:622-626). */
void rt_renderer_stop(struct rt_renderer *r);

#ifdef __cplusplus
}
#endif
#endif
