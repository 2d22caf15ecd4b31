/* rt_renderer.c -- see rt_renderer.h. */
#include "rt_renderer.h"

#include <stdlib.h>
#include <string.h>

int rt_renderer_init(struct rt_renderer *r, rt_ctx *ctx, int width, int height) {
    memset(r, 0, sizeof *r);
    r->ctx = ctx;
    r->tileSize = 64;
    r->maxIterations = 3; /* ContentView.swift:268 */
    int rc = rt_set_target(ctx, width, height);
    if (rc < 0) return rc;
    r->tileCount = rt_generate_tiles(width, height, r->tileSize, NULL, 0);
    r->tiles = (struct rt_tile *)malloc(sizeof(struct rt_tile) * (size_t)(r->tileCount > 0 ? r->tileCount : 1));
    if (!r->tiles) return RT_ERR_INVALID;
    rt_generate_tiles(width, height, r->tileSize, r->tiles, r->tileCount);
    return RT_OK;
}

void rt_renderer_free(struct rt_renderer *r) {
    free(r->tiles);
    r->tiles = NULL;
}

int rt_renderer_start(struct rt_renderer *r) {
    if (r->isRendering) return RT_OK;
    r->isRendering = 1;
    r->currentSample = 0;
    r->currentTileIndex = 0;
    r->renderProgress = 0.0f;
    return rt_clear(r->ctx);
}

void rt_renderer_stop(struct rt_renderer *r) { r->isRendering = 0; }

int rt_renderer_step(struct rt_renderer *r) {
    if (!r->isRendering) return 0;
    if (r->currentTileIndex >= r->tileCount) {
        r->currentSample += 1;
        if (r->maxIterations > 500) {
            /* the reference never stops in this case */
        } else if (r->currentSample >= r->maxIterations) {
            rt_renderer_stop(r);
            return 0;
        }
        r->currentTileIndex = 0;
        return 1;
    }
    struct rt_tile tile = r->tiles[r->currentTileIndex];
    /* runComputeShaderForTile (:666-709) */
    struct Uniforms sampleUniforms = r->uniforms;
    sampleUniforms.currentSample = r->currentSample;
    sampleUniforms.totalSamples = r->maxIterations;
    sampleUniforms.objCount = r->objectCount;
    sampleUniforms.tileX = tile.x;
    sampleUniforms.tileY = tile.y;
    sampleUniforms.tileWidth = tile.width;
    sampleUniforms.tileHeight = tile.height;
    struct Object dummy;
    memset(&dummy, 0, sizeof dummy);
    int rc = rt_render_tile(r->ctx, &sampleUniforms, r->objectCount ? r->objects : &dummy, r->objectCount ? r->objectCount : 1);
    if (rc < 0) return rc;
    r->currentTileIndex += 1;
    int totalWork = r->tileCount * r->maxIterations;
    int completedWork = r->currentSample * r->tileCount + r->currentTileIndex;
    r->renderProgress = (float)completedWork / (float)totalWork;
    return 1;
}
