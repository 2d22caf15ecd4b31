/*
 * rt_cli.c -- command-line driver: renders one of the named scenes on one GPU and writes a PPM.
 *   rt_render [--scene cover|three_spheres|cornell|nested_glass|random_spheres] [--width W] [--height H]
 *             [--spp N] [--depth D] [--progressive PASSES] [--chunk C] [--fetch 0..3] [--filter 0..3] [--wavefront] [--stamps] [--out file.ppm]
 * --progressive runs the reference's tile-by-tile loop (PASSES passes of --spp samples each);
 * otherwise the whole image is traced in one batch call.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rt_renderer.h"

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv) {
    const char *scene = "cover", *out = NULL;
    int width = 1200, height = 675, spp = 100, depth = 50, passes = 0, chunk = -1, fetch = 0, stamps = 0, filter = 0, bpc = 0, wavefront = 0, slots = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--scene") && i + 1 < argc) scene = argv[++i];
        else if (!strcmp(argv[i], "--width") && i + 1 < argc) width = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--height") && i + 1 < argc) height = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--spp") && i + 1 < argc) spp = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--depth") && i + 1 < argc) depth = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--progressive") && i + 1 < argc) passes = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--chunk") && i + 1 < argc) chunk = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--fetch") && i + 1 < argc) fetch = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--stamps")) stamps = 1;
        else if (!strcmp(argv[i], "--wavefront")) wavefront = 1;
        else if (!strcmp(argv[i], "--slots") && i + 1 < argc) slots = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--filter") && i + 1 < argc) filter = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--blocks-per-cu") && i + 1 < argc) bpc = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    int n = 0;
    if (!strcmp(scene, "cover")) n = rt_scene_cover(NULL, 0, 20250711);
    else if (!strcmp(scene, "three_spheres")) n = rt_scene_three_spheres(NULL, 0);
    else if (!strcmp(scene, "cornell")) n = rt_scene_cornell(NULL, 0);
    else if (!strcmp(scene, "nested_glass")) n = rt_scene_nested_glass(NULL, 0, 400);
    else if (!strcmp(scene, "random_spheres")) n = rt_scene_random_spheres(NULL, 0, 100000, 50.0f, 20250711);
    else { fprintf(stderr, "unknown scene %s\n", scene); return 2; }
    struct Object *objs = (struct Object *)calloc((size_t)n + 1, sizeof *objs);
    if (!strcmp(scene, "cover")) rt_scene_cover(objs, n, 20250711);
    else if (!strcmp(scene, "three_spheres")) rt_scene_three_spheres(objs, n);
    else if (!strcmp(scene, "cornell")) rt_scene_cornell(objs, n);
    else if (!strcmp(scene, "nested_glass")) rt_scene_nested_glass(objs, n, 400);
    else rt_scene_random_spheres(objs, n, 100000, 50.0f, 20250711);

    struct Uniforms u;
    if (rt_config_uniforms(scene, &u, width, height, spp, depth) != 0) return 2;
    u.objCount = n;

    rt_ctx *ctx = NULL;
    int rc = rt_create(&ctx, 0, NULL);
    if (rc != RT_OK) { fprintf(stderr, "rt_create failed: %d\n", rc); return 1; }
    if (chunk >= 0) rt_set_option(ctx, RT_OPT_CHUNK, chunk);
    rt_set_option(ctx, RT_OPT_SCENE_FETCH, fetch);
    if (stamps) rt_set_option(ctx, RT_OPT_KERNEL, 2);
    if (wavefront) rt_set_option(ctx, RT_OPT_KERNEL, 1);
    if (slots) rt_set_option(ctx, RT_OPT_WAVEFRONT_SLOTS, slots);
    rt_set_option(ctx, RT_OPT_FILTER, filter);
    rt_set_option(ctx, RT_OPT_BLOCKS_PER_CU, bpc);

    double t0, t1;
    if (passes > 0) {
        struct rt_renderer r;
        if ((rc = rt_renderer_init(&r, ctx, width, height)) < 0) goto fail;
        r.uniforms = u;
        r.objects = objs;
        r.objectCount = n;
        r.maxIterations = passes;
        t0 = now_s();
        if ((rc = rt_renderer_start(&r)) < 0) goto fail;
        while ((rc = rt_renderer_step(&r)) > 0) {}
        if (rc < 0) goto fail;
        t1 = now_s();
        rt_renderer_free(&r);
        printf("progressive: %d tiles x %d passes x %d spp in %.3f s\n", r.tileCount, passes, spp, t1 - t0);
    } else {
        if ((rc = rt_set_target(ctx, width, height)) < 0) goto fail;
        if ((rc = rt_set_scene(ctx, objs, n)) < 0) goto fail;
        t0 = now_s();
        if ((rc = rt_accum_clear(ctx, NULL)) < 0) goto fail;
        if ((rc = rt_render(ctx, &u, NULL, 0, 0, spp, NULL)) < 0) goto fail;
        if ((rc = rt_resolve(ctx, NULL, spp)) < 0) goto fail;
        if ((rc = rt_synchronize(ctx)) < 0) goto fail;
        t1 = now_s();
        struct rt_stats st;
        rt_last_stats(ctx, &st);
        double msamples = (double)width * height * spp * 1e-6;
        printf("grid %d blocks x %d threads\n", st.grid_blocks, st.block_threads);
        printf("%s %dx%d %d spp depth %d: %.3f s wall, kernel %.3f ms, %.2f Msamples/s, %.2f segments/sample, "
               "%.2f TFLOP/s algorithmic (23 flop x %d objects x segments)\n",
               scene, width, height, spp, depth, t1 - t0, st.kernel_ms, msamples / (t1 - t0),
               (double)st.segments / ((double)width * height * spp), 23.0 * (double)st.tests / (st.kernel_ms * 1e-3) * 1e-12, n);
        if (stamps) {
            double tot = 0;
            for (int i = 0; i < 5; ++i) tot += (double)st.phase_cycles[i];
            const char *names[5] = {"generate", "filter", "resolve", "quads+record", "shade"};
            for (int i = 0; i < 5; ++i)
                printf("  %-13s %6.2f %% of wave cycles, %.0f cycles per loop trip\n", names[i], 100.0 * (double)st.phase_cycles[i] / tot,
                       (double)st.phase_cycles[i] / (double)st.phase_cycles[5]);
            printf("  resolve loop: %.2f trips per search, %.2f candidates per lane-segment, %.1f lanes per trip\n",
                   (double)st.phase_cycles[6] / (double)st.phase_cycles[5], (double)st.phase_cycles[7] / (double)st.segments,
                   (double)st.phase_cycles[7] / (double)(st.phase_cycles[6] ? st.phase_cycles[6] : 1));
        }
    }
    if (out) {
        float *img = (float *)malloc((size_t)width * height * 4 * sizeof(float));
        if ((rc = rt_read_target(ctx, img)) < 0) goto fail;
        rt_write_ppm(out, img, width, height);
        free(img);
    }
    rt_destroy(ctx);
    free(objs);
    return 0;
fail:
    fprintf(stderr, "error %d: %s\n", rc, rt_last_error(ctx));
    rt_destroy(ctx);
    return 1;
}
