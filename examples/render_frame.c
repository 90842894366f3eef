/* examples/render_frame.c — the whole drop-in path from plain C (C99) on one MI355X: loadScene, the radiosity pre-pass,
 * allocateBuffers, renderFrame with MIS-guided sampling, the radiosity view, Save PNG.
 *   gcc -std=c99 -Iinclude examples/render_frame.c -Lcuda-pathtracer_amd -lptmi -Wl,-rpath,$PWD/cuda-pathtracer_amd -o render_frame
 *   ./render_frame tests/golden/scenes/cbox.obj out.png */
#include <stdio.h>
#include <stdlib.h>

#include "ptmi.h"

#define CHECK(call) do { if ((call) != PTMI_OK) { fprintf(stderr, "%s: %s\n", #call, ptmi_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s scene.obj out.png\n", argv[0]); return 2; }
    const int W = 160, H = 120;
    ptmi_ctx* ctx = NULL;
    CHECK(ptmi_ctx_create(0, &ctx));
    CHECK(ptmi_load_scene(ctx, argv[1], /*subdivision*/1, /*convert_quads*/0));
    int n_prims = 0;
    CHECK(ptmi_scene_info(ctx, &n_prims, NULL, NULL, NULL, NULL));

    ptmi_radiosity_params rp; ptmi_default_radiosity_params(&rp);
    rp.mc_samples = 16; rp.num_iterations = 4;
    ptmi_radiosity_stats rs;
    CHECK(ptmi_run_radiosity_solver(ctx, &rp, &rs));
    printf("radiosity: %d primitives, %llu pairs, %llu shadow rays, %.2f ms\n", n_prims, (unsigned long long)rs.pairs,
           (unsigned long long)rs.rays, rs.seconds * 1e3);

    CHECK(ptmi_update_resolution(ctx, W, H, NULL));
    ptmi_config cfg; ptmi_default_config(&cfg);
    cfg.spp = 8; cfg.sampling_mode = 3;                       /* MIS between the BSDF and the radiosity grids */
    CHECK(ptmi_set_config(ctx, &cfg));
    ptmi_stats st;
    CHECK(ptmi_render_frame(ctx, &st));
    unsigned char* rgb = (unsigned char*)malloc((size_t)W * H * 3);
    float* radiance = (float*)malloc((size_t)W * H * 3 * sizeof(float));
    CHECK(ptmi_read_image(ctx, rgb, radiance));
    double mean = 0.0;
    for (int i = 0; i < W * H * 3; i++) mean += radiance[i];
    printf("frame: %llu samples in %.3f ms, mean radiance %.6f\n", (unsigned long long)st.samples, st.seconds * 1e3, mean / (W * H * 3));
    CHECK(ptmi_write_png(argv[2], W, H, rgb));

    cfg.integrator = 1; cfg.spp = 2;                          /* the "Radiosity" integrator of the UI */
    CHECK(ptmi_set_config(ctx, &cfg));
    CHECK(ptmi_render_frame(ctx, NULL));
    CHECK(ptmi_read_image(ctx, rgb, NULL));
    long lit = 0;
    for (int i = 0; i < W * H; i++) lit += (rgb[3 * i] | rgb[3 * i + 1] | rgb[3 * i + 2]) != 0;
    printf("radiosity view: %ld of %d pixels lit\n", lit, W * H);
    free(rgb); free(radiance);
    ptmi_ctx_destroy(ctx);
    return 0;
}
