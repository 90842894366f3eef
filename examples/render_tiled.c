/* examples/render_tiled.c — a frame tiled over the GPUs of one node from plain C (C99): one PROCESS per GPU, rows dealt in
 * interleaved blocks (ptmi_tiling), no data-path collective, ONE RCCL gather at frame end (ptmi_gather_frame) to rank 0.
 * The ncclUniqueId travels through a file here; MPI_Bcast or any other channel does as well.
 *
 *   gcc -std=c99 -Iinclude examples/render_tiled.c -Lcuda-pathtracer_amd -lptmi -Wl,-rpath,$PWD/cuda-pathtracer_amd -o render_tiled
 *   for r in 0 1 2 3 4 5 6 7; do ./render_tiled tests/golden/scenes/cbox.obj 8 $r /tmp/ptmi.id out.png & done; wait
 *
 * Pixel RNG streams are keyed by the GLOBAL pixel index (integrator.h:278-279), so the assembled frame is bit-identical to
 * the one GPU's. */
#define _DEFAULT_SOURCE      /* usleep */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "ptmi.h"

#define CHECK(call) do { if ((call) != PTMI_OK) { fprintf(stderr, "rank %d: %s: %s\n", rank, #call, ptmi_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s scene n_ranks rank id_file out.png [width height spp]\n", argv[0]); return 2; }
    const int n_ranks = atoi(argv[2]), rank = atoi(argv[3]);
    const char* id_file = argv[4];
    const int W = argc > 6 ? atoi(argv[6]) : 512, H = argc > 7 ? atoi(argv[7]) : 512, spp = argc > 8 ? atoi(argv[8]) : 16;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) { fprintf(stderr, "bad rank\n"); return 2; }

    ptmi_ctx* ctx = NULL;
    CHECK(ptmi_ctx_create(rank, &ctx));                       /* GPU `rank` of this node */
    CHECK(ptmi_load_scene(ctx, argv[1], 0, 0));               /* every rank holds the whole scene */

    unsigned char id[PTMI_UNIQUE_ID_BYTES];
    char tmp[4096];
    if (rank == 0) {                                          /* rank 0 draws the id and publishes it atomically */
        CHECK(ptmi_dist_unique_id(id));
        snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
        FILE* f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { perror(tmp); return 1; }
        fclose(f);
        if (rename(tmp, id_file) != 0) { perror(id_file); return 1; }
    } else {
        FILE* f = NULL;
        for (int tries = 0; tries < 600 && !(f = fopen(id_file, "rb")); tries++) usleep(100000);
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "rank %d: no id in %s\n", rank, id_file); return 1; }
        fclose(f);
    }
    CHECK(ptmi_dist_init(ctx, id, n_ranks, rank));            /* collective: ncclCommInitRank */

    ptmi_tiling tiling; ptmi_default_tiling(&tiling);
    tiling.n_ranks = n_ranks; tiling.rank = rank; tiling.row_block = 8;
    CHECK(ptmi_update_resolution(ctx, W, H, &tiling));
    ptmi_config cfg; ptmi_default_config(&cfg);
    cfg.spp = spp;
    CHECK(ptmi_set_config(ctx, &cfg));

    ptmi_stats st;
    CHECK(ptmi_render_frame(ctx, &st));                       /* this rank's rows */
    CHECK(ptmi_gather_frame(ctx, 0, PTMI_GATHER_RGB8));       /* the one exchange of the frame; only enqueues */
    CHECK(ptmi_dist_barrier(ctx));
    int rows = 0;
    CHECK(ptmi_local_rows(ctx, &rows));
    printf("rank %d of %d: %d rows, %llu samples in %.3f ms\n", rank, n_ranks, rows, (unsigned long long)st.samples, st.seconds * 1e3);
    if (rank == 0) {
        unsigned char* rgb = (unsigned char*)malloc((size_t)W * H * 3);
        CHECK(ptmi_read_frame(ctx, rgb, NULL));               /* the assembled W x H frame */
        CHECK(ptmi_write_png(argv[5], W, H, rgb));
        unsigned long sum = 0;
        for (size_t i = 0; i < (size_t)W * H * 3; i++) sum += rgb[i];
        printf("frame %dx%d written to %s, byte sum %lu\n", W, H, argv[5], sum);
        free(rgb);
        remove(id_file);
    }
    CHECK(ptmi_dist_finalize(ctx));
    ptmi_ctx_destroy(ctx);
    return 0;
}
