/* examples/host_only.c — the C ABI from plain C (C99), host-only entry points: no GPU needed.
 *   gcc -std=c99 -Iinclude examples/host_only.c -Lcuda-pathtracer_amd -lptmi -Wl,-rpath,$PWD/cuda-pathtracer_amd -o host_only
 *   ./host_only tests/golden/scenes/cbox.obj
 * With a GPU the same handle-free pattern continues with ptmi_ctx_create / ptmi_load_scene /
 * ptmi_update_resolution / ptmi_render_frame / ptmi_read_image. */
#include <stdio.h>
#include <stdlib.h>

#include "ptmi.h"

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s scene.obj\n", argv[0]); return 2; }
    ptmi_host_scene* sc = NULL;
    if (ptmi_host_scene_load(argv[1], 0, 0, &sc) != PTMI_OK) { fprintf(stderr, "load failed: %s\n", ptmi_last_error()); return 1; }
    int n_prims, n_tris, n_quads, n_nodes, depth;
    ptmi_host_scene_info(sc, &n_prims, &n_tris, &n_quads, &n_nodes, &depth);
    printf("prims %d (tris %d quads %d) bvh nodes %d depth %d\n", n_prims, n_tris, n_quads, n_nodes, depth);

    ptmi_camera cam;
    ptmi_default_camera(&cam);
    float f[12];
    if (ptmi_host_camera_frame(&cam, 1024, 1024, f) != PTMI_OK) return 1;
    printf("origin %.9g %.9g %.9g\n", f[0], f[1], f[2]);

    ptmi_tiling t; ptmi_default_tiling(&t); t.n_ranks = 8; t.rank = 3;
    int n_rows = 0;
    ptmi_host_local_row_map(4096, &t, &n_rows, NULL);
    printf("rank 3 of 8 owns %d of 4096 rows\n", n_rows);
    ptmi_host_scene_free(sc);

    ptmi_ctx* ctx = NULL;                       /* on a machine without an MI355X this fails loudly: there is no CPU fallback */
    if (ptmi_ctx_create(0, &ctx) != PTMI_OK) printf("no device: %s\n", ptmi_last_error());
    else ptmi_ctx_destroy(ctx);
    return 0;
}
