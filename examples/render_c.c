/* Minimal C client of the C ABI (include/splat_hip.h): the four naive_gaussians() splats
 * (src/gaussians.rs:319-374) rendered like src/bin/01_naive_gaussian.rs would after
 * update_camera_pose(), written as a PPM.  Plain C99, no C++ runtime needed on the caller's side.
 *   gcc -std=c99 -Iinclude examples/render_c.c -Lsplat_amd -lsplat_hip -Wl,-rpath,$PWD/splat_amd -lm -o render_c */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "splat_hip.h"

int main(int argc, char** argv) {
    const int W = 320, H = 240;
    const float pos[4][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    const float scl[4][3] = {{.03f, .03f, .03f}, {.2f, .03f, .03f}, {.03f, .2f, .03f}, {.03f, .03f, .2f}};
    const float col[4][3] = {{1, 0, 1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    float pos4[16], scales[12], rot[16], opacity[4], sh[4 * 48], cov3d[36];
    memset(sh, 0, sizeof sh);
    for (int g = 0; g < 4; ++g) {
        for (int a = 0; a < 3; ++a) {
            pos4[4 * g + a] = pos[g][a];
            scales[3 * g + a] = scl[g][a];
            sh[48 * g + a] = (col[g][a] - 0.5f) / 0.28209f;
        }
        pos4[4 * g + 3] = 1.0f;
        rot[4 * g + 0] = rot[4 * g + 1] = rot[4 * g + 2] = 0.0f; rot[4 * g + 3] = 1.0f;   /* (i,j,k,w) */
        opacity[g] = 1.0f;
    }
    if (splat_abi_version() != SPLAT_ABI_VERSION || splat_stats_size() != sizeof(splat_stats)) {
        fprintf(stderr, "libsplat_hip.so was built from another splat_hip.h (ABI %u, this client %d)\n", splat_abi_version(), SPLAT_ABI_VERSION);
        return 1;
    }
    splat_ctx* ctx = NULL;
    if (splat_create(NULL, &ctx) != SPLAT_OK) { fprintf(stderr, "splat_create: %s\n", splat_last_error(NULL)); return 1; }
    if (splat_compute_cov3d(ctx, 4, scales, rot, cov3d) != SPLAT_OK ||
        splat_upload_scene(ctx, 4, pos4, cov3d, opacity, sh) != SPLAT_OK) {
        fprintf(stderr, "%s\n", splat_last_error(ctx)); return 1;
    }
    /* Camera::new(H, W, (0,0,5)) + compute_matrices, written out: eye on +z looking at the origin, up = (0,-1,0) */
    splat_camera cam;
    memset(&cam, 0, sizeof cam);
    cam.view[0] = -1.0f; cam.view[5] = -1.0f; cam.view[10] = 1.0f; cam.view[14] = -5.0f; cam.view[15] = 1.0f;
    const float znear = 0.01f, zfar = 100.0f, t = tanf(3.14159265358979f / 4.0f);
    cam.proj[5] = 1.0f / t; cam.proj[0] = cam.proj[5] / ((float)W / (float)H);
    cam.proj[10] = (zfar + znear) / (znear - zfar); cam.proj[14] = zfar * znear * 2.0f / (znear - zfar); cam.proj[11] = -1.0f;
    cam.w = (float)W; cam.h = (float)H;
    cam.htany = t; cam.htanx = t / (float)H * (float)W; cam.focal = (float)H / (2.0f * t);
    cam.cam_pos[2] = 5.0f;
    cam.lowpass = 0.3f;      /* GaussianSplatPipeline02 */
    cam.sh_dim = 15;
    uint32_t* argb = (uint32_t*)calloc((size_t)W * H, 4);
    splat_stats st;
    if (splat_render(ctx, &cam, argb, &st) != SPLAT_OK) { fprintf(stderr, "%s\n", splat_last_error(ctx)); return 1; }
    printf("rendered %llu visible splats, %llu (splat, tile) pairs, %.3f ms on the GPU\n",
           (unsigned long long)st.n_visible, (unsigned long long)st.n_pairs, st.ms_total);
    if (argc > 1) {
        FILE* f = fopen(argv[1], "wb");
        if (!f) { perror(argv[1]); return 1; }
        fprintf(f, "P6\n%d %d\n255\n", W, H);
        for (int i = 0; i < W * H; ++i) { unsigned char p[3] = {argb[i] >> 16, argb[i] >> 8, argb[i]}; fwrite(p, 1, 3, f); }
        fclose(f);
    }
    unsigned long long lit = 0;
    for (int i = 0; i < W * H; ++i) lit += (argb[i] & 0xffffff) != 0;
    printf("%llu pixels lit\n", lit);
    /* the viewer loop's frame (clear + render_to_buffer, src/main.rs:73-74) as one call: the image is written, never read --
     * into a pageable image (a copy behind the frame) and into a page-locked one (the compositor stores into it directly);
     * both must be the frame above, which started from zeros */
    uint32_t* again = (uint32_t*)malloc((size_t)W * H * 4);
    uint32_t* pinned = (uint32_t*)splat_host_alloc((uint64_t)W * H * 4);
    if (!again || !pinned) { fprintf(stderr, "out of memory\n"); return 1; }
    memset(again, 0xab, (size_t)W * H * 4); memset(pinned, 0xcd, (size_t)W * H * 4);
    if (splat_render_frame(ctx, &cam, again, NULL) != SPLAT_OK || splat_render_frame(ctx, &cam, pinned, NULL) != SPLAT_OK) {
        fprintf(stderr, "%s\n", splat_last_error(ctx)); return 1;
    }
    if (memcmp(again, argb, (size_t)W * H * 4) != 0 || memcmp(pinned, argb, (size_t)W * H * 4) != 0) {
        fprintf(stderr, "splat_render_frame differs from clear + splat_render\n"); return 3;
    }
    printf("splat_render_frame: same frame (pageable and page-locked image)\n");
    free(again); splat_host_free(pinned);
    free(argb);
    splat_destroy(ctx);
    return lit ? 0 : 2;
}
