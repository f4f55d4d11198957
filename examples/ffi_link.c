/* ffi_link.c -- a plain C host of libtrayracer_hip.a, linked the way integration/rust/build.rs links it into the
 * reference's crate (/root/reference/build.rs:1-11 + the added lines): the static archive, -lamdhip64 and -lstdc++, NO hipcc
 * at link time.  It proves that the archive's device code (fat binary) registers itself under a foreign linker, and it makes
 * the calls the Rust shim makes in place of raytracer/render.rs:105-152: trr_create -> trr_scene_create_ex -> setters ->
 * trr_render_blocks with a block sink, a progress callback and an int32 cancel flag (integration/rust/hip.rs).
 *
 * usage: ffi_link scene.bin frame.ppm     (scene.bin: the flat "TRRFFI01" file tests/test_gpu_ffi_link.py writes)
 *   magic[8]; u32 n_objects; per object { u32 n_tris; f32 tris[n_tris*24]; i32 mat_index[n_tris]; u32 n_mats;
 *   trr_material_desc mats[n_mats]; f32 transform[16]; f32 inverse[16] }; u32 n_lights; trr_light lights[n_lights];
 *   f32 ambient[3], ambient_intensity, sky_color[3]; i32 anti_aliasing; u32 samples; u64 seed; trr_camera; f32 delta; u32 max_depth */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "trayracer_hip.h"

typedef struct Sink {
    uint8_t* frame;
    uint32_t width, blocks, progress_calls, last_done, last_total;
    volatile int32_t cancel;
    int cancel_after;   /* blocks after which the sink raises the cancel flag (-1: never) */
} Sink;

static void on_block(void* user, uint32_t xb, uint32_t yb, uint32_t bw, uint32_t bh, const uint8_t* rgb) {
    Sink* s = (Sink*)user;
    for (uint32_t y = 0; y < bh; ++y)   /* render.rs:170-183: paint the block into the image */
        memcpy(s->frame + ((size_t)(yb * bh + y) * s->width + (size_t)xb * bw) * 3, rgb + (size_t)y * bw * 3, (size_t)bw * 3);
    s->blocks++;
    if (s->cancel_after >= 0 && (int)s->blocks >= s->cancel_after) s->cancel = 1;
}

static void on_progress(void* user, uint32_t done, uint32_t total) {
    Sink* s = (Sink*)user;
    s->progress_calls++;
    s->last_done = done;
    s->last_total = total;
}

static int read_exact(FILE* f, void* dst, size_t bytes) { return bytes == 0 || fread(dst, 1, bytes, f) == bytes; }

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != TRR_OK) {                                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, trr_last_error(ctx));                      \
            return 2;                                                                                \
        }                                                                                            \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s scene.bin frame.ppm\n", argv[0]);
        return 1;
    }
    FILE* f = fopen(argv[1], "rb");
    char magic[8];
    if (!f || !read_exact(f, magic, 8) || memcmp(magic, "TRRFFI01", 8) != 0) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 1;
    }
    uint32_t n_objects = 0;
    if (!read_exact(f, &n_objects, 4) || n_objects > 1024) return 1;
    trr_object_desc* objs = (trr_object_desc*)calloc(n_objects ? n_objects : 1, sizeof *objs);
    for (uint32_t i = 0; i < n_objects; ++i) {
        uint32_t nt = 0, nm = 0;
        if (!read_exact(f, &nt, 4)) return 1;
        float* tris = (float*)malloc((size_t)(nt ? nt : 1) * 24 * sizeof(float));
        int32_t* mi = (int32_t*)malloc((size_t)(nt ? nt : 1) * sizeof(int32_t));
        if (!read_exact(f, tris, (size_t)nt * 24 * sizeof(float)) || !read_exact(f, mi, (size_t)nt * sizeof(int32_t)) || !read_exact(f, &nm, 4))
            return 1;
        trr_material_desc* mats = (trr_material_desc*)calloc(nm ? nm : 1, sizeof *mats);
        if (!read_exact(f, mats, (size_t)nm * sizeof *mats)) return 1;
        objs[i].triangles = tris;
        objs[i].material_index = mi;
        objs[i].n_triangles = nt;
        objs[i].materials = mats;
        objs[i].n_materials = nm;
        if (!read_exact(f, objs[i].transform, 64) || !read_exact(f, objs[i].inverse, 64)) return 1;
    }
    uint32_t n_lights = 0;
    if (!read_exact(f, &n_lights, 4) || n_lights > 65535) return 1;
    trr_light* lights = (trr_light*)calloc(n_lights ? n_lights : 1, sizeof *lights);
    if (!read_exact(f, lights, (size_t)n_lights * sizeof *lights)) return 1;
    trr_settings st;
    memset(&st, 0, sizeof st);
    trr_camera cam;
    float delta = 0.0f;
    uint32_t max_depth = 0;
    if (!read_exact(f, st.ambient_color, 12) || !read_exact(f, &st.ambient_intensity, 4) || !read_exact(f, st.skybox_color, 12) ||
        !read_exact(f, &st.anti_aliasing, 4) || !read_exact(f, &st.samples, 4) || !read_exact(f, &st.seed, 8) ||
        !read_exact(f, &cam, sizeof cam) || !read_exact(f, &delta, 4) || !read_exact(f, &max_depth, 4))
        return 1;
    fclose(f);

    trr_ctx* ctx = NULL;
    trr_scene* scene = NULL;
    printf("%s\n", trr_version());
    CHECK(trr_create(0, &ctx));
    CHECK(trr_scene_create_ex(ctx, objs, n_objects, NULL, 0, TRR_BUILD_HOST, &scene));
    CHECK(trr_scene_set_lights(scene, lights, n_lights));
    CHECK(trr_scene_set_settings(scene, &st));
    CHECK(trr_scene_set_camera(scene, &cam));

    Sink sink;
    memset(&sink, 0, sizeof sink);
    sink.frame = (uint8_t*)calloc((size_t)cam.width * cam.height, 3);   /* render.rs:51: the image starts black */
    sink.width = cam.width;
    sink.cancel_after = -1;
    trr_stats stats;
    CHECK(trr_render_blocks(ctx, scene, delta, max_depth, 0, 0, &sink.cancel, on_block, on_progress, &sink, &stats));
    printf("blocks=%u progress_calls=%u last=%u/%u rays=%llu\n", sink.blocks, sink.progress_calls, sink.last_done, sink.last_total,
           (unsigned long long)(stats.rays_primary + stats.rays_shadow + stats.rays_reflection + stats.rays_march));
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 1;
    fprintf(o, "P6\n%u %u\n255\n", cam.width, cam.height);
    fwrite(sink.frame, 3, (size_t)cam.width * cam.height, o);
    fclose(o);

    /* the cancel flag, raised from inside the sink after two rows of blocks: render.rs:113,154 */
    Sink c2;
    memset(&c2, 0, sizeof c2);
    c2.frame = (uint8_t*)calloc((size_t)cam.width * cam.height, 3);
    c2.width = cam.width;
    c2.cancel_after = 40;
    int rc = trr_render_blocks(ctx, scene, delta, max_depth, 0, 0, &c2.cancel, on_block, on_progress, &c2, NULL);
    printf("cancelled: rc=%d blocks=%u\n", rc, c2.blocks);

    trr_scene_destroy(scene);
    trr_destroy(ctx);
    return rc == TRR_ERR_CANCELLED ? 0 : 3;
}
