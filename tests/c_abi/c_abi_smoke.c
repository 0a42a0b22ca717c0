/* The C-ABI used from plain C: no Python, no torch.  hipMalloc'd buffers in, libsquidstitch calls,
 * results checked against a loop on the host.  Built and run by tests/test_c_abi_gpu.py:
 *   gcc -D__HIP_PLATFORM_AMD__ c_abi_smoke.c -I include -I /opt/rocm/include -L image-stitcher_amd/csrc
 *       -lsquidstitch -L /opt/rocm/lib -lamdhip64
 */
#define _POSIX_C_SOURCE 200809L   /* mkstemp under -std=c11 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "squidstitch.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_SQ(x) do { int r_ = (x); if (r_ < 0) { printf("sq error %d: %s\n", r_, sq_last_error()); return 3; } } while (0)

int main(void) {
    enum { TH = 48, TW = 80, N = 5, HC = 130, WC = 211 };
    if (sq_version() != SQ_VERSION) { printf("version mismatch\n"); return 1; }
    /* tiles: deterministic pattern */
    static uint16_t tiles[N][TH][TW];
    for (int t = 0; t < N; ++t) for (int y = 0; y < TH; ++y) for (int x = 0; x < TW; ++x)
        tiles[t][y][x] = (uint16_t)(1 + ((t * 7919u + y * 131u + x * 17u) % 65000u));
    sq_rect rects[N] = {{0, 0, TH, TW, 0, 0}, {3, 5, 40, 70, 30, 60}, {0, 0, TH, TW, 90, 150},
                        {10, 0, 30, TW, 60, 3}, {0, 0, TH, TW, 20, 100}};
    /* host expectation: last writer wins, clip at the canvas */
    static uint16_t want[HC][WC];
    memset(want, 0, sizeof want);
    for (int t = 0; t < N; ++t)
        for (int y = 0; y < rects[t].h; ++y) for (int x = 0; x < rects[t].w; ++x) {
            int dy = rects[t].dst_y + y, dx = rects[t].dst_x + x;
            if (dy < HC && dx < WC) want[dy][dx] = tiles[t][rects[t].src_y0 + y][rects[t].src_x0 + x];
        }
    sq_fuse_plan *plan = sq_fuse_plan_create(rects, N, TH, TW, HC, WC, SQ_FUSE_OVERWRITE);
    if (!plan) { printf("plan: %s\n", sq_last_error()); return 3; }
    int64_t nbytes = sq_fuse_plan_table_bytes(plan);
    void *table = malloc((size_t)nbytes);
    CHECK_SQ(sq_fuse_plan_export(plan, table, nbytes));
    void *d_table, *d_tiles, *d_canvas; uint32_t *d_mm;
    CHECK_HIP(hipMalloc(&d_table, (size_t)nbytes));
    CHECK_HIP(hipMalloc(&d_tiles, sizeof tiles));
    CHECK_HIP(hipMalloc(&d_canvas, sizeof want));
    CHECK_HIP(hipMalloc((void **)&d_mm, 2 * N * sizeof(uint32_t)));
    CHECK_HIP(hipMemcpy(d_table, table, (size_t)nbytes, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_tiles, tiles, sizeof tiles, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(d_canvas, 0xAA, sizeof want));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    sq_fuse_args a;
    memset(&a, 0, sizeof a);
    a.plan = plan; a.table_dev = d_table; a.table_bytes = nbytes;
    a.tile_base_dev = d_tiles; a.tile_plane_stride = (int64_t)N * TH * TW; a.tile_stride = TH * TW;
    a.n_tiles = N; a.tile_h = TH; a.tile_w = TW; a.tile_pitch = TW; a.tile_dtype = SQ_U16;
    a.canvas_dev = d_canvas; a.canvas_plane_stride = (int64_t)HC * WC; a.canvas_h = HC; a.canvas_w = WC;
    a.canvas_pitch = WC; a.canvas_dtype = SQ_U16; a.n_planes = 1; a.mode = SQ_FUSE_OVERWRITE;
    CHECK_SQ(sq_fuse_planes(&a, stream));
    CHECK_SQ(sq_tile_minmax(NULL, d_tiles, TH * TW, N, TH, TW, TW, SQ_U16, d_mm, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    static uint16_t got[HC][WC];
    uint32_t mm[2 * N];
    CHECK_HIP(hipMemcpy(got, d_canvas, sizeof got, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(mm, d_mm, sizeof mm, hipMemcpyDeviceToHost));
    if (memcmp(got, want, sizeof want) != 0) { printf("canvas differs\n"); return 4; }
    for (int t = 0; t < N; ++t) {
        uint32_t lo = 65535, hi = 0;
        for (int y = 0; y < TH; ++y) for (int x = 0; x < TW; ++x) { uint32_t v = tiles[t][y][x]; if (v < lo) lo = v; if (v > hi) hi = v; }
        if (mm[2 * t] != lo || mm[2 * t + 1] != hi) { printf("minmax of tile %d differs\n", t); return 5; }
    }
    /* the same plan with its work list produced on the device: the table must be the exported one byte for byte */
    {
        sq_fuse_plan *spans = sq_fuse_plan_create_spans(rects, N, TH, TW, HC, WC, SQ_FUSE_OVERWRITE);
        if (!spans) { printf("spans plan: %s\n", sq_last_error()); return 7; }
        if (sq_fuse_plan_table_bytes(spans) != nbytes) { printf("spans plan: table size differs\n"); return 7; }
        int64_t sbytes = sq_fuse_plan_expand_scratch_bytes(spans);
        if (sbytes < 0) { printf("scratch: %s\n", sq_last_error()); return 7; }
        void *d_table2, *d_scratch;
        CHECK_HIP(hipMalloc(&d_table2, (size_t)nbytes));
        CHECK_HIP(hipMalloc(&d_scratch, (size_t)(sbytes ? sbytes : 1)));
        sq_fuse_args b = a;
        b.plan = spans; b.table_dev = d_table2;
        if (sq_fuse_planes(&b, stream) != SQ_ERR_INVALID || !strstr(sq_last_error(), "sq_fuse_plan_expand")) { printf("unexpanded plan accepted\n"); return 7; }
        CHECK_SQ(sq_fuse_plan_expand(spans, d_table2, nbytes, d_scratch, sbytes, stream));
        void *table2 = malloc((size_t)nbytes);
        CHECK_HIP(hipMemcpy(table2, d_table2, (size_t)nbytes, hipMemcpyDeviceToHost));
        if (memcmp(table, table2, (size_t)nbytes) != 0) { printf("device-expanded table differs from the host planner's\n"); return 7; }
        CHECK_HIP(hipMemset(d_canvas, 0x55, sizeof want));
        CHECK_SQ(sq_fuse_planes(&b, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_HIP(hipMemcpy(got, d_canvas, sizeof got, hipMemcpyDeviceToHost));
        if (memcmp(got, want, sizeof want) != 0) { printf("canvas through the device-expanded plan differs\n"); return 7; }
        sq_fuse_plan_destroy(spans);
        free(table2);
        hipFree(d_table2); hipFree(d_scratch);
    }
    /* the canvas in an arena (sq_arena_create: physical slices classified and mapped round-robin over the card's memory
       classes): plain device memory to the fusion call, the same voxels; then the canvas as a chunk file by native threads */
    {
        sq_arena_info info;
        memset(&info, 0, sizeof info);
        sq_arena *arena = sq_arena_create(64 << 20, 192 << 20, 8 << 20, 32 << 20, 0, stream, &info);
        if (!arena) { printf("arena: %s\n", sq_last_error()); return 8; }
        if (!info.base_dev || info.bytes != (64 << 20) || info.n_slices != 8 || info.n_classes < 1 || info.n_candidates < 8) { printf("arena info\n"); return 8; }
        sq_fuse_args c = a;
        c.canvas_dev = (char *)info.base_dev + 4096;
        CHECK_HIP(hipMemset(info.base_dev, 0x33, 1 << 20));
        CHECK_SQ(sq_fuse_planes(&c, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_HIP(hipMemcpy(got, c.canvas_dev, sizeof got, hipMemcpyDeviceToHost));
        if (memcmp(got, want, sizeof want) != 0) {
            uint32_t w0[4], w1[4];
            hipMemcpy(w0, info.base_dev, 16, hipMemcpyDeviceToHost);
            hipMemcpy(w1, (char *)info.base_dev + 4096 + sizeof got + 64, 16, hipMemcpyDeviceToHost);
            uint16_t s16[8];
            hipMemcpy(s16, c.canvas_dev, 16, hipMemcpyDeviceToHost);
            static uint16_t again[HC][WC];
            hipMemcpy(again, c.canvas_dev, sizeof again, hipMemcpyDeviceToHost);
            printf("16-byte read of the canvas start: %u %u %u %u; second full read: %u %u %u %u (%s the first)\n", s16[0], s16[1], s16[2], s16[3],
                   again[0][0], again[0][1], again[0][2], again[0][3], memcmp(again, got, sizeof got) ? "differs from" : "equals");
            printf("arena base %p (%lld bytes, %d slices of %d candidates, %d classes), canvas %p; base holds %08x %08x, behind the canvas %08x %08x\n",
                   info.base_dev, (long long)info.bytes, info.n_slices, info.n_candidates, info.n_classes, c.canvas_dev, w0[0], w0[1], w1[0], w1[1]);
            int shown = 0;
            for (int y = 0; y < HC && shown < 4; ++y)
                for (int x = 0; x < WC && shown < 4; ++x)
                    if (got[y][x] != want[y][x]) printf("canvas in the arena differs at (%d, %d): %u, expected %u\n", y, x, got[y][x], want[y][x]), ++shown;
            return 8;
        }
        sq_arena_info again;
        CHECK_SQ(sq_arena_info_get(arena, &again));
        if (again.base_dev != info.base_dev) { printf("arena info_get\n"); return 8; }
        CHECK_SQ(sq_arena_destroy(arena));
        char path[] = "/tmp/sq_c_abi_chunk_XXXXXX";
        int fd = mkstemp(path);
        if (fd < 0) { printf("mkstemp\n"); return 8; }
        close(fd);
        const int64_t poff[1] = {0}, doff[2] = {0, (int64_t)sizeof got};
        int64_t done = 0;
        CHECK_SQ(sq_write_files(path, poff, got, doff, 1, 2, &done));
        FILE *fh = fopen(path, "rb");
        static uint16_t back[HC][WC];
        const size_t rd = fh ? fread(back, 1, sizeof back, fh) : 0;
        if (fh) fclose(fh);
        remove(path);
        if (done != (int64_t)sizeof got || rd != sizeof got || memcmp(back, want, sizeof want) != 0) { printf("chunk file differs\n"); return 8; }
    }
    /* error path: a wrong canvas size must be refused with a message, not crash */
    a.canvas_w = WC + 1;
    if (sq_fuse_planes(&a, stream) != SQ_ERR_INVALID || !strstr(sq_last_error(), "geometry differs")) { printf("bad error path\n"); return 6; }
    sq_fuse_plan_destroy(plan);
    free(table);
    hipFree(d_table); hipFree(d_tiles); hipFree(d_canvas); hipFree(d_mm);
    hipStreamDestroy(stream);
    printf("c-abi smoke ok\n");
    return 0;
}
