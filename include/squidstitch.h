/*
 * libsquidstitch -- C-ABI of the MI355X (gfx950) registration-and-fusion core.
 *
 * The reference (sohamazing/image-stitcher) has no FFI: its hot path is Python methods of
 * `Stitcher` (stitcher.py:31).  Each entry point below replaces the arithmetic inside one or
 * more of those methods; the integer geometry stays on the host, written exactly as the
 * reference writes it, and only rectangles / crop origins cross this boundary.
 *
 * Conventions
 *  - `extern "C"`, plain C types.  No torch types, no C++ types.
 *  - Every `*_dev` pointer is a DEVICE pointer owned by the caller (e.g. a PyTorch-ROCm
 *    tensor's data_ptr()).  The library never allocates, frees or keeps caller memory
 *    (the one exception is explicit: sq_arena_create hands out device memory the caller asked
 *    for and gives back with sq_arena_destroy).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All device work
 *    is enqueued on it and is asynchronous; no entry point synchronises.
 *  - Return value: 0 = OK, negative = sq_status.  sq_last_error() gives the thread-local
 *    message of the last failure on the calling thread.
 *  - No global mutable state: handles are immutable after creation; safe to use from a
 *    QThread or a multiprocessing child as the reference's front-ends do.
 */
#ifndef SQUIDSTITCH_H
#define SQUIDSTITCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQ_VERSION 108 /* 0.1.7: sq_write_files (chunk files by native threads); 0.1.6: sq_arena_* (canvas memory mapped over all memory classes of the card); 0.1.5: sq_selftest_normalise_divide; 0.1.4: sq_fuse_plan_create_spans / sq_fuse_plan_expand (work list of an overwrite plan produced on the device) */

typedef enum sq_status {
    SQ_OK = 0,
    SQ_ERR_INVALID = -1,     /* bad argument (NULL, negative size, unknown enum, misaligned) */
    SQ_ERR_HIP = -2,         /* a HIP runtime call or launch failed                          */
    SQ_ERR_UNSUPPORTED = -3, /* valid request this build cannot serve                        */
    SQ_ERR_WORKSPACE = -4    /* caller workspace too small                                   */
} sq_status;

typedef enum sq_dtype { SQ_U8 = 1, SQ_U16 = 2, SQ_F32 = 4, SQ_F64 = 8 } sq_dtype;

int sq_version(void);
const char *sq_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Fusion: replaces the per-file loop of Stitcher.stitch_region (stitcher.py:652-681) together
 * with place_tile / place_single_channel_tile (:544-605) and apply_flatfield_correction
 * (:607-611), for all (channel, z) planes of one (timepoint, region) in one launch.
 * ---------------------------------------------------------------------------------------- */

/* One tile's rectangle after the reference's crop (stitcher.py:577-587), BEFORE the canvas
 * clip of :590-594 (the library applies that clip).  Array order = the reference's write
 * order (sorted-filename order within a plane, stitcher.py:168,652): a later rect overwrites
 * an earlier one where they overlap. */
typedef struct sq_rect {
    int32_t src_y0, src_x0; /* first tile pixel used (top_crop, left_crop)              */
    int32_t h, w;           /* size of the cropped tile                                 */
    int32_t dst_y, dst_x;   /* canvas position of that first pixel (y_pixel, x_pixel)   */
} sq_rect;

typedef enum sq_fuse_mode {
    SQ_FUSE_OVERWRITE = 0, /* the reference: last writer wins, output dtype = tile dtype   */
    SQ_FUSE_FEATHER = 1    /* extension: distance-to-edge weighted mean of all covering tiles */
} sq_fuse_mode;

/* Opaque host-side plan: the canvas of one plane cut into disjoint spans, each with the
 * tile(s) that own it, plus the launch work-list.  Depends only on integer geometry, so one
 * plan serves every (c, z) plane of a region, every timepoint and every region that share
 * shifts (the reference computes shifts once per run, stitcher.py:1244-1246). */
typedef struct sq_fuse_plan sq_fuse_plan;

/* Build a plan.  Returns NULL on error (see sq_last_error).
 * tile_h/tile_w: full tile size (needed for feather weights and bounds checks). */
sq_fuse_plan *sq_fuse_plan_create(const sq_rect *rects, int32_t n_rects, int32_t tile_h, int32_t tile_w,
                                  int32_t canvas_h, int32_t canvas_w, int32_t mode);
void sq_fuse_plan_destroy(sq_fuse_plan *plan);
/* Size of the device table and a copy of it into caller host memory; the caller uploads it
 * (e.g. torch.from_numpy(...).cuda()) and passes the device copy to sq_fuse_planes. */
int64_t sq_fuse_plan_table_bytes(const sq_fuse_plan *plan);
int sq_fuse_plan_export(const sq_fuse_plan *plan, void *host_buf, int64_t host_bytes);

/* Copy the table to caller-owned device memory (table_bytes >= sq_fuse_plan_table_bytes) on `stream`
 * and wait for the copy: the plan may be destroyed right after.  The plan keeps its table in
 * page-locked host memory when a device is present, so this is one DMA at link speed; prefer it to
 * sq_fuse_plan_export + a copy of your own. */
int sq_fuse_plan_upload(const sq_fuse_plan *plan, void *table_dev, int64_t table_bytes, void *stream);
/* The same plan with its work list produced ON THE DEVICE (overwrite mode; tiles up to 8176 rows): the host stops after
 * the sweep into spans (1.2 of the 5 ms a 32 x 32 grid's plan takes) and uploads those -- ~100 KB instead of the 14.7 MB
 * table; sq_fuse_plan_expand cuts them into items, finds the seam owners and orders the list with five small kernels into the
 * caller's table buffer (>= sq_fuse_plan_table_bytes) using a scratch buffer (>= sq_fuse_plan_expand_scratch_bytes, free
 * again on return), and waits for them.  The table is sq_fuse_plan_create's byte for byte.  Such a plan has no host copy
 * of its items: sq_fuse_plan_export / _upload refuse it, and sq_fuse_planes refuses it until it has been expanded.
 * This is the first thing a job does after registration (reference: stitch_region's tile loop, stitcher.py:652-681,
 * starts from the shifts the same way). */
sq_fuse_plan *sq_fuse_plan_create_spans(const sq_rect *rects, int32_t n_rects, int32_t tile_h, int32_t tile_w,
                                        int32_t canvas_h, int32_t canvas_w, int32_t mode);
int64_t sq_fuse_plan_expand_scratch_bytes(const sq_fuse_plan *plan);
int sq_fuse_plan_expand(sq_fuse_plan *plan, void *table_dev, int64_t table_bytes, void *scratch_dev, int64_t scratch_bytes,
                        void *stream);
/* Introspection (tests, DESIGN.md numbers): n_spans, n_items, covered voxels, max tiles/span. */
int sq_fuse_plan_stats(const sq_fuse_plan *plan, int64_t *n_spans, int64_t *n_items, int64_t *covered_voxels,
                       int32_t *max_refs);

typedef struct sq_fuse_args {
    /* plan */
    const sq_fuse_plan *plan; /* host handle (sizes, launch geometry)                       */
    const void *table_dev;    /* device copy of sq_fuse_plan_export()                       */
    int64_t table_bytes;
    /* tiles: either a device array of n_planes*n_tiles device pointers (plane-major), or    */
    /* tile_base_dev + (plane*plane_stride + tile*tile_stride) elements when tile_ptrs_dev==0 */
    const void *const *tile_ptrs_dev;
    const void *tile_base_dev;
    int64_t tile_plane_stride, tile_stride; /* in elements                                   */
    int32_t n_tiles, tile_h, tile_w;
    int32_t tile_pitch;                     /* elements between tile rows (>= tile_w)        */
    int32_t tile_dtype;                     /* SQ_U8 | SQ_U16                                */
    /* flatfield (apply_flatfield_correction): device array of n_planes device pointers to   */
    /* tile_h x tile_w gains, NULL array or NULL entry = identity for that plane             */
    const void *const *flat_ptrs_dev;
    int32_t flat_dtype; /* SQ_F32 | SQ_F64: decides the division precision like numpy     */
    /* canvas */
    void *canvas_dev;            /* plane p at canvas_dev + p*canvas_plane_stride elements */
    int64_t canvas_plane_stride; /* elements                                               */
    int32_t canvas_h, canvas_w, canvas_pitch;
    int32_t canvas_dtype; /* overwrite: == tile_dtype; feather: tile_dtype or SQ_F32       */
    int32_t n_planes;
    int32_t mode; /* must match the plan                                               */
    /* optional scratch, sq_fuse_scratch_bytes(n_planes) bytes, 128-byte aligned: lets the call classify */
    /* each plane's float32 gains (all normal floats? -> shortened exact divide) and hand out work      */
    /* through per-XCD device queues; NULL = generic divide, static work split                          */
    void *scratch_dev;
    int64_t scratch_bytes;
    /* work distribution: 0 = the library chooses (device queues for big launches, a static walk for     */
    /* small ones); tests force either one.  grid_blocks > 0 caps the persistent grid (0 = resident WGs) */
    int32_t flags; /* sq_fuse_flags */
    int32_t grid_blocks;
} sq_fuse_args;

typedef enum sq_fuse_flags {
    SQ_FUSE_FORCE_QUEUES = 1, /* device work queues whatever the launch size (needs scratch_dev) */
    SQ_FUSE_FORCE_STATIC = 2, /* static grid-stride walk whatever the launch size                */
    SQ_FUSE_NO_PLANE_GROUPS = 4, /* uint16 / float32 gains: one plane at a time even where planes share a gain image */
    SQ_FUSE_NO_SEAM_OWNERS = 8,  /* plane groups: both items at a vertical seam write their part of the shared cache line */
    SQ_FUSE_CONSECUTIVE_GROUPS = 16 /* plane groups: the planes that share a gain image are taken ZB consecutive ones at a time
                                       (round 2's grouping) instead of being dealt round-robin to the key's groups -- for A/B runs:
                                       a group writes fastest when its planes lie far apart in device memory */
} sq_fuse_flags;

int64_t sq_fuse_scratch_bytes(int32_t n_planes);

/* Fuse n_planes planes.  Every canvas voxel is written exactly once (uncovered voxels = 0,
 * the reference starts from da.zeros, stitcher.py:362); no atomics; deterministic. */
int sq_fuse_planes(const sq_fuse_args *args, void *stream);

/* ------------------------------------------------------------------------------------------
 * Canvas memory.  Replaces the allocation behind Stitcher.init_output (stitcher.py:356-362: the reference's canvas is a
 * lazy dask array; here it is device memory the fusion kernel writes once).  WHERE that memory lies decides how fast the
 * kernel can write it: MI355X device memory falls into a few classes of tens of GiB each (thirds of the card where it was scanned), a row-segment write
 * stream confined to one class runs at 0.55 of the HBM peak and at 0.73-0.76 when spread over the classes, and hipMalloc
 * hands out runs of tens of GiB of one class (csrc/arena.hip).  sq_arena_create takes `bytes` of device
 * memory in physical slices (hipMemCreate), measures which class every 512 MiB unit of them lies in (a pair-fill probe,
 * ~0.3 s for 100 GiB, on `stream`), and maps the slices into ONE contiguous virtual range round-robin over the classes:
 * any canvas laid out in [base_dev, base_dev + bytes) then has all classes under every ~200 MB of it, whatever its plane
 * stride and however the planes are grouped.  The caller sub-allocates (the arena is a flat range; canvases want their
 * planes on 128-byte lines, see sq_fuse_args) and destroys it when no kernel uses it any more.  Synchronises `stream`.
 * Tiles, gains and plans may live anywhere: reads do not depend on the class.
 *   candidate_bytes: the most memory the call may take while it looks for a balanced arena (0 = bytes: whatever comes).  Memory
 *       comes in runs of tens of GiB of one class, so candidates are taken chunk by chunk and classified until the three largest
 *       classes hold a third of `bytes` each (or, after 2.5 x bytes, the two largest half each); the rest is given back -- and
 *       is back -- before the call returns.  Pass what is free (less a reserve) and create the arena FIRST, while the card is
 *       still empty: typically 1.5-2.5 x bytes are taken; the driver clears every slice it hands out and takes back (0.5-6 s
 *       for 80 GiB).
 *   slice_bytes: 0 = 64 MiB (a multiple of 2 MiB);  unit_bytes: 0 = 512 MiB (a multiple of the slice, >= 16 MiB)
 *   flags: SQ_ARENA_NATURAL_ORDER = skip the probe and keep the slices in creation order (the control of A/B runs);
 *          SQ_ARENA_TWO_CLASSES = map slices of the two largest classes only (a measurement aid: two halves against three thirds)
 * Returns NULL on failure (sq_last_error; SQ_ERR_UNSUPPORTED in the message when the platform lacks virtual memory
 * management: allocate the canvas any other way then -- every entry point takes plain device pointers).
 * ---------------------------------------------------------------------------------------- */
#define SQ_ARENA_MAX_CLASSES 8
typedef enum sq_arena_flags { SQ_ARENA_NATURAL_ORDER = 1, SQ_ARENA_TWO_CLASSES = 2 } sq_arena_flags;
typedef struct sq_arena sq_arena;
typedef struct sq_arena_info {
    void *base_dev;       /* first byte of the arena (2 MiB aligned at least)                     */
    int64_t bytes;        /* size, rounded up to whole slices                                     */
    int64_t slice_bytes;
    int32_t n_slices;
    int32_t n_candidates; /* slices taken and classified (>= n_slices); the ones not chosen were given back */
    int32_t n_classes;    /* populations the probe told apart (3 on an MI355X when the arena spans them; 1 = nothing to interleave) */
    int32_t class_slices[SQ_ARENA_MAX_CLASSES];     /* slices of the arena per class     */
    int32_t class_candidates[SQ_ARENA_MAX_CLASSES]; /* slices taken per class            */
    int32_t interleaved;  /* 1: slices mapped round-robin over the classes; 0: creation order */
    float probe_ms;       /* device time of the classification                                   */
    float create_ms;      /* host time of the whole call                                         */
    float min_pair_gbs, max_pair_gbs; /* slowest / fastest pair fill seen by the probe, GB/s written */
} sq_arena_info;
sq_arena *sq_arena_create(int64_t bytes, int64_t candidate_bytes, int64_t slice_bytes, int64_t unit_bytes, int32_t flags,
                          void *stream, sq_arena_info *info);
int sq_arena_info_get(const sq_arena *arena, sq_arena_info *info);
int sq_arena_destroy(sq_arena *arena);

/* ------------------------------------------------------------------------------------------
 * Pyramid: one level of the OME-Zarr multiscale image from the level before.  Replaces
 * ome_zarr.scale.Scaler(max_layer=n-1).nearest(stitched_region) at stitcher.py:797-798 (level
 * count: stitcher.py:346-352), i.e. skimage.transform.resize(plane, (Y//2, X//2), order=0,
 * preserve_range=True, anti_aliasing=False) per plane and level:
 *     dst[p][y][x] = src[p][2*y + 1][2*x + 1],   dst is (src_h / 2) x (src_w / 2), floor.
 * Strides and pitches in elements; dtype SQ_U8 or SQ_U16; src and dst must not overlap.
 * ---------------------------------------------------------------------------------------- */
int sq_downsample2(const void *src_dev, int64_t src_plane_stride, int32_t src_h, int32_t src_w, int64_t src_pitch,
                   void *dst_dev, int64_t dst_plane_stride, int64_t dst_pitch, int32_t n_planes, int32_t dtype,
                   void *stream);

/* ------------------------------------------------------------------------------------------
 * Registration: replaces normalize_image (stitcher.py:613-617), the crops of
 * calculate_horizontal_shift / calculate_vertical_shift (:504-506, :517-519) and
 * skimage.registration.phase_cross_correlation(upsample_factor=10) (:510, :523), batched
 * over tile pairs.  The python round() and the "- crop width" of :511/:524 stay on the host.
 * ---------------------------------------------------------------------------------------- */

/* Per-tile min and max (normalize_image's reductions): out_minmax_dev[2*i] = min,
 * [2*i+1] = max as uint32, for n_tiles tiles given by pointer table or base+stride. */
int sq_tile_minmax(const void *const *tile_ptrs_dev, const void *tile_base_dev, int64_t tile_stride, int32_t n_tiles,
                   int32_t tile_h, int32_t tile_w, int32_t tile_pitch, int32_t tile_dtype,
                   uint32_t *out_minmax_dev, void *stream);

/* normalize_image (stitcher.py:613-617) on whole tiles: out_dev[i] (dense tile_h x tile_w, same dtype) =
 * ((tile - min) / (max - min) * dtype_max) truncated, in float64 like numpy; minmax from sq_tile_minmax.
 * The registration pipeline fuses the same arithmetic into its first kernel; this entry point exists for
 * callers of the method itself. */
int sq_normalize_tiles(const void *const *tile_ptrs_dev, const void *tile_base_dev, int64_t tile_stride, int32_t n_tiles,
                       int32_t tile_h, int32_t tile_w, int32_t tile_pitch, int32_t tile_dtype,
                       const uint32_t *minmax_dev, void *out_dev, void *stream);

typedef enum sq_normalization {
    SQ_NORM_NONE = 0, /* scikit-image <= 0.18                                       */
    SQ_NORM_PHASE = 1 /* scikit-image >= 0.19 default: P /= max(|P|, 100 eps)       */
} sq_normalization;

/* One pair: crop origin inside the reference tile and inside the moving tile; both crops are
 * n0 x n1 (shared by the whole batch). */
typedef struct sq_pair {
    int32_t ref_tile, mov_tile; /* indices into the tile table / minmax table          */
    int32_t ref_y0, ref_x0;     /* crop origin in the reference tile                   */
    int32_t mov_y0, mov_x0;     /* crop origin in the moving tile                      */
} sq_pair;

/* Result per pair.  shift = round(coarse*u)/u + (fine - fix(ceil(1.5u)/2))/u is formed on the
 * host in float64 exactly as skimage does (_phase_cross_correlation.py:232-250). */
typedef struct sq_pair_result {
    int32_t coarse[2]; /* whole-pixel peak after wrap-around (skimage :215-220); INT32_MIN  */
                       /* in both if the pair's tile index or crop lies outside its tile    */
    int32_t fine[2];   /* argmax index in the 15x15 upsampled neighbourhood (:244)     */
    double ccmax_re, ccmax_im; /* cross-correlation value at the refined peak          */
    double src_amp, tgt_amp;   /* sum |F|^2, sum |G|^2 (:252-254)                      */
} sq_pair_result;

typedef struct sq_register_args {
    const void *const *tile_ptrs_dev; /* or NULL + base/stride                          */
    const void *tile_base_dev;
    int64_t tile_stride;
    int32_t n_tiles, tile_h, tile_w, tile_pitch, tile_dtype;
    const uint32_t *minmax_dev; /* from sq_tile_minmax (2 per tile); an entry with min > max
                                   means "do not normalise this tile" (plain skimage call)  */
    const sq_pair *pairs_dev;
    int32_t n_pairs;
    int32_t n0, n1; /* crop size (rows, cols)                                           */
    int32_t upsample_factor; /* 10 in the reference; >= 1                                */
    int32_t normalization;   /* sq_normalization                                         */
    sq_pair_result *results_dev;
    void *workspace_dev;
    int64_t workspace_bytes;
} sq_register_args;

/* Crop lengths: 2 ... 65535 pixels a side (the reference's pocketfft takes any length, stitcher.py:503-510, 516-523).
 *   - a power of two: radix-2 FFT;
 *   - any other length whose prime factors are all <= 13 (1500, 3000, 6000 ...): mixed-radix Cooley-Tukey (radix 4 / 2 / 3 /
 *     5 / 7 / 11 / 13), directly;
 *   - any other length n: Bluestein's chirp-z form through a smooth length M >= 2n - 1 (2084 -> 4320 points);
 * float64 throughout -- the factorisations pocketfft (the reference's FFT, via scipy / numpy) uses for such lengths.
 * A line of up to 9728 points (complex128) is transformed in the 160 KB of LDS: smooth sides up to 9720, any side up to 4860
 * (crops are about half a tile side long, stitcher.py:504-506 / :517-519: every sensor up to 9720 pixels a side -- a 9568 x 6380
 * one gives 4784 and 3190).  A longer line runs the same transform in a scratch line of the workspace (through the L2; slower,
 * and the workspace grows by 512 such lines).  Beyond 65535: SQ_ERR_UNSUPPORTED.
 * sq_register_line_supported: 1 when a crop side of n pixels is accepted, else 0 (no device needed). */
int sq_register_line_supported(int32_t n);
/* Bytes of workspace sq_register_pairs needs for (n_pairs, n0, n1). */
int64_t sq_register_workspace_bytes(int32_t n_pairs, int32_t n0, int32_t n1, int32_t upsample_factor);
int sq_register_pairs(const sq_register_args *args, void *stream);

/* ------------------------------------------------------------------------------------------
 * Chunk encoding on the device: the chunks of n_planes (t, c, z) planes (chunk_h x chunk_w elements,
 * zero-padded past the plane's edge like zarr pads edge chunks) as Blosc-1 frames -- byte shuffle + LZ4, the
 * default codec of the store the reference writes (zarr.storage.default_compressor, stitcher.py:814-818) -- packed
 * densely into out_dev: chunk i = plane-major, then chunk row, then chunk column, lives at
 * out_dev[offsets_dev[i] .. offsets_dev[i + 1]) (n_chunks + 1 offsets; an all-zero chunk has size 0: the store's
 * fill_value stands for it).  *status_dev != 0 afterwards: out_capacity was too small (sq_blosc_out_bound() never is).
 * Strides in elements; dtype SQ_U8 or SQ_U16.  Asynchronous on `stream` like the other entry points.
 * ---------------------------------------------------------------------------------------- */
int64_t sq_blosc_chunk_count(int32_t n_planes, int32_t h, int32_t w, int32_t chunk_h, int32_t chunk_w);
int64_t sq_blosc_out_bound(int32_t n_planes, int32_t h, int32_t w, int32_t dtype, int32_t chunk_h, int32_t chunk_w);
int64_t sq_blosc_scratch_bytes(int32_t n_planes, int32_t h, int32_t w, int32_t dtype, int32_t chunk_h, int32_t chunk_w);
int sq_blosc_encode_planes(const void *planes_dev, int64_t plane_stride, int64_t pitch, int32_t n_planes, int32_t h, int32_t w,
                           int32_t dtype, int32_t chunk_h, int32_t chunk_w, void *scratch_dev, int64_t scratch_bytes,
                           uint64_t *offsets_dev, void *out_dev, int64_t out_capacity, uint32_t *status_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * Chunk files.  The store of save_region_ome_zarr (stitcher.py:771-859; zarr v2, one file per chunk, chunks (1,1,1,512,512)) is
 * hundreds of thousands of half-MB files per region; written one by one from the interpreter they were the wall of a files ->
 * store run once the codec ran on the device.  sq_write_files writes n_files files from ONE host buffer with native threads:
 * file i is the NUL-terminated string at paths + path_offsets[i] and holds data[data_offsets[i] .. data_offsets[i + 1]) (created
 * or truncated, mode 0644; an empty range makes an empty file).  The directories must exist.  n_threads <= 0: 16.  Stops at the
 * first failure (SQ_ERR_INVALID, the path and errno text in sq_last_error); *bytes_written = bytes that reached the files.
 * Host-only: no device, no stream.
 * ---------------------------------------------------------------------------------------- */
int sq_write_files(const char *paths, const int64_t *path_offsets, const void *data, const int64_t *data_offsets, int64_t n_files,
                   int32_t n_threads, int64_t *bytes_written);

/* ------------------------------------------------------------------------------------------
 * Flatfield ESTIMATE: replaces basicpy.BaSiC(get_darkfield=False, smoothness_flatfield=s).fit(images).flatfield
 * in Stitcher.get_flatfields (stitcher.py:365-419; the call is :374-377) for one channel's sample of tiles
 * (<= 80: the reference adds at most 32 per timepoint and stops once it holds MORE than 48, :381-395 -> up to 48 + 32).
 * PARITY UNPINNED: basicpy is an absent, un-pinned third-party package; this is the published BaSiC algorithm
 * (Peng et al. 2017; LADMAP + re-weighted L1, no darkfield, basicpy's documented defaults) as defined by
 * oracle/basic_oracle.py.  Not on the hot path (the divide by the result is: sq_fuse_planes).
 * Unlike the other entry points this one SYNCHRONISES `stream` (its iteration count is decided by the data).
 * flatfield_dev: tile_h x tile_w float32, dense.  workspace: sq_basic_workspace_bytes(), 256-byte aligned.
 * ---------------------------------------------------------------------------------------- */
typedef struct sq_basic_info {
    int32_t reweight_iterations; /* outer re-weighted-L1 rounds run (<= 10)          */
    int32_t ladmap_iterations;   /* inner iterations summed over the rounds          */
    int32_t working_size;        /* 128: the images are resampled to this before the fit */
} sq_basic_info;

int64_t sq_basic_workspace_bytes(int32_t n_images, int32_t tile_h, int32_t tile_w);
int sq_basic_fit(const void *const *tile_ptrs_dev, const void *tile_base_dev, int64_t tile_stride, int32_t n_images,
                 int32_t tile_h, int32_t tile_w, int32_t tile_pitch, int32_t tile_dtype, float smoothness_flatfield,
                 float *flatfield_dev, void *workspace_dev, int64_t workspace_bytes, sq_basic_info *info, void *stream);

/* Self-test (tests only): the fusion kernels divide uint16 pixels by float32 gains with a shortened
 * sequence that is exact for gains with 2^-100 <= |g| < 2^100.  This compares its
 * final clipped integers, truncated (overwrite mode) and rounded (feather mode), with the IEEE path for ALL 2^23 gain
 * mantissas x all 65536 numerators in n_binades consecutive binades starting at 2^exponent (allowed:
 * -100..99), either sign, and leaves the number of differing results in *mismatches_dev (must be 0). */
int sq_selftest_flat_divide(int32_t exponent, int32_t n_binades, int32_t negative, uint64_t *mismatches_dev,
                            void *stream);

/* The same for float64 gains (their shortened sequence = the compiler's IEEE division without its range
 * handling): 2^15 pseudo-random gains per binade (from `seed`; all-zero, all-one and single-bit mantissas
 * included) x all 65536 numerators; compares the quotient doubles and the clipped integers. */
int sq_selftest_flat_divide_f64(int32_t exponent, int32_t n_binades, int32_t negative, uint64_t seed,
                                uint64_t *mismatches_dev, void *stream);

/* normalize_image's division (stitcher.py:615-617: (img - min) / (max - min) in float64) is computed in the registration
 * kernels as a multiply by the reciprocal of the range plus Markstein's correction (two fused multiply-adds).  This
 * compares that with the IEEE division, bit for bit, for ALL numerators 0..65535 x ALL ranges 0..65535 and ADDS the
 * number of differing quotients to *mismatches_dev (zero it first; must stay 0). */
int sq_selftest_normalise_divide(uint64_t *mismatches_dev, void *stream);

/* The grouped feather blend divides the weighted sum of two quotients by the sum of their weights with the IEEE
 * sequence minus its range handling (the reciprocal of the weight sum is shared by the planes of a group).  This
 * compares it with the compiler's division, bit for bit, for ALL 2^23 mantissas of the numerator in n_binades
 * consecutive binades starting at 2^exponent (allowed: -44..52, what moderate gains can produce) x every weight sum
 * 2..16384, either sign; *mismatches_dev must come back 0. */
int sq_selftest_blend_divide(int32_t exponent, int32_t n_binades, int32_t negative, uint64_t *mismatches_dev,
                             void *stream);

/* ------------------------------------------------------------------------------------------
 * Synthetic tiles on the device (bench / tests only): the generator of
 * image-stitcher_amd/synth.py, bit for bit.  out_dev[i] is tile i (tile_h x tile_w, dense).
 * ---------------------------------------------------------------------------------------- */
typedef struct sq_synth_tile {
    uint64_t scene_seed, noise_seed;
    int64_t oy, ox; /* scene origin of the tile */
} sq_synth_tile;

int sq_synth_tiles(const sq_synth_tile *tiles_dev, int32_t n_tiles, int32_t tile_h, int32_t tile_w, int32_t noise_amp,
                   int32_t tile_dtype, void *out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SQUIDSTITCH_H */
