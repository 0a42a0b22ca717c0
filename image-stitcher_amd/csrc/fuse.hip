// Fusion kernels for gfx950: tiles -> canvas planes, every canvas voxel written exactly once.
//
// Replaces the per-file loop of the reference (stitcher.py:652-681 -> :544-611).  The host
// planner (plan.cpp) has already cut the canvas into disjoint spans with their owning tile(s),
// so the device work is a batch of rectangle copies with an optional per-pixel flatfield divide
// (overwrite mode), or a weighted mean of up to MAX_REFS tiles (feather mode).
//
// Roofline: HBM.  Algorithmic traffic is 2 B read + 2 B write per uint16 voxel (+ the flatfield,
// which is re-used by every tile of a channel and lives in L2 / Infinity Cache).
//
// Mapping: a persistent grid (resident workgroups only) walks the (plane, item) list with a grid
// stride; one item = up to BLOCK_ROWS x BLOCK_COLS of one span, described by one 32-byte record;
// each of the 4 waves takes whole rows, 64 lanes x 16 B per step, so every wave-instruction
// stores one contiguous 1 KiB run of a canvas row.  Wave stores start on 128-byte lines of the
// canvas (its pitch is arbitrary, so the phase is recomputed per row); the tile side is read with
// 16-byte loads at whatever 2-byte phase the placement leaves (measured cost: ~2 %).
//
// Measured on MI355X: 0.53-0.55 of the 8 TB/s HBM peak with float32 gains,
// 0.58-0.63 without, 0.70 on one huge aligned tile (the ceiling of this structure).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <map>

#include "common.h"

using namespace sq;

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct __attribute__((packed)) U32x4U {  // 16 bytes at any alignment
    u32x4 v;
};
struct __attribute__((packed)) F32x4U {
    f32x4 v;
};
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed)) U32x2U {  // 8 bytes at any alignment
    u32x2 v;
};
struct __attribute__((packed)) F64x2U {
    f64x2 v;
};

#define SQ_GLOBAL __attribute__((address_space(1)))
// Explicit global-address-space accessors: pointers that come out of a table are generic to the
// compiler and would be lowered to flat_* instructions.
template <typename V>
__device__ __forceinline__ auto ldg(const void *p) {
    return ((const SQ_GLOBAL V *)p)->v;
}
template <typename S>
__device__ __forceinline__ S ldg_s(const void *p) {
    return *(const SQ_GLOBAL S *)p;
}
#ifndef SQ_STORE_POLICY
#define SQ_STORE_POLICY 0
#endif
__device__ __forceinline__ void stg_nt(void *p, u32x4 v) {
#if SQ_STORE_POLICY == 0
    __builtin_nontemporal_store(v, (SQ_GLOBAL u32x4 *)p);
#elif SQ_STORE_POLICY == 1
    asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(p), "v"(v) : "memory");
#elif SQ_STORE_POLICY == 2
    asm volatile("global_store_dwordx4 %0, %1, off nt sc0 sc1" ::"v"(p), "v"(v) : "memory");
#elif SQ_STORE_POLICY == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#else
    *(SQ_GLOBAL u32x4 *)p = v;
#endif
}
// 16-byte non-temporal store at scalar base + 32-bit lane offset (bytes): no 64-bit address pair in vector registers
// (nt measured best here too: 0.623 against 0.599 plain, 0.622 "sc1 nt", 0.602 "sc0 sc1"; profiles/r02_exp21_store_policy.log)
__device__ __forceinline__ void stg_nt_at(void *base, uint32_t byte_off, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(byte_off), "v"(v), "s"(base) : "memory");
}
template <typename S>
__device__ __forceinline__ void stg_s(void *p, S v) {
    *(SQ_GLOBAL S *)p = v;
}

struct FuseParams {
    const Span *spans;
    const Ref *refs;
    const Item *items;
    const Seam *seams;            // per item, who writes the canvas line a vertical seam falls in (overwrite plans), or NULL
    const void *const *tile_ptrs;
    const void *tile_base;
    int64_t tile_plane_stride, tile_stride;
    const void *const *flat_ptrs;
    void *canvas;
    int64_t canvas_plane_stride;
    int32_t n_tiles, tile_h, tile_w, tile_pitch;
    int32_t canvas_pitch;
    const uint32_t *flat_class;   // per plane: bit 0 clear = every gain is a normal float in the fast divide's range; bit 1 clear =
                                  // every gain is also moderate (2^-20 <= |g| < 2^20: what the grouped feather blend asks for)
    uint32_t *queue;              // 9 chunk counters (8 XCD lanes + the rest), one 128-byte line each; NULL = static stride
    int32_t lane_items;           // list positions [0, 8 * lane_items) of a plane are lane-interleaved
    int32_t n_planes;
    int32_t chunk;                // consecutive lane positions a workgroup takes per atomic, 1..QUEUE_CHUNK
    const struct PlaneGroup *groups;   // plane groups of the float32-gain kernel (build_groups_kernel), else NULL
    const uint32_t *n_groups;          // how many there are (device side: the host never learns it)
};
// Planes that are divided by the SAME gain image (the z planes of a channel) and whose canvas rows sit at the same
// phase inside a 128-byte line are carried through an item together: the gains and their reciprocals are loaded /
// computed once per group instead of once per plane.  32 bytes.
#ifndef SQ_ZB
#define SQ_ZB 5
#endif
constexpr int ZB = SQ_ZB;            // most planes in a group
static_assert(ZB >= 1 && ZB <= 7, "a PlaneGroup holds at most 7 planes");
struct PlaneGroup {
    int32_t n;          // 1..ZB
    int32_t plane[7];
};
static_assert(sizeof(PlaneGroup) == 32, "PlaneGroup layout");
constexpr int QUEUE_STRIDE = 32;   // uint32 words between the counters
#ifndef SQ_QUEUE_CHUNK
#define SQ_QUEUE_CHUNK 8
#endif
constexpr int QUEUE_CHUNK = SQ_QUEUE_CHUNK;   // most consecutive lane positions a workgroup takes per atomic

template <typename T>
__device__ __forceinline__ const T *tile_ptr(const FuseParams &P, int plane, int tile) {
    if (P.tile_ptrs) return static_cast<const T *>(P.tile_ptrs[(int64_t)plane * P.n_tiles + tile]);
    return static_cast<const T *>(P.tile_base) + plane * P.tile_plane_stride + tile * P.tile_stride;
}

// divide -> clip -> truncating cast of apply_flatfield_correction (stitcher.py:609-610), in the
// flatfield's own precision like numpy's uint16 / floatXX promotion.  NaN (0/0) -> 0, which is
// what the x86 cast of the reference produces; +inf -> dtype max through the clip.
// RND = 0: truncate (the reference's astype).  RND = 1: round half to even first -- feather mode's
// integer output (np.rint) for a voxel a single tile covers.
template <typename T, int RND = 0>
__device__ __forceinline__ T flat_f32(T v, float g) {
    float q = __fdiv_rn((float)v, g);
    if (RND) q = rintf(q);
    const float hi = sizeof(T) == 1 ? 255.0f : 65535.0f;
    q = fminf(fmaxf(q, 0.0f), hi);
    return (T)q;
}
// Fast exact flatfield divide for THIS operand class: numerator an integer in [0, 65535], gain a
// float32 with 2^-100 <= |g| < 2^100 (either sign).  Markstein's scheme -- the hardware reciprocal
// (v_rcp_f32, 1 ulp), one quotient, one exact-residual correction -- gives the correctly rounded
// quotient here: with r = (1/g)(1 + e), q = n r has relative error h <= |e| + 2^-24, the residual
// n - g q is exact in one FMA, and q + rem r = (n/g)(1 - h e), i.e. wrong by < 2^-44 relative
// before its single rounding, while a 16-bit numerator keeps n/g at least 2^-41 (relative) away
// from every rounding boundary.  A Newton step on r (two more FMAs) is therefore not needed; it was
// there in earlier versions and is kept behind SQ_DIV_NEWTON.  Below 2^-112 the first quotient
// would overflow and the correction turn into inf - inf; the guard leaves a wide margin.  None of
// this is taken on faith: sq_selftest_flat_divide compares the final clipped integers with the
// IEEE path for ALL 2^23 mantissas x 65536 numerators in every binade of the range, on the GPU the
// tests run on (tests/test_fuse_gpu.py).  Zeros, denormals, tiny gains, infinities and NaNs among
// a plane's gains are found by a pre-pass (flat_classify_kernel) and send that plane through the
// generic IEEE sequence instead.
// 4 VALU slots + the reciprocal instead of the 11 of the IEEE sequence (two v_div_scale, v_div_fmas,
// v_div_fixup, two refinements).  Doing two pixels per instruction on the packed-float32 pipe
// (v_pk_mul_f32 / v_pk_fma_f32) was tried: it needs 86 VGPRs (5 waves) and measured no faster.
#ifndef SQ_DIV_NEWTON
#define SQ_DIV_NEWTON 0
#endif
__device__ __forceinline__ float div_u16_normal(float n, float g) {
    float r = __builtin_amdgcn_rcpf(g);
#if SQ_DIV_NEWTON
    const float e = fmaf(-g, r, 1.0f);
    r = fmaf(e, r, r);
#endif
    const float q = n * r;
    const float rem = fmaf(-g, q, n);
    return fmaf(rem, r, q);
}
constexpr int FAST_MIN_EXP = -100;   // fast divide allowed for 2^FAST_MIN_EXP <= |g| < 2^FAST_END_EXP:
constexpr int FAST_END_EXP = 100;    // (every non-zero quotient n/g is then a normal float)
constexpr int BLEND_ACC_MIN_EXP = -44, BLEND_ACC_END_EXP = 53, BLEND_WSUM_MAX = 16384;   // what the grouped blend's last division sees
constexpr int MODERATE_EXP = 20;     // grouped feather blend: 2^-20 <= |g| < 2^20 keeps sums of weighted quotients far from the range ends

// float -> uint32 the way the hardware does it: negative and NaN -> 0, too large -> 0xFFFFFFFF.
// (C++'s (uint32_t)f is undefined outside the range, so say the instruction.)
__device__ __forceinline__ uint32_t cvt_u32_sat(float f) {
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

// RND = 1 (feather mode rounds the float32 quotient half to even): the reference's result then hangs
// on how n/g rounds to float32 right next to a representable k + 0.5, and n/g can be within 2^-48 of
// that float midpoint -- with the RAW hardware reciprocal the short sequence is not enough there (401 of
// the 2^39 operand pairs of a binade come out one ulp off, e.g. 4075 / 0x1.dbf3fep-3).  With ONE Newton
// step on the reciprocal it is: on gfx950 v_rcp_f32 + one step IS the correctly rounded reciprocal for
// every one of the 2^23 mantissas, and Markstein's theorem then makes quotient + one exact-residual
// correction the correctly rounded quotient (tools/div_probe.hip: 0 of 2^39 pairs differ in every binade
// tried, either sign; the second correction of rounds 1-3 -- the compiler's own IEEE sequence has it --
// changed nothing).  6 slots, not the compiler's 11; sq_selftest_flat_divide compares it with the
// compiler's division for every operand pair, on the GPU the tests run on.
__device__ __forceinline__ float div_u16_normal_ieee(float n, float g) {
    float r = __builtin_amdgcn_rcpf(g);
    r = fmaf(fmaf(-g, r, 1.0f), r, r);
    const float q = n * r;
    return fmaf(fmaf(-g, q, n), r, q);
}
template <int RND>
__device__ __forceinline__ float quotient_u16_normal(float n, float g) {
    if (!RND) return div_u16_normal(n, g);
    return __builtin_rintf(div_u16_normal_ieee(n, g));   // v_rndne_f32
}
template <typename T, int RND = 0>
__device__ __forceinline__ T flat_f32_fast(T v, float g) {
    // clip(q, 0, max) then truncate == saturating conversions: NaN never occurs on this path
    const uint32_t k = cvt_u32_sat(quotient_u16_normal<RND>((float)v, g));
    return (T)min(k, sizeof(T) == 1 ? 255u : 65535u);
}
// two pixels of one 32-bit word at once: v_cvt_pk_u16_u32 saturates to 65535 and packs
template <int RND = 0>
__device__ __forceinline__ uint32_t flat_f32_fast_pair(uint32_t word, float g_lo, float g_hi) {
    const uint32_t a = cvt_u32_sat(quotient_u16_normal<RND>((float)(word & 0xFFFFu), g_lo));
    const uint32_t b = cvt_u32_sat(quotient_u16_normal<RND>((float)(word >> 16), g_hi));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 p = __builtin_amdgcn_cvt_pk_u16(a, b);
    return (uint32_t)p[0] | ((uint32_t)p[1] << 16);
}

// float64 gains: the compiler's IEEE sequence is v_div_scale x2, v_rcp_f64, two Newton steps, quotient,
// residual, v_div_fmas, v_div_fixup.  For a numerator in [0, 65535] and a gain with 2^-100 <= |g| < 2^100
// the scaling is the identity and the fix-up never fires, so the same arithmetic without those three
// instructions yields the same double, bit for bit (sq_selftest_flat_divide_f64 compares the doubles and
// the clipped integers for 2^15 random gains per binade x every numerator; tests/test_fuse_gpu.py).
// v_cvt_u32_f64 saturates like its float32 sibling: clip + truncate in one instruction.
__device__ __forceinline__ double div_u16_normal_f64(double n, double g) {
    double r = __builtin_amdgcn_rcp(g);
    r = fma(r, fma(-g, r, 1.0), r);
    r = fma(r, fma(-g, r, 1.0), r);
    const double q = n * r;
    return fma(fma(-g, q, n), r, q);
}
__device__ __forceinline__ uint32_t cvt_u32_sat(double f) {
    uint32_t r;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(f));
    return r;
}
template <typename T>
__device__ __forceinline__ T flat_f64_fast(T v, double g) {
    return (T)min(cvt_u32_sat(div_u16_normal_f64((double)v, g)), sizeof(T) == 1 ? 255u : 65535u);
}

template <typename T>
__device__ __forceinline__ T flat_f64(T v, double g) {
    double q = __ddiv_rn((double)v, g);
    const double hi = sizeof(T) == 1 ? 255.0 : 65535.0;
    q = fmin(fmax(q, 0.0), hi);
    return (T)q;
}

template <typename T, int FLAT>
__device__ __forceinline__ T correct_one(T v, const void *frow, int p) {
    if (FLAT == 1) return flat_f32<T>(v, ldg_s<float>(static_cast<const float *>(frow) + p));
    if (FLAT == 2) return flat_f64<T>(v, ldg_s<double>(static_cast<const double *>(frow) + p));
    return v;
}

template <typename T>
struct Pix;  // 16 bytes of pixels
template <>
struct Pix<uint16_t> {
    static constexpr int N = 8;
    __device__ static uint16_t get(const u32x4 &v, int e) { return (uint16_t)(v[e >> 1] >> ((e & 1) * 16)); }
    __device__ static void set(u32x4 &v, int e, uint16_t x) {
        v[e >> 1] = (e & 1) ? ((v[e >> 1] & 0x0000FFFFu) | ((uint32_t)x << 16)) : ((v[e >> 1] & 0xFFFF0000u) | x);
    }
};
template <>
struct Pix<uint8_t> {
    static constexpr int N = 16;
    __device__ static uint8_t get(const u32x4 &v, int e) { return (uint8_t)(v[e >> 2] >> ((e & 3) * 8)); }
    __device__ static void set(u32x4 &v, int e, uint8_t x) {
        const int sh = (e & 3) * 8;
        v[e >> 2] = (v[e >> 2] & ~(0xFFu << sh)) | ((uint32_t)x << sh);
    }
};

// ---------------------------------------------------------------------------------------------
// overwrite mode (the reference's semantics)
// ---------------------------------------------------------------------------------------------
// One wave moves RB = BLOCK_ROWS/4 rows of an item as a software pipeline over (row, slot) steps,
// a slot being one 16-byte vector per lane (64 lanes = 1 KiB of a canvas row).  Step s issues the
// loads of slot s (pixels, flatfield gains, and -- with the row's first slot -- the row's edge
// pixels) and then finishes slot s - DEPTH (divide, clip, pack, store).  vmcnt is in issue order,
// so a finishing slot waits only for loads at least as old as its own: younger loads and stores
// stay in flight.  DEPTH = all steps for the plain copy (every load of the item is in flight
// before the first store); 1-2 with a flatfield, where registers buy occupancy that hides the
// divide behind other waves' memory time.
// A row's edges (canvas pixels before the first / after the last whole 16-byte vector) are done
// one pixel per lane: the head with the row's first slot, the tail with its second -- one 2-byte load and
// one 2-byte store per wave and row each.  Where the item owns the seam on its left (Seam in common.h;
// uint16 only: a line is 64 pixels = the wave) the head is the seam's whole line: lane j writes byte 2j
// of it, the pixels before the item's first from the left neighbour's tile (or zero fill).
constexpr int NO_EDGE = -(1 << 20);
template <typename T>
struct Row {
    T *drow;
    const T *srow;
    const char *frow;
    const T *lsrow;       // seam owner: the left neighbour's pixel that would land on drow[0] ...
    const char *lfrow;    // ... and its gain
    int mis, n;
    int v_first, v_end;   // whole vectors are v in [v_first, v_end)
    int edge_p;           // this lane's head pixel: before the first whole vector, or lane - mis over the seam's line
                          // (< 0: the left neighbour's); NO_EDGE: none
    int tail_p;           // this lane's pixel after the last whole vector (or -1)
    bool lzero;           // the left neighbour is zero fill
};

template <typename T, int FLAT>
struct Slot {
    static constexpr int VEC = Pix<T>::N;
    u32x4 px;
    f32x4 g32[FLAT == 1 ? VEC / 4 : 1];
    f64x2 g64[FLAT == 2 ? VEC / 2 : 1];
    T edge;
    float eg32;
    double eg64;
};

template <typename T>
__device__ __forceinline__ void row_setup(Row<T> &J, int lane, int seam_flags = 0) {
    constexpr int VEC = Pix<T>::N;
    // Vector v covers row pixels [v*VEC - mis, +VEC).  mis is the row's phase inside a 128-byte
    // line, not just inside 16 bytes: vector 0 then starts ON a line boundary, so every 1 KiB
    // wave-store covers 8 whole lines instead of straddling 9 (measured +10-15 % on canvases
    // whose pitch is not a multiple of 128 bytes, which is the normal case).
    constexpr int LINE = 128 / (int)sizeof(T);
    J.mis = (int)((reinterpret_cast<uintptr_t>(J.drow) / sizeof(T)) & (LINE - 1));
    const bool seams = sizeof(T) == 2 && J.n > 0;
    const bool head_line = seams && (seam_flags & SEAM_HAS_LEFT) && J.mis > 0;
    const int n_own = (seams && (seam_flags & SEAM_LEAVE_TAIL)) ? J.n - ((J.n + J.mis) & (LINE - 1)) : J.n;   // n >= LINE there
    J.v_first = head_line ? LINE / VEC : (J.mis + VEC - 1) / VEC;
    J.v_end = (n_own + J.mis) / VEC;
    const int head_end = head_line ? 0 : min(n_own, J.v_first * VEC - J.mis);   // pixels [0, head_end)
    const int tail_start = max(head_end, J.v_end * VEC - J.mis);               // pixels [tail_start, n_own)
    J.edge_p = head_line ? lane - J.mis : (lane < head_end ? lane : NO_EDGE);
    J.tail_p = (lane < VEC && tail_start + lane < n_own) ? tail_start + lane : -1;
    J.lzero = (seam_flags & SEAM_LEFT_ZERO) != 0;
}

// plain loads: the tile is read once, but non-temporal loads measured 3-8 % slower here
template <typename T, int FLAT>
__device__ __forceinline__ void slot_load(Slot<T, FLAT> &S, const Row<T> &J, int lane, int k) {
    constexpr int VEC = Pix<T>::N;
    const int v = lane + 64 * k;
    if (v >= J.v_first && v < J.v_end) {
        const int p0 = v * VEC - J.mis;
        S.px = ldg<U32x4U>(J.srow + p0);
        if (FLAT == 1 && J.frow) {
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) S.g32[q] = ldg<F32x4U>(reinterpret_cast<const float *>(J.frow) + p0 + 4 * q);
        }
        if (FLAT == 2 && J.frow) {
#pragma unroll
            for (int q = 0; q < VEC / 2; ++q) S.g64[q] = ldg<F64x2U>(reinterpret_cast<const double *>(J.frow) + p0 + 2 * q);
        }
    }
    if (k == 0 && J.edge_p > NO_EDGE) {
        const int p = J.edge_p;
        const bool left = p < 0;
        if (left && J.lzero) {
            S.edge = 0;
        } else {
            S.edge = ldg_s<T>((left ? J.lsrow : J.srow) + p);
            const char *fr = left ? J.lfrow : J.frow;
            if (FLAT == 1 && J.frow) S.eg32 = ldg_s<float>(reinterpret_cast<const float *>(fr) + p);
            if (FLAT == 2 && J.frow) S.eg64 = ldg_s<double>(reinterpret_cast<const double *>(fr) + p);
        }
    }
    if (k == 1 && J.tail_p >= 0) {
        S.edge = ldg_s<T>(J.srow + J.tail_p);
        if (FLAT == 1 && J.frow) S.eg32 = ldg_s<float>(reinterpret_cast<const float *>(J.frow) + J.tail_p);
        if (FLAT == 2 && J.frow) S.eg64 = ldg_s<double>(reinterpret_cast<const double *>(J.frow) + J.tail_p);
    }
}

// non-temporal stores: the canvas is written once and never read back by this kernel
template <typename T, int FLAT, bool FAST, int RND>
__device__ __forceinline__ void slot_store(Slot<T, FLAT> &S, const Row<T> &J, int lane, int k) {
    constexpr int VEC = Pix<T>::N;
    const int v = lane + 64 * k;
    if (v >= J.v_first && v < J.v_end) {
        u32x4 px = S.px;
        if (FLAT == 1 && J.frow) {
            if (FAST && sizeof(T) == 2) {   // every gain of this plane is inside the fast range (pre-pass flag)
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) {
                    px[2 * q] = flat_f32_fast_pair<RND>(px[2 * q], S.g32[q][0], S.g32[q][1]);
                    px[2 * q + 1] = flat_f32_fast_pair<RND>(px[2 * q + 1], S.g32[q][2], S.g32[q][3]);
                }
            } else if (FAST) {
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        Pix<T>::set(px, 4 * q + e, flat_f32_fast<T, RND>(Pix<T>::get(px, 4 * q + e), S.g32[q][e]));
            } else {
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        Pix<T>::set(px, 4 * q + e, flat_f32<T, RND>(Pix<T>::get(px, 4 * q + e), S.g32[q][e]));
            }
        } else if (FLAT == 2 && J.frow) {
#pragma unroll
            for (int q = 0; q < VEC / 2; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    Pix<T>::set(px, 2 * q + e, FAST ? flat_f64_fast<T>(Pix<T>::get(px, 2 * q + e), S.g64[q][e])
                                                    : flat_f64<T>(Pix<T>::get(px, 2 * q + e), S.g64[q][e]));
        }
        stg_nt(J.drow + (v * VEC - J.mis), px);
    }
    const bool head = k == 0 && J.edge_p > NO_EDGE, tail = k == 1 && J.tail_p >= 0;
    if (head || tail) {
        T t = S.edge;
        if (!(head && J.edge_p < 0 && J.lzero)) {   // (zero fill left of the seam stays 0)
            if (FLAT == 1 && J.frow) t = FAST ? flat_f32_fast<T, RND>(t, S.eg32) : flat_f32<T, RND>(t, S.eg32);
            if (FLAT == 2 && J.frow) t = FAST ? flat_f64_fast<T>(t, S.eg64) : flat_f64<T>(t, S.eg64);
        }
        stg_s<T>(J.drow + (head ? J.edge_p : J.tail_p), t);
    }
}

template <typename T>
__device__ __forceinline__ void row_zero(T *drow, int n, int lane, bool leave_tail = false) {
    constexpr int VEC = Pix<T>::N;
    constexpr int SLOTS = BLOCK_COLS / VEC / 64 + 1;
    Row<T> J;
    J.drow = drow;
    J.n = n;
    // leave_tail: the line the row ends in is written by the right neighbour (Seam in common.h; n >= one line)
    row_setup<T>(J, lane, leave_tail ? SEAM_LEAVE_TAIL : 0);
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int v = lane + 64 * k;
        // (plain instead of non-temporal stores for the zeros: 0.634 against 0.642 for the whole launch)
        if (v >= J.v_first && v < J.v_end) stg_nt(drow + (v * VEC - J.mis), u32x4{0, 0, 0, 0});
    }
    if (J.edge_p > NO_EDGE) stg_s<T>(drow + J.edge_p, 0);
    if (J.tail_p >= 0) stg_s<T>(drow + J.tail_p, 0);
}

#ifndef SQ_DEPTH_PLAIN
#define SQ_DEPTH_PLAIN 16
#endif
#ifndef SQ_DEPTH_F32
#define SQ_DEPTH_F32 2
#endif
#ifndef SQ_DEPTH_F64
#define SQ_DEPTH_F64 1
#endif

// the software pipeline over the (row, slot) steps of RB rows (see the comment above Row)
template <typename T, int FLAT, bool FAST, int RND, int RB, int SLOTS, int DEPTH>
__device__ __forceinline__ void pipeline_rows(const Row<T> (&J)[RB], int lane) {
    constexpr int NSTEP = RB * SLOTS;
    Slot<T, FLAT> buf[DEPTH + 1];
#pragma unroll
    for (int s = 0; s < NSTEP + DEPTH; ++s) {
        if (s < NSTEP) slot_load<T, FLAT>(buf[s % (DEPTH + 1)], J[s / SLOTS], lane, s % SLOTS);
        if (s >= DEPTH) {
            const int d = s - DEPTH;
            slot_store<T, FLAT, FAST, RND>(buf[d % (DEPTH + 1)], J[d / SLOTS], lane, d % SLOTS);
        }
    }
}

// minimum waves per SIMD asked of the register allocator: 6 caps the float32-gain kernel at 80 VGPRs
// without spilling (86 otherwise: 5 waves), measured +3 %; 8 spills and loses 20 %
#ifndef SQ_WAVES_PLAIN
#define SQ_WAVES_PLAIN 1
#endif
// (round 4: 5, not 6 -- at 80 VGPRs the per-plane kernels with gains kept spill code inside their item loops, which this build
//  does not ship any more: tools/barrier_scan.py scan_spills, tests/test_isa_cpu.py; these kernels serve groups of one)
#ifndef SQ_WAVES_F32
#define SQ_WAVES_F32 5
#endif
#ifndef SQ_WAVES_F64
#define SQ_WAVES_F64 1
#endif
// one edge pixel through the plane's flatfield divide (the slot pipeline's arithmetic, one pixel at a time)
template <typename T, int FLAT, int RND>
__device__ __forceinline__ T correct_edge(T t, const char *gain, bool fast) {
    if (FLAT == 1 && gain) {
        const float g = ldg_s<float>(gain);
        return fast ? flat_f32_fast<T, RND>(t, g) : flat_f32<T, RND>(t, g);
    }
    if (FLAT == 2 && gain) {
        const double g = ldg_s<double>(gain);
        return fast ? flat_f64_fast<T>(t, g) : flat_f64<T>(t, g);
    }
    return t;
}

// seam: who writes the cache line at this item's left / right end (Seam in common.h; uint16 only -- a kernel honours the
// records for every item of a plane or for none); ltile: the plane's tile left of the seam
template <typename T, int FLAT, int RND = 0>
__device__ __forceinline__ void process_item(const FuseParams &P, int plane, const Item &it, const T *tile, int wave,
                                             int lane, const Seam seam = Seam{-1, 0, 0, 0}, const T *ltile = nullptr) {
    constexpr int VEC = Pix<T>::N;
    constexpr int FSZ = FLAT == 2 ? 8 : 4;
    constexpr int LINE = 128 / (int)sizeof(T);
#ifndef SQ_RB_FLAT
#define SQ_RB_FLAT 1   // with gains a wave pipelines one row at a time (measured 0.8 % faster than two: less row state)
#endif
    constexpr int RB = FLAT ? SQ_RB_FLAT : (BLOCK_ROWS >= 4 ? BLOCK_ROWS / 4 : 1);   // rows a wave pipelines together
    constexpr int SLOTS = BLOCK_COLS / VEC / 64 + 1;         // vectors per lane per row (+1: alignment phase)
    constexpr int NSTEP = RB * SLOTS;
    constexpr int WANT = FLAT == 0 ? SQ_DEPTH_PLAIN : (FLAT == 1 ? SQ_DEPTH_F32 : SQ_DEPTH_F64);
    constexpr int DEPTH = WANT < NSTEP ? WANT : NSTEP;
    static_assert(SLOTS >= 2, "the tail pixels ride on a row's second slot");
    const int rows = it.hw >> 16, n = it.hw & 0xFFFF;
    const int sflags = sizeof(T) == 2 ? seam.flags : 0;
    const bool has_left = sflags & SEAM_HAS_LEFT, lzero = sflags & SEAM_LEFT_ZERO, leave_tail = sflags & SEAM_LEAVE_TAIL;
    T *canvas = static_cast<T *>(P.canvas) + plane * P.canvas_plane_stride;
    const char *flat = (FLAT && P.flat_ptrs) ? static_cast<const char *>(P.flat_ptrs[plane]) : nullptr;
    const bool fast = FLAT != 0 && P.flat_class && (P.flat_class[plane] & 1u) == 0;
    if (!it.nref) {   // uncovered canvas: zeros (da.zeros, stitcher.py:362)
        for (int r = wave; r < rows; r += 4) {
            T *d = canvas + (int64_t)(it.dst_y + r) * P.canvas_pitch + it.dst_x;
            const int mis = (int)((reinterpret_cast<uintptr_t>(d) / sizeof(T)) & (LINE - 1));
            const int head = (has_left && mis > 0) ? mis : 0;   // pixels of the left neighbour in the seam's line
            if (head && !lzero) {   // the seam's line: the left tile's pixels, then zeros
                const int p = lane - mis;
                const int po = p < 0 ? p : -1;
                const int64_t at = (int64_t)(seam.b + r) * P.tile_pitch + seam.c + po;
                const T e = correct_edge<T, FLAT, RND>(ldg_s<T>(ltile + at), flat ? flat + ((int64_t)(seam.b + r) * P.tile_w + seam.c + po) * FSZ : nullptr, fast);
                stg_s<T>(d + p, p < 0 ? e : (T)0);
                row_zero<T>(d + (LINE - head), n - (LINE - head), lane, leave_tail);    // after the seam's line
            } else if (head) {
                row_zero<T>(d - head, n + head, lane, leave_tail);                      // zeros from the line boundary
            } else {
                row_zero<T>(d, n, lane, leave_tail);
            }
        }
        return;
    }
    for (int rb = wave; rb < rows; rb += 4 * RB) {
        Row<T> J[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int r = rb + 4 * j;
            J[j].n = r < rows ? n : 0;
            J[j].drow = canvas + (int64_t)(it.dst_y + r) * P.canvas_pitch + it.dst_x;
            J[j].srow = tile + (int64_t)(it.b + r) * P.tile_pitch + it.c;
            J[j].frow = flat ? flat + ((int64_t)(it.b + r) * P.tile_w + it.c) * FSZ : nullptr;
            const bool lt = has_left && !lzero;
            J[j].lsrow = lt ? ltile + (int64_t)(seam.b + r) * P.tile_pitch + seam.c : J[j].srow;
            J[j].lfrow = (lt && flat) ? flat + ((int64_t)(seam.b + r) * P.tile_w + seam.c) * FSZ : J[j].frow;
            row_setup<T>(J[j], lane, sflags);
        }
        if (fast) pipeline_rows<T, FLAT, true, RND, RB, SLOTS, DEPTH>(J, lane);
        else pipeline_rows<T, FLAT, false, RND, RB, SLOTS, DEPTH>(J, lane);
    }
}

// Work distribution.
//  * static (no scratch given): a persistent grid-stride walk, block b takes items b, b + G, ...
//  * dynamic: per plane the item list is 8 interleaved lanes (one per XCD: lane x holds the items of the
//    tile-row blocks == x mod 8, see plan.cpp) followed by a short rest.  Nine device counters hand out
//    chunks of QUEUE_CHUNK consecutive positions of a lane; a workgroup reads the XCD it really runs on
//    (HW_REG_XCC_ID), pulls from THAT lane, and moves on to the next lane / the rest once its own is
//    drained.  The items in flight on an XCD are then always one contiguous window of its lane -- same
//    flatfield rows, fetched into that XCD's L2 once, however unevenly workgroups progress (with the
//    static stride they drift apart over a 35 ms launch: PMC, 52 GB of gains re-fetched per launch)
//    -- and the launch ends with every workgroup busy.  One atomic and one barrier per chunk (the
//    first attempt paid both per item and lost 7 %); the atomic for the next chunk is issued before
//    the current chunk is processed and its result only looked at afterwards.
struct Chunk {
    int q;         // 0..7 lane, 8 rest, -1 none
    uint32_t c;    // chunk index inside the queue
};

// wave-uniform values the compiler cannot prove uniform (they come out of LDS): pin them to scalar registers
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ Item sgpr(Item it) {
    it.dst_y = sgpr(it.dst_y);
    it.dst_x = sgpr(it.dst_x);
    it.hw = sgpr(it.hw);
    it.nref = sgpr(it.nref);
    it.a = sgpr(it.a);
    it.b = sgpr(it.b);
    it.c = sgpr(it.c);
    it.span = sgpr(it.span);
    return it;
}
template <typename T>
__device__ __forceinline__ const T *sgpr(const T *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)sgpr((int)(uint32_t)v), hi = (uint32_t)sgpr((int)(uint32_t)(v >> 32));
    return reinterpret_cast<const T *>(((uint64_t)hi << 32) | lo);
}

// The queue walk shared by the fusion kernels: calls body(plane, item, aux) for every (plane, item) this
// workgroup is handed, all threads of the workgroup together, arguments in scalar registers.
// aux = pre(plane, item, list position) is evaluated by the thread that loads the descriptor (the overwrite kernel
// fetches the tile pointer there, so that eight of them are in flight at once).
template <typename Aux, typename Pre, typename Body>
__device__ __forceinline__ void for_each_queued_item(const FuseParams &P, const int64_t n_items, const uint32_t n_units, Pre pre,
                                                     Body body) {
    __shared__ int s_q[2];
    __shared__ uint32_t s_c[2];
    __shared__ Item s_item[QUEUE_CHUNK];          // the chunk's descriptors, loaded by QUEUE_CHUNK threads at once
    __shared__ int s_plane[QUEUE_CHUNK];
    __shared__ Aux s_aux[QUEUE_CHUNK];
    const int home = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);   // HW_REG_XCC_ID[3:0]
    // queue q holds n_planes * per_plane(q) positions; 32-bit arithmetic (the host checks the sizes)
    auto per_plane_of = [&](int q) { return (uint32_t)(q < 8 ? (int64_t)P.lane_items : n_items - 8 * (int64_t)P.lane_items); };
    auto total_of = [&](int q) { return n_units * per_plane_of(q); };
    int given_up = 0;   // thread 0: queues found empty so far (own lane first, then the others, then the rest)
    auto queue_of = [&](int k) { return k < 8 ? ((home + k) & 7) : 8; };
    auto settle = [&](uint32_t c) -> Chunk {   // thread 0: make (given_up, c) a real chunk or move on
        while (true) {
            const int q = queue_of(given_up);
            if ((uint64_t)c * (uint32_t)P.chunk < total_of(q)) return {q, c};
            if (++given_up > 8) return {-1, 0u};
            c = atomicAdd(&P.queue[queue_of(given_up) * QUEUE_STRIDE], 1u);
        }
    };
    // lds_written(): s_waitcnt lgkmcnt(0) by the wave that has just written the NEXT chunk's (queue, index) into LDS.  The
    // barrier at the top of the loop is what publishes them, and a barrier only orders what has completed: the compiler
    // (ROCm 7.2) puts the wait in front of the barrier after the descriptor stores below but NOT in front of the one at the
    // top of the loop, which it reaches round the back edge straight after thread 0's ds_write -- the other waves could
    // then read the slot before the write landed, i.e. the (queue, index) of two chunks ago: they repeated an old chunk
    // (harmless) and skipped their share of the new one.  Found in round 3 as 28 ... 508 unwritten voxels in 1-2 % of the
    // launches of the per-plane feather kernel on a small plan (tools/queue_stress.py; the plane-group kernels never
    // showed it in thousands of launches, but their code had the same gap).
    // 0xc07f is the s_waitcnt immediate of the gfx9 family (vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8]): lgkmcnt(0) with
    // vmcnt / expcnt at their maxima.  gfx10+ lay the fields out differently -- there the same bits would wait on something else
    // and the race would be back, silently -- so a device pass for anything but gfx9 stops here (tools/barrier_scan.py,
    // run by tests/test_isa_cpu.py, checks the listing of the build that ships).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "lds_written(): the s_waitcnt immediate below is the gfx9 encoding; re-derive it for this target"
#endif
    auto lds_written = [] { __builtin_amdgcn_s_waitcnt(0xc07f); };
    if (threadIdx.x == 0) {
        const Chunk first = settle(atomicAdd(&P.queue[queue_of(0) * QUEUE_STRIDE], 1u));
        s_q[0] = first.q;
        s_c[0] = first.c;
        lds_written();
    }
    for (int iter = 0;; ++iter) {
        __syncthreads();
        const int q = sgpr(s_q[iter & 1]);
        if (q < 0) break;
        const uint32_t c = (uint32_t)sgpr((int)s_c[iter & 1]);
        uint32_t pending = 0;
        const bool pull = threadIdx.x == 0 && given_up <= 8;
        if (pull) pending = atomicAdd(&P.queue[queue_of(given_up) * QUEUE_STRIDE], 1u);   // next chunk; looked at after this one
        const uint32_t u0 = c * (uint32_t)P.chunk;
        const int count = (int)min((uint32_t)P.chunk, total_of(q) - u0);
        if ((int)threadIdx.x < count) {   // one descriptor per thread: queue position -> (plane, list position)
            const uint32_t per_plane = per_plane_of(q);
            const uint32_t u = u0 + threadIdx.x;
            // (unit-major: all items of unit 0, then unit 1 ...  The other way round -- position r of every unit, then r + 1,
            // so that the chip writes into ALL groups' planes at once -- was measured in round 3: 0.680 whatever the grouping,
            // between consecutive groups (0.647) and spread + dealt ones (0.692) on the same buffers;
            // profiles/r03_exp_unit_minor_*.log)
            const int plane = (int)(u / per_plane);
            const uint32_t r = u - (uint32_t)plane * per_plane;
            const int64_t pos = q < 8 ? (int64_t)r * 8 + q : 8 * (int64_t)P.lane_items + r;
            const Item it = P.items[pos];
            s_item[threadIdx.x] = it;
            s_plane[threadIdx.x] = plane;
            s_aux[threadIdx.x] = pre(plane, it, pos);
        }
        __syncthreads();
        for (int j = 0; j < count; ++j) body(sgpr(s_plane[j]), sgpr(s_item[j]), s_aux[j]);
        if (threadIdx.x == 0) {
            const Chunk nxt = pull ? settle(pending) : Chunk{-1, 0u};
            s_q[(iter + 1) & 1] = nxt.q;
            s_c[(iter + 1) & 1] = nxt.c;
            lds_written();
        }
    }
}

// what a (plane, item) of the per-plane kernel needs beside its descriptor
struct PlaneAux {
    const void *tile, *ltile;
    Seam seam;
};
template <typename T>
__device__ __forceinline__ PlaneAux plane_aux(const FuseParams &P, int plane, const Item &it, int64_t pos) {
    PlaneAux A;
    A.tile = it.nref ? tile_ptr<T>(P, plane, it.a) : nullptr;
    A.seam = (sizeof(T) == 2 && P.seams) ? P.seams[pos] : Seam{-1, 0, 0, 0};
    A.ltile = ((A.seam.flags & SEAM_HAS_LEFT) && !(A.seam.flags & SEAM_LEFT_ZERO)) ? tile_ptr<T>(P, plane, A.seam.a) : nullptr;
    return A;
}
__device__ __forceinline__ Seam sgpr(Seam s) {
    s.a = sgpr(s.a);
    s.b = sgpr(s.b);
    s.c = sgpr(s.c);
    s.flags = sgpr(s.flags);
    return s;
}

template <typename T, int FLAT, bool DYN>
__global__ __launch_bounds__(256, (FLAT == 1 ? (sizeof(T) == 2 ? SQ_WAVES_F32 : 1) : (FLAT == 2 ? SQ_WAVES_F64 : SQ_WAVES_PLAIN)))
void fuse_overwrite_kernel(const FuseParams P, const int64_t n_items, const int64_t n_work) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    if (!DYN) {
        // persistent grid-stride walk: block b takes items b, b + G, ...; the next descriptor, seam record and tile
        // pointers are fetched while the current item streams
        int64_t work = blockIdx.x;
        if (work >= n_work) return;
        int plane = (int)(work / n_items);
        Item it = P.items[work - plane * n_items];
        PlaneAux A = plane_aux<T>(P, plane, it, work - plane * n_items);
        while (true) {
            const int64_t nwork = work + gridDim.x;
            const bool more = nwork < n_work;
            int nplane = plane;
            Item nit = it;
            PlaneAux nA = A;
            if (more) {
                nplane = (int)(nwork / n_items);
                nit = P.items[nwork - nplane * n_items];
                nA = plane_aux<T>(P, nplane, nit, nwork - nplane * n_items);
            }
            process_item<T, FLAT>(P, plane, it, static_cast<const T *>(A.tile), wave, lane, A.seam, static_cast<const T *>(A.ltile));
            if (!more) break;
            work = nwork;
            plane = nplane;
            it = nit;
            A = nA;
        }
        return;
    }
    for_each_queued_item<PlaneAux>(
        P, n_items, (uint32_t)P.n_planes,
        [&](int plane, const Item &it, int64_t pos) -> PlaneAux { return plane_aux<T>(P, plane, it, pos); },
        [&](int plane, const Item &it, const PlaneAux &A) {
            process_item<T, FLAT>(P, plane, it, sgpr(static_cast<const T *>(A.tile)), wave, lane, sgpr(A.seam),
                                  sgpr(static_cast<const T *>(A.ltile)));
        });
}

// ---------------------------------------------------------------------------------------------
// overwrite mode, uint16 tiles, float32 gains: plane groups
// ---------------------------------------------------------------------------------------------
// The z planes of a channel share one gain image and one plan, so a workgroup carries up to ZB of them through
// an item together: per 16-byte slot a lane loads its 8 gains ONCE, takes their reciprocals ONCE (v_rcp_f32 is
// a quarter-rate instruction: it was 4 of the ~10 issue cycles per pixel), and then streams the ZB planes'
// pixels through the same 5-instruction Markstein divide (div_u16_normal above: the arithmetic and therefore
// every bit of the result are the per-plane kernel's).  Per pixel and plane that leaves convert, multiply, two
// FMAs, convert and half a pack; the gain traffic from L2 and the gain registers shrink by the group size, so
// the ZB independent pixel loads of a slot are in flight together at 4 VGPRs each.
// Loads are unconditional (lanes outside the row read a clamped, valid address and only their store is masked):
// no divergent branch sits between the loads of a slot and their use.
// Groups are built on the device (build_groups_kernel: the gain pointers live in device memory); a plane whose
// gains need the generic divide, or that shares its gain image with nobody, is a group of one and takes the
// per-plane pipeline (process_item).
struct UnitAux {
    PlaneGroup g;
    const void *tile[ZB];
    const void *ltile[ZB];  // the planes' tiles left of the seam this item owns (Seam in common.h); feather: the second reference's
    Seam seam;              // feather: tile and source origin of a blended item's second reference (flags 0)
    Seam first;             // feather: tile and source origin of the item's first reference
};
// feather: + the third and fourth reference of an item at a corner of the grid (blend_item_zgn)
struct FeatherAux : UnitAux {
    Seam xref[2];
    const void *xtile[2][ZB];
};

// RND = 0 (overwrite: truncate): Markstein with r = v_rcp_f32(g), the arithmetic of div_u16_normal.
// RND = 1 (feather, a voxel one tile covers: round half to even): the arithmetic of div_u16_normal_ieee with its Newton
// step on the reciprocal hoisted -- r arrives refined (recip_for), two exact-residual corrections here, v_rndne.
template <int RND>
__device__ __forceinline__ float recip_for(float g) {
    float r = __builtin_amdgcn_rcpf(g);
    if (RND) r = fmaf(fmaf(-g, r, 1.0f), r, r);
    return r;
}
template <int RND>
__device__ __forceinline__ float quot_one(float n, float g, float r) {
    float q = n * r;
    q = fmaf(fmaf(-g, q, n), r, q);      // r refined (RND = 1): the correctly rounded quotient (div_u16_normal_ieee)
    if (RND) q = __builtin_rintf(q);
    return q;
}
// float64 gains (overwrite mode): the arithmetic of div_u16_normal_f64 with its reciprocal -- v_rcp_f64 and two Newton
// steps, 5 of its 8 instructions -- hoisted out of the planes' loop
__device__ __forceinline__ double recip_f64(double g) {
    double r = __builtin_amdgcn_rcp(g);
    r = fma(r, fma(-g, r, 1.0), r);
    return fma(r, fma(-g, r, 1.0), r);
}
template <int RND>
__device__ __forceinline__ float recip_of(float g) { return recip_for<RND>(g); }
template <int RND>
__device__ __forceinline__ double recip_of(double g) { return recip_f64(g); }
template <int RND>
__device__ __forceinline__ float quot_of(float n, float g, float r) { return quot_one<RND>(n, g, r); }
template <int RND>
__device__ __forceinline__ double quot_of(double n, double g, double r) {
    const double q = n * r;
    return fma(fma(-g, q, n), r, q);
}
// Two float32 lanes per instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: IEEE per component, so every bit is the scalar
// form's).  The overwrite quotient (RND = 0: 3 of its 5.8 instructions per pixel) measured no faster packed, twice (round 1 on the
// per-plane kernel, round 4 on the grouped structure in a mixed arena: tools/membw_gains "A2", 0.712 against 0.713) -- that path
// waits on memory.  The feather paths do 8.5 (one tile, rounded) to 27 (two-tile strips) instructions per pixel: there it pays.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// div_u16_normal_ieee / div_by_refined on two lanes: n / d correctly rounded, r = the refined reciprocal of d
__device__ __forceinline__ f32x2 div_by_refined2(f32x2 n, f32x2 d, f32x2 r) {
    const f32x2 q = n * r;
    return pk_fma(pk_fma(-d, q, n), r, q);
}
#ifndef SQ_FEATHER_PACKED
#define SQ_FEATHER_PACKED 1
#endif
template <int RND, typename G>
__device__ __forceinline__ uint32_t quot_pair(uint32_t word, G g_lo, G g_hi, G r_lo, G r_hi) {
    if constexpr (RND == 1 && std::is_same<G, float>::value && SQ_FEATHER_PACKED) {
        const f32x2 n = {(float)(word & 0xFFFFu), (float)(word >> 16)};
        const f32x2 q = div_by_refined2(n, f32x2{g_lo, g_hi}, f32x2{r_lo, r_hi});
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        const u16x2 p = __builtin_amdgcn_cvt_pk_u16(cvt_u32_sat(__builtin_rintf(q[0])), cvt_u32_sat(__builtin_rintf(q[1])));
        return (uint32_t)p[0] | ((uint32_t)p[1] << 16);
    }
    const G n0 = (G)(word & 0xFFFFu), n1 = (G)(word >> 16);
    const uint32_t a = cvt_u32_sat(quot_of<RND>(n0, g_lo, r_lo));
    const uint32_t b = cvt_u32_sat(quot_of<RND>(n1, g_hi, r_hi));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 p = __builtin_amdgcn_cvt_pk_u16(a, b);
    return (uint32_t)p[0] | ((uint32_t)p[1] << 16);
}
// four uint8 pixels of one 32-bit word (uint8 planes in groups, round 4): the same quotient -- the exhaustive proof of the
// shortened divide covers every numerator below 65536 -- clipped to 255 and packed
template <int RND, typename G>
__device__ __forceinline__ uint32_t quot_quad(uint32_t word, const G *g, const G *r) {
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const G n = (G)((word >> (8 * k)) & 0xFFu);
        out |= min(cvt_u32_sat(quot_of<RND>(n, g[k], r[k])), 255u) << (8 * k);
    }
    return out;
}
// 8 consecutive gains at any alignment
__device__ __forceinline__ void load_gains(const char *p, float (&g)[8]) {
    const f32x4 a = ldg<F32x4U>(p), b = ldg<F32x4U>(p + 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        g[e] = a[e];
        g[4 + e] = b[e];
    }
}
__device__ __forceinline__ void load_gains(const char *p, double (&g)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f64x2 a = ldg<F64x2U>(p + 16 * q);
        g[2 * q] = a[0];
        g[2 * q + 1] = a[1];
    }
}

// 16 consecutive gains (a uint8 plane's 16-byte pixel vector)
template <typename GT>
__device__ __forceinline__ void load_gains(const char *p, GT (&g)[16]) {
    GT lo[8], hi[8];
    load_gains(p, lo);
    load_gains(p + 8 * sizeof(GT), hi);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        g[e] = lo[e];
        g[8 + e] = hi[e];
    }
}

// (One row per workgroup -- wave w taking slot w, 2 + ZB loads and ZB stores per thread -- launched one-shot, the regime
// in which a bare copy gains 10 %, was built and measured with groups of 5 / 3 / 2: 0.53 / 0.43 / 0.39 against 0.62-0.64
// for this form on the same box; the descriptor, its LDS hand-over and the tile pointers cost more per 14-36 KB workgroup
// than the access pattern gives back.  profiles/r02_exp20_row_per_workgroup.log)
// G = NoGain: planes WITHOUT a flatfield carried through an item together -- nothing is shared between them but the
// geometry, yet five planes per thread write faster than one plane per launch when the planes lie in different stretches
// of device memory (DESIGN.md 5.1 point 9: the bare 5-plane copy 0.72-0.75 of peak against 0.64-0.65 plane by plane)
struct NoGain {};
// T = uint8_t (round 4): the same row loop with 16 pixels per lane and slot and 128 pixels per line.  uint8 planes have no seam
// owners (a line is 128 pixels, two per lane in the whole-line pass: not built) -- the kernel ignores the plan's seam records for
// the whole plane, which is a partition of the canvas like honouring them for the whole plane is.
template <bool FULL, int RND = 0, typename G = float, typename T = uint16_t>
__device__ __forceinline__ void process_item_zg(const FuseParams &P, const UnitAux &A, const int gn, const Item &it,
                                                const int wave, const int lane) {
    constexpr bool GAINS = !std::is_same<G, NoGain>::value;
    typedef typename std::conditional<GAINS, G, float>::type GT;      // the arithmetic type where there is arithmetic
    constexpr uint32_t GSZ = sizeof(GT), TSZ = sizeof(T), TMAX = sizeof(T) == 1 ? 255u : 65535u;
    constexpr int VEC = 16 / (int)sizeof(T), LINE = 128 / (int)sizeof(T);
    constexpr int SLOTS = BLOCK_COLS / VEC / 64 + 1;
    const int rows = it.hw >> 16, n = it.hw & 0xFFFF;
    const int sflags = sizeof(T) == 2 ? sgpr(A.seam.flags) : 0;
    const bool leave_tail = sflags & SEAM_LEAVE_TAIL;
    T *cplane[ZB];
    const T *tiles[ZB];
#pragma unroll
    for (int z = 0; z < ZB; ++z) {
        const int zz = (FULL || z < gn) ? z : 0;
        cplane[z] = static_cast<T *>(P.canvas) + (int64_t)sgpr(A.g.plane[zz]) * P.canvas_plane_stride;
        tiles[z] = sgpr(static_cast<const T *>(A.tile[zz]));
    }
    if (!it.nref) {   // uncovered canvas: zeros (da.zeros, stitcher.py:362)
        const bool has_left = sflags & SEAM_HAS_LEFT, lzero = sflags & SEAM_LEFT_ZERO;
        for (int r = wave; r < rows; r += 4) {
            const int64_t doff = (int64_t)(it.dst_y + r) * P.canvas_pitch + it.dst_x;
            const int mis = (int)((reinterpret_cast<uintptr_t>(cplane[0] + doff) / sizeof(T)) & (LINE - 1));
            const int head = (has_left && mis > 0) ? mis : 0;   // pixels of the left neighbour in the seam's line
            if (head && !lzero) {   // the seam's line: the left tile's pixels, then zeros (see head_line below)
                const int p = lane - mis;
                const bool left = p < 0;
                const int po = left ? p : -1;
                const int sb = sgpr(A.seam.b), sc = sgpr(A.seam.c);
                GT eg = GT(1), er = GT(1);
                if constexpr (GAINS) {
                    const GT *lflat = static_cast<const GT *>(P.flat_ptrs[sgpr(A.g.plane[0])]);
                    eg = ldg_s<GT>(reinterpret_cast<const char *>(lflat + (int64_t)(sb + r) * P.tile_w + sc) + po * (int)GSZ);
                    er = recip_of<RND>(eg);
                }
#pragma unroll
                for (int z = 0; z < ZB; ++z)
                    if (FULL || z < gn) {
                        const T *lt = sgpr(static_cast<const T *>(A.ltile[z]));
                        const T e = ldg_s<T>(reinterpret_cast<const char *>(lt + (int64_t)(sb + r) * P.tile_pitch + sc) + po * (int)TSZ);
                        uint32_t kq = e;
                        if constexpr (GAINS) kq = min(cvt_u32_sat(quot_of<RND>((GT)e, eg, er)), TMAX);
                        if (!left) kq = 0u;
                        stg_s<T>(reinterpret_cast<char *>(cplane[z] + doff) + p * (int)TSZ, (T)kq);
                    }
            }
#pragma unroll
            for (int z = 0; z < ZB; ++z)
                if (FULL || z < gn) {
                    T *d = cplane[z] + doff;
                    if (head && lzero) row_zero<T>(d - head, n + head, lane, leave_tail);                // zeros from the line boundary
                    else if (head) row_zero<T>(d + (LINE - head), n - (LINE - head), lane, leave_tail);    // after the seam's line
                    else row_zero<T>(d, n, lane, leave_tail);
                }
        }
        return;
    }
    const GT *flat = nullptr;
    if constexpr (GAINS) flat = static_cast<const GT *>(P.flat_ptrs[sgpr(A.g.plane[0])]);
    for (int r = wave; r < rows; r += 4) {
        const int64_t doff = (int64_t)(it.dst_y + r) * P.canvas_pitch + it.dst_x;
        const int64_t soff = (int64_t)(it.b + r) * P.tile_pitch + it.c;
        // row bases are wave-uniform (scalar registers); a lane adds ONE 32-bit byte offset, shared by the planes, so the
        // loads and stores take the scalar-base + vector-offset form and no 64-bit address pair per plane lives in VGPRs
        const char *frow = GAINS ? reinterpret_cast<const char *>(flat + (int64_t)(it.b + r) * P.tile_w + it.c) : nullptr;
        const char *srow[ZB];
        char *drow[ZB];
#pragma unroll
        for (int z = 0; z < ZB; ++z) {
            srow[z] = reinterpret_cast<const char *>(tiles[z] + soff);
            drow[z] = reinterpret_cast<char *>(cplane[z] + doff);
        }
        // the phase of the row inside a 128-byte line: the same for every plane of the group (build_groups_kernel)
        const int mis = (int)((reinterpret_cast<uintptr_t>(drow[0]) / sizeof(T)) & (LINE - 1));
        // seams (common.h): with head_line this item writes the whole line its first pixel falls in (below, after the
        // vectors); with leave_tail it stops at the last line boundary and the right neighbour writes the rest
        const bool head_line = (sflags & SEAM_HAS_LEFT) && mis > 0;
        const int n_own = leave_tail ? n - ((n + mis) & (LINE - 1)) : n;   // n >= LINE where a flag is set
        const int v_first = head_line ? LINE / VEC : (mis + VEC - 1) / VEC, v_end = (n_own + mis) / VEC;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            if (64 * k >= v_end || v_end <= v_first) break;   // wave-uniform: no whole vector (left) in this row
            const int v = lane + 64 * k;
            const bool act = v >= v_first && v < v_end;
            const int p0 = v * VEC - mis;
            // Lanes without a vector of their own load the row's first / last WHOLE vector (their store is masked): no source line is
            // touched that the row's own vectors do not touch.  Clamping to the row's first / last PIXELS instead (rounds 1-3) made
            // every LEAVE_TAIL item fetch the source line its neighbour owns, and every row the line before its first vector:
            // 26.52 -> 26.25 ms per 40-plane launch of config 3 (0.7235 -> 0.7307), profiles/r04_exp_clamp_inside.log.
            const uint32_t o = (uint32_t)min(max(p0, v_first * VEC - mis), (v_end - 1) * VEC - mis);
            GT g[VEC], rc[VEC];
            if constexpr (GAINS) load_gains(frow + o * GSZ, g);
            u32x4 px[ZB];
#pragma unroll
            for (int z = 0; z < ZB; ++z)
                if (FULL || z < gn) px[z] = ldg<U32x4U>(srow[z] + o * TSZ);
            if constexpr (GAINS) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) rc[c] = recip_of<RND>(g[c]);
            }
#pragma unroll
            for (int z = 0; z < ZB; ++z)
                if (FULL || z < gn) {
                    u32x4 ov = px[z];
                    if constexpr (GAINS) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if constexpr (sizeof(T) == 2) ov[c] = quot_pair<RND, GT>(px[z][c], g[2 * c], g[2 * c + 1], rc[2 * c], rc[2 * c + 1]);
                            else ov[c] = quot_quad<RND, GT>(px[z][c], &g[4 * c], &rc[4 * c]);
                        }
                    }
                    if (act) stg_nt_at(drow[z], o * TSZ, ov);
                }
        }
        if (head_line) {
            // the seam's line, one pixel per lane: lane j writes byte 2j of the line, the pixels before dst_x from the
            // left neighbour's tile (or zero fill), the rest from this item's -- one whole-line store per plane
            const int p = lane - mis;   // pixel relative to dst_x: -mis .. 63 - mis (< n)
            const bool left = p < 0, lzero = sflags & SEAM_LEFT_ZERO;
            const int sb = sgpr(A.seam.b), sc = sgpr(A.seam.c);
            const int po = (left && lzero) ? 0 : p;   // left of a zero-fill seam: any valid address, the value is replaced
            GT eg = GT(1), er = GT(1);
            if constexpr (GAINS) {
                const char *lfrow = lzero ? frow : reinterpret_cast<const char *>(flat + (int64_t)(sb + r) * P.tile_w + sc);
                eg = ldg_s<GT>((left ? lfrow : frow) + po * (int)GSZ);
            }
            T e[ZB];
#pragma unroll
            for (int z = 0; z < ZB; ++z)
                if (FULL || z < gn) {
                    const T *lt = sgpr(static_cast<const T *>(A.ltile[z]));
                    const char *lrow = lzero ? srow[z] : reinterpret_cast<const char *>(lt + (int64_t)(sb + r) * P.tile_pitch + sc);
                    e[z] = ldg_s<T>((left ? lrow : srow[z]) + po * (int)TSZ);
                }
            if constexpr (GAINS) er = recip_of<RND>(eg);
#pragma unroll
            for (int z = 0; z < ZB; ++z)
                if (FULL || z < gn) {
                    uint32_t kq = e[z];
                    if constexpr (GAINS) kq = min(cvt_u32_sat(quot_of<RND>((GT)e[z], eg, er)), TMAX);
                    if (left && lzero) kq = 0;
                    stg_s<T>(drow[z] + p * (int)TSZ, (T)kq);
                }
        }
        // the row's other edges (canvas pixels before the first / after the last whole 16-byte vector that no seam line
        // covers): lanes 0..7 the head, 8..15 the tail, one pixel each -- after the vectors, so that nothing of them lives
        // across the slot loop
        const int head_end = head_line ? 0 : min(n_own, v_first * VEC - mis);
        const int tail_start = max(head_end, v_end * VEC - mis);
        int ep = -1;
        if (lane < VEC) {
            if (lane < head_end) ep = lane;
        } else if (lane < 2 * VEC) {
            if (tail_start + (lane - VEC) < n_own) ep = tail_start + (lane - VEC);
        }
        if (__builtin_amdgcn_ballot_w64(ep >= 0)) {   // wave-uniform: many rows have no edge pixels at all
            const uint32_t eo = (uint32_t)max(ep, 0);
            GT eg = GT(1), er = GT(1);
            if constexpr (GAINS) eg = ldg_s<GT>(frow + eo * GSZ);
            T e[ZB];
#pragma unroll
            for (int z = 0; z < ZB; ++z)
                if (FULL || z < gn) e[z] = ldg_s<T>(srow[z] + eo * TSZ);
            if constexpr (GAINS) er = recip_of<RND>(eg);
#pragma unroll
            for (int z = 0; z < ZB; ++z)
                if (FULL || z < gn) {
                    uint32_t kq = e[z];
                    if constexpr (GAINS) kq = min(cvt_u32_sat(quot_of<RND>((GT)e[z], eg, er)), TMAX);
                    if (ep >= 0) stg_s<T>(drow[z] + eo * TSZ, (T)kq);
                }
        }
    }
}

#ifndef SQ_WAVES_ZG
#define SQ_WAVES_ZG 1
#endif
template <typename G, bool DYN, typename T = uint16_t>
__global__ __launch_bounds__(256, SQ_WAVES_ZG) void fuse_overwrite_zg_kernel(const FuseParams P, const int64_t n_items) {
    constexpr int FLAT = std::is_same<G, NoGain>::value ? 0 : (sizeof(G) == 8 ? 2 : 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint32_t n_groups = *P.n_groups;
    auto pre = [&](int unit, const Item &it, int64_t pos) -> UnitAux {
        UnitAux A;
        A.g = P.groups[unit];
        A.seam = P.seams ? P.seams[pos] : Seam{-1, 0, 0, 0};
        const bool left_tile = (A.seam.flags & SEAM_HAS_LEFT) && !(A.seam.flags & SEAM_LEFT_ZERO);
#pragma unroll
        for (int z = 0; z < ZB; ++z) {
            A.tile[z] = (it.nref && z < A.g.n) ? tile_ptr<T>(P, A.g.plane[z], it.a) : nullptr;
            A.ltile[z] = (left_tile && z < A.g.n) ? tile_ptr<T>(P, A.g.plane[z], A.seam.a) : nullptr;
        }
        return A;
    };
    auto body = [&](int, const Item &it, const UnitAux &A) {
        const int gn = sgpr(A.g.n);
        if (gn == 1) {   // the per-plane pipeline (a group of one through process_item_zg measured the same)
            process_item<T, FLAT>(P, sgpr(A.g.plane[0]), it, sgpr(static_cast<const T *>(A.tile[0])), wave, lane, sgpr(A.seam),
                                  sgpr(static_cast<const T *>(A.ltile[0])));
        } else if (gn == ZB) {
            process_item_zg<true, 0, G, T>(P, A, gn, it, wave, lane);
        } else {
            process_item_zg<false, 0, G, T>(P, A, gn, it, wave, lane);
        }
    };
    if (DYN) {
        for_each_queued_item<UnitAux>(P, n_items, n_groups, pre, body);
    } else {
        __shared__ Item s_it;
        __shared__ UnitAux s_A;
        const int64_t n_work = (int64_t)n_groups * n_items;
        for (int64_t work = blockIdx.x; work < n_work; work += gridDim.x) {
            __syncthreads();
            if (threadIdx.x == 0) {
                const int unit = (int)(work / n_items);
                s_it = P.items[work - unit * n_items];
                s_A = pre(unit, s_it, work - unit * n_items);
            }
            __syncthreads();
            body(0, sgpr(s_it), s_A);
        }
    }
}

// One block: deal the planes into groups of <= zb that share a gain image (and a fast gain class, and the phase
// of their canvas inside a 128-byte line).  Planes are few (C x Z of a region); more than CAP of them, or zb = 1,
// and every plane is its own group.
__global__ __launch_bounds__(256) void build_groups_kernel(const void *const *flat_ptrs, const uint32_t *cls, uint32_t cls_mask,
                                                          int n_planes, int64_t plane_stride_bytes, int zb, uint32_t *n_groups,
                                                          PlaneGroup *groups, bool consecutive) {
    // flat_ptrs == NULL (feather without gains): the planes share their geometry only, any of them may go together
    constexpr int CAP = 1024;
    __shared__ uint64_t key[CAP];
    __shared__ int first[CAP], open_group[CAP];
    __shared__ short members[CAP], dealt[CAP];
    __shared__ PlaneGroup out[CAP];
    const int tid = threadIdx.x;
    if (n_planes > CAP || zb <= 1) {
        for (int p = tid; p < n_planes; p += 256) {
            PlaneGroup g{};
            g.n = 1;
            g.plane[0] = p;
            groups[p] = g;
        }
        if (tid == 0) *n_groups = (uint32_t)n_planes;
        return;
    }
    for (int p = tid; p < n_planes; p += 256) {
        const void *f = flat_ptrs ? flat_ptrs[p] : nullptr;
        const bool ok = flat_ptrs ? (f && (cls[p] & cls_mask) == 0) : true;
        // 0 = "groups with nobody"; else the gain pointer (< 2^56) and the canvas plane's phase in a line
        key[p] = ok ? ((reinterpret_cast<uint64_t>(f) << 7) | (uint64_t)(((int64_t)p * plane_stride_bytes) & 127) | (1ull << 63)) : 0;
        open_group[p] = -1;
    }
    __syncthreads();
    for (int p = tid; p < n_planes; p += 256) {
        int f = p;
        if (key[p])
            for (int q = 0; q < p; ++q)
                if (key[q] == key[p]) {
                    f = q;
                    break;
                }
        first[p] = f;
    }
    __syncthreads();
    if (tid == 0) {
        // The m planes of one key become ceil(m / zb) groups, and the planes are DEALT to them round-robin in index order
        // (plane j of the key -> the key's group j mod its group count), not taken five consecutive ones at a time: plane
        // index = position in the canvas allocation, and the write rate of a group depends on how far apart its planes
        // lie -- stretches of tens of GiB of device memory behave like separate banks of resources for this access pattern
        // (row segments at the canvas pitch): planes inside one stretch 0.56 of peak for the bare row fill, planes from two
        // stretches 0.74 (tools/membw_gains, profiles/r03_exp_*placement*.log, DESIGN.md 5.1 point 8).  Dealing spreads every
        // group over the whole run of its key's planes; the group sizes come out balanced (4 + 4 + 4 instead of 5 + 5 + 2).
        int ng = 0;
        for (int p = 0; p < n_planes; ++p) members[p] = dealt[p] = 0;
        for (int p = 0; p < n_planes; ++p)
            if (key[p]) ++members[first[p]];           // planes of the key, counted at its first plane (its "lead")
        for (int p = 0; p < n_planes; ++p) {           // every key its run of group slots, keys in index order
            if (!key[p]) {
                open_group[p] = ng++;                  // groups with nobody
            } else if (first[p] == p) {
                open_group[p] = ng;
                ng += (members[p] + zb - 1) / zb;
            }
        }
        for (int g = 0; g < ng; ++g) out[g].n = 0;
        for (int p = 0; p < n_planes; ++p) {
            int g = open_group[p];
            if (key[p]) {
                const int lead = first[p];
                const int j = dealt[lead]++;           // consecutive (round 2, for A/B runs): planes j zb .. j zb + zb - 1 together
                g = open_group[lead] + (consecutive ? j / zb : j % ((members[lead] + zb - 1) / zb));
            }
            out[g].plane[out[g].n++] = p;
        }
        // unused slots of a group repeat its first plane (the kernels read plane[z] for z < n only; keep the rest valid)
        for (int g = 0; g < ng; ++g)
            for (int z = out[g].n; z < 7; ++z) out[g].plane[z] = out[g].plane[0];
        *n_groups = (uint32_t)ng;
        first[0] = ng;
    }
    __syncthreads();
    const int ng = first[0];
    for (int g = tid; g < ng; g += 256) groups[g] = out[g];
}

// ---------------------------------------------------------------------------------------------
// feather mode (extension; definition = oracle/stitch_oracle.py fuse_plane_feather)
//   voxels covered by ONE tile:  out = v
//   voxels covered by several:   out = sum_i w_i v_i / sum_i w_i   (float32, tiles in write order,
//                                multiply and add separate like numpy: -ffp-contract=off)
//   w_i = min(sx+1, W-sx, sy+1, H-sy) in the tile's own coordinates,
//   v_i = tile / float32(flatfield) when a flatfield is given (no clip).
//   integer canvases: rint + clip.  Uncovered voxels: 0.
// Persistent grid over (plane, item) like the overwrite kernel.  Items with nothing to blend (no
// tile, or one) go through the overwrite kernel's pipelined row copy; blended items deal their
// (row, 8-pixel group) pairs to the workgroup's threads (blend_item / blend_group below).
// ---------------------------------------------------------------------------------------------
template <typename OutT>
__device__ __forceinline__ OutT feather_out(float o) {
    if (sizeof(OutT) == 4) return (OutT)o;
    const float hi = sizeof(OutT) == 1 ? 255.0f : 65535.0f;
    return (OutT)fminf(fmaxf(rintf(o), 0.0f), hi);
}

// One 8-pixel group (p0 .. p0+7 of item row r) blended from the item's references.  The loads of up
// to four references (pixels and gains) are issued before the first use; references are
// wave-uniform, so the loops over them run on scalar registers.
constexpr int BLEND_MAXR = 4;   // references whose pixel vectors are loaded ahead (more than 4 tiles meet nowhere in a grid)

template <typename T, typename OutT, int FLAT, bool FAST>
__device__ __forceinline__ void blend_group(const FuseParams &P, int plane, const Item &it, const char *flat, int r, int p0,
                                            OutT *dst) {
    constexpr int VEC = 8;
    constexpr int MAXR = BLEND_MAXR;
    const int nref = it.nref;
    float acc[VEC], wsum[VEC], last[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = wsum[e] = last[e] = 0.0f;
    auto blend = [&](const float (&px)[VEC], int y, int x0) {
        const int wy = min(y + 1, P.tile_h - y);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int x = x0 + e;
            const float w = (float)min(min(x + 1, P.tile_w - x), wy);
            acc[e] = __fadd_rn(acc[e], __fmul_rn(w, px[e]));
            wsum[e] = __fadd_rn(wsum[e], w);
            last[e] = px[e];
        }
    };
    if (sizeof(T) == 2 && FLAT != 2 && nref <= MAXR) {
        u32x4 raw[MAXR];
        f32x4 g[MAXR][2];
        int ys[MAXR], xs[MAXR];
#pragma unroll
        for (int k = 0; k < MAXR; ++k)
            if (k < nref) {
                const Ref rf = P.refs[it.a + k];
                ys[k] = rf.src_y + it.b + r;
                xs[k] = rf.src_x + it.c + p0;
                raw[k] = ldg<U32x4U>(tile_ptr<T>(P, plane, rf.tile) + (int64_t)ys[k] * P.tile_pitch + xs[k]);
                if (FLAT == 1 && flat) {
                    const float *gp = reinterpret_cast<const float *>(flat) + (int64_t)ys[k] * P.tile_w + xs[k];
                    g[k][0] = ldg<F32x4U>(gp);
                    g[k][1] = ldg<F32x4U>(gp + 4);
                }
            }
#pragma unroll
        for (int k = 0; k < MAXR; ++k)
            if (k < nref) {
                float px[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    px[e] = (float)Pix<uint16_t>::get(raw[k], e);
                    if (FLAT == 1 && flat) px[e] = FAST ? div_u16_normal_ieee(px[e], g[k][e >> 2][e & 3]) : __fdiv_rn(px[e], g[k][e >> 2][e & 3]);
                }
                blend(px, ys[k], xs[k]);
            }
    } else {
        for (int k = 0; k < nref; ++k) {
            const Ref rf = P.refs[it.a + k];
            const T *tile = tile_ptr<T>(P, plane, rf.tile);
            const int y = rf.src_y + it.b + r;
            const int x0 = rf.src_x + it.c + p0;
            const T *src = tile + (int64_t)y * P.tile_pitch + x0;
            float px[VEC];
            if (sizeof(T) == 2) {
                const u32x4 raw = ldg<U32x4U>(src);
#pragma unroll
                for (int e = 0; e < VEC; ++e) px[e] = (float)Pix<uint16_t>::get(raw, e);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) px[e] = (float)ldg_s<T>(src + e);
            }
            if (FLAT && flat) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const char *gp = flat + ((int64_t)y * P.tile_w + x0 + e) * (FLAT == 2 ? 8 : 4);
                    const float gain = FLAT == 2 ? (float)ldg_s<double>(gp) : ldg_s<float>(gp);
                    px[e] = FAST ? div_u16_normal_ieee(px[e], gain) : __fdiv_rn(px[e], gain);
                }
            }
            blend(px, y, x0);
        }
    }
    OutT o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = feather_out<OutT>(nref == 1 ? last[e] : (nref ? __fdiv_rn(acc[e], wsum[e]) : 0.0f));
    if (sizeof(OutT) == 2) {
        u32x4 out;
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = (uint32_t)(uint16_t)o[2 * q] | ((uint32_t)(uint16_t)o[2 * q + 1] << 16);
        stg_nt(dst, out);
    } else if (sizeof(OutT) == 4) {
        u32x4 lo, hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lo[q] = __float_as_uint((float)o[q]);
            hi[q] = __float_as_uint((float)o[4 + q]);
        }
        stg_nt(dst, lo);
        stg_nt(dst + 4, hi);
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) stg_s<OutT>(dst + e, o[e]);
    }
}

// A blended item: its (row, 8-pixel group) pairs are dealt to the 256 threads of the workgroup, so a
// narrow overlap strip (244 pixels = 30 groups per row) keeps every lane busy; stores are aligned to
// the 8-pixel group of the canvas row, the pixels before / after the aligned body go one per thread.
template <typename T, typename OutT, int FLAT, bool FAST>
__device__ __forceinline__ void blend_item(const FuseParams &P, int plane, const Item &it, int tid) {
    constexpr int VEC = 8;
    const int rows = it.hw >> 16, n = it.hw & 0xFFFF;
    const int nref = it.nref;
    OutT *canvas = static_cast<OutT *>(P.canvas) + plane * P.canvas_plane_stride;
    const char *flat = (FLAT && P.flat_ptrs) ? static_cast<const char *>(P.flat_ptrs[plane]) : nullptr;
    const int G = n / VEC + 1;   // upper bound of the whole groups of a row
    auto locate = [&](int idx, int &r, int &p0, OutT *&dst) -> bool {   // (row, group) pair idx -> where it lives; false: no such group
        r = idx / G;
        const int j = idx - r * G;
        OutT *drow = canvas + (int64_t)(it.dst_y + r) * P.canvas_pitch + it.dst_x;
        const int mis = (int)((reinterpret_cast<uintptr_t>(drow) / sizeof(OutT)) & (VEC - 1));
        const int v = (mis ? 1 : 0) + j;
        p0 = v * VEC - mis;
        dst = drow + p0;
        return v < (n + mis) / VEC;
    };
    // (A two-stage software pipeline over a thread's pairs -- the next pair's pixel vectors loaded before the current
    // pair is blended -- was built and measured in round 2: 0.45-0.46 against 0.51-0.52 for this plain loop at equal item
    // height, profiles/r02_exp9_feather.log: the second set of vectors costs the occupancy that hides the latency.)
    for (int idx = tid; idx < rows * G; idx += 256) {
        int r, p0;
        OutT *dst;
        if (!locate(idx, r, p0, dst)) continue;
        blend_group<T, OutT, FLAT, FAST>(P, plane, it, flat, r, p0, dst);
    }
    for (int idx = tid; idx < rows * 2 * VEC; idx += 256) {
        const int r = idx / (2 * VEC), l = idx - r * 2 * VEC;
        OutT *drow = canvas + (int64_t)(it.dst_y + r) * P.canvas_pitch + it.dst_x;
        const int mis = (int)((reinterpret_cast<uintptr_t>(drow) / sizeof(OutT)) & (VEC - 1));
        const int v_first = mis ? 1 : 0, v_end = (n + mis) / VEC;
        const int head_end = min(n, v_first * VEC - mis);
        const int tail_start = max(head_end, v_end * VEC - mis);
        int p = -1;
        if (l < VEC) {
            if (l < head_end) p = l;
        } else if (tail_start + (l - VEC) < n) {
            p = tail_start + (l - VEC);
        }
        if (p < 0) continue;
        float acc = 0.0f, wsum = 0.0f, last = 0.0f;
        for (int k = 0; k < nref; ++k) {
            const Ref rf = P.refs[it.a + k];
            const int y = rf.src_y + it.b + r, x = rf.src_x + it.c + p;
            float v = (float)ldg_s<T>(tile_ptr<T>(P, plane, rf.tile) + (int64_t)y * P.tile_pitch + x);
            if (FLAT && flat) {
                const char *gp = flat + ((int64_t)y * P.tile_w + x) * (FLAT == 2 ? 8 : 4);
                const float gain = FLAT == 2 ? (float)ldg_s<double>(gp) : ldg_s<float>(gp);
                v = FAST ? div_u16_normal_ieee(v, gain) : __fdiv_rn(v, gain);
            }
            const float w = (float)min(min(x + 1, P.tile_w - x), min(y + 1, P.tile_h - y));
            acc = __fadd_rn(acc, __fmul_rn(w, v));
            wsum = __fadd_rn(wsum, w);
            last = v;
        }
        stg_s<OutT>(drow + p, feather_out<OutT>(nref == 1 ? last : (nref ? __fdiv_rn(acc, wsum) : 0.0f)));
    }
}

// one (plane, item) of a feather plan, all threads of the workgroup together
template <typename T, typename OutT, int FLAT>
__device__ __forceinline__ void feather_one_item(const FuseParams &P, int plane, const Item &it, int wave, int lane) {
    const int nref = it.nref;
    if (sizeof(OutT) == sizeof(T) && FLAT != 2 && nref <= 1) {
        // nothing to blend: uncovered canvas, or one tile -> the overwrite kernel's pipelined copy
        // (with float32 gains: divide, round half to even, clip -- what the blend yields for a
        // single reference)
        Item one = it;
        const T *tile = nullptr;
        if (nref) {
            const Ref rf = P.refs[it.a];
            tile = tile_ptr<T>(P, plane, rf.tile);
            one.a = rf.tile;
            one.b = rf.src_y + it.b;
            one.c = rf.src_x + it.c;
        }
        process_item<T, FLAT == 2 ? 0 : FLAT, 1>(P, plane, one, tile, wave, lane);
        return;
    }
    // float32 gains the pre-pass found all normal: the 8-slot divide, bit-identical to the IEEE
    // quotient for this operand class (exhaustive self-test), instead of the 11-slot generic one
    const bool fast = FLAT == 1 && P.flat_class && (P.flat_class[plane] & 1u) == 0;
    if (fast) blend_item<T, OutT, FLAT, true>(P, plane, it, threadIdx.x);
    else blend_item<T, OutT, FLAT, false>(P, plane, it, threadIdx.x);
}

#ifndef SQ_WAVES_FEATHER_F32
#define SQ_WAVES_FEATHER_F32 4      // (5 until round 4: spill code inside the item loops, see SQ_WAVES_F32)
#endif
template <typename T, typename OutT, int FLAT, bool DYN>
__global__ __launch_bounds__(256, (FLAT == 1 && sizeof(T) == 2 && sizeof(OutT) == 2 ? SQ_WAVES_FEATHER_F32 : 1))
void fuse_feather_kernel(const FuseParams P, const int64_t n_items, const int64_t n_work) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (DYN) {
        // single-tile and blended items cost very differently: the queues (one "rest" queue here, the
        // feather plan is not lane-interleaved) keep every workgroup busy until the end
        for_each_queued_item<int>(P, n_items, (uint32_t)P.n_planes, [](int, const Item &, int64_t) -> int { return 0; },
                                  [&](int plane, const Item &it, const int &) { feather_one_item<T, OutT, FLAT>(P, plane, it, wave, lane); });
    } else {
        for (int64_t work = blockIdx.x; work < n_work; work += gridDim.x) {
            const int plane = (int)(work / n_items);
            feather_one_item<T, OutT, FLAT>(P, plane, P.items[work - plane * n_items], wave, lane);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// feather mode, uint16 tiles and canvas, float32 gains: plane groups
// ---------------------------------------------------------------------------------------------
// The z planes of a channel share the gain image AND the geometry, so everything of a blended voxel that does not
// depend on the pixel values is the same for them: the two gains and their refined reciprocals, the two weights, their
// sum and ITS refined reciprocal.  A thread works that out once per 8-pixel group and then runs the planes' pixels
// through it: per plane and voxel two quotients of 3 instructions (div_u16_normal_ieee with the reciprocal handed in),
// two products, one sum, the division by the weight sum in the same 3-instruction form, round, clip, pack -- about 20
// VALU instructions (half as many on the packed-float32 pipe) against 45 for a plane on its own.  The arithmetic is the
// per-plane blend's, operation for operation: v_k = n_k / g_k correctly rounded, acc = w_0 v_0 + w_1 v_1 (multiply and add
// separate), acc / wsum correctly rounded.  Each quotient is Markstein's: refined reciprocal (one Newton step: the correctly
// rounded reciprocal on this hardware), n r, one exact-residual correction -- without the compiler's v_div_scale /
// v_div_fmas / v_div_fixup range handling, which cannot trigger here: wsum is an integer in [2, 2^14], and with every gain
// of the plane moderate (2^-20 <= |g| < 2^20: bit 1 of the gain class, else the plane is a group of one) acc is 0 or
// 2^-44 <= |acc| < 2^53.  sq_selftest_blend_divide compares acc / wsum with the compiler's division on the device for every
// mantissa of acc x every weight sum (rounds 1-3 ran a second correction; tools/div_probe.hip: it never changed a bit).
// Items of one tile go through process_item_zg<.., RND = 1>; spans that three or four tiles cover (the corners of a
// grid, (overlap / tile)^2 of the canvas) take the per-plane blend, plane after plane.
__device__ __forceinline__ float div_by_refined(float n, float d, float r) {   // r = recip_for<1>(d)
    const float q = n * r;
    return fmaf(fmaf(-d, q, n), r, q);
}

// OutT = uint16_t (round, clip, pack) or float (the blended value as it is: two 16-byte stores per 8 voxels).
// NREF = 2: a strip two tiles cover.  NREF = 1 / 0 (float canvases only): a voxel ONE tile covers (its value, divided by its
// gain) / none (zero) -- the uint16 canvas sends those through process_item_zg's pipelined rows, which write the tile dtype.
// A float32 canvas is held to the north star's tolerance for fused float voxels, 1e-5 relative, not to the bits of the
// definition: its quotients are n * r and acc * r with r the reciprocal after one Newton step (each within 2^-22 of the
// correctly rounded quotient) instead of the three-instruction exact sequence -- 3 instructions per blended voxel instead of
// 9.  Integer canvases stay bit-equal to the definition (rounding half to even hangs on the exact quotient).
template <int FLAT, bool FULL, typename OutT = uint16_t, int NREF = 2>
__device__ __forceinline__ void blend_item_zg(const FuseParams &P, const UnitAux &A, const int gn, const Item &it, const int tid) {
    typedef uint16_t T;
    constexpr int VEC = 8;
    constexpr bool F32OUT = sizeof(OutT) == 4;
    static_assert(NREF == 2 || F32OUT, "one-tile and empty items of an integer canvas take process_item_zg");
    const int rows = it.hw >> 16, n = it.hw & 0xFFFF;
    OutT *cplane[ZB];
    const T *t0[ZB], *t1[ZB];
#pragma unroll
    for (int z = 0; z < ZB; ++z) {
        const int zz = (FULL || z < gn) ? z : 0;
        cplane[z] = static_cast<OutT *>(P.canvas) + (int64_t)sgpr(A.g.plane[zz]) * P.canvas_plane_stride;
        t0[z] = sgpr(static_cast<const T *>(A.tile[zz]));
        t1[z] = sgpr(static_cast<const T *>(A.ltile[zz]));
    }
    // FLAT = 2: float64 gains, which feather mode takes as float32 (the per-plane blend casts every gain it loads: the
    // arithmetic is float32 in either case) -- loaded as doubles here, cast once per 8-pixel group
    typedef typename std::conditional<FLAT == 2, double, float>::type GM;
    const GM *flat = (FLAT && NREF) ? static_cast<const GM *>(P.flat_ptrs[sgpr(A.g.plane[0])]) : nullptr;
    const int ya = sgpr(A.first.b), xa = sgpr(A.first.c), yb = sgpr(A.seam.b), xb = sgpr(A.seam.c);
    const int G = n / VEC + 1;   // upper bound of the whole groups of a row
    auto load_gain8 = [&](int y, int x, f32x2 (&g)[4]) {
        if constexpr (FLAT == 1) {
            const float *gp = reinterpret_cast<const float *>(flat) + (int64_t)y * P.tile_w + x;
            const f32x4 a = ldg<F32x4U>(gp), b = ldg<F32x4U>(gp + 4);
            g[0] = f32x2{a[0], a[1]}, g[1] = f32x2{a[2], a[3]}, g[2] = f32x2{b[0], b[1]}, g[3] = f32x2{b[2], b[3]};
        } else if constexpr (FLAT == 2) {
            const double *gp = reinterpret_cast<const double *>(flat) + (int64_t)y * P.tile_w + x;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const f64x2 a = ldg<F64x2U>(gp + 2 * h);
                g[h] = f32x2{(float)a[0], (float)a[1]};
            }
        }
    };
    auto refined = [](f32x2 d) {      // recip_for<1> on two lanes (v_rcp_f32 has no packed form)
        const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        return pk_fma(pk_fma(-d, r, f32x2{1.0f, 1.0f}), r, r);
    };
    for (int idx = tid; idx < rows * G; idx += 256) {
        const int r = idx / G, j = idx - r * G;
        const int64_t doff = (int64_t)(it.dst_y + r) * P.canvas_pitch + it.dst_x;
        // the planes of a group sit at the same phase of a 128-byte line, so also of 16 (32) bytes
        const int mis = (int)((reinterpret_cast<uintptr_t>(cplane[0] + doff) / sizeof(OutT)) & (VEC - 1));
        const int v = (mis ? 1 : 0) + j;
        if (v >= (n + mis) / VEC) continue;
        const int p0 = v * VEC - mis;
        if constexpr (NREF == 0) {      // uncovered canvas: zeros (float canvases; da.zeros, stitcher.py:362)
            const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int z = 0; z < ZB; ++z)
                if (FULL || z < gn) {
                    stg_nt(cplane[z] + doff + p0, zero);
                    stg_nt(cplane[z] + doff + p0 + 4, zero);
                }
            continue;
        }
        const int y0 = ya + r, x0 = xa + p0, y1 = yb + r, x1 = xb + p0;
        const int64_t s0 = (int64_t)y0 * P.tile_pitch + x0, s1 = (int64_t)y1 * P.tile_pitch + x1;
        f32x2 g0[4], g1[4];
        if constexpr (FLAT != 0) {
            load_gain8(y0, x0, g0);
            if constexpr (NREF == 2) load_gain8(y1, x1, g1);
        }
        u32x4 ra[ZB], rb[ZB];
#pragma unroll
        for (int z = 0; z < ZB; ++z)
            if (FULL || z < gn) {
                ra[z] = ldg<U32x4U>(t0[z] + s0);
                if constexpr (NREF == 2) rb[z] = ldg<U32x4U>(t1[z] + s1);
            }
        f32x2 r0[4], r1[4], w0[4], w1[4], ws[4], rw[4];
        const int wy0 = min(y0 + 1, P.tile_h - y0), wy1 = min(y1 + 1, P.tile_h - y1);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if constexpr (FLAT != 0) {
                r0[h] = refined(g0[h]);
                if constexpr (NREF == 2) r1[h] = refined(g1[h]);
            }
            if constexpr (NREF == 2) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int e = 2 * h + c;
                    w0[h][c] = (float)min(min(x0 + e + 1, P.tile_w - (x0 + e)), wy0);
                    w1[h][c] = (float)min(min(x1 + e + 1, P.tile_w - (x1 + e)), wy1);
                }
                ws[h] = w0[h] + w1[h];
                rw[h] = refined(ws[h]);
            }
        }
#pragma unroll
        for (int z = 0; z < ZB; ++z)
            if (FULL || z < gn) {
                f32x2 o[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    f32x2 va = {(float)(ra[z][h] & 0xFFFFu), (float)(ra[z][h] >> 16)}, vb = {0.0f, 0.0f};
                    if constexpr (NREF == 2) vb = f32x2{(float)(rb[z][h] & 0xFFFFu), (float)(rb[z][h] >> 16)};
                    if constexpr (FLAT != 0) {
                        if constexpr (F32OUT) {
                            va = va * r0[h];
                            if constexpr (NREF == 2) vb = vb * r1[h];
                        } else {
                            va = div_by_refined2(va, g0[h], r0[h]);
                            vb = div_by_refined2(vb, g1[h], r1[h]);
                        }
                    }
                    if constexpr (NREF == 2) {
                        const f32x2 acc = w0[h] * va + w1[h] * vb;      // multiply and add separate (-ffp-contract=off), like numpy
                        o[h] = F32OUT ? acc * rw[h] : div_by_refined2(acc, ws[h], rw[h]);
                    } else {
                        o[h] = va;
                    }
                }
                if constexpr (F32OUT) {
                    u32x4 lo, hi;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        lo[2 * q] = __float_as_uint(o[q][0]), lo[2 * q + 1] = __float_as_uint(o[q][1]);
                        hi[2 * q] = __float_as_uint(o[2 + q][0]), hi[2 * q + 1] = __float_as_uint(o[2 + q][1]);
                    }
                    stg_nt(cplane[z] + doff + p0, lo);
                    stg_nt(cplane[z] + doff + p0 + 4, hi);
                } else {
                    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                    u32x4 out;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // round half to even, negative -> 0, pack saturates to 65535
                        const u16x2 pk = __builtin_amdgcn_cvt_pk_u16(cvt_u32_sat(__builtin_rintf(o[q][0])), cvt_u32_sat(__builtin_rintf(o[q][1])));
                        out[q] = (uint32_t)pk[0] | ((uint32_t)pk[1] << 16);
                    }
                    stg_nt(cplane[z] + doff + p0, out);
                }
            }
    }
    // the pixels before / after the 16-byte-aligned body of each row, one per thread and plane
    for (int idx = tid; idx < rows * 2 * VEC; idx += 256) {
        const int r = idx / (2 * VEC), l = idx - r * 2 * VEC;
        const int64_t doff = (int64_t)(it.dst_y + r) * P.canvas_pitch + it.dst_x;
        const int mis = (int)((reinterpret_cast<uintptr_t>(cplane[0] + doff) / sizeof(OutT)) & (VEC - 1));
        const int v_first = mis ? 1 : 0, v_end = (n + mis) / VEC;
        const int head_end = min(n, v_first * VEC - mis);
        const int tail_start = max(head_end, v_end * VEC - mis);
        int p = -1;
        if (l < VEC) {
            if (l < head_end) p = l;
        } else if (tail_start + (l - VEC) < n) {
            p = tail_start + (l - VEC);
        }
        if (p < 0) continue;
        if constexpr (NREF == 0) {
#pragma unroll
            for (int z = 0; z < ZB; ++z)
                if (FULL || z < gn) stg_s<OutT>(cplane[z] + doff + p, (OutT)0);
            continue;
        }
        const int y0 = ya + r, x0 = xa + p, y1 = yb + r, x1 = xb + p;
        const float ga = FLAT ? (float)ldg_s<GM>(flat + (int64_t)y0 * P.tile_w + x0) : 1.0f;
        const float gb = (FLAT && NREF == 2) ? (float)ldg_s<GM>(flat + (int64_t)y1 * P.tile_w + x1) : 1.0f;
        const float fa = FLAT ? recip_for<1>(ga) : 1.0f, fb = (FLAT && NREF == 2) ? recip_for<1>(gb) : 1.0f;
        const float wa = (float)min(min(x0 + 1, P.tile_w - x0), min(y0 + 1, P.tile_h - y0));
        const float wb = (float)min(min(x1 + 1, P.tile_w - x1), min(y1 + 1, P.tile_h - y1));
        const float wsum = __fadd_rn(wa, wb), rws = recip_for<1>(wsum);
#pragma unroll
        for (int z = 0; z < ZB; ++z)
            if (FULL || z < gn) {
                float va = (float)ldg_s<T>(t0[z] + (int64_t)y0 * P.tile_pitch + x0), vb = 0.0f;
                if constexpr (NREF == 2) vb = (float)ldg_s<T>(t1[z] + (int64_t)y1 * P.tile_pitch + x1);
                if (FLAT) {
                    va = F32OUT ? __fmul_rn(va, fa) : div_by_refined(va, ga, fa);
                    if constexpr (NREF == 2) vb = F32OUT ? __fmul_rn(vb, fb) : div_by_refined(vb, gb, fb);
                }
                float o = va;
                if constexpr (NREF == 2) {
                    const float acc = __fadd_rn(__fmul_rn(wa, va), __fmul_rn(wb, vb));
                    o = F32OUT ? __fmul_rn(acc, rws) : div_by_refined(acc, wsum, rws);
                }
                if constexpr (F32OUT) stg_s<OutT>(cplane[z] + doff + p, o);
                else stg_s<OutT>(cplane[z] + doff + p, (OutT)min(cvt_u32_sat(__builtin_rintf(o)), 65535u));
            }
    }
}

// The float32 canvas in groups of FOUR voxels per thread (16 bytes out, 8 bytes of pixels in): every store instruction of a wave
// then writes 1 KiB of a canvas row without gaps.  With eight voxels per thread a store instruction writes 16 of every 32 bytes
// and the instruction after it the other 16: the L2 sends such half-written 64-byte pieces on as they are (WRITE_SIZE 99 GB for 85 GB
// of canvas, profiles/r04_feather_counters.log).  Arithmetic and tolerance as in blend_item_zg's float path.  FLAT: 0 or 1.
template <int FLAT, bool FULL, int NREF>
__device__ __forceinline__ void blend_item_zg4(const FuseParams &P, const UnitAux &A, const int gn, const Item &it, const int tid) {
    static_assert(FLAT == 0 || FLAT == 1, "float32 gains or none");
    typedef uint16_t T;
    constexpr int VEC = 4;
    const int rows = it.hw >> 16, n = it.hw & 0xFFFF;
    float *cplane[ZB];
    const T *t0[ZB], *t1[ZB];
#pragma unroll
    for (int z = 0; z < ZB; ++z) {
        const int zz = (FULL || z < gn) ? z : 0;
        cplane[z] = static_cast<float *>(P.canvas) + (int64_t)sgpr(A.g.plane[zz]) * P.canvas_plane_stride;
        t0[z] = sgpr(static_cast<const T *>(A.tile[zz]));
        t1[z] = sgpr(static_cast<const T *>(A.ltile[zz]));
    }
    const float *flat = (FLAT && NREF) ? static_cast<const float *>(P.flat_ptrs[sgpr(A.g.plane[0])]) : nullptr;
    const int ya = sgpr(A.first.b), xa = sgpr(A.first.c), yb = sgpr(A.seam.b), xb = sgpr(A.seam.c);
    const int G = n / VEC + 1;   // upper bound of the whole groups of a row
    auto refined2 = [](f32x2 d) {
        const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        return pk_fma(pk_fma(-d, r, f32x2{1.0f, 1.0f}), r, r);
    };
    for (int idx = tid; idx < rows * G; idx += 256) {
        const int r = idx / G, j = idx - r * G;
        const int64_t doff = (int64_t)(it.dst_y + r) * P.canvas_pitch + it.dst_x;
        const int mis = (int)((reinterpret_cast<uintptr_t>(cplane[0] + doff) / sizeof(float)) & (VEC - 1));
        const int v = (mis ? 1 : 0) + j;
        if (v >= (n + mis) / VEC) continue;
        const int p0 = v * VEC - mis;
        if constexpr (NREF == 0) {
#pragma unroll
            for (int z = 0; z < ZB; ++z)
                if (FULL || z < gn) stg_nt(cplane[z] + doff + p0, u32x4{0u, 0u, 0u, 0u});
            continue;
        }
        const int y0 = ya + r, x0 = xa + p0, y1 = yb + r, x1 = xb + p0;
        const int64_t s0 = (int64_t)y0 * P.tile_pitch + x0, s1 = (int64_t)y1 * P.tile_pitch + x1;
        f32x2 r0[2], r1[2], w0[2], w1[2], rw[2];
        if constexpr (FLAT != 0) {
            const f32x4 a = ldg<F32x4U>(flat + (int64_t)y0 * P.tile_w + x0);
            r0[0] = refined2(f32x2{a[0], a[1]}), r0[1] = refined2(f32x2{a[2], a[3]});
            if constexpr (NREF == 2) {
                const f32x4 b = ldg<F32x4U>(flat + (int64_t)y1 * P.tile_w + x1);
                r1[0] = refined2(f32x2{b[0], b[1]}), r1[1] = refined2(f32x2{b[2], b[3]});
            }
        }
        u32x2 ra[ZB], rb[ZB];
#pragma unroll
        for (int z = 0; z < ZB; ++z)
            if (FULL || z < gn) {
                ra[z] = ldg<U32x2U>(t0[z] + s0);
                if constexpr (NREF == 2) rb[z] = ldg<U32x2U>(t1[z] + s1);
            }
        if constexpr (NREF == 2) {
            const int wy0 = min(y0 + 1, P.tile_h - y0), wy1 = min(y1 + 1, P.tile_h - y1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int e = 2 * h + c;
                    w0[h][c] = (float)min(min(x0 + e + 1, P.tile_w - (x0 + e)), wy0);
                    w1[h][c] = (float)min(min(x1 + e + 1, P.tile_w - (x1 + e)), wy1);
                }
                rw[h] = refined2(w0[h] + w1[h]);
            }
        }
#pragma unroll
        for (int z = 0; z < ZB; ++z)
            if (FULL || z < gn) {
                u32x4 out;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2 va = {(float)(ra[z][h] & 0xFFFFu), (float)(ra[z][h] >> 16)};
                    if constexpr (FLAT != 0) va = va * r0[h];
                    f32x2 o = va;
                    if constexpr (NREF == 2) {
                        f32x2 vb = {(float)(rb[z][h] & 0xFFFFu), (float)(rb[z][h] >> 16)};
                        if constexpr (FLAT != 0) vb = vb * r1[h];
                        o = (w0[h] * va + w1[h] * vb) * rw[h];      // multiply and add separate (-ffp-contract=off)
                    }
                    out[2 * h] = __float_as_uint(o[0]), out[2 * h + 1] = __float_as_uint(o[1]);
                }
                stg_nt(cplane[z] + doff + p0, out);
            }
    }
    // the voxels before / after the 16-byte-aligned body of each row, one per thread and plane
    for (int idx = tid; idx < rows * 2 * VEC; idx += 256) {
        const int r = idx / (2 * VEC), l = idx - r * 2 * VEC;
        const int64_t doff = (int64_t)(it.dst_y + r) * P.canvas_pitch + it.dst_x;
        const int mis = (int)((reinterpret_cast<uintptr_t>(cplane[0] + doff) / sizeof(float)) & (VEC - 1));
        const int v_first = mis ? 1 : 0, v_end = (n + mis) / VEC;
        const int head_end = min(n, v_first * VEC - mis);
        const int tail_start = max(head_end, v_end * VEC - mis);
        int p = -1;
        if (l < VEC) {
            if (l < head_end) p = l;
        } else if (tail_start + (l - VEC) < n) {
            p = tail_start + (l - VEC);
        }
        if (p < 0) continue;
        if constexpr (NREF == 0) {
#pragma unroll
            for (int z = 0; z < ZB; ++z)
                if (FULL || z < gn) stg_s<float>(cplane[z] + doff + p, 0.0f);
            continue;
        }
        const int y0 = ya + r, x0 = xa + p, y1 = yb + r, x1 = xb + p;
        const float fa = FLAT ? recip_for<1>(ldg_s<float>(flat + (int64_t)y0 * P.tile_w + x0)) : 1.0f;
        const float fb = (FLAT && NREF == 2) ? recip_for<1>(ldg_s<float>(flat + (int64_t)y1 * P.tile_w + x1)) : 1.0f;
        const float wa = (float)min(min(x0 + 1, P.tile_w - x0), min(y0 + 1, P.tile_h - y0));
        const float wb = (float)min(min(x1 + 1, P.tile_w - x1), min(y1 + 1, P.tile_h - y1));
        const float rws = recip_for<1>(__fadd_rn(wa, wb));
#pragma unroll
        for (int z = 0; z < ZB; ++z)
            if (FULL || z < gn) {
                float va = (float)ldg_s<T>(t0[z] + (int64_t)y0 * P.tile_pitch + x0);
                if (FLAT) va = __fmul_rn(va, fa);
                float o = va;
                if constexpr (NREF == 2) {
                    float vb = (float)ldg_s<T>(t1[z] + (int64_t)y1 * P.tile_pitch + x1);
                    if (FLAT) vb = __fmul_rn(vb, fb);
                    o = __fmul_rn(__fadd_rn(__fmul_rn(wa, va), __fmul_rn(wb, vb)), rws);
                }
                stg_s<float>(cplane[z] + doff + p, o);
            }
    }
}

// Spans that THREE or FOUR tiles cover (the corners of a grid: 1.3 % of config 3's canvas, and 12 % of the launch while they went
// through the per-plane blend plane after plane -- every plane a chain of descriptor, pointer and pixel loads of its own:
// profiles/r04_exp_feather_by_cover.log).  The grouped form: a thread takes one 8-voxel group through all references and planes --
// weights and gain reciprocals once per reference, no store before the last load -- with blend_item_zg's arithmetic per reference and
// blend_group's order of summation (acc = w_0 v_0, then + w_k v_k; multiply and add separate), so an integer canvas is the
// per-plane blend's bit for bit and a float canvas is held to 1e-5 relative like the two-tile strips.  FLAT: 0 or 1.
template <int FLAT, bool FULL, typename OutT>
__device__ __forceinline__ void blend_item_zgn(const FuseParams &P, const FeatherAux &A, const int gn, const Item &it, const int tid) {
    static_assert(FLAT == 0 || FLAT == 1, "float32 gains or none");
    typedef uint16_t T;
    constexpr int VEC = 8, MAXR = 4;
    constexpr bool F32OUT = sizeof(OutT) == 4;
    const int rows = it.hw >> 16, n = it.hw & 0xFFFF;
    const bool four = sgpr((int)it.nref) == 4;      // else three
    OutT *cplane[ZB];
    const T *tp[MAXR][ZB];
#pragma unroll
    for (int z = 0; z < ZB; ++z) {
        const int zz = (FULL || z < gn) ? z : 0;
        cplane[z] = static_cast<OutT *>(P.canvas) + (int64_t)sgpr(A.g.plane[zz]) * P.canvas_plane_stride;
        tp[0][z] = sgpr(static_cast<const T *>(A.tile[zz]));
        tp[1][z] = sgpr(static_cast<const T *>(A.ltile[zz]));
        tp[2][z] = sgpr(static_cast<const T *>(A.xtile[0][zz]));
        tp[3][z] = sgpr(static_cast<const T *>(A.xtile[1][zz]));
    }
    const float *flat = FLAT ? static_cast<const float *>(P.flat_ptrs[sgpr(A.g.plane[0])]) : nullptr;
    const int ry[MAXR] = {sgpr(A.first.b), sgpr(A.seam.b), sgpr(A.xref[0].b), sgpr(A.xref[1].b)};
    const int rx[MAXR] = {sgpr(A.first.c), sgpr(A.seam.c), sgpr(A.xref[0].c), sgpr(A.xref[1].c)};
    const int G = n / VEC + 1;   // upper bound of the whole groups of a row
    auto refined = [](f32x2 d) {
        const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        return pk_fma(pk_fma(-d, r, f32x2{1.0f, 1.0f}), r, r);
    };
    for (int idx = tid; idx < rows * G; idx += 256) {
        const int r = idx / G, j = idx - r * G;
        const int64_t doff = (int64_t)(it.dst_y + r) * P.canvas_pitch + it.dst_x;
        const int mis = (int)((reinterpret_cast<uintptr_t>(cplane[0] + doff) / sizeof(OutT)) & (VEC - 1));
        const int v = (mis ? 1 : 0) + j;
        if (v >= (n + mis) / VEC) continue;
        const int p0 = v * VEC - mis;
        f32x2 acc[ZB][4], ws[4];
#pragma unroll
        for (int k = 0; k < MAXR; ++k)
            if (k < 3 || four) {
                const int y = ry[k] + r, x = rx[k] + p0;
                const int64_t so = (int64_t)y * P.tile_pitch + x;
                f32x2 g[4], rr[4], w[4];
                if constexpr (FLAT != 0) {
                    const float *gp = flat + (int64_t)y * P.tile_w + x;
                    const f32x4 a = ldg<F32x4U>(gp), b = ldg<F32x4U>(gp + 4);
                    g[0] = f32x2{a[0], a[1]}, g[1] = f32x2{a[2], a[3]}, g[2] = f32x2{b[0], b[1]}, g[3] = f32x2{b[2], b[3]};
                }
                u32x4 raw[ZB];
#pragma unroll
                for (int z = 0; z < ZB; ++z)
                    if (FULL || z < gn) raw[z] = ldg<U32x4U>(tp[k][z] + so);
                const int wy = min(y + 1, P.tile_h - y);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    if constexpr (FLAT != 0) rr[h] = refined(g[h]);
#pragma unroll
                    for (int c = 0; c < 2; ++c) w[h][c] = (float)min(min(x + 2 * h + c + 1, P.tile_w - (x + 2 * h + c)), wy);
                    ws[h] = k == 0 ? w[h] : ws[h] + w[h];
                }
#pragma unroll
                for (int z = 0; z < ZB; ++z)
                    if (FULL || z < gn) {
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            f32x2 px = {(float)(raw[z][h] & 0xFFFFu), (float)(raw[z][h] >> 16)};
                            if constexpr (FLAT != 0) px = F32OUT ? px * rr[h] : div_by_refined2(px, g[h], rr[h]);
                            const f32x2 t = w[h] * px;      // multiply and add separate (-ffp-contract=off), like numpy
                            acc[z][h] = k == 0 ? t : acc[z][h] + t;
                        }
                    }
            }
        f32x2 rw[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) rw[h] = refined(ws[h]);
#pragma unroll
        for (int z = 0; z < ZB; ++z)
            if (FULL || z < gn) {
                f32x2 o[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) o[h] = F32OUT ? acc[z][h] * rw[h] : div_by_refined2(acc[z][h], ws[h], rw[h]);
                if constexpr (F32OUT) {
                    u32x4 lo, hi;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        lo[2 * q] = __float_as_uint(o[q][0]), lo[2 * q + 1] = __float_as_uint(o[q][1]);
                        hi[2 * q] = __float_as_uint(o[2 + q][0]), hi[2 * q + 1] = __float_as_uint(o[2 + q][1]);
                    }
                    stg_nt(cplane[z] + doff + p0, lo);
                    stg_nt(cplane[z] + doff + p0 + 4, hi);
                } else {
                    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                    u32x4 out;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u16x2 pk = __builtin_amdgcn_cvt_pk_u16(cvt_u32_sat(__builtin_rintf(o[q][0])), cvt_u32_sat(__builtin_rintf(o[q][1])));
                        out[q] = (uint32_t)pk[0] | ((uint32_t)pk[1] << 16);
                    }
                    stg_nt(cplane[z] + doff + p0, out);
                }
            }
    }
    // the voxels before / after the 16-byte-aligned body of each row, one per thread and plane
    for (int idx = tid; idx < rows * 2 * VEC; idx += 256) {
        const int r = idx / (2 * VEC), l = idx - r * 2 * VEC;
        const int64_t doff = (int64_t)(it.dst_y + r) * P.canvas_pitch + it.dst_x;
        const int mis = (int)((reinterpret_cast<uintptr_t>(cplane[0] + doff) / sizeof(OutT)) & (VEC - 1));
        const int v_first = mis ? 1 : 0, v_end = (n + mis) / VEC;
        const int head_end = min(n, v_first * VEC - mis);
        const int tail_start = max(head_end, v_end * VEC - mis);
        int p = -1;
        if (l < VEC) {
            if (l < head_end) p = l;
        } else if (tail_start + (l - VEC) < n) {
            p = tail_start + (l - VEC);
        }
        if (p < 0) continue;
        float acc[ZB], wsum = 0.0f;
#pragma unroll
        for (int k = 0; k < MAXR; ++k)
            if (k < 3 || four) {
                const int y = ry[k] + r, x = rx[k] + p;
                const float gk = FLAT ? ldg_s<float>(flat + (int64_t)y * P.tile_w + x) : 1.0f;
                const float fk = FLAT ? recip_for<1>(gk) : 1.0f;
                const float w = (float)min(min(x + 1, P.tile_w - x), min(y + 1, P.tile_h - y));
                wsum = k == 0 ? w : __fadd_rn(wsum, w);
#pragma unroll
                for (int z = 0; z < ZB; ++z)
                    if (FULL || z < gn) {
                        float px = (float)ldg_s<T>(tp[k][z] + (int64_t)y * P.tile_pitch + x);
                        if (FLAT) px = F32OUT ? __fmul_rn(px, fk) : div_by_refined(px, gk, fk);
                        const float t = __fmul_rn(w, px);
                        acc[z] = k == 0 ? t : __fadd_rn(acc[z], t);
                    }
            }
        const float rws = recip_for<1>(wsum);
#pragma unroll
        for (int z = 0; z < ZB; ++z)
            if (FULL || z < gn) {
                const float o = F32OUT ? __fmul_rn(acc[z], rws) : div_by_refined(acc[z], wsum, rws);
                if constexpr (F32OUT) stg_s<OutT>(cplane[z] + doff + p, o);
                else stg_s<OutT>(cplane[z] + doff + p, (OutT)min(cvt_u32_sat(__builtin_rintf(o)), 65535u));
            }
    }
}

// waves per SIMD asked of the register allocator: with gains 3 (168 VGPRs instead of 171: 0.555 against 0.532 for 2 waves,
// 4 waves / 128 VGPRs 0.551); without gains the allocator's own choice measured best (0.588 against 0.574 / 0.581 at 3 / 4)
#ifndef SQ_WAVES_FEATHER_ZG
#define SQ_WAVES_FEATHER_ZG 3
#endif
template <int FLAT, bool DYN, typename OutT = uint16_t>
__global__ __launch_bounds__(256, FLAT ? SQ_WAVES_FEATHER_ZG : 1) void fuse_feather_zg_kernel(const FuseParams P, const int64_t n_items) {
    typedef uint16_t T;
    constexpr bool F32OUT = sizeof(OutT) == 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint32_t n_groups = *P.n_groups;
    auto pre = [&](int unit, const Item &it, int64_t) -> FeatherAux {
        FeatherAux A;
        A.g = P.groups[unit];
        A.seam = Seam{-1, 0, 0, 0};
        A.first = Seam{-1, 0, 0, 0};
        A.xref[0] = A.xref[1] = Seam{-1, 0, 0, 0};
        const bool one = it.nref >= 1 && it.nref <= 4, two = it.nref >= 2 && it.nref <= 4;      // (more than four: the per-plane blend)
        Ref ra{}, rb{};
        if (one) {
            ra = P.refs[it.a];
            A.first = Seam{ra.tile, ra.src_y + it.b, ra.src_x + it.c, 0};
        }
        if (two) {
            rb = P.refs[it.a + 1];
            A.seam = Seam{rb.tile, rb.src_y + it.b, rb.src_x + it.c, 0};
        }
#pragma unroll
        for (int z = 0; z < ZB; ++z) {
            A.tile[z] = (one && z < A.g.n) ? tile_ptr<T>(P, A.g.plane[z], ra.tile) : nullptr;
            A.ltile[z] = (two && z < A.g.n) ? tile_ptr<T>(P, A.g.plane[z], rb.tile) : nullptr;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool have = it.nref >= 3 + k && it.nref <= 4;
            Ref rk{};
            if (have) {
                rk = P.refs[it.a + 2 + k];
                A.xref[k] = Seam{rk.tile, rk.src_y + it.b, rk.src_x + it.c, 0};
            }
#pragma unroll
            for (int z = 0; z < ZB; ++z) A.xtile[k][z] = (have && z < A.g.n) ? tile_ptr<T>(P, A.g.plane[z], rk.tile) : nullptr;
        }
        return A;
    };
    auto body = [&](int, const Item &it, const FeatherAux &A) {
        const int gn = sgpr(A.g.n);
#ifdef SQ_FEATHER_ONLY      // experiment builds (tools/build_variant.sh): only the items that 0 / 1 / 2 / 3-or-4 tiles cover
        if ((it.nref > 3 ? 3 : it.nref) != SQ_FEATHER_ONLY) return;
#endif
        if (gn == 1) {
            feather_one_item<T, OutT, FLAT>(P, sgpr(A.g.plane[0]), it, wave, lane);
        } else if (F32OUT && it.nref <= 1) {
            // float32 canvas: one-tile and empty items through the grouped form too (gains and their reciprocals once per group,
            // a multiply per voxel and plane -- a float canvas is held to 1e-5 relative, see blend_item_zg)
            // (a row-wise form of this -- a wave per canvas row, scalar row bases, line-aligned slots like process_item_zg --
            //  measured SLOWER for a float canvas: 33.3 against 29.6 ms for 20 config-3 planes, profiles/r04_exp_feather.log)
            if constexpr (F32OUT && FLAT != 2) {
                if (it.nref == 1) {
                    if (gn == ZB) blend_item_zg4<FLAT, true, 1>(P, A, gn, it, threadIdx.x);
                    else blend_item_zg4<FLAT, false, 1>(P, A, gn, it, threadIdx.x);
                } else {
                    if (gn == ZB) blend_item_zg4<FLAT, true, 0>(P, A, gn, it, threadIdx.x);
                    else blend_item_zg4<FLAT, false, 0>(P, A, gn, it, threadIdx.x);
                }
            } else if constexpr (F32OUT) {      // float64 gains: cast once per 8-voxel group
                if (it.nref == 1) {
                    if (gn == ZB) blend_item_zg<FLAT, true, OutT, 1>(P, A, gn, it, threadIdx.x);
                    else blend_item_zg<FLAT, false, OutT, 1>(P, A, gn, it, threadIdx.x);
                } else {
                    if (gn == ZB) blend_item_zg<FLAT, true, OutT, 0>(P, A, gn, it, threadIdx.x);
                    else blend_item_zg<FLAT, false, OutT, 0>(P, A, gn, it, threadIdx.x);
                }
            }
        } else if (FLAT == 2 && it.nref <= 1) {
            // float64 gains: the grouped one-tile path does not take gains as doubles -- the per-plane path, plane after plane
            for (int z = 0; z < gn; ++z) feather_one_item<T, OutT, FLAT>(P, sgpr(A.g.plane[z]), it, wave, lane);
        } else if (it.nref <= 1) {
            Item one = it;
            one.a = sgpr(A.first.a);
            one.b = sgpr(A.first.b);
            one.c = sgpr(A.first.c);
            if constexpr (FLAT == 2) {
                // (taken by the branch above)
            } else if (!FLAT) {   // nothing to share: the plain pipelined copy, plane after plane
                for (int z = 0; z < gn; ++z)
                    process_item<T, 0, 1>(P, sgpr(A.g.plane[z]), one, sgpr(static_cast<const T *>(A.tile[z])), wave, lane);
            } else if (gn == ZB) {
                process_item_zg<true, 1>(P, A, gn, one, wave, lane);
            } else {
                process_item_zg<false, 1>(P, A, gn, one, wave, lane);
            }
        } else if (it.nref == 2) {
            if constexpr (F32OUT && FLAT != 2) {
                if (gn == ZB) blend_item_zg4<FLAT, true, 2>(P, A, gn, it, threadIdx.x);
                else blend_item_zg4<FLAT, false, 2>(P, A, gn, it, threadIdx.x);
            } else {
                if (gn == ZB) blend_item_zg<FLAT, true, OutT>(P, A, gn, it, threadIdx.x);
                else blend_item_zg<FLAT, false, OutT>(P, A, gn, it, threadIdx.x);
            }
        } else if (FLAT != 2 && it.nref <= 4) {      // three or four tiles: a corner of the grid
            if constexpr (FLAT != 2) {
                if (gn == ZB) blend_item_zgn<FLAT, true, OutT>(P, A, gn, it, threadIdx.x);
                else blend_item_zgn<FLAT, false, OutT>(P, A, gn, it, threadIdx.x);
            }
        } else {
            for (int z = 0; z < gn; ++z) blend_item<T, OutT, FLAT, true>(P, sgpr(A.g.plane[z]), it, threadIdx.x);
        }
    };
    if (DYN) {
        for_each_queued_item<FeatherAux>(P, n_items, n_groups, pre, body);
    } else {
        __shared__ Item s_it;
        __shared__ FeatherAux s_A;
        const int64_t n_work = (int64_t)n_groups * n_items;
        for (int64_t work = blockIdx.x; work < n_work; work += gridDim.x) {
            __syncthreads();
            if (threadIdx.x == 0) {
                const int unit = (int)(work / n_items);
                s_it = P.items[work - unit * n_items];
                s_A = pre(unit, s_it, work - unit * n_items);
            }
            __syncthreads();
            body(0, sgpr(s_it), s_A);
        }
    }
}

// Persistent launch: as many workgroups as the chip keeps resident (queried once per kernel),
// each walking the (plane, item) list with a grid stride.
template <typename K>
int launch(K kernel, const FuseParams &P, int64_t n_items, int n_planes, hipStream_t stream, int grid_override = 0) {
    const int64_t n_work = n_items * n_planes;
    if (n_work == 0) return SQ_OK;
    static thread_local std::map<const void *, int> resident;
    const void *key = reinterpret_cast<const void *>(kernel);
    auto it = resident.find(key);
    if (it == resident.end()) {
        int dev = 0, cus = 256, per_cu = 8;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        it = resident.emplace(key, cus * std::min(per_cu, 8)).first;
    }
    const int64_t blocks = std::min<int64_t>(n_work, grid_override > 0 ? grid_override : it->second);
    // work-queue chunk: QUEUE_CHUNK items per atomic when every workgroup gets many chunks, fewer for small
    // launches so that the last round does not leave workgroups idle
    FuseParams Q = P;
    Q.chunk = (int32_t)std::max<int64_t>(1, std::min<int64_t>(QUEUE_CHUNK, n_work / (blocks * 16)));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), 0, stream, Q, n_items, n_work);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_fuse_planes: launch failed: %s", hipGetErrorString(e));
    return SQ_OK;
}

// the plane-group kernel: the number of work units (groups x items) is only known on the device, so the grid is
// the resident workgroups (fewer only when even one plane per group would not fill them)
template <typename K>
int launch_zg(K kernel, const FuseParams &P, int64_t n_items, int n_planes, hipStream_t stream, int grid_override) {
    if (n_items == 0 || n_planes == 0) return SQ_OK;
    static thread_local std::map<const void *, int> resident;
    const void *key = reinterpret_cast<const void *>(kernel);
    auto it = resident.find(key);
    if (it == resident.end()) {
        int dev = 0, cus = 256, per_cu = 8;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        it = resident.emplace(key, cus * std::min(per_cu, 8)).first;
    }
    const int64_t min_units = n_items * ((n_planes + ZB - 1) / ZB);
    const int64_t blocks = std::min<int64_t>(min_units, grid_override > 0 ? grid_override : it->second);
    FuseParams Q = P;
    Q.chunk = (int32_t)std::max<int64_t>(1, std::min<int64_t>(QUEUE_CHUNK, min_units / (blocks * 16)));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), 0, stream, Q, n_items);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_fuse_planes: launch failed: %s", hipGetErrorString(e));
    return SQ_OK;
}

// pre-pass: does a plane's flatfield hold anything outside the fast divide's range?
template <typename G>
__global__ __launch_bounds__(256) void flat_classify_kernel(const void *const *flat_ptrs, int64_t n, uint32_t *cls) {
    const int plane = blockIdx.y;
    const G *f = static_cast<const G *>(flat_ptrs[plane]);
    if (!f) return;
    for (int q = 0; q < plane; ++q)          // the z planes of a channel name the same image: read it once
        if (flat_ptrs[q] == f) return;       // (flat_class_share_kernel hands the class on)
    const G lo = (G)__builtin_ldexp(1.0, FAST_MIN_EXP), hi = (G)__builtin_ldexp(1.0, FAST_END_EXP);
    const G mlo = (G)__builtin_ldexp(1.0, -MODERATE_EXP), mhi = (G)__builtin_ldexp(1.0, MODERATE_EXP);
    bool odd = false, wide = false, neg = false;
    auto look = [&](G g) {
        const G a = g < 0 ? -g : g;
        odd |= !(a >= lo && a < hi);   // NaN fails both
        wide |= !(a >= mlo && a < mhi);
        neg |= !(g > 0);               // a gain that is not positive: weighted quotients could cancel in a blend
    };
    // 16 bytes per lane and load (the image is read at 1 TB/s with 4-byte loads from 64 workgroups: 66 us of every launch)
    constexpr int PER = 16 / sizeof(G);
    const int64_t head = min<int64_t>(n, (int64_t)((16 - (reinterpret_cast<uintptr_t>(f) & 15)) & 15) / (int64_t)sizeof(G));
    const int64_t nvec = (n - head) / PER;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = tid; v < nvec; v += nthr) {
        if constexpr (sizeof(G) == 4) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(f + head + v * PER);
            look(q[0]), look(q[1]), look(q[2]), look(q[3]);
        } else {
            const F64x2U q = *reinterpret_cast<const F64x2U *>(f + head + v * PER);
            look(q.v[0]), look(q.v[1]);
        }
    }
    for (int64_t i = tid; i < head; i += nthr) look(ldg_s<G>(f + i));
    for (int64_t i = head + nvec * PER + tid; i < n; i += nthr) look(ldg_s<G>(f + i));
    const uint32_t bits = (__builtin_amdgcn_ballot_w64(odd) ? 1u : 0u) | (__builtin_amdgcn_ballot_w64(wide) ? 2u : 0u) |
                          (__builtin_amdgcn_ballot_w64(neg) ? 4u : 0u);
    if (bits && (threadIdx.x & 63) == 0) atomicOr(&cls[plane], bits);
}

// after the pre-pass: planes that share a gain image share its class
__global__ __launch_bounds__(256) void flat_class_share_kernel(const void *const *flat_ptrs, int n_planes, uint32_t *cls) {
    for (int p = threadIdx.x; p < n_planes; p += 256) {
        const void *f = flat_ptrs[p];
        if (!f) continue;
        for (int q = 0; q < p; ++q)
            if (flat_ptrs[q] == f) {
                cls[p] = cls[q];     // q is the first plane with this image: the one the pre-pass read (never written here)
                break;
            }
    }
}

// exhaustive check of the fast divide against the IEEE path: final clipped integers, one binade of
// the gain per blockIdx.y, every mantissa, every uint16 numerator
__global__ __launch_bounds__(256) void selftest_divide_kernel(int exponent0, int negative, unsigned long long *bad) {
    const uint32_t mant = blockIdx.x * 256u + threadIdx.x;   // 2^23 mantissas
    const int exponent = exponent0 + (int)blockIdx.y;
    const uint32_t bits = ((uint32_t)(exponent + 127) << 23) | mant | (negative ? 0x80000000u : 0u);
    const float g = __uint_as_float(bits);
    unsigned long long local = 0;
    for (int v = 0; v < 65536; ++v)
    {
        const uint32_t want = flat_f32<uint16_t>((uint16_t)v, g);   // the IEEE path
        const uint32_t pair = flat_f32_fast_pair((uint32_t)v | ((uint32_t)v << 16), g, g);
        local += want != flat_f32_fast<uint16_t>((uint16_t)v, g);
        local += want != (pair & 0xFFFFu);
        local += want != (pair >> 16);
        // rounded instead of truncated (feather mode, voxels one tile covers).  NB the quotient float
        // itself is NOT always the IEEE one: n/g can sit within 2^-48 of a float midpoint, closer than the
        // sequence's 2^-46 error, while integer and half-integer boundaries are >= 2^-41 away -- so only
        // truncated results may use it; rounded and blended ones take div_u16_normal_ieee.
        local += flat_f32<uint16_t, 1>((uint16_t)v, g) != flat_f32_fast<uint16_t, 1>((uint16_t)v, g);
        // and the 8-slot sequence yields the IEEE quotient itself, bit for bit (the blend of several
        // tiles uses the quotient as a float)
        local += __fdiv_rn((float)v, g) != div_u16_normal_ieee((float)v, g);
#ifdef SQ_SELFTEST_DEBUG
        if (flat_f32<uint16_t, 1>((uint16_t)v, g) != flat_f32_fast<uint16_t, 1>((uint16_t)v, g) && atomicAdd(bad + 1, 1ull) < 8)
            printf("rint mismatch v=%d g=%a (%08x): ieee q=%a -> %u, fast q=%a -> %u\n", v, g, bits, __fdiv_rn((float)v, g),
                   (unsigned)flat_f32<uint16_t, 1>((uint16_t)v, g), quotient_u16_normal<1>((float)v, g),
                   (unsigned)flat_f32_fast<uint16_t, 1>((uint16_t)v, g));
#endif
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(bad, local);
}

}  // namespace

extern "C" int sq_selftest_flat_divide(int32_t exponent, int32_t n_binades, int32_t negative, uint64_t *mismatches_dev,
                                       void *stream) {
    if (!mismatches_dev || exponent < FAST_MIN_EXP || n_binades < 1 || exponent + n_binades > FAST_END_EXP)
        return fail(SQ_ERR_INVALID, "sq_selftest_flat_divide: binades [%d, %d] outside the fast path's range [%d, %d]",
                    exponent, exponent + n_binades - 1, FAST_MIN_EXP, FAST_END_EXP - 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(mismatches_dev, 0, sizeof(uint64_t), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(selftest_divide_kernel, dim3(1u << 15, n_binades), dim3(256), 0, s, exponent, negative,
                           reinterpret_cast<unsigned long long *>(mismatches_dev));
        e = hipGetLastError();
    }
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_selftest_flat_divide: %s", hipGetErrorString(e));
    return SQ_OK;
}

// the grouped feather blend's division acc / wsum (div_by_refined with recip_for<1>(wsum)) against the compiler's IEEE
// division: every mantissa of acc in one binade per blockIdx.y, every weight sum the kernel can meet
__global__ __launch_bounds__(256) void selftest_blend_divide_kernel(int exponent0, int negative, unsigned long long *bad) {
    const uint32_t mant = blockIdx.x * 256u + threadIdx.x;   // 2^23 mantissas
    const int exponent = exponent0 + (int)blockIdx.y;
    const float acc = __uint_as_float(((uint32_t)(exponent + 127) << 23) | mant | (negative ? 0x80000000u : 0u));
    unsigned long long local = 0;
    for (int ws = 2; ws <= BLEND_WSUM_MAX; ++ws) {
        const float d = (float)ws;
        const float want = __fdiv_rn(acc, d), got = div_by_refined(acc, d, recip_for<1>(d));
        local += __float_as_uint(want) != __float_as_uint(got);
    }
    if (local) atomicAdd(bad, local);
}

extern "C" int sq_selftest_blend_divide(int32_t exponent, int32_t n_binades, int32_t negative, uint64_t *mismatches_dev,
                                        void *stream) {
    if (!mismatches_dev || exponent < BLEND_ACC_MIN_EXP || n_binades < 1 || exponent + n_binades > BLEND_ACC_END_EXP)
        return fail(SQ_ERR_INVALID, "sq_selftest_blend_divide: binades [%d, %d] outside the blend's range [%d, %d]", exponent,
                    exponent + n_binades - 1, BLEND_ACC_MIN_EXP, BLEND_ACC_END_EXP - 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(mismatches_dev, 0, sizeof(uint64_t), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(selftest_blend_divide_kernel, dim3(1u << 15, n_binades), dim3(256), 0, s, exponent, negative,
                           reinterpret_cast<unsigned long long *>(mismatches_dev));
        e = hipGetLastError();
    }
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_selftest_blend_divide: %s", hipGetErrorString(e));
    return SQ_OK;
}

// the float64 divide against the compiler's IEEE division: 2^15 pseudo-random gains per binade (all 52
// mantissa bits from a counter hash) x every uint16 numerator; the doubles and the clipped integers
__global__ __launch_bounds__(256) void selftest_divide_f64_kernel(int exponent0, int negative, uint64_t seed, unsigned long long *bad) {
    const uint64_t id = ((uint64_t)blockIdx.y << 32) | (blockIdx.x * 256u + threadIdx.x);
    uint64_t x = seed + id * 0x9E3779B97F4A7C15ull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    const int exponent = exponent0 + (int)blockIdx.y;
    // a few structured mantissas first: all zeros, all ones, single bits -- then the hash
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    uint64_t mant = x & ((1ull << 52) - 1);
    if (t == 0) mant = 0;
    else if (t == 1) mant = (1ull << 52) - 1;
    else if (t < 54) mant = 1ull << (t - 2);
    else if (t < 106) mant = ((1ull << 52) - 1) ^ (1ull << (t - 54));
    const uint64_t bits = ((uint64_t)(exponent + 1023) << 52) | mant | (negative ? 0x8000000000000000ull : 0ull);
    const double g = __longlong_as_double((long long)bits);
    unsigned long long local = 0;
    for (int v = 0; v < 65536; ++v) {
        const double want = __ddiv_rn((double)v, g);
        const double got = div_u16_normal_f64((double)v, g);
        local += __double_as_longlong(want) != __double_as_longlong(got) && !(want == 0.0 && got == 0.0);
        local += flat_f64<uint16_t>((uint16_t)v, g) != flat_f64_fast<uint16_t>((uint16_t)v, g);
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(bad, local);
}

extern "C" int sq_selftest_flat_divide_f64(int32_t exponent, int32_t n_binades, int32_t negative, uint64_t seed,
                                           uint64_t *mismatches_dev, void *stream) {
    if (!mismatches_dev || exponent < FAST_MIN_EXP || n_binades < 1 || exponent + n_binades > FAST_END_EXP)
        return fail(SQ_ERR_INVALID, "sq_selftest_flat_divide_f64: binades [%d, %d] outside the fast path's range [%d, %d]",
                    exponent, exponent + n_binades - 1, FAST_MIN_EXP, FAST_END_EXP - 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(mismatches_dev, 0, sizeof(uint64_t), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(selftest_divide_f64_kernel, dim3(128, n_binades), dim3(256), 0, s, exponent, negative, seed,
                           reinterpret_cast<unsigned long long *>(mismatches_dev));
        e = hipGetLastError();
    }
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_selftest_flat_divide_f64: %s", hipGetErrorString(e));
    return SQ_OK;
}

// scratch: one uint32 gain class per plane | the nine chunk counters of the work queues, a 128-byte line each |
// the number of plane groups (a line) | the plane groups (32 bytes per plane at most)
namespace {
struct ScratchLayout {
    int64_t queue, n_groups, groups, total;
};
ScratchLayout scratch_layout(int64_t n_planes) {
    ScratchLayout L;
    L.queue = (n_planes * 4 + 127) & ~int64_t(127);
    L.n_groups = L.queue + 9 * QUEUE_STRIDE * 4;
    L.groups = L.n_groups + 128;
    L.total = L.groups + n_planes * (int64_t)sizeof(PlaneGroup);
    return L;
}
}  // namespace

extern "C" int64_t sq_fuse_scratch_bytes(int32_t n_planes) {
    if (n_planes < 0) return fail(SQ_ERR_INVALID, "sq_fuse_scratch_bytes: n_planes %d", n_planes);
    return scratch_layout(n_planes).total;
}

extern "C" int sq_fuse_planes(const sq_fuse_args *a, void *stream_) {
    if (!a || !a->plan || !a->table_dev || !a->canvas_dev)
        return fail(SQ_ERR_INVALID, "sq_fuse_planes: NULL plan/table/canvas");
    const TableHeader &h = a->plan->header();
    if (a->plan->spans_only && !a->plan->expanded)
        return fail(SQ_ERR_INVALID, "sq_fuse_planes: the plan of sq_fuse_plan_create_spans has not been through sq_fuse_plan_expand");
    if (a->table_bytes != a->plan->device_bytes())
        return fail(SQ_ERR_INVALID, "sq_fuse_planes: table_bytes %lld != plan %lld", (long long)a->table_bytes,
                    (long long)a->plan->device_bytes());
    if (a->mode != h.mode) return fail(SQ_ERR_INVALID, "sq_fuse_planes: mode %d but plan was built for %d", a->mode, h.mode);
    if (a->n_tiles != h.n_tiles || a->tile_h != h.tile_h || a->tile_w != h.tile_w || a->canvas_h != h.canvas_h ||
        a->canvas_w != h.canvas_w)
        return fail(SQ_ERR_INVALID, "sq_fuse_planes: geometry differs from the plan (tiles %d/%d %dx%d/%dx%d canvas %dx%d/%dx%d)",
                    a->n_tiles, h.n_tiles, a->tile_h, a->tile_w, h.tile_h, h.tile_w, a->canvas_h, a->canvas_w, h.canvas_h,
                    h.canvas_w);
    if (!a->tile_ptrs_dev && !a->tile_base_dev && h.n_refs > 0)
        return fail(SQ_ERR_INVALID, "sq_fuse_planes: no tile table and no tile base");
    if (a->tile_pitch < a->tile_w || a->canvas_pitch < a->canvas_w)
        return fail(SQ_ERR_INVALID, "sq_fuse_planes: pitch smaller than width");
    if (a->n_planes < 0) return fail(SQ_ERR_INVALID, "sq_fuse_planes: n_planes %d out of range", a->n_planes);
    if ((a->flags & ~(SQ_FUSE_FORCE_QUEUES | SQ_FUSE_FORCE_STATIC | SQ_FUSE_NO_PLANE_GROUPS | SQ_FUSE_NO_SEAM_OWNERS | SQ_FUSE_CONSECUTIVE_GROUPS)) || a->grid_blocks < 0 ||
        ((a->flags & SQ_FUSE_FORCE_QUEUES) && (a->flags & SQ_FUSE_FORCE_STATIC)))
        return fail(SQ_ERR_INVALID, "sq_fuse_planes: flags %d / grid_blocks %d", a->flags, a->grid_blocks);
    if (a->tile_dtype != SQ_U8 && a->tile_dtype != SQ_U16)
        return fail(SQ_ERR_UNSUPPORTED, "sq_fuse_planes: tile dtype %d (uint8/uint16 only)", a->tile_dtype);
    if (a->flat_ptrs_dev && a->flat_dtype != SQ_F32 && a->flat_dtype != SQ_F64)
        return fail(SQ_ERR_UNSUPPORTED, "sq_fuse_planes: flatfield dtype %d (float32/float64 only)", a->flat_dtype);
    if (a->canvas_plane_stride < (int64_t)a->canvas_h * a->canvas_pitch && a->n_planes > 1)
        return fail(SQ_ERR_INVALID, "sq_fuse_planes: canvas planes overlap");
    const size_t esz = a->canvas_dtype == SQ_F32 ? 4 : (size_t)a->canvas_dtype;
    if (reinterpret_cast<uintptr_t>(a->canvas_dev) % esz)
        return fail(SQ_ERR_INVALID, "sq_fuse_planes: canvas pointer not aligned to its element size");

    FuseParams P{};
    const char *base = static_cast<const char *>(a->table_dev);
    P.spans = reinterpret_cast<const Span *>(base + h.off_spans);
    P.refs = reinterpret_cast<const Ref *>(base + h.off_refs);
    P.items = reinterpret_cast<const Item *>(base + h.off_items);
    P.seams = (h.off_seams && !(a->flags & SQ_FUSE_NO_SEAM_OWNERS)) ? reinterpret_cast<const Seam *>(base + h.off_seams) : nullptr;
    P.tile_ptrs = a->tile_ptrs_dev;
    P.tile_base = a->tile_base_dev;
    P.tile_plane_stride = a->tile_plane_stride;
    P.tile_stride = a->tile_stride;
    P.flat_ptrs = a->flat_ptrs_dev;
    P.canvas = a->canvas_dev;
    P.canvas_plane_stride = a->canvas_plane_stride;
    P.n_tiles = a->n_tiles;
    P.tile_h = a->tile_h;
    P.tile_w = a->tile_w;
    P.tile_pitch = a->tile_pitch;
    P.canvas_pitch = a->canvas_pitch;
    P.flat_class = nullptr;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int flat = a->flat_ptrs_dev ? (a->flat_dtype == SQ_F64 ? 2 : 1) : 0;
    P.queue = nullptr;
    P.lane_items = (int32_t)h.lane_items;
    P.n_planes = a->n_planes;
    const ScratchLayout SL = scratch_layout(a->n_planes);
    if (a->scratch_dev && a->n_planes > 0) {
        if (a->scratch_bytes < sq_fuse_scratch_bytes(a->n_planes))
            return fail(SQ_ERR_WORKSPACE, "sq_fuse_planes: scratch %lld < %lld bytes", (long long)a->scratch_bytes,
                        (long long)sq_fuse_scratch_bytes(a->n_planes));
        if (reinterpret_cast<uintptr_t>(a->scratch_dev) % 128) return fail(SQ_ERR_INVALID, "sq_fuse_planes: scratch not 128-byte aligned");
        if (hipMemsetAsync(a->scratch_dev, 0, (size_t)SL.groups, stream) != hipSuccess)   // classes, counters, group count
            return fail(SQ_ERR_HIP, "sq_fuse_planes: cannot clear the scratch");
        if (flat) {
            // classify every plane's gains once per call (reads H*W*4 (8) B per plane, ~0.4 % of the launch)
            if (flat == 1)
                hipLaunchKernelGGL(flat_classify_kernel<float>, dim3(256, a->n_planes), dim3(256), 0, stream, a->flat_ptrs_dev,
                                   (int64_t)a->tile_h * a->tile_w, static_cast<uint32_t *>(a->scratch_dev));
            else
                hipLaunchKernelGGL(flat_classify_kernel<double>, dim3(256, a->n_planes), dim3(256), 0, stream, a->flat_ptrs_dev,
                                   (int64_t)a->tile_h * a->tile_w, static_cast<uint32_t *>(a->scratch_dev));
            if (a->n_planes > 1)
                hipLaunchKernelGGL(flat_class_share_kernel, dim3(1), dim3(256), 0, stream, a->flat_ptrs_dev, a->n_planes,
                                   static_cast<uint32_t *>(a->scratch_dev));
            P.flat_class = static_cast<const uint32_t *>(a->scratch_dev);
        }
        // the queues count in 32 bits; a launch with fewer than ~64 items per resident workgroup is over
        // before the queues pay for their barriers (measured on the 8x8-grid, one-plane case): static walk
        const int64_t n_work = (int64_t)a->n_planes * h.n_items;
        // (args->flags can force either one whatever the size -- tests)
        if ((n_work >= 100000 || (a->flags & SQ_FUSE_FORCE_QUEUES)) && n_work < (int64_t(1) << 31) && !(a->flags & SQ_FUSE_FORCE_STATIC))
            P.queue = reinterpret_cast<uint32_t *>(static_cast<char *>(a->scratch_dev) + SL.queue);
    }
    const bool u16 = a->tile_dtype == SQ_U16;

    if (a->mode == SQ_FUSE_OVERWRITE) {
        if (a->canvas_dtype != a->tile_dtype)
            return fail(SQ_ERR_INVALID, "sq_fuse_planes: overwrite mode keeps the tile dtype (canvas %d, tile %d)",
                        a->canvas_dtype, a->tile_dtype);
#define SQ_OVERWRITE(T, F)                                                                                  \
    do {                                                                                                      \
        if (P.queue) return launch(fuse_overwrite_kernel<T, F, true>, P, h.n_items, a->n_planes, stream, a->grid_blocks);     \
        return launch(fuse_overwrite_kernel<T, F, false>, P, h.n_items, a->n_planes, stream, a->grid_blocks);                 \
    } while (0)
        const int64_t esz = u16 ? 2 : 1;
        if (flat && a->scratch_dev && a->n_planes > 1 && ZB > 1 && !(a->flags & SQ_FUSE_NO_PLANE_GROUPS)) {
            // planes that share a gain image go through the items together (fuse_overwrite_zg_kernel)
            char *sc = static_cast<char *>(a->scratch_dev);
            uint32_t *n_groups = reinterpret_cast<uint32_t *>(sc + SL.n_groups);
            PlaneGroup *groups = reinterpret_cast<PlaneGroup *>(sc + SL.groups);
            hipLaunchKernelGGL(build_groups_kernel, dim3(1), dim3(256), 0, stream, a->flat_ptrs_dev, P.flat_class, 1u, a->n_planes,
                               a->canvas_plane_stride * esz, ZB, n_groups, groups,
                               (a->flags & SQ_FUSE_CONSECUTIVE_GROUPS) != 0);
            P.groups = groups;
            P.n_groups = n_groups;
#define SQ_ZG(G, T)                                                                                                                  \
    do {                                                                                                                              \
        if (P.queue) return launch_zg(fuse_overwrite_zg_kernel<G, true, T>, P, h.n_items, a->n_planes, stream, a->grid_blocks);       \
        return launch_zg(fuse_overwrite_zg_kernel<G, false, T>, P, h.n_items, a->n_planes, stream, a->grid_blocks);                   \
    } while (0)
            if (u16) {
                if (flat == 2) SQ_ZG(double, uint16_t);
                SQ_ZG(float, uint16_t);
            }
            if (flat == 2) SQ_ZG(double, uint8_t);      // uint8 planes in groups (round 4; no seam owners: see process_item_zg)
            SQ_ZG(float, uint8_t);
        }
        // (uint8 planes without gains stay with the per-plane pipeline: 0.651 against 0.643 through the groups on the arena,
        //  profiles/r04_exp_uint8_plane_groups.log -- with gains the groups take them from 0.367 to 0.601)
        if (u16 && !flat && a->scratch_dev && a->n_planes > 1 && ZB > 1 && !(a->flags & SQ_FUSE_NO_PLANE_GROUPS)) {
            // no flatfield: the planes still go through the items ZB at a time, dealt over the canvas allocation -- they
            // share nothing but the geometry, but a group's stores land in different stretches of device memory
            char *sc = static_cast<char *>(a->scratch_dev);
            uint32_t *n_groups = reinterpret_cast<uint32_t *>(sc + SL.n_groups);
            PlaneGroup *groups = reinterpret_cast<PlaneGroup *>(sc + SL.groups);
            hipLaunchKernelGGL(build_groups_kernel, dim3(1), dim3(256), 0, stream, (const void *const *)nullptr, (const uint32_t *)nullptr, 0u,
                               a->n_planes, a->canvas_plane_stride * esz, ZB, n_groups, groups,
                               (a->flags & SQ_FUSE_CONSECUTIVE_GROUPS) != 0);
            P.groups = groups;
            P.n_groups = n_groups;
            SQ_ZG(NoGain, uint16_t);
#undef SQ_ZG
        }
        if (u16) {
            if (flat == 0) SQ_OVERWRITE(uint16_t, 0);
            if (flat == 1) SQ_OVERWRITE(uint16_t, 1);
            SQ_OVERWRITE(uint16_t, 2);
        }
        if (flat == 0) SQ_OVERWRITE(uint8_t, 0);
        if (flat == 1) SQ_OVERWRITE(uint8_t, 1);
        SQ_OVERWRITE(uint8_t, 2);
#undef SQ_OVERWRITE
    }
    // feather
    const bool f32out = a->canvas_dtype == SQ_F32;
    if (!f32out && a->canvas_dtype != a->tile_dtype)
        return fail(SQ_ERR_INVALID, "sq_fuse_planes: feather canvas must be float32 or the tile dtype");
#define SQ_FEATHER_F(T, O, F)                                                                              \
    do {                                                                                                    \
        if (P.queue) return launch(fuse_feather_kernel<T, O, F, true>, P, h.n_items, a->n_planes, stream, a->grid_blocks);  \
        return launch(fuse_feather_kernel<T, O, F, false>, P, h.n_items, a->n_planes, stream, a->grid_blocks);              \
    } while (0)
#define SQ_FEATHER(T, O)                 \
    do {                                 \
        if (flat == 0) SQ_FEATHER_F(T, O, 0); \
        if (flat == 1) SQ_FEATHER_F(T, O, 1); \
        SQ_FEATHER_F(T, O, 2);           \
    } while (0)
    if (u16 && a->scratch_dev && a->n_planes > 1 && ZB > 1 && !(a->flags & SQ_FUSE_NO_PLANE_GROUPS) &&
        std::min(a->tile_h, a->tile_w) <= BLEND_WSUM_MAX) {   // a weight is at most half the shorter tile side, a weight sum twice that
        // planes that share a gain image (every gain moderate), or that have none, go through the items together
        // (fuse_feather_zg_kernel)
        char *sc = static_cast<char *>(a->scratch_dev);
        uint32_t *n_groups = reinterpret_cast<uint32_t *>(sc + SL.n_groups);
        PlaneGroup *groups = reinterpret_cast<PlaneGroup *>(sc + SL.groups);
        // class bits (flat_classify_kernel): 1 outside the fast divide's range, 2 not moderate, 4 not all positive.  A float32
        // canvas takes its grouped quotients as n * (1 / g) within the north star's 1e-5 relative (blend_item_zg) -- a bound
        // that holds for sums of same-signed terms, so planes with a non-positive gain stay with the exact per-plane blend
        hipLaunchKernelGGL(build_groups_kernel, dim3(1), dim3(256), 0, stream, a->flat_ptrs_dev, P.flat_class, f32out ? 7u : 3u, a->n_planes,
                           a->canvas_plane_stride * (int64_t)(f32out ? sizeof(float) : sizeof(uint16_t)), ZB, n_groups, groups,
                               (a->flags & SQ_FUSE_CONSECUTIVE_GROUPS) != 0);
        P.groups = groups;
        P.n_groups = n_groups;
#define SQ_FEATHER_ZG(F, O)                                                                                                      \
    do {                                                                                                                          \
        if (P.queue) return launch_zg(fuse_feather_zg_kernel<F, true, O>, P, h.n_items, a->n_planes, stream, a->grid_blocks);    \
        return launch_zg(fuse_feather_zg_kernel<F, false, O>, P, h.n_items, a->n_planes, stream, a->grid_blocks);                \
    } while (0)
        if (f32out) {      // float32 canvas: every item through the grouped form (blend_item_zg with NREF = 2 / 1 / 0)
            if (flat == 2) SQ_FEATHER_ZG(2, float);
            if (flat == 1) SQ_FEATHER_ZG(1, float);
            SQ_FEATHER_ZG(0, float);
        }
        if (flat == 2) SQ_FEATHER_ZG(2, uint16_t);      // float64 gains: likewise (the blend takes them as float32)
        if (flat == 1) SQ_FEATHER_ZG(1, uint16_t);
        SQ_FEATHER_ZG(0, uint16_t);
#undef SQ_FEATHER_ZG
    }
    if (u16) {
        if (f32out) SQ_FEATHER(uint16_t, float);
        SQ_FEATHER(uint16_t, uint16_t);
    }
    if (f32out) SQ_FEATHER(uint8_t, float);
    SQ_FEATHER(uint8_t, uint8_t);
#undef SQ_FEATHER
#undef SQ_FEATHER_F
}
