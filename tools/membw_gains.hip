// membw_gains: the two structural experiments VERDICT r2 (item 2) asks for, "membw first": the fusion kernel's inner
// work -- config-3-shaped row segments of Z planes that share one float32 gain image, uint16 pixel / gain -> uint16,
// the product's Markstein divide -- in three structures, alternated in ONE process on the same buffers:
//
//   A  "regs"      the shipped structure (fuse_overwrite_zg_kernel): a workgroup of 4 waves per item of 8 rows, wave w
//                  rows w, w + 4; per 16-byte slot a lane loads its 8 gains, takes 8 reciprocals, issues the Z planes'
//                  pixel loads and streams them through the divide: Z loads + Z stores per thread and slot, gains and
//                  reciprocals in 16 VGPRs.
//   B  "lds gains" gains staged in LDS per row: the workgroup's threads load the row's gains once, take the
//                  reciprocals once and write {g, 1/g} to LDS (lane-linear, conflict-free 16-byte units); then every
//                  WAVE takes a different plane: ONE pixel load + ONE store per thread and slot, the gain pairs by
//                  ds_read_b128.  Two LDS row buffers, one barrier per row.  ROWS rows per workgroup (8 = an item,
//                  1 / 2 = the short-lived workgroups of profiles/r02_membw_2d.log's row-wise copy).
//   C  "lds dma"   structure A with the pixel stream through LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
//                  instruction straight into a per-wave LDS ring, no VGPR destination), read back by ds_read_b128
//                  when the slot is computed.  (Sources are 16-byte aligned here; the real tiles sit at 2-byte
//                  phases, which an LDS-DMA could only serve through aligned chunks + v_alignbyte: this variant is
//                  the upper bound of what that could give.)
//   (P  plain copy of the same segments, Z planes per thread, no gains: the yardstick.)
//
// Geometry: G x G tiles of T x T uint16 (dense), every tile contributes `rows` row segments of S bytes starting at
// byte src_off; the canvas has an arbitrary pitch (rows at every 16-byte phase inside a 128-byte line); stores are
// issued so that lane 0 of every store instruction sits on a 128-byte line ("line slots", what the product does).
// Results of B and C are compared with A's on the device (bit for bit) before anything is timed.
//   hipcc --offload-arch=gfx950 -O3 membw_gains.hip -o membw_gains ;  ./membw_gains [Z=5] [reps=5]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <dlfcn.h>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);     \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define G1 __attribute__((address_space(1)))
struct __attribute__((packed)) U4U { u32x4 v; };
struct __attribute__((packed)) F4U { f32x4 v; };

constexpr int ZMAX = 5;
struct Geo {
    const char *src;      // plane z at src + z * src_plane
    char *dst;            // plane z at dst + z * dst_plane
    const float *gain;    // T x T
    size_t src_plane, dst_plane, src_tile, src_pitch, src_off, dst_pitch, dst_ty, dst_tx;
    int G, T, rows, S, gy0, gx0;   // gain of a segment's first pixel: (gy0, gx0)
};

__device__ __forceinline__ uint32_t cvt_u32_sat(float f) {
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}
__device__ __forceinline__ float quot(float n, float g, float r) {   // the product's div_u16_normal with the reciprocal given
    float q = n * r;
    return fmaf(fmaf(-g, q, n), r, q);
}
__device__ __forceinline__ uint32_t quot_pair(uint32_t word, float g0, float g1, float r0, float r1) {
    const float n0 = (float)(word & 0xFFFFu), n1 = (float)(word >> 16);
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 p = __builtin_amdgcn_cvt_pk_u16(cvt_u32_sat(quot(n0, g0, r0)), cvt_u32_sat(quot(n1, g1, r1)));
    return (uint32_t)p[0] | ((uint32_t)p[1] << 16);
}
__device__ __forceinline__ void st_nt(char *p, u32x4 v) { __builtin_nontemporal_store(v, (G1 u32x4 *)p); }
// the same quotient on the PACKED float32 pipe: two pixels per v_pk_mul_f32 / v_pk_fma_f32 (IEEE per component: the same bits)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t quot_pair_pk(uint32_t word, f32x2 g, f32x2 r) {
    const f32x2 n = {(float)(word & 0xFFFFu), (float)(word >> 16)};
    f32x2 q = n * r;
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(-g, q, n), r, q);
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 p = __builtin_amdgcn_cvt_pk_u16(cvt_u32_sat(q[0]), cvt_u32_sat(q[1]));
    return (uint32_t)p[0] | ((uint32_t)p[1] << 16);
}

struct RowAddr {
    const char *s;   // plane 0 source row
    char *d;         // plane 0 destination row
    const char *g;   // gain row (float)
    int shift;       // 16-byte units between the line boundary before d and d
};
__device__ __forceinline__ RowAddr row_of(const Geo &P, int tile, int r) {
    const int ty = tile / P.G, tx = tile % P.G;
    RowAddr A;
    A.s = P.src + (size_t)tile * P.src_tile + P.src_off + (size_t)r * P.src_pitch;
    A.d = P.dst + (size_t)ty * P.dst_ty + (size_t)tx * P.dst_tx + (size_t)r * P.dst_pitch;
    A.g = reinterpret_cast<const char *>(P.gain + (size_t)(P.gy0 + r) * P.T + P.gx0);
    A.shift = (int)(((uintptr_t)A.d & 127) >> 4);
    return A;
}

// ---- A: the shipped structure --------------------------------------------------------------------------------
template <int Z, bool GAINS>
__global__ __launch_bounds__(256) void k_regs(const Geo P) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nblk = (P.rows + 7) / 8;
    const int tile = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int nvec = P.S / 16;
    for (int j = 0; j < 2; ++j) {
        const int r = blk * 8 + wave + 4 * j;
        if (r >= P.rows) break;
        const RowAddr A = row_of(P, tile, r);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (64 * k - A.shift >= nvec) break;
            const int i = lane + 64 * k - A.shift;
            const bool act = i >= 0 && i < nvec;
            const uint32_t o = (uint32_t)min(max(i, 0), nvec - 1) * 16u;
            float g[8], rc[8];
            if (GAINS) {
                const f32x4 a = ((const G1 F4U *)(A.g + o * 2u))->v, b = ((const G1 F4U *)(A.g + o * 2u + 16))->v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    g[e] = a[e];
                    g[4 + e] = b[e];
                }
            }
            u32x4 px[Z];
#pragma unroll
            for (int z = 0; z < Z; ++z) px[z] = ((const G1 U4U *)(A.s + z * P.src_plane + o))->v;
            if (GAINS) {
#pragma unroll
                for (int c = 0; c < 8; ++c) rc[c] = __builtin_amdgcn_rcpf(g[c]);
            }
#pragma unroll
            for (int z = 0; z < Z; ++z) {
                u32x4 ov = px[z];
                if (GAINS) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ov[c] = quot_pair(px[z][c], g[2 * c], g[2 * c + 1], rc[2 * c], rc[2 * c + 1]);
                }
                if (act) st_nt(A.d + z * P.dst_plane + o, ov);
            }
        }
    }
}

// ---- A3: the shipped arithmetic, but the four waves of a workgroup on ONE row at a time (wave w the row's slot w): the workgroup
// writes 4 KiB of one canvas row together instead of 1 KiB of four rows
// ---- (derived from A) --------------------------------------------------------------------------------
template <int Z, bool GAINS>
__global__ __launch_bounds__(256) void k_regs_rowwise(const Geo P) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nblk = (P.rows + 7) / 8;
    const int tile = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int nvec = P.S / 16;
    for (int j = 0; j < 8; ++j) {
        const int r = blk * 8 + j;
        if (r >= P.rows) break;
        const RowAddr A = row_of(P, tile, r);
        {
            const int k = wave;
            if (64 * k - A.shift >= nvec) continue;
            const int i = lane + 64 * k - A.shift;
            const bool act = i >= 0 && i < nvec;
            const uint32_t o = (uint32_t)min(max(i, 0), nvec - 1) * 16u;
            float g[8], rc[8];
            if (GAINS) {
                const f32x4 a = ((const G1 F4U *)(A.g + o * 2u))->v, b = ((const G1 F4U *)(A.g + o * 2u + 16))->v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    g[e] = a[e];
                    g[4 + e] = b[e];
                }
            }
            u32x4 px[Z];
#pragma unroll
            for (int z = 0; z < Z; ++z) px[z] = ((const G1 U4U *)(A.s + z * P.src_plane + o))->v;
            if (GAINS) {
#pragma unroll
                for (int c = 0; c < 8; ++c) rc[c] = __builtin_amdgcn_rcpf(g[c]);
            }
#pragma unroll
            for (int z = 0; z < Z; ++z) {
                u32x4 ov = px[z];
                if (GAINS) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ov[c] = quot_pair(px[z][c], g[2 * c], g[2 * c + 1], rc[2 * c], rc[2 * c + 1]);
                }
                if (act) st_nt(A.d + z * P.dst_plane + o, ov);
            }
        }
    }
}

// ---- A2: the shipped structure with the quotient on the packed float32 pipe --------------------------------------------------------------------------------
template <int Z, bool GAINS>
__global__ __launch_bounds__(256) void k_regs_pk(const Geo P) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nblk = (P.rows + 7) / 8;
    const int tile = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int nvec = P.S / 16;
    for (int j = 0; j < 2; ++j) {
        const int r = blk * 8 + wave + 4 * j;
        if (r >= P.rows) break;
        const RowAddr A = row_of(P, tile, r);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (64 * k - A.shift >= nvec) break;
            const int i = lane + 64 * k - A.shift;
            const bool act = i >= 0 && i < nvec;
            const uint32_t o = (uint32_t)min(max(i, 0), nvec - 1) * 16u;
            float g[8], rc[8];
            if (GAINS) {
                const f32x4 a = ((const G1 F4U *)(A.g + o * 2u))->v, b = ((const G1 F4U *)(A.g + o * 2u + 16))->v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    g[e] = a[e];
                    g[4 + e] = b[e];
                }
            }
            u32x4 px[Z];
#pragma unroll
            for (int z = 0; z < Z; ++z) px[z] = ((const G1 U4U *)(A.s + z * P.src_plane + o))->v;
            if (GAINS) {
#pragma unroll
                for (int c = 0; c < 8; ++c) rc[c] = __builtin_amdgcn_rcpf(g[c]);
            }
#pragma unroll
            for (int z = 0; z < Z; ++z) {
                u32x4 ov = px[z];
                if (GAINS) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ov[c] = quot_pair_pk(px[z][c], f32x2{g[2 * c], g[2 * c + 1]}, f32x2{rc[2 * c], rc[2 * c + 1]});
                }
                if (act) st_nt(A.d + z * P.dst_plane + o, ov);
            }
        }
    }
}

// ---- B: gains staged in LDS, one wave per plane ---------------------------------------------------------------
// LDS image of a row: [4 components][256 units] float4; unit u = 64 * slot + lane; components g0-3, g4-7, r0-3, r4-7.
// Lane l of slot k reads unit 64k + l of every component: consecutive lanes, consecutive 16 bytes -- conflict-free.
template <int Z, int ROWS>
__global__ __launch_bounds__(64 * Z) void k_lds_gains(const Geo P) {
    __shared__ f32x4 s_g[2][4][256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nblk = (P.rows + ROWS - 1) / ROWS;
    const int tile = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int nvec = P.S / 16;
    const int r0 = blk * ROWS, nr = min(ROWS, P.rows - r0);
    const int u = threadIdx.x;          // staging unit of this thread (threads >= 256 stage nothing)
    auto stage_load = [&](const RowAddr &A, f32x4 &a, f32x4 &b) {
        const int i = (u & 63) + 64 * (u >> 6) - A.shift;
        const uint32_t o = (uint32_t)min(max(i, 0), nvec - 1) * 16u;
        a = ((const G1 F4U *)(A.g + o * 2u))->v;
        b = ((const G1 F4U *)(A.g + o * 2u + 16))->v;
    };
    auto stage_write = [&](int buf, const f32x4 &a, const f32x4 &b) {
        f32x4 ra, rb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ra[e] = __builtin_amdgcn_rcpf(a[e]);
            rb[e] = __builtin_amdgcn_rcpf(b[e]);
        }
        s_g[buf][0][u] = a;
        s_g[buf][1][u] = b;
        s_g[buf][2][u] = ra;
        s_g[buf][3][u] = rb;
    };
    RowAddr A = row_of(P, tile, r0);
    f32x4 ga, gb;
    if (u < 256) {
        stage_load(A, ga, gb);
        stage_write(0, ga, gb);
    }
    u32x4 px[4];
    auto px_load = [&](const RowAddr &R) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k - R.shift;
            const uint32_t o = (uint32_t)min(max(i, 0), nvec - 1) * 16u;
            px[k] = ((const G1 U4U *)(R.s + wave * P.src_plane + o))->v;
        }
    };
    px_load(A);
    for (int j = 0; j < nr; ++j) {
        __syncthreads();   // row j's gains are in s_g[j & 1]; everybody is done with s_g[(j + 1) & 1]
        const bool more = j + 1 < nr;
        RowAddr N = A;
        if (more) {
            N = row_of(P, tile, r0 + j + 1);
            if (u < 256) stage_load(N, ga, gb);
        }
        u32x4 ov[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int un = 64 * k + lane;
            const f32x4 a = s_g[j & 1][0][un], b = s_g[j & 1][1][un], ra = s_g[j & 1][2][un], rb = s_g[j & 1][3][un];
            ov[k][0] = quot_pair(px[k][0], a[0], a[1], ra[0], ra[1]);
            ov[k][1] = quot_pair(px[k][1], a[2], a[3], ra[2], ra[3]);
            ov[k][2] = quot_pair(px[k][2], b[0], b[1], rb[0], rb[1]);
            ov[k][3] = quot_pair(px[k][3], b[2], b[3], rb[2], rb[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k - A.shift;
            if (i >= 0 && i < nvec) st_nt(A.d + wave * P.dst_plane + (uint32_t)i * 16u, ov[k]);
        }
        if (more) {
            px_load(N);
            if (u < 256) stage_write((j + 1) & 1, ga, gb);
            A = N;
        }
    }
}

// ---- C: structure A, pixel stream through LDS-DMA -------------------------------------------------------------
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
template <int Z>
__global__ __launch_bounds__(256) void k_lds_dma(const Geo P) {
    // per wave: a ring of 2 slots x Z planes x 1 KiB
    __shared__ u32x4 s_px[4][2][Z][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nblk = (P.rows + 7) / 8;
    const int tile = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int nvec = P.S / 16;
    // steps of this wave: (row j, slot k), j = 0, 1; the DMA of step t + 1 is issued before step t is computed
    RowAddr R[2];
    int nrow = 0;
    for (int j = 0; j < 2; ++j) {
        const int r = blk * 8 + wave + 4 * j;
        if (r < P.rows) R[nrow++] = row_of(P, tile, r);
    }
    if (!nrow) return;
    const int steps = nrow * 4;
    auto issue = [&](int t) {
        const RowAddr &A = R[t >> 2];
        const int i = lane + 64 * (t & 3) - A.shift;
        const uint32_t o = (uint32_t)min(max(i, 0), nvec - 1) * 16u;
#pragma unroll
        for (int z = 0; z < Z; ++z) {
            const uint32_t dst = (uint32_t)(uintptr_t)&s_px[wave][t & 1][z][0];
            glds16(A.s + z * P.src_plane + o, dst);
        }
    };
    issue(0);
    for (int t = 0; t < steps; ++t) {
        const RowAddr &A = R[t >> 2];
        const int i = lane + 64 * (t & 3) - A.shift;
        const bool act = i >= 0 && i < nvec;
        const uint32_t o = (uint32_t)min(max(i, 0), nvec - 1) * 16u;
        const f32x4 a = ((const G1 F4U *)(A.g + o * 2u))->v, b = ((const G1 F4U *)(A.g + o * 2u + 16))->v;
        if (t + 1 < steps) {
            issue(t + 1);
            // the Z DMAs of step t are older than the Z of step t + 1 (and than the two gain loads? no: those were issued
            // before) -- wait until only step t + 1's are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Z) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        float g[8], rc[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            g[e] = a[e];
            g[4 + e] = b[e];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) rc[c] = __builtin_amdgcn_rcpf(g[c]);
#pragma unroll
        for (int z = 0; z < Z; ++z) {
            const u32x4 px = s_px[wave][t & 1][z][lane];
            u32x4 ov;
#pragma unroll
            for (int c = 0; c < 4; ++c) ov[c] = quot_pair(px[c], g[2 * c], g[2 * c + 1], rc[2 * c], rc[2 * c + 1]);
            if (act) st_nt(A.d + z * P.dst_plane + o, ov);
        }
        // the ring slot (t & 1) is overwritten by the DMA of step t + 2, issued in iteration t + 1 after this wave's own
        // ds_reads above have returned (their results were consumed by the stores' operands)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// yardsticks for the placement experiment: a linear copy and a linear fill of the same byte count
__global__ __launch_bounds__(256) void k_linear_copy(const u32x4 *src, u32x4 *dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(((const G1 u32x4 *)src)[i], (G1 u32x4 *)dst + i);
}
__global__ __launch_bounds__(256) void k_linear_fill(u32x4 *dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32x4 v = {1u, 2u, 3u, 4u};
    if (i < n) __builtin_nontemporal_store(v, (G1 u32x4 *)dst + i);
}
// canvas rows only: zero fill of the same row segments (no source at all)
template <int Z>
__global__ __launch_bounds__(256) void k_fill_rows(const Geo P) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nblk = (P.rows + 7) / 8;
    const int tile = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int nvec = P.S / 16;
    const u32x4 v = {1u, 2u, 3u, 4u};
    for (int j = 0; j < 2; ++j) {
        const int r = blk * 8 + wave + 4 * j;
        if (r >= P.rows) break;
        const RowAddr A = row_of(P, tile, r);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k - A.shift;
            if (i >= 0 && i < nvec)
#pragma unroll
                for (int z = 0; z < Z; ++z) st_nt(A.d + z * P.dst_plane + (uint32_t)i * 16u, v);
        }
    }
}

// per-plane canvas pointers (each plane its own allocation): the row fill and the plain 5-planes-per-thread copy
struct PlanePtrs { char *d[ZMAX]; };
template <int Z, bool COPY>
__device__ __forceinline__ void k_rows_pp_body(const Geo &P, const PlanePtrs &D) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nblk = (P.rows + 7) / 8;
    const int tile = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int nvec = P.S / 16;
    for (int j = 0; j < 2; ++j) {
        const int r = blk * 8 + wave + 4 * j;
        if (r >= P.rows) break;
        const RowAddr A = row_of(P, tile, r);      // A.d relative to P.dst = nullptr: an offset
        const size_t doff = (size_t)(A.d - P.dst);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k - A.shift;
            const bool act = i >= 0 && i < nvec;
            const uint32_t o = (uint32_t)min(max(i, 0), nvec - 1) * 16u;
            u32x4 px[Z];
#pragma unroll
            for (int z = 0; z < Z; ++z) px[z] = COPY ? ((const G1 U4U *)(A.s + z * P.src_plane + o))->v : u32x4{1u, 2u, 3u, 4u};
#pragma unroll
            for (int z = 0; z < Z; ++z)
                if (act) st_nt(D.d[z] + doff + o, px[z]);
        }
    }
}

template <int Z, bool COPY>
__global__ __launch_bounds__(256) void k_rows_pp(const Geo P, const PlanePtrs D) { k_rows_pp_body<Z, COPY>(P, D); }

// ---- mode 500: the same row fill under three NAMES, so that per-dispatch PMC values can be told apart by kernel name ----
__global__ __launch_bounds__(256) void k_pair_same_class(const Geo P, const PlanePtrs D) { k_rows_pp_body<2, false>(P, D); }
__global__ __launch_bounds__(256) void k_pair_other_class(const Geo P, const PlanePtrs D) { k_rows_pp_body<2, false>(P, D); }
__global__ __launch_bounds__(256) void k_single_plane(const Geo P, const PlanePtrs D) { k_rows_pp_body<1, false>(P, D); }
__global__ __launch_bounds__(256) void k_pair_same_class_copy(const Geo P, const PlanePtrs D) { k_rows_pp_body<2, true>(P, D); }
__global__ __launch_bounds__(256) void k_pair_other_class_copy(const Geo P, const PlanePtrs D) { k_rows_pp_body<2, true>(P, D); }

// ---- the canvas' all-zero TAIL (the reference's oversize rows: 20 % of the voxels): a dense block of full-width rows ----
// (a) cut like the plan cuts it today: items of 8 rows x 4096-byte pieces, wave w rows w and w + 4
template <int Z>
__global__ __launch_bounds__(256) void k_tail_items(const PlanePtrs D, size_t pitch, int rows, int pieces) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int band = blockIdx.x / pieces, piece = blockIdx.x % pieces;
    const u32x4 v = {0u, 0u, 0u, 0u};
    for (int j = 0; j < 2; ++j) {
        const int r = band * 8 + wave + 4 * j;
        if (r >= rows) break;
        const size_t row0 = (size_t)r * pitch;
        const size_t seg0 = row0 + (size_t)piece * 4096, seg1 = min(row0 + pitch, seg0 + 4096);
#pragma unroll
        for (int z = 0; z < Z; ++z) {
            char *d = D.d[z];
            const size_t a0 = (seg0 + ((uintptr_t)(d + seg0) & 127 ? 128 - ((uintptr_t)(d + seg0) & 127) : 0));   // first line boundary
            // head bytes before the first line boundary (multiples of 2 here: pitch even) -- one 16-byte store per lane where whole
            for (size_t q = seg0 + lane * 16; q + 16 <= a0 && q + 16 <= seg1; q += 64 * 16) st_nt(d + q, v);
            for (int k = 0; k < 5; ++k) {
                const size_t q = a0 + ((size_t)lane + 64 * k) * 16;
                if (q + 16 <= seg1) st_nt(d + q, v);
            }
        }
    }
}
// (b) one full row per workgroup, the four waves a contiguous quarter each (line-aligned split)
template <int Z>
__global__ __launch_bounds__(256) void k_tail_rows(const PlanePtrs D, size_t pitch, int rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t row0 = (size_t)blockIdx.x * pitch;
    const u32x4 v = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int z = 0; z < Z; ++z) {
        char *d = D.d[z];
        const size_t first = (row0 + 15) / 16 * 16, last = (row0 + pitch) / 16 * 16;     // whole 16-byte vectors of the row
        const size_t nvec = (last - first) / 16, per = (nvec + 3) / 4;
        const size_t v0 = wave * per, v1 = min(nvec, v0 + per);
        for (size_t i = v0 + lane; i < v1; i += 64) st_nt(d + first + i * 16, v);
    }
}
// (c) the block as what it is: one contiguous run, 4 KiB per workgroup
template <int Z>
__global__ __launch_bounds__(256) void k_tail_linear(const PlanePtrs D, size_t bytes) {
    const size_t q = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    const u32x4 v = {0u, 0u, 0u, 0u};
    if (q + 16 <= bytes)
#pragma unroll
        for (int z = 0; z < Z; ++z) st_nt(D.d[z] + q, v);
}

__global__ void k_compare(const uint32_t *a, const uint32_t *b, size_t n, unsigned long long *bad) {
    unsigned long long mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) mine += a[i] != b[i];
    if (mine) atomicAdd(bad, mine);
}
__global__ void k_init(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 15;
        x *= 2246822519u;
        x ^= x >> 13;
        p[i] = x;
    }
}
__global__ void k_init_gain(float *g, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        g[i] = 0.8f + 0.4f * (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.0f;
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    double best = 1e30, sum = 0;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
        sum += ms;
    }
    CK(hipGetLastError());
    (void)sum;
    return best;
}

int main(int argc, char **argv) {
    constexpr int Z = ZMAX;
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    // plane-stride experiments: extra bytes between consecutive planes of the tiles / of the canvas (multiples of 16);
    // quick = 1: only P, A and B(8) per round
    const size_t src_pad = argc > 2 ? (size_t)atoll(argv[2]) : 0, dst_pad = argc > 3 ? (size_t)atoll(argv[3]) : 0;
    const bool quick = argc > 4 && atoi(argv[4]);
    const int G = 16, T = 2048;
    Geo P{};
    P.G = G;
    P.T = T;
    P.rows = 1800;
    P.S = 3600;
    P.src_tile = (size_t)T * T * 2;
    P.src_pitch = 4096;
    P.src_off = 124 * 4096 + 240;            // 16-byte aligned source (C needs it); the phase costs ~2 % (r02_membw_2d.log)
    P.gy0 = 124;
    P.gx0 = 120;
    P.dst_pitch = (size_t)G * P.S + 560;     // rows at every 16-byte phase of a line
    P.dst_tx = P.S;
    P.dst_ty = (size_t)P.rows * P.dst_pitch;
    P.src_plane = (size_t)G * G * P.src_tile + src_pad;
    P.dst_plane = (((size_t)G * P.rows * P.dst_pitch) + 4095) / 4096 * 4096 + dst_pad;
    char *src, *dst, *ref;
    float *gain;
    unsigned long long *bad;
    if (getenv("MEMBW_MIXED")) {
        // dst and ref in a MIXED arena (libsquidstitch's sq_arena_create through its C ABI: physical slices classified and mapped
        // round-robin over the card's memory classes) -- created FIRST, while the card is empty, so that it can choose
        void *lib = dlopen(getenv("MEMBW_MIXED"), RTLD_NOW);
        if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
        typedef void *(*create_t)(int64_t, int64_t, int64_t, int64_t, int32_t, void *, void *);
        create_t create = (create_t)dlsym(lib, "sq_arena_create");
        struct { void *base; int64_t bytes, slice; int32_t n_slices, n_cand, n_classes, cs[8], cc[8], inter; float probe_ms, create_ms, lo, hi; } info;
        const size_t need = 2 * Z * P.dst_plane + (64 << 20);
        if (!create || !create((int64_t)need, (int64_t)(3 * need), 0, 0, 0, nullptr, &info)) { printf("sq_arena_create failed\n"); return 1; }
        printf("dst and ref in a mixed arena: %d slices of %ld MiB from %d candidates, %d classes (%d %d %d %d ...), created in %.0f ms\n", info.n_slices,
               (long)(info.slice >> 20), info.n_cand, info.n_classes, info.cs[0], info.cs[1], info.cs[2], info.cs[3], info.create_ms);
        dst = (char *)info.base;
        ref = dst + Z * P.dst_plane;
    } else {
        CK(hipMalloc(&dst, Z * P.dst_plane));
        CK(hipMalloc(&ref, Z * P.dst_plane));
    }
    CK(hipMalloc(&src, Z * P.src_plane));
    CK(hipMalloc(&gain, (size_t)T * T * 4));
    CK(hipMalloc(&bad, 8));
    hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, (uint32_t *)src, Z * P.src_plane / 4, 12345u);
    hipLaunchKernelGGL(k_init_gain, dim3(1024), dim3(256), 0, 0, gain, (size_t)T * T);
    CK(hipMemset(dst, 0, Z * P.dst_plane));
    CK(hipMemset(ref, 0, Z * P.dst_plane));
    CK(hipDeviceSynchronize());
    P.src = src;
    P.gain = gain;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; %d planes of a %dx%d grid of %d x %d-byte segments, canvas pitch %zu; plane strides: tiles %zu (+%zu), canvas %zu (+%zu)\n",
           prop.name, prop.multiProcessorCount, Z, G, G, P.rows, P.S, P.dst_pitch, P.src_plane, src_pad, P.dst_plane, dst_pad);
    const double moved = 2.0 * Z * G * G * (double)P.rows * P.S;   // pixel bytes read + written (gains: + T*T*4 once, not counted)
    const unsigned items8 = (unsigned)(G * G * ((P.rows + 7) / 8));
    auto report = [&](const char *name, double ms, const char *check) {
        printf("%-86s %8.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)  %s\n", name, ms, moved / ms / 1e6, moved / ms / 1e6 / 8000, check);
        fflush(stdout);
    };
    auto check = [&](const char *what) -> const char * {
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(k_compare, dim3(4096), dim3(256), 0, 0, (const uint32_t *)dst, (const uint32_t *)ref, Z * P.dst_plane / 4, bad);
        unsigned long long h;
        CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        static char buf[96];
        snprintf(buf, sizeof buf, h ? "MISMATCH vs A: %llu words" : "== A bit for bit", h);
        if (h) printf("!! %s: %s\n", what, buf);
        return buf;
    };
    if (argc > 5 && atoi(argv[5]) == 700) {
        // A MIXED ARENA: device memory taken in SLICE-sized physical pieces (hipMemCreate), classified UNIT by UNIT (a unit =
        // 512 MiB of consecutively created slices: physical memory is handed out in long runs) by the pair fill, and the slices
        // mapped into ONE virtual range round-robin over the classes (hipMemAddressReserve / hipMemMap): every plane laid out in
        // that range then has all classes under it at SLICE granularity.  Compared in this process with the same kernels on a
        // plain hipMalloc of the same size: single plane, 5 consecutive planes, kernel A, linear fill.
        //   membw_gains reps 0 0 1 700 [GiB=96] [slice MiB=512] [order: 0 = round-robin, 1 = natural (control)]
        const size_t gib = argc > 6 ? (size_t)atoll(argv[6]) : 96;
        const size_t SLICE = (argc > 7 ? (size_t)atoll(argv[7]) : 512) << 20;
        const int natural = argc > 8 ? atoi(argv[8]) : 0;
        const size_t UNIT = SLICE > ((size_t)512 << 20) ? SLICE : ((size_t)512 << 20);
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        const int spu = (int)(UNIT / SLICE);
        const int nu = (int)((gib << 30) / UNIT), n = nu * spu;
        printf("mixed arena: %d slices of %zu MiB in %d units of %zu MiB (allocation granularity %zu KiB)\n", n, SLICE >> 20, nu, UNIT >> 20, gran >> 10);
        hipMemGenericAllocationHandle_t *h = (hipMemGenericAllocationHandle_t *)malloc(sizeof(hipMemGenericAllocationHandle_t) * n);
        int *order = (int *)malloc(sizeof(int) * n);
        auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
        double t0 = now();
        for (int i = 0; i < n; ++i) CK(hipMemCreate(&h[i], SLICE, &prop, 0));
        double t1 = now();
        char *va = nullptr;
        CK(hipMemAddressReserve((void **)&va, (size_t)n * SLICE, (size_t)1 << 30, nullptr, 0));
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        for (int i = 0; i < n; ++i) CK(hipMemMap(va + (size_t)i * SLICE, SLICE, 0, h[i], 0));
        double t2 = now();
        CK(hipMemSetAccess(va, (size_t)n * SLICE, &acc, 1));
        double t3 = now();
        printf("host: hipMemCreate x %d %.1f ms, hipMemMap x %d %.1f ms, hipMemSetAccess %.1f ms\n", n, (t1 - t0) * 1e3, n, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
        // a probe geometry that fits one unit
        Geo Q = P;
        Q.src = src;
        Q.dst = nullptr;
        for (Q.G = 16; Q.G > 1; --Q.G) {
            Q.dst_pitch = (size_t)Q.G * Q.S + 560;
            Q.dst_ty = (size_t)Q.rows * Q.dst_pitch;
            if ((size_t)Q.G * Q.rows * Q.dst_pitch <= UNIT) break;
        }
        const unsigned qitems = (unsigned)(Q.G * Q.G * ((Q.rows + 7) / 8));
        const double qmoved = (double)Q.G * Q.G * Q.rows * Q.S;      // bytes ONE plane's fill writes
        printf("probe: %d x %d tiles, %.0f MB per unit\n", Q.G, Q.G, qmoved / 1e6);
        int *cls = (int *)malloc(sizeof(int) * nu);
        double *rt = (double *)malloc(sizeof(double) * nu);
        for (int k = 0; k < nu; ++k) cls[k] = -1;
        int ncls = 0;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        int probes = 0;
        for (; ncls < 8; ++ncls) {
            int ref = -1;
            for (int k = 0; k < nu; ++k)
                if (cls[k] < 0) { ref = k; break; }
            if (ref < 0) break;
            cls[ref] = ncls;
            double lo = 1e9, hi = 0;
            for (int k = 0; k < nu; ++k) {
                if (cls[k] >= 0) continue;
                PlanePtrs D{};
                D.d[0] = va + (size_t)ref * UNIT;
                D.d[1] = va + (size_t)k * UNIT;
                const double ms2 = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<2, false>), dim3(qitems), dim3(256), 0, 0, Q, D); }, 2);
                rt[k] = 2 * qmoved / ms2 / 1e6 / 8000;
                lo = rt[k] < lo ? rt[k] : lo;
                hi = rt[k] > hi ? rt[k] : hi;
                ++probes;
            }
            int members = 1;
            const double cut = hi - lo > 0.06 ? 0.5 * (lo + hi) : (ncls == 0 ? 1e9 : (lo < 0.62 ? 1e9 : -1));
            for (int k = 0; k < nu; ++k)
                if (cls[k] < 0 && rt[k] < cut) { cls[k] = ncls; ++members; }
            printf("class %c: reference unit %d, pair rates %.3f .. %.3f, %d units\n", 'A' + ncls, ref, lo, hi, members);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float pms;
        CK(hipEventElapsedTime(&pms, e0, e1));
        printf("%d probes, %.1f ms on the device\nunits in allocation order: ", probes, pms);
        for (int k = 0; k < nu; ++k) printf("%c", cls[k] >= 0 ? 'A' + cls[k] : '?');
        printf("\n");
        // remap round-robin over the classes, slice by slice
        CK(hipDeviceSynchronize());
        t0 = now();
        for (int i = 0; i < n; ++i) CK(hipMemUnmap(va + (size_t)i * SLICE, SLICE));      // one call per mapping (a call over many undoes the first only)
        int no = 0;
        if (natural) {
            for (int i = 0; i < n; ++i) order[no++] = i;
        } else {
            int next[8] = {};
            bool any = true;
            while (any) {
                any = false;
                for (int c = 0; c < ncls; ++c) {
                    while (next[c] < n && cls[next[c] / spu] != c) ++next[c];
                    if (next[c] < n) { order[no++] = next[c]++; any = true; }
                }
            }
        }
        for (int i = 0; i < no; ++i) CK(hipMemMap(va + (size_t)i * SLICE, SLICE, 0, h[order[i]], 0));
        CK(hipMemSetAccess(va, (size_t)no * SLICE, &acc, 1));
        printf("remapped %s in %.1f ms; first slices: ", natural ? "in NATURAL order (control)" : "round-robin over the classes", (now() - t0) * 1e3);
        for (int i = 0; i < no && i < 96; ++i) printf("%c", 'A' + cls[order[i] / spu]);
        printf("\n");
        // the plain arena of the same size
        char *plain;
        CK(hipMalloc(&plain, (size_t)n * SLICE));
        const double moved1 = moved / Z;
        auto bench = [&](const char *name, char *base) {
            P.src = src;
            for (int round = 0; round < 2; ++round) {
                // single planes at a few offsets, 5 consecutive planes at a few offsets
                printf("%s round %d: single plane fill", name, round);
                for (size_t off : {(size_t)0, (size_t)11 << 30, (size_t)37 << 30, (size_t)61 << 30}) {
                    if (off + P.dst_plane > (size_t)n * SLICE) continue;
                    PlanePtrs D{};
                    D.d[0] = base + off;
                    P.dst = nullptr;
                    const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<1, false>), dim3(items8), dim3(256), 0, 0, P, D); }, reps);
                    printf(" %.3f", 0.5 * moved1 / ms / 1e6 / 8000);
                }
                printf(" | 5 consecutive planes: fill / copy / kernel A");
                for (size_t off : {(size_t)0, (size_t)23 << 30, (size_t)52 << 30}) {
                    if (off + Z * P.dst_plane > (size_t)n * SLICE) continue;
                    PlanePtrs D{};
                    for (int z = 0; z < Z; ++z) D.d[z] = base + off + z * P.dst_plane;
                    P.dst = nullptr;
                    const double msr = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<Z, false>), dim3(items8), dim3(256), 0, 0, P, D); }, reps);
                    const double msc = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<Z, true>), dim3(items8), dim3(256), 0, 0, P, D); }, reps);
                    P.dst = base + off;
                    const double msa = time_ms([&] { hipLaunchKernelGGL((k_regs<Z, true>), dim3(items8), dim3(256), 0, 0, P); }, reps);
                    printf("  %.3f / %.3f / %.3f", 0.5 * moved / msr / 1e6 / 8000, moved / msc / 1e6 / 8000, moved / msa / 1e6 / 8000);
                }
                const size_t nlin = (size_t)8 << 30 >> 4;
                const double msf = time_ms([&] { hipLaunchKernelGGL(k_linear_fill, dim3((unsigned)((nlin + 255) / 256)), dim3(256), 0, 0, (u32x4 *)base, nlin); }, reps);
                const double msl = time_ms([&] { hipLaunchKernelGGL(k_linear_copy, dim3((unsigned)((nlin + 255) / 256)), dim3(256), 0, 0, (const u32x4 *)src, (u32x4 *)base, nlin); }, reps);
                printf(" | linear fill / copy of 8 GiB %.3f / %.3f\n", (double)nlin * 16 / msf / 1e6 / 8000, 2.0 * nlin * 16 / msl / 1e6 / 8000);
                fflush(stdout);
            }
        };
        bench("plain hipMalloc", plain);
        bench("mixed arena    ", va);
        bench("plain hipMalloc", plain);
        bench("mixed arena    ", va);
        // the SOURCE in mixed memory too: kernel A reading its tiles from the mixed arena (upper half) and writing the lower half
        if ((size_t)n * SLICE >= Z * (P.src_plane + P.dst_plane) + ((size_t)2 << 30)) {
            char *msrc = va + (((Z * P.dst_plane) + ((size_t)1 << 30)) & ~(((size_t)1 << 30) - 1));
            CK(hipMemcpy(msrc, src, Z * P.src_plane, hipMemcpyDeviceToDevice));
            for (int round = 0; round < 2; ++round) {
                P.src = msrc;
                P.dst = va;
                const double m1 = time_ms([&] { hipLaunchKernelGGL((k_regs<Z, true>), dim3(items8), dim3(256), 0, 0, P); }, reps);
                P.src = src;
                const double m2 = time_ms([&] { hipLaunchKernelGGL((k_regs<Z, true>), dim3(items8), dim3(256), 0, 0, P); }, reps);
                printf("kernel A, canvas in the mixed arena: tiles in the mixed arena too %.3f | tiles in a plain hipMalloc %.3f\n", moved / m1 / 1e6 / 8000, moved / m2 / 1e6 / 8000);
            }
        }
        return 0;
    }
    if (argc > 5 && atoi(argv[5]) == 600) {
        // HOW MANY CLASSES, HOW BIG: as much device memory as one allocation gives, plane k at +k GiB, every plane classified
        // against reference planes: class 0 = what collides with plane 0 (the pair fills at one plane's rate), class 1 = what
        // collides with the first plane outside class 0, and so on.  Prints the class of every GiB and the GiB per class.
        P.src = src;
        P.dst = nullptr;
        size_t gib = argc > 6 ? (size_t)atoll(argv[6]) : 240;
        char *big = nullptr;
        while (gib >= 32 && hipMalloc(&big, gib << 30) != hipSuccess) { (void)hipGetLastError(); gib -= 8; }
        if (!big) { printf("no big allocation\n"); return 1; }
        const size_t STEP = (size_t)1 << 30;
        const int nk = (int)(((gib << 30) - P.dst_plane) / STEP);
        const double moved1 = moved / Z;
        printf("one allocation of %zu GiB at %p; plane k at +k GiB (a plane is %.2f GiB)\n", gib, (void *)big, P.dst_plane / 1073741824.0);
        static int cls[512];
        for (int k = 0; k < nk; ++k) cls[k] = -1;
        int ncls = 0;
        for (;;) {
            int ref = -1;
            for (int k = 0; k < nk; ++k)
                if (cls[k] < 0) { ref = k; break; }
            if (ref < 0 || ncls >= 12) break;
            cls[ref] = ncls;
            if (ref + 1 < nk && cls[ref + 1] < 0) cls[ref + 1] = -2;   // overlaps the reference plane: decided by its neighbours below
            int members = 1;
            for (int k = 0; k < nk; ++k) {
                if (cls[k] != -1) continue;
                if (abs(k - ref) < 2) continue;
                PlanePtrs D{};
                D.d[0] = big + ref * STEP;
                D.d[1] = big + k * STEP;
                const double ms2 = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<2, false>), dim3(items8), dim3(256), 0, 0, P, D); }, 2);
                const double r = moved1 / ms2 / 1e6 / 8000;
                if (r < 0.64) { cls[k] = ncls; ++members; }
            }
            printf("class %d: reference plane at +%d GiB, %d planes collide with it\n", ncls, ref, members);
            fflush(stdout);
            ++ncls;
        }
        for (int k = 0; k < nk; ++k)
            if (cls[k] == -2) cls[k] = k + 1 < nk && cls[k + 1] >= 0 ? cls[k + 1] : cls[k - 1];
        int size[16] = {};
        printf("class of every GiB:");
        for (int k = 0; k < nk; ++k) {
            if (k % 64 == 0) printf("\n  +%3d GiB: ", k);
            printf("%c", cls[k] >= 0 ? 'A' + cls[k] : '?');
            if (cls[k] >= 0) ++size[cls[k]];
        }
        printf("\nGiB per class:");
        for (int c = 0; c < ncls; ++c) printf(" %c %d", 'A' + c, size[c]);
        printf("\n");
        // cross-check: one plane from each of the first three classes, pairwise and all together
        return 0;
    }
    if (argc > 5 && atoi(argv[5]) == 500) {
        // CLASSES UNDER THE COUNTERS: one big allocation, plane k at +k GiB.  (1) row fill of the pair (0, k) for every k: which
        // stretches collide with the stretch at 0 ("same class": the pair runs at one plane's rate) and which do not;
        // (2) the same fill launched under three kernel NAMES -- k_pair_same_class, k_pair_other_class, k_single_plane -- a few
        // times each, so that a rocprofv3 --pmc pass of this very process reports the memory-side counters per case.
        P.src = src;
        P.dst = nullptr;
        const size_t gib = argc > 6 ? (size_t)atoll(argv[6]) : 128;
        const size_t total = gib << 30, STEP = (size_t)1 << 30;
        char *big;
        CK(hipMalloc(&big, total));
        const int nk = (int)((total - P.dst_plane) / STEP);
        const double moved1 = moved / Z;
        printf("one allocation of %zu GiB at %p; plane k at +k GiB (a plane is %.2f GiB); row fill of the pair (0, k), fraction of 8 TB/s\n",
               total >> 30, (void *)big, P.dst_plane / 1073741824.0);
        static double rate[512];
        double lo = 1e9, hi = 0;
        for (int k = 2; k < nk; ++k) {
            PlanePtrs D{};
            D.d[0] = big;
            D.d[1] = big + k * STEP;
            const double ms2 = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<2, false>), dim3(items8), dim3(256), 0, 0, P, D); }, 2);
            rate[k] = 0.5 * 2 * moved1 / ms2 / 1e6 / 8000;
            lo = rate[k] < lo ? rate[k] : lo;
            hi = rate[k] > hi ? rate[k] : hi;
            printf(" %.3f", rate[k]);
            if (k % 16 == 1) printf("\n");
        }
        printf("\n");
        const double mid = 0.5 * (lo + hi);
        // a "same class" k and an "other class" k, each with like neighbours (inside a stretch, not at its edge)
        int ks = -1, kf = -1;
        for (int k = 3; k + 1 < nk; ++k) {
            const bool s = rate[k - 1] < mid && rate[k] < mid && rate[k + 1] < mid, f = rate[k - 1] > mid && rate[k] > mid && rate[k + 1] > mid;
            if (s && ks < 0) ks = k;
            if (f && kf < 0) kf = k;
        }
        printf("lowest %.3f, highest %.3f; same-class plane at +%d GiB, other-class plane at +%d GiB\n", lo, hi, ks, kf);
        if (ks < 0 || kf < 0 || hi - lo < 0.08) {
            printf("no two classes inside this allocation (spread %.3f): nothing to compare\n", hi - lo);
            return 0;
        }
        PlanePtrs S{}, F{}, O{};
        S.d[0] = F.d[0] = O.d[0] = big;
        S.d[1] = big + ks * STEP;
        F.d[1] = big + kf * STEP;
        for (int round = 0; round < 3; ++round) {
            const double a = time_ms([&] { hipLaunchKernelGGL(k_pair_same_class, dim3(items8), dim3(256), 0, 0, P, S); }, reps);
            const double b = time_ms([&] { hipLaunchKernelGGL(k_pair_other_class, dim3(items8), dim3(256), 0, 0, P, F); }, reps);
            const double c = time_ms([&] { hipLaunchKernelGGL(k_single_plane, dim3(items8), dim3(256), 0, 0, P, O); }, reps);
            const double d = time_ms([&] { hipLaunchKernelGGL(k_pair_same_class_copy, dim3(items8), dim3(256), 0, 0, P, S); }, reps);
            const double e = time_ms([&] { hipLaunchKernelGGL(k_pair_other_class_copy, dim3(items8), dim3(256), 0, 0, P, F); }, reps);
            printf("round %d: fill pair same class %.3f ms = %.3f | pair other class %.3f ms = %.3f | single plane %.3f ms = %.3f | copy pair same %.3f ms = %.3f | copy pair other %.3f ms = %.3f of 8 TB/s\n",
                   round, a, moved1 / a / 1e6 / 8000, b, moved1 / b / 1e6 / 8000, c, 0.5 * moved1 / c / 1e6 / 8000, d, 2 * moved1 / d / 1e6 / 8000, e,
                   2 * moved1 / e / 1e6 / 8000);
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 5 && atoi(argv[5]) == 400) {
        // THE ZERO TAIL: 5 planes 16 GiB apart in one allocation (the arena's spacing), each with a dense tail of 15 128
        // rows x 116 068 bytes (the headline grid's); the three ways of writing it
        const size_t pitch = 116068, spacing = (size_t)16 << 30;
        const int rows = 15128;
        const size_t bytes = pitch * rows;
        char *big;
        CK(hipMalloc(&big, 4 * spacing + bytes + 4096));
        PlanePtrs D;
        for (int z = 0; z < Z; ++z) D.d[z] = big + z * spacing;
        const int pieces = (int)((pitch + 4095) / 4096);
        for (int round = 0; round < 3; ++round) {
            const double ma = time_ms([&] { hipLaunchKernelGGL((k_tail_items<Z>), dim3((unsigned)((rows + 7) / 8 * pieces)), dim3(256), 0, 0, D, pitch, rows, pieces); }, reps);
            const double mb = time_ms([&] { hipLaunchKernelGGL((k_tail_rows<Z>), dim3((unsigned)rows), dim3(256), 0, 0, D, pitch, rows); }, reps);
            const double mc = time_ms([&] { hipLaunchKernelGGL((k_tail_linear<Z>), dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, D, bytes); }, reps);
            printf("round %d: zero tail of %d planes x %.2f GB: 8-row x 4-KiB items %.3f | one full row per workgroup %.3f | linear 4-KiB chunks %.3f of 8 TB/s\n",
                   round, Z, bytes / 1e9, Z * bytes / ma / 1e6 / 8000, Z * bytes / mb / 1e6 / 8000, Z * bytes / mc / 1e6 / 8000);
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 5 && atoi(argv[5]) == 300) {
        // ADDRESS AXIS: one 64 GiB allocation; "plane" k starts k * STEP bytes into it.  Row fill on the pair (ref, k): which
        // stretches of the allocation collide with the stretch at `ref`?
        P.src = src;
        P.dst = nullptr;
        const size_t total = (size_t)64 << 30, STEP = (size_t)512 << 20;
        char *big;
        CK(hipMalloc(&big, total));
        const int nk = (int)((total - P.dst_plane) / STEP);
        printf("one allocation of %zu GiB at %p; plane k at +k * %zu MiB (a plane is %.2f GiB); row fill of the pair (ref, k), fraction of 8 TB/s\n",
               total >> 30, (void *)big, STEP >> 20, P.dst_plane / 1073741824.0);
        const double moved1 = moved / Z;
        const int refs[] = {0, 13, 40};
        for (int ref : refs) {
            printf("ref %3d:", ref);
            for (int k = 0; k < nk; ++k) {
                if ((size_t)abs(k - ref) * STEP < P.dst_plane) {   // overlapping: skip
                    printf("  --- ");
                    continue;
                }
                PlanePtrs D{};
                D.d[0] = big + ref * STEP;
                D.d[1] = big + k * STEP;
                const double ms2 = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<2, false>), dim3(items8), dim3(256), 0, 0, P, D); }, 2);
                printf(" %.3f", 0.5 * 2 * moved1 / ms2 / 1e6 / 8000);
            }
            printf("\n");
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 5 && atoi(argv[5]) == 200) {
        // PAIRWISE: N separately allocated canvas planes; the row fill on every PAIR of them (2 planes per thread), on single
        // planes, and on a few sets of five chosen from the pair matrix
        P.src = src;
        P.dst = nullptr;
        constexpr int N = 14;
        char *pl[N];
        size_t sp = 7;
        for (int i = 0; i < N; ++i) {
            if (i >= N / 2) {
                char *spacer;
                sp = sp * 1103515245u + 12345u;
                CK(hipMalloc(&spacer, ((sp >> 8) % 997 + 3) << 20));
            }
            CK(hipMalloc(&pl[i], P.dst_plane));
        }
        const double moved1 = moved / Z;   // pixel bytes read + written of ONE plane
        static double rate[N][N];
        for (int i = 0; i < N; ++i) {
            PlanePtrs D{};
            D.d[0] = pl[i];
            const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<1, false>), dim3(items8), dim3(256), 0, 0, P, D); }, reps);
            rate[i][i] = 0.5 * moved1 / ms / 1e6 / 8000;
            for (int j = i + 1; j < N; ++j) {
                D.d[1] = pl[j];
                const double ms2 = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<2, false>), dim3(items8), dim3(256), 0, 0, P, D); }, reps);
                rate[i][j] = rate[j][i] = 0.5 * 2 * moved1 / ms2 / 1e6 / 8000;
            }
        }
        printf("row fill, fraction of 8 TB/s: diagonal = one plane per thread, off-diagonal = the pair (2 planes per thread)\n      ");
        for (int j = 0; j < N; ++j) printf("  p%02d ", j);
        printf("\n");
        for (int i = 0; i < N; ++i) {
            printf("p%02d %p ", i, (void *)pl[i]);
            for (int j = 0; j < N; ++j) printf("%.3f ", rate[i][j]);
            printf("\n");
        }
        // greedy sets of five: start from each plane, add the plane whose worst pair rate with the set is best
        for (int start = 0; start < N; start += 3) {
            int set[Z] = {start};
            bool used[N] = {};
            used[start] = true;
            for (int k = 1; k < Z; ++k) {
                int best = -1;
                double bv = -1;
                for (int c = 0; c < N; ++c) {
                    if (used[c]) continue;
                    double worst = 1e9;
                    for (int q = 0; q < k; ++q) worst = rate[set[q]][c] < worst ? rate[set[q]][c] : worst;
                    if (worst > bv) { bv = worst; best = c; }
                }
                set[k] = best;
                used[best] = true;
            }
            PlanePtrs D;
            for (int z = 0; z < Z; ++z) D.d[z] = pl[set[z]];
            const double msr = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<Z, false>), dim3(items8), dim3(256), 0, 0, P, D); }, reps);
            const double msc = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<Z, true>), dim3(items8), dim3(256), 0, 0, P, D); }, reps);
            printf("chosen set {%d %d %d %d %d}: row fill %.3f   copy 5 planes/thread %.3f\n", set[0], set[1], set[2], set[3], set[4],
                   0.5 * moved / msr / 1e6 / 8000, moved / msc / 1e6 / 8000);
        }
        {   // and the worst: the first five in allocation order
            PlanePtrs D;
            for (int z = 0; z < Z; ++z) D.d[z] = pl[z];
            const double msr = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<Z, false>), dim3(items8), dim3(256), 0, 0, P, D); }, reps);
            printf("planes {0 1 2 3 4} as allocated: row fill %.3f\n", 0.5 * moved / msr / 1e6 / 8000);
        }
        return 0;
    }
    if (argc > 5 && atoi(argv[5]) == 100) {
        // PER-PLANE ALLOCATIONS: every canvas plane its own hipMalloc (2 MiB-aligned base, so the row phases inside a
        // 128-byte line are those of the single allocation), a few sets per process, with and without spacer
        // allocations of odd sizes in between
        P.src = src;
        P.dst = nullptr;
        size_t sp = 0;
        for (int set = 0; set < 8; ++set) {
            PlanePtrs D;
            for (int z = 0; z < Z; ++z) {
                if (set >= 4) {   // a spacer of a different size before every plane
                    char *spacer;
                    sp = sp * 1103515245u + 12345u;
                    CK(hipMalloc(&spacer, ((sp >> 8) % 997 + 3) << 20));
                }
                CK(hipMalloc(&D.d[z], P.dst_plane));
            }
            const double msr = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<Z, false>), dim3(items8), dim3(256), 0, 0, P, D); }, reps);
            const double msc = time_ms([&] { hipLaunchKernelGGL((k_rows_pp<Z, true>), dim3(items8), dim3(256), 0, 0, P, D); }, reps);
            printf("set %d (%s): planes at %p %p %p %p %p   row fill %.3f   copy 5 planes/thread %.3f of 8 TB/s\n", set, set >= 4 ? "spacers" : "back to back",
                   (void *)D.d[0], (void *)D.d[1], (void *)D.d[2], (void *)D.d[3], (void *)D.d[4], 0.5 * moved / msr / 1e6 / 8000, moved / msc / 1e6 / 8000);
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 5 && atoi(argv[5]) < 0) {
        // STRIDE-IN-ONE-ALLOCATION experiment: one canvas allocation with slack; the row fill (the pattern that shows the
        // effect most: 0.56 / 0.73) and kernel A against the byte distance between consecutive canvas planes
        const size_t slack = (size_t)320 << 20;
        const int nalloc = -atoi(argv[5]);
        for (int a = 0; a < nalloc; ++a) {
            char *big;
            CK(hipMalloc(&big, Z * (P.dst_plane + slack)));
            P.src = src;
            P.dst = big;
            printf("allocation %d: canvas %p\n", a, (void *)big);
            const size_t pads[] = {0, 256, 512, 1024, 4096, 16384, 65536, 65536 + 256, 1 << 20, (1 << 20) + 4352, 2 << 20, 4 << 20, 16 << 20, (16 << 20) + 65792,
                                   64 << 20, (64 << 20) + 256, 128 << 20, 256 << 20, (256 << 20) + (1 << 20) + 4352, 317 << 20};
            const size_t base_plane = P.dst_plane;
            for (size_t pad : pads) {
                P.dst_plane = base_plane + pad;
                const double msr = time_ms([&] { hipLaunchKernelGGL((k_fill_rows<Z>), dim3(items8), dim3(256), 0, 0, P); }, reps);
                const double ms = time_ms([&] { hipLaunchKernelGGL((k_regs<Z, true>), dim3(items8), dim3(256), 0, 0, P); }, reps);
                printf("  canvas plane stride %zu (+%zu): row fill %.3f   A %.3f of 8 TB/s\n", P.dst_plane, pad, 0.5 * moved / msr / 1e6 / 8000, moved / ms / 1e6 / 8000);
                fflush(stdout);
            }
            P.dst_plane = base_plane;
        }
        return 0;
    }
    if (argc > 5 && atoi(argv[5])) {
        // PLACEMENT experiment: same sizes, same kernel, fresh allocations in one process.  trial t re-allocates the canvas
        // (odd t) or the tile stacks (even t > 0); earlier allocations stay alive so that new memory is handed out
        printf("placement: kernel A on fresh allocations (sizes fixed); pointers modulo 2 MiB / 1 GiB in brackets\n");
        for (int trial = 0; trial < atoi(argv[5]); ++trial) {
            if (trial > 0 && (trial & 1)) {
                char *nd;
                CK(hipMalloc(&nd, Z * P.dst_plane));
                dst = nd;
            } else if (trial > 0) {
                char *ns;
                CK(hipMalloc(&ns, Z * P.src_plane));
                CK(hipMemcpy(ns, src, Z * P.src_plane, hipMemcpyDeviceToDevice));
                src = ns;
            }
            P.src = src;
            P.dst = dst;
            const double ms = time_ms([&] { hipLaunchKernelGGL((k_regs<Z, true>), dim3(items8), dim3(256), 0, 0, P); }, reps);
            const double msp = time_ms([&] { hipLaunchKernelGGL((k_regs<Z, false>), dim3(items8), dim3(256), 0, 0, P); }, reps);
            // one plane per launch (the per-plane kernel's pattern), five launches
            const double ms1 = time_ms([&] {
                for (int z = 0; z < Z; ++z) {
                    Geo Q = P;
                    Q.src = P.src + z * P.src_plane;
                    Q.dst = P.dst + z * P.dst_plane;
                    hipLaunchKernelGGL((k_regs<1, false>), dim3(items8), dim3(256), 0, 0, Q);
                }
            }, reps);
            const size_t nlin = Z * P.dst_plane / 16;
            const double msl = time_ms([&] { hipLaunchKernelGGL(k_linear_copy, dim3((unsigned)((nlin + 255) / 256)), dim3(256), 0, 0, (const u32x4 *)src, (u32x4 *)dst, nlin); }, reps);
            const double msf = time_ms([&] { hipLaunchKernelGGL(k_linear_fill, dim3((unsigned)((nlin + 255) / 256)), dim3(256), 0, 0, (u32x4 *)dst, nlin); }, reps);
            const double msr = time_ms([&] { hipLaunchKernelGGL((k_fill_rows<Z>), dim3(items8), dim3(256), 0, 0, P); }, reps);
            printf("trial %d (%s): canvas %p [%4zu MiB in its GiB]  A %.3f  P(5 planes/thread) %.3f  P(1 plane/launch) %.3f  row fill %.3f | linear copy %.3f  linear fill %.3f of 8 TB/s\n",
                   trial, trial == 0 ? "first" : ((trial & 1) ? "new canvas" : "new tiles"), (void *)dst, ((uintptr_t)dst & ((1u << 30) - 1)) >> 20,
                   moved / ms / 1e6 / 8000, moved / msp / 1e6 / 8000, moved / ms1 / 1e6 / 8000, 0.5 * moved / msr / 1e6 / 8000,
                   2.0 * nlin * 16 / msl / 1e6 / 8000, 1.0 * nlin * 16 / msf / 1e6 / 8000);
            fflush(stdout);
        }
        return 0;
    }
    // reference = A into `ref`
    P.dst = ref;
    hipLaunchKernelGGL((k_regs<Z, true>), dim3(items8), dim3(256), 0, 0, P);
    CK(hipDeviceSynchronize());
    P.dst = dst;
    for (int round = 0; round < (quick ? 2 : 3); ++round) {
        printf("-- round %d\n", round);
        double ms;
        ms = time_ms([&] { hipLaunchKernelGGL((k_regs<Z, false>), dim3(items8), dim3(256), 0, 0, P); }, reps);
        report("P  plain copy, 5 planes per thread (no gains)", ms, "");
        CK(hipMemset(dst, 0, Z * P.dst_plane));
        ms = time_ms([&] { hipLaunchKernelGGL((k_regs<Z, true>), dim3(items8), dim3(256), 0, 0, P); }, reps);
        report("A  regs: gains + reciprocals in VGPRs, 5 planes per thread (shipped structure)", ms, check("A"));
        CK(hipMemset(dst, 0, Z * P.dst_plane));
        ms = time_ms([&] { hipLaunchKernelGGL((k_regs_pk<Z, true>), dim3(items8), dim3(256), 0, 0, P); }, reps);
        report("A2 regs, quotient on the packed float32 pipe (v_pk_mul_f32 / v_pk_fma_f32)", ms, check("A2"));
        CK(hipMemset(dst, 0, Z * P.dst_plane));
        ms = time_ms([&] { hipLaunchKernelGGL((k_regs_rowwise<Z, true>), dim3(items8), dim3(256), 0, 0, P); }, reps);
        report("A3 regs, the 4 waves of a workgroup on ONE row at a time (wave w = slot w)", ms, check("A3"));
        ms = time_ms([&] { hipLaunchKernelGGL((k_regs_rowwise<Z, false>), dim3(items8), dim3(256), 0, 0, P); }, reps);
        report("P3 plain copy in the A3 form", ms, "");
#define RUN_B(ROWS)                                                                                                       \
    do {                                                                                                                  \
        CK(hipMemset(dst, 0, Z *P.dst_plane));                                                                            \
        const unsigned nb = (unsigned)(G * G * ((P.rows + ROWS - 1) / ROWS));                                             \
        ms = time_ms([&] { hipLaunchKernelGGL((k_lds_gains<Z, ROWS>), dim3(nb), dim3(64 * Z), 0, 0, P); }, reps);         \
        report("B  lds gains: {g, 1/g} staged in LDS per row, a wave per plane, " #ROWS " rows per workgroup", ms, check("B")); \
    } while (0)
        RUN_B(8);
        if (quick) continue;
        RUN_B(4);
        RUN_B(2);
        RUN_B(1);
        RUN_B(16);
        CK(hipMemset(dst, 0, Z * P.dst_plane));
        ms = time_ms([&] { hipLaunchKernelGGL((k_lds_dma<Z>), dim3(items8), dim3(256), 0, 0, P); }, reps);
        report("C  lds dma: structure A, pixels by global_load_lds_dwordx4 into a per-wave ring", ms, check("C"));
    }
    return 0;
}
