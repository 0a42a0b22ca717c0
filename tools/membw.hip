// membw: what this MI355X box moves per second with plain streaming kernels -- the yardstick the fusion
// kernel's rates are read against (tools/, not product code).   hipcc --offload-arch=gfx950 -O3 membw.hip -o membw
//   1. linear 16-byte-per-lane copy / fill / read, one-shot and persistent grids, nt or plain policy
//   2. the same copy dealt over Z (source, destination) pairs 2 GiB apart, visited round-robin per chunk
//      (does interleaving the streams of several planes cost anything?)
//   3. a row-segment copy shaped like config 3 (3608-byte segments, source pitch 4096, canvas pitch 58276)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);     \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define G1 __attribute__((address_space(1)))
struct __attribute__((packed)) U4U { u32x4 v; };

template <bool NT>
__device__ __forceinline__ u32x4 ld(const u32x4 *p) {
    if (NT) return __builtin_nontemporal_load((const G1 u32x4 *)p);
    return *(const G1 u32x4 *)p;
}
template <bool NT>
__device__ __forceinline__ void st(u32x4 *p, u32x4 v) {
    if (NT) __builtin_nontemporal_store(v, (G1 u32x4 *)p);
    else *(G1 u32x4 *)p = v;
}

// chunk = 256 threads x U vectors, contiguous U KiB x 4; block b takes chunks b, b + grid, ...
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_linear(const u32x4 *src, u32x4 *dst, size_t n_chunks) {
    for (size_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const size_t base = c * (256 * U) + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = ld<NTL>(src + base + k * 256);
#pragma unroll
        for (int k = 0; k < U; ++k) st<NTS>(dst + base + k * 256, v[k]);
    }
}
template <int U, bool NTS>
__global__ __launch_bounds__(256) void fill_linear(u32x4 *dst, size_t n_chunks) {
    for (size_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const size_t base = c * (256 * U) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < U; ++k) st<NTS>(dst + base + k * 256, u32x4{0, 0, 0, 0});
    }
}
template <int U>
__global__ __launch_bounds__(256) void read_linear(const u32x4 *src, size_t n_chunks, uint32_t *sink) {
    u32x4 acc{0, 0, 0, 0};
    for (size_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const size_t base = c * (256 * U) + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = ld<false>(src + base + k * 256);
#pragma unroll
        for (int k = 0; k < U; ++k) acc ^= v[k];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}
// persistent grid, every WAVE takes the next U KiB in address order from one atomic counter (what the hardware's
// own workgroup dispatcher does for a one-shot grid)
template <int U, bool NTS>
__global__ __launch_bounds__(256) void copy_dynamic(const u32x4 *src, u32x4 *dst, size_t n_chunks, unsigned *counter) {
    const int lane = threadIdx.x & 63;
    unsigned c = 0;
    if (lane == 0) c = atomicAdd(counter, 1u);
    c = __builtin_amdgcn_readfirstlane(c);
    while (c < n_chunks) {
        unsigned nxt = 0;
        if (lane == 0) nxt = atomicAdd(counter, 1u);   // asked for before this chunk is moved
        const size_t base = (size_t)c * (64 * U) + lane;
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = ld<false>(src + base + k * 64);
#pragma unroll
        for (int k = 0; k < U; ++k) st<NTS>(dst + base + k * 64, v[k]);
        c = __builtin_amdgcn_readfirstlane(nxt);
    }
}
// one-shot grid whose blocks take chunks in a scattered order: S streams, each sequential
template <int U>
__global__ __launch_bounds__(256) void copy_perm(const u32x4 *src, u32x4 *dst, size_t n_chunks, unsigned S) {
    const size_t b = blockIdx.x;
    const size_t c = (b % S) * (n_chunks / S) + b / S;
    const size_t base = c * (256 * U) + threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = ld<false>(src + base + k * 256);
#pragma unroll
    for (int k = 0; k < U; ++k) st<true>(dst + base + k * 256, v[k]);
}
// one-shot U=1 whose waves wait for their store's acknowledgement before they end
__global__ __launch_bounds__(256) void copy_oneshot_wait(const u32x4 *src, u32x4 *dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    st<true>(dst + i, ld<false>(src + i));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// persistent grid-stride whose waves wait for every store's acknowledgement at the end of each chunk
template <int U>
__global__ __launch_bounds__(256) void copy_linear_wait(const u32x4 *src, u32x4 *dst, size_t n_chunks) {
    for (size_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const size_t base = c * (256 * U) + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = ld<false>(src + base + k * 256);
#pragma unroll
        for (int k = 0; k < U; ++k) st<true>(dst + base + k * 256, v[k]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
// persistent grid, chunks of 256 x U vectors handed to WORKGROUPS in address order by one atomic counter; the
// atomic for the next chunk is issued before the current one is moved
template <int U>
__global__ __launch_bounds__(256) void copy_dynamic_wg(const u32x4 *src, u32x4 *dst, size_t n_chunks, unsigned *counter) {
    __shared__ unsigned s_c[2];
    if (threadIdx.x == 0) s_c[0] = atomicAdd(counter, 1u);
    for (int it = 0;; ++it) {
        __syncthreads();
        const unsigned c = s_c[it & 1];
        if (c >= n_chunks) break;
        unsigned nxt = 0;
        if (threadIdx.x == 0) nxt = atomicAdd(counter, 1u);
        const size_t base = (size_t)c * (256 * U) + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = ld<false>(src + base + k * 256);
#pragma unroll
        for (int k = 0; k < U; ++k) st<true>(dst + base + k * 256, v[k]);
        if (threadIdx.x == 0) s_c[(it + 1) & 1] = nxt;
    }
}
// loader / storer split: waves 0,1 only LOAD (global -> registers -> an LDS ring), waves 2,3 only STORE (ring ->
// registers -> global).  A loader's vmcnt then never covers a store, so no wave ever waits for a store to be
// acknowledged.  Pair i = (wave i, wave i + 2) has a private ring of NS 1-KiB slots and two monotonic LDS counters.
// Stream s = 2 * block + pair moves the 1-KiB units s, s + S, s + 2S, ... (S = 2 * gridDim.x).
template <int NS, int K>
__global__ __launch_bounds__(256) void copy_split(const u32x4 *src, u32x4 *dst, size_t n_units) {
    __shared__ u32x4 ring[2][NS][64];
    __shared__ volatile unsigned filled[2], drained[2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, pair = wave & 1;
    if (threadIdx.x < 2) filled[threadIdx.x] = drained[threadIdx.x] = 0;
    __syncthreads();
    const size_t S = 2 * (size_t)gridDim.x, s0 = 2 * (size_t)blockIdx.x + pair;
    const unsigned total = s0 < n_units ? (unsigned)((n_units - s0 + S - 1) / S) : 0;
    if (wave < 2) {
        u32x4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k)
            if ((unsigned)k < total) v[k] = ld<false>(src + (s0 + k * S) * 64 + lane);
        for (unsigned j = 0; j < total; j += K) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const unsigned jj = j + k;
                if (jj < total) {
                    while (jj - drained[pair] >= (unsigned)NS) __builtin_amdgcn_s_sleep(1);
                    ring[pair][jj % NS][lane] = v[k];
                    asm volatile("" ::: "memory");
                    if (lane == 0) filled[pair] = jj + 1;   // LDS operations of one wave execute in order
                    if (jj + K < total) v[k] = ld<false>(src + (s0 + (size_t)(jj + K) * S) * 64 + lane);
                }
            }
        }
    } else {
        for (unsigned j = 0; j < total; ++j) {
            while (filled[pair] <= j) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            const u32x4 x = ring[pair][j % NS][lane];
            asm volatile("" ::: "memory");
            if (lane == 0) drained[pair] = j + 1;
            st<true>(dst + (s0 + (size_t)j * S) * 64 + lane, x);
        }
    }
}
// Z streams `zstride` vectors apart; a block copies chunk c of stream 0, 1, ..., Z-1 before moving on
template <int U, int Z>
__global__ __launch_bounds__(256) void copy_zinter(const u32x4 *src, u32x4 *dst, size_t n_chunks, size_t zstride) {
    for (size_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const size_t base = c * (256 * U) + threadIdx.x;
        u32x4 v[Z][U];
#pragma unroll
        for (int z = 0; z < Z; ++z)
#pragma unroll
            for (int k = 0; k < U; ++k) v[z][k] = ld<false>(src + z * zstride + base + k * 256);
#pragma unroll
        for (int z = 0; z < Z; ++z)
#pragma unroll
            for (int k = 0; k < U; ++k) st<true>(dst + z * zstride + base + k * 256, v[z][k]);
    }
}

// config-3-like row segments: grid of GxG tiles (T x T uint16, dense), each contributing rows [CR, T-CR) x
// columns [CR, T-CR) to a canvas of pitch `pitch` bytes; item = 8 rows of one tile, wave w takes rows w, w+4;
// 16-byte vectors at whatever phase both sides have (ALIGN 0) or destination-aligned to 16 B (ALIGN 1).
// order 0: tile-major items; order 1: row-block-major (all tiles' block b, then b+1): the product's order
template <int ALIGN, int Z>
__global__ __launch_bounds__(256) void copy_rows(const char *src, char *dst, int G, int T, int CR, size_t pitch, int order,
                                                 size_t n_items, size_t src_z, size_t dst_z, unsigned *counter) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rows = T - 2 * CR, seg = rows * 2;   // bytes per row segment
    const int nblk = (rows + 7) / 8;
    __shared__ unsigned s_next;
    for (size_t it = blockIdx.x; it < n_items;) {
        if (counter) {   // dynamic: items in list order from one atomic counter
            __syncthreads();
            if (threadIdx.x == 0) s_next = atomicAdd(counter, 1u);
            __syncthreads();
            it = s_next;
            if (it >= n_items) break;
        }
        int tile, blk;
        if (order == 0) { tile = (int)(it / nblk); blk = (int)(it % nblk); }
        else if (order == 1) { blk = (int)(it / (G * G)); tile = (int)(it % (G * G)); }
        else { const int tx = (int)(it % G); blk = (int)((it / G) % nblk); tile = (int)(it / G / nblk) * G + tx; }   // canvas raster
        const int ty = tile / G, tx = tile % G;
        for (int r = blk * 8 + wave; r < min(rows, blk * 8 + 8); r += 4) {
            const char *s = src + (size_t)tile * T * T * 2 + (size_t)(CR + r) * T * 2 + CR * 2;
            char *d = dst + ((size_t)ty * rows + r) * pitch + (size_t)tx * seg;
            const int mis = ALIGN ? (int)((16 - ((uintptr_t)d & 15)) & 15) : 0;   // head bytes before the aligned body
            const int nvec = (seg - mis) / 16;
            u32x4 v[Z][4];
#pragma unroll
            for (int z = 0; z < Z; ++z)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = lane + 64 * k;
                    if (i < nvec) v[z][k] = ((const G1 U4U *)(s + z * src_z + mis + i * 16))->v;
                }
#pragma unroll
            for (int z = 0; z < Z; ++z)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = lane + 64 * k;
                    if (i < nvec) {
                        if (ALIGN) __builtin_nontemporal_store(v[z][k], (G1 u32x4 *)(d + z * dst_z + mis + i * 16));
                        else ((G1 U4U *)(d + z * dst_z + mis + i * 16))->v = v[z][k];
                    }
                }
            // head and tail, 2 bytes per lane
            const int tail0 = mis + nvec * 16;
#pragma unroll
            for (int z = 0; z < Z; ++z) {
                if (lane * 2 < mis) *(G1 uint16_t *)(d + z * dst_z + lane * 2) = *(const G1 uint16_t *)(s + z * src_z + lane * 2);
                if (tail0 + lane * 2 < seg)
                    *(G1 uint16_t *)(d + z * dst_z + tail0 + lane * 2) = *(const G1 uint16_t *)(s + z * src_z + tail0 + lane * 2);
            }
        }
        if (!counter) it += gridDim.x;
    }
}


// Decomposition of the row-segment pattern: tiles of `rows` row segments of S bytes (S % 16 == 0) at src_pitch,
// first byte src_off into a tile of src_tile bytes; tile (ty, tx) lands at dst + ty*dst_ty + tx*dst_tx, rows at
// dst_pitch.  One workgroup per (tile, block of 8 rows), one-shot; wave w copies rows w and w+4 with 16-byte
// vectors (plain loads, nt stores).  Every address is 16-byte aligned; whether it is 128-byte aligned is the
// experiment.
template <int line_slots>
__global__ __launch_bounds__(256) void copy2d(const char *src, char *dst, int G, int rows, int S, size_t src_tile, size_t src_pitch,
                                              size_t src_off, size_t dst_pitch, size_t dst_ty, size_t dst_tx, int rpw) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = 4 * rpw, nblk = (rows + per - 1) / per;
    const size_t it = blockIdx.x;
    const int tile = (int)(it / nblk), blk = (int)(it % nblk);
    const int ty = tile / G, tx = tile % G;
    const int nvec = S / 16;
    for (int j = 0; j < rpw; ++j) {
        const int r = blk * per + wave + 4 * j;
        if (r >= rows) break;
        const char *s = src + (size_t)tile * src_tile + src_off + (size_t)r * src_pitch;
        char *d = dst + (size_t)ty * dst_ty + (size_t)tx * dst_tx + (size_t)r * dst_pitch;
        // line_slots 1: lane 0 of every store instruction sits on a 128-byte line of the DESTINATION (lanes before the
        // segment masked off), so no instruction straddles a line; 0: lane 0 starts at the segment.
        // line_slots 2: as 1, and every line has ONE writer: a segment owns the line its first byte falls in (the bytes
        // before it come from the left neighbour's source row) and leaves the line its end falls in to the right neighbour
        if (line_slots == 2) {
            const long x0 = 0, x1 = S;   // relative to d
            const long a = tx == 0 ? x0 : -(long)((uintptr_t)d & 127);
            const long b = tx == G - 1 ? x1 : x1 - (long)((uintptr_t)(d + x1) & 127);
            const char *sl = s - (long)src_tile + S;   // left neighbour's row: its byte S + q pairs with our q < 0
            const int shift = tx == 0 ? (int)(((uintptr_t)d & 127) >> 4) : 0;
            u32x4 v[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const long q = a + ((long)lane + 64 * k - shift) * 16;
                if (q >= a && q < b) v[k] = ((const G1 U4U *)((q >= 0 ? s : sl) + q))->v;
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const long q = a + ((long)lane + 64 * k - shift) * 16;
                if (q >= a && q < b) __builtin_nontemporal_store(v[k], (G1 u32x4 *)(d + q));
            }
            continue;
        }
        const int shift = line_slots ? (int)(((uintptr_t)d & 127) >> 4) : 0;
        u32x4 v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = lane + 64 * k - shift;
            if (i >= 0 && i < nvec) v[k] = ((const G1 U4U *)(s + (long)i * 16))->v;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = lane + 64 * k - shift;
            if (i >= 0 && i < nvec) __builtin_nontemporal_store(v[k], (G1 u32x4 *)(d + (long)i * 16));
        }
    }
}

// the same tiles, the other way round: the four waves of a workgroup share ONE row at a time (wave w copies the w-th
// KiB of the segment, one vector per lane) and walk down the block's rows together -- a workgroup then touches 4 KiB
// of one row at once instead of 1 KiB of four rows
template <int ROWS_PER_WG, int Z>
__global__ __launch_bounds__(256) void copy2d_rowwise(const char *src, char *dst, int G, int rows, int S, size_t src_tile, size_t src_pitch,
                                                      size_t src_off, size_t dst_pitch, size_t dst_ty, size_t dst_tx, size_t src_z, size_t dst_z) {
    const int nblk = (rows + ROWS_PER_WG - 1) / ROWS_PER_WG;
    const size_t it = blockIdx.x;
    const int tile = (int)(it / nblk), blk = (int)(it % nblk);
    const int ty = tile / G, tx = tile % G;
    const int nvec = S / 16;
    u32x4 v[ROWS_PER_WG][Z];
    const int r0 = blk * ROWS_PER_WG;
    const char *s = src + (size_t)tile * src_tile + src_off + (size_t)r0 * src_pitch;
    char *d = dst + (size_t)ty * dst_ty + (size_t)tx * dst_tx + (size_t)r0 * dst_pitch;
#pragma unroll
    for (int j = 0; j < ROWS_PER_WG; ++j) {
        const int shift = (int)(((uintptr_t)(d + j * dst_pitch) & 127) >> 4);
        const int i = (int)threadIdx.x - shift;
        if (r0 + j < rows && i >= 0 && i < nvec)
#pragma unroll
            for (int z = 0; z < Z; ++z) v[j][z] = ((const G1 U4U *)(s + z * src_z + j * src_pitch + (long)i * 16))->v;
    }
#pragma unroll
    for (int j = 0; j < ROWS_PER_WG; ++j) {
        const int shift = (int)(((uintptr_t)(d + j * dst_pitch) & 127) >> 4);
        const int i = (int)threadIdx.x - shift;
        if (r0 + j < rows && i >= 0 && i < nvec)
#pragma unroll
            for (int z = 0; z < Z; ++z) __builtin_nontemporal_store(v[j][z], (G1 u32x4 *)(d + z * dst_z + j * dst_pitch + (long)i * 16));
    }
}

// the row-wise workgroup again, R rows per workgroup one after the other, but with at most DEPTH loads of a thread in
// flight: load row j + DEPTH, then wait for row j (vmcnt in issue order) and store it
template <int R, int DEPTH>
__global__ __launch_bounds__(256) void copy2d_rowwise_depth(const char *src, char *dst, int G, int rows, int S, size_t src_tile, size_t src_pitch,
                                                            size_t src_off, size_t dst_pitch, size_t dst_ty, size_t dst_tx) {
    const int nblk = (rows + R - 1) / R;
    const size_t it = blockIdx.x;
    const int tile = (int)(it / nblk), blk = (int)(it % nblk);
    const int ty = tile / G, tx = tile % G;
    const int nvec = S / 16;
    const int r0 = blk * R;
    const char *s = src + (size_t)tile * src_tile + src_off + (size_t)r0 * src_pitch;
    char *d = dst + (size_t)ty * dst_ty + (size_t)tx * dst_tx + (size_t)r0 * dst_pitch;
    u32x4 v[DEPTH];
    auto ld = [&](int j) {
        const int shift = (int)(((uintptr_t)(d + j * dst_pitch) & 127) >> 4);
        const int i = min(max((int)threadIdx.x - shift, 0), nvec - 1);
        return ((const G1 U4U *)(s + j * src_pitch + (long)i * 16))->v;
    };
#pragma unroll
    for (int j = 0; j < DEPTH && j < R; ++j) v[j] = ld(min(j, rows - 1 - r0));
#pragma unroll
    for (int j = 0; j < R; ++j) {
        u32x4 cur = v[j % DEPTH];
        // wait until only the younger DEPTH - 1 loads are outstanding (stores count too: they are older or this row's)
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur));
        const int shift = (int)(((uintptr_t)(d + j * dst_pitch) & 127) >> 4);
        const int i = (int)threadIdx.x - shift;
        if (r0 + j < rows && i >= 0 && i < nvec) __builtin_nontemporal_store(cur, (G1 u32x4 *)(d + j * dst_pitch + (long)i * 16));
        if (j + DEPTH < R) v[j % DEPTH] = ld(min(j + DEPTH, rows - 1 - r0));
    }
}

template <typename F>
static double time_ms(F launch, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}

int main(int argc, char **argv) {
    const size_t GiB = size_t(1) << 30;
    const size_t bytes = (argc > 1 ? atoi(argv[1]) : 8) * GiB;   // per buffer
    char *src, *dst;
    uint32_t *sink;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&dst, bytes + (1 << 20)));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(src, 3, bytes));
    CK(hipMemset(dst, 1, bytes));
    const size_t nvec = bytes / 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, buffers 2 x %zu GiB\n", prop.name, cus, bytes / GiB);
    auto report = [&](const char *name, double ms, double moved) {
        printf("%-64s %8.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, ms, moved / ms / 1e6, moved / ms / 1e6 / 8000);
        fflush(stdout);
    };
    char name[256];

    if (argc > 2 && !strcmp(argv[2], "2d")) {
        // one plane = 16 x 16 tiles; each case copies G*G*rows*S bytes
        const int G = 16, T = 2048;
        struct Case { const char *what; int rows, S; size_t src_tile, src_pitch, src_off; int dst_kind; int rpw; int line_slots = 0; };
        // dst_kind 0: canvas (pitch G*S rounded per `pad`), 1: canvas with the unpadded 58276-like pitch (G*S + 560),
        //          2: packed tile-major (a linear destination), 3: canvas pitch rounded up to 128
        const Case cases[] = {
            {"S=4096 rows contiguous -> packed (a linear copy through this kernel)", 2048, 4096, (size_t)T * T * 2, 4096, 0, 2, 2},
            {"S=4096 whole tiles -> canvas pitch 65536 (all lines whole)", 2048, 4096, (size_t)T * T * 2, 4096, 0, 0, 2},
            {"S=3584 src off 256 -> canvas pitch 57344 (all lines whole)", 1792, 3584, (size_t)T * T * 2, 4096, 128 * 4096 + 256, 0, 2},
            {"S=3584 src off 256 -> packed", 1792, 3584, (size_t)T * T * 2, 4096, 128 * 4096 + 256, 2, 2},
            {"S=3584 src packed -> canvas pitch 57344", 1792, 3584, (size_t)1792 * 3584, 3584, 0, 0, 2},
            {"S=3584 src off 256 -> canvas pitch 57344+560 (rows at every 16-byte phase)", 1792, 3584, (size_t)T * T * 2, 4096, 128 * 4096 + 256, 1, 2},
            {"S=3600 src off 240 -> canvas pitch 57600 (tiles at 16-byte phases, rows too)", 1800, 3600, (size_t)T * T * 2, 4096, 124 * 4096 + 240, 0, 2},
            {"S=3600 src off 240 -> canvas pitch 57728 (128-multiple; tiles at 16-byte phases)", 1800, 3600, (size_t)T * T * 2, 4096, 124 * 4096 + 240, 3, 2},
            {"S=3600 src off 240 -> packed", 1800, 3600, (size_t)T * T * 2, 4096, 124 * 4096 + 240, 2, 2},
            {"S=3584 src off 256 -> canvas pitch 57344, 4 rows per wave", 1792, 3584, (size_t)T * T * 2, 4096, 128 * 4096 + 256, 0, 4},
            {"S=3584 src off 256 -> canvas pitch 57344, 1 row per wave", 1792, 3584, (size_t)T * T * 2, 4096, 128 * 4096 + 256, 0, 1},
            {"S=3584 src off 240 (16-byte phase) -> canvas pitch 57344 (dst lines whole)", 1792, 3584, (size_t)T * T * 2, 4096, 128 * 4096 + 240, 0, 2},
            {"S=3584 src off 242 (2-byte phase) -> canvas pitch 57344 (dst lines whole)", 1792, 3584, (size_t)T * T * 2, 4096, 128 * 4096 + 242, 0, 2},
            {"S=3584 src off 256 -> canvas pitch 57344+560, LINE SLOTS", 1792, 3584, (size_t)T * T * 2, 4096, 128 * 4096 + 256, 1, 2, 1},
            {"S=3600 src off 240 -> canvas pitch 57728, LINE SLOTS", 1800, 3600, (size_t)T * T * 2, 4096, 124 * 4096 + 240, 3, 2, 1},
            {"S=3600 src off 242 -> canvas pitch 57600+560, LINE SLOTS", 1800, 3600, (size_t)T * T * 2, 4096, 124 * 4096 + 242, 1, 2, 1},
            {"S=3600 src off 242 -> canvas pitch 57600+560", 1800, 3600, (size_t)T * T * 2, 4096, 124 * 4096 + 242, 1, 2, 0},
            {"S=3584 src off 256 -> canvas pitch 57344 (all lines whole), LINE SLOTS (control)", 1792, 3584, (size_t)T * T * 2, 4096, 128 * 4096 + 256, 0, 2, 1},
            {"S=3584 src off 256 -> canvas pitch 57344+560, ONE WRITER PER LINE", 1792, 3584, (size_t)T * T * 2, 4096, 128 * 4096 + 256, 1, 2, 2},
            {"S=3600 src off 240 -> canvas pitch 57728, ONE WRITER PER LINE", 1800, 3600, (size_t)T * T * 2, 4096, 124 * 4096 + 240, 3, 2, 2},
            {"S=3600 src off 242 -> canvas pitch 57600+560, ONE WRITER PER LINE", 1800, 3600, (size_t)T * T * 2, 4096, 124 * 4096 + 242, 1, 2, 2},
            {"S=3600 src off 242 -> canvas pitch 57600+560, ONE WRITER PER LINE, 1 row per wave", 1800, 3600, (size_t)T * T * 2, 4096, 124 * 4096 + 242, 1, 1, 2},
            {"S=3600 src off 242 -> canvas pitch 57600+560, LINE SLOTS, 1 row per wave", 1800, 3600, (size_t)T * T * 2, 4096, 124 * 4096 + 242, 1, 1, 1},
            {"S=2048 half rows -> canvas pitch 32768", 2048, 2048, (size_t)T * T * 2, 4096, 0, 0, 2},
            {"S=1024 quarter rows -> canvas pitch 16384", 2048, 1024, (size_t)T * T * 2, 4096, 0, 0, 2},
        };
        for (const Case &c : cases) {
            size_t dst_pitch, dst_ty, dst_tx;
            if (c.dst_kind == 2) { dst_pitch = c.S; dst_tx = (size_t)c.rows * c.S; dst_ty = (size_t)G * dst_tx; }
            else {
                dst_pitch = (size_t)G * c.S + (c.dst_kind == 1 ? 560 : 0);
                if (c.dst_kind == 3) dst_pitch = (dst_pitch + 127) / 128 * 128;
                dst_tx = c.S; dst_ty = (size_t)c.rows * dst_pitch;
            }
            const size_t plane_src = (size_t)G * G * c.src_tile, plane_dst = ((size_t)G * c.rows * (c.dst_kind == 2 ? (size_t)G * c.S : dst_pitch) + 4095) / 4096 * 4096;
            const int nplanes = (int)std::min(bytes / plane_src, bytes / plane_dst);
            const int per = 4 * c.rpw;
            const size_t n_items = (size_t)G * G * ((c.rows + per - 1) / per);
            double ms = time_ms([&] {
                for (int p = 0; p < nplanes; ++p)
                {
                    auto k = c.line_slots == 2 ? copy2d<2> : (c.line_slots == 1 ? copy2d<1> : copy2d<0>);
                    hipLaunchKernelGGL(k, dim3((unsigned)n_items), dim3(256), 0, 0, src + p * plane_src, dst + p * plane_dst, G, c.rows, c.S,
                                       c.src_tile, c.src_pitch, c.src_off, dst_pitch, dst_ty, dst_tx, c.rpw);
                }
            });
            snprintf(name, sizeof name, "2d %s, %d planes", c.what, nplanes);
            printf("%-100s %8.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, ms, 2.0 * G * G * c.rows * c.S * nplanes / ms / 1e6, 2.0 * G * G * c.rows * c.S * nplanes / ms / 1e6 / 8000);
            fflush(stdout);
        }

        // row-wise workgroups (S <= 4096 - 128: 256 lanes cover the segment at any phase)
        {
            struct RC { const char *what; int rows, S; size_t src_off; int dst_kind; };
            const RC rcs[] = {
                {"S=3584 src off 256 -> canvas pitch 57344 (all lines whole)", 1792, 3584, 128 * 4096 + 256, 0},
                {"S=3584 src off 256 -> canvas pitch 57344+560 (line slots)", 1792, 3584, 128 * 4096 + 256, 1},
                {"S=3600 src off 242 -> canvas pitch 57600+560 (line slots)", 1800, 3600, 124 * 4096 + 242, 1},
            };
            for (const RC &c : rcs) {
                const size_t src_tile = (size_t)T * T * 2;
                size_t dst_pitch = (size_t)G * c.S + (c.dst_kind == 1 ? 560 : 0), dst_tx = c.S, dst_ty = (size_t)c.rows * dst_pitch;
                const size_t plane_src = (size_t)G * G * src_tile, plane_dst = ((size_t)G * c.rows * dst_pitch + 4095) / 4096 * 4096;
                const int nplanes = (int)std::min(bytes / plane_src, bytes / plane_dst);
#define RW(N, Z)                                                                                                                 \
    do {                                                                                                                         \
        const size_t n_items = (size_t)G * G * ((c.rows + N - 1) / N);                                                           \
        double ms = time_ms([&] {                                                                                                \
            for (int p = 0; p + Z <= nplanes; p += Z)                                                                            \
                hipLaunchKernelGGL((copy2d_rowwise<N, Z>), dim3((unsigned)n_items), dim3(256), 0, 0, src + p * plane_src, dst + p * plane_dst, G, \
                                   c.rows, c.S, src_tile, (size_t)4096, c.src_off, dst_pitch, dst_ty, dst_tx, plane_src, plane_dst); \
        });                                                                                                                      \
        snprintf(name, sizeof name, "2d ROW-WISE WG, %d rows per workgroup, %d planes per thread: %s, %d planes", N, Z, c.what, nplanes / Z * Z); \
        printf("%-125s %8.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, ms, 2.0 * G * G * c.rows * c.S * (nplanes / Z * Z) / ms / 1e6, \
               2.0 * G * G * c.rows * c.S * (nplanes / Z * Z) / ms / 1e6 / 8000);                                                \
        fflush(stdout);                                                                                                          \
    } while (0)
#define RD(N, D)                                                                                                                 \
    do {                                                                                                                         \
        const size_t n_items = (size_t)G * G * ((c.rows + N - 1) / N);                                                           \
        double ms = time_ms([&] {                                                                                                \
            for (int p = 0; p < nplanes; ++p)                                                                                    \
                hipLaunchKernelGGL((copy2d_rowwise_depth<N, D>), dim3((unsigned)n_items), dim3(256), 0, 0, src + p * plane_src, dst + p * plane_dst, G, \
                                   c.rows, c.S, src_tile, (size_t)4096, c.src_off, dst_pitch, dst_ty, dst_tx);                  \
        });                                                                                                                      \
        snprintf(name, sizeof name, "2d ROW-WISE WG, %d rows one after the other, <= %d loads in flight per thread: %s, %d planes", N, D, c.what, nplanes); \
        printf("%-125s %8.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, ms, 2.0 * G * G * c.rows * c.S * nplanes / ms / 1e6,    \
               2.0 * G * G * c.rows * c.S * nplanes / ms / 1e6 / 8000);                                                          \
        fflush(stdout);                                                                                                          \
    } while (0)
                RD(4, 1);
                RD(4, 2);
                RD(8, 1);
                RD(8, 2);
                RD(16, 2);
                RD(16, 4);
                RD(64, 2);
                RD(64, 4);
                RW(1, 1);
                RW(2, 1);
                RW(4, 1);
                RW(1, 2);
                RW(1, 4);
                RW(2, 4);
            }
        }
        return 0;
    }
#define COPY(U, NTL, NTS, PER_CU)                                                                                   \
    do {                                                                                                            \
        const size_t nch = nvec / (256 * U);                                                                        \
        const unsigned grid = PER_CU ? (unsigned)(cus * PER_CU) : (unsigned)nch;                                    \
        double ms = time_ms([&] { hipLaunchKernelGGL((copy_linear<U, NTL, NTS>), dim3(grid), dim3(256), 0, 0, (const u32x4 *)src, (u32x4 *)dst, nch); }); \
        snprintf(name, sizeof name, "copy U=%d loads %s stores %s grid %s", U, NTL ? "nt" : "plain", NTS ? "nt" : "plain", PER_CU ? #PER_CU "/CU persistent" : "one-shot"); \
        report(name, ms, 2.0 * bytes);                                                                              \
    } while (0)
    COPY(1, false, false, 0);
    COPY(1, false, true, 0);
    COPY(4, false, false, 0);
    COPY(4, false, true, 0);
    COPY(4, true, true, 0);
    COPY(8, false, true, 0);
    COPY(4, false, true, 8);
    COPY(4, false, true, 4);
    COPY(8, false, true, 8);
    COPY(8, false, false, 8);
    COPY(16, false, true, 4);
    unsigned *counter;
    CK(hipMalloc(&counter, 4));
#define DYN(U, PER_CU)                                                                                              \
    do {                                                                                                            \
        const size_t nch = nvec / (64 * U);                                                                         \
        double ms = time_ms([&] { CK(hipMemsetAsync(counter, 0, 4, 0)); hipLaunchKernelGGL((copy_dynamic<U, true>), dim3(cus * PER_CU), dim3(256), 0, 0, (const u32x4 *)src, (u32x4 *)dst, nch, counter); }); \
        snprintf(name, sizeof name, "copy U=%d per WAVE from one atomic counter, %d/CU persistent", U, PER_CU);     \
        report(name, ms, 2.0 * bytes);                                                                              \
    } while (0)
#define PERM(U, S)                                                                                                  \
    do {                                                                                                            \
        const size_t nch = nvec / (256 * U);                                                                        \
        double ms = time_ms([&] { hipLaunchKernelGGL((copy_perm<U>), dim3((unsigned)nch), dim3(256), 0, 0, (const u32x4 *)src, (u32x4 *)dst, nch, S); }); \
        snprintf(name, sizeof name, "copy U=%d one-shot, blocks dealt over %d sequential streams", U, S);           \
        report(name, ms, 2.0 * bytes);                                                                              \
    } while (0)
    {
        double ms = time_ms([&] { hipLaunchKernelGGL(copy_oneshot_wait, dim3((unsigned)(nvec / 256)), dim3(256), 0, 0, (const u32x4 *)src, (u32x4 *)dst); });
        report("copy U=1 one-shot, vmcnt(0) before the wave ends", ms, 2.0 * bytes);
        ms = time_ms([&] { hipLaunchKernelGGL((copy_linear_wait<4>), dim3(cus * 8), dim3(256), 0, 0, (const u32x4 *)src, (u32x4 *)dst, nvec / 1024); });
        report("copy U=4 8/CU persistent, vmcnt(0) after every chunk", ms, 2.0 * bytes);
        ms = time_ms([&] { hipLaunchKernelGGL((copy_linear_wait<8>), dim3(cus * 8), dim3(256), 0, 0, (const u32x4 *)src, (u32x4 *)dst, nvec / 2048); });
        report("copy U=8 8/CU persistent, vmcnt(0) after every chunk", ms, 2.0 * bytes);
    }
#define DYNWG(U, PER_CU)                                                                                            \
    do {                                                                                                            \
        const size_t nch = nvec / (256 * U);                                                                        \
        double ms = time_ms([&] { CK(hipMemsetAsync(counter, 0, 4, 0)); hipLaunchKernelGGL((copy_dynamic_wg<U>), dim3(cus * PER_CU), dim3(256), 0, 0, (const u32x4 *)src, (u32x4 *)dst, nch, counter); }); \
        snprintf(name, sizeof name, "copy U=%d per WORKGROUP from one atomic counter, %d/CU persistent", U, PER_CU); \
        report(name, ms, 2.0 * bytes);                                                                              \
    } while (0)
    DYNWG(4, 8);
    DYNWG(8, 8);
    DYNWG(16, 4);
#define SPLIT(NS, K, PER_CU)                                                                                        \
    do {                                                                                                            \
        double ms = time_ms([&] { hipLaunchKernelGGL((copy_split<NS, K>), dim3(cus * PER_CU), dim3(256), 0, 0, (const u32x4 *)src, (u32x4 *)dst, nvec / 64); }); \
        snprintf(name, sizeof name, "copy loader/storer waves, ring %d x 1 KiB, %d loads in flight, %d/CU persistent", NS, K, PER_CU); \
        report(name, ms, 2.0 * bytes);                                                                              \
    } while (0)
    SPLIT(16, 4, 4);
    SPLIT(16, 4, 5);
    SPLIT(16, 8, 4);
    SPLIT(8, 4, 8);
    SPLIT(8, 2, 8);
    SPLIT(8, 8, 8);
    SPLIT(4, 4, 8);
    PERM(1, 1);
    PERM(1, 16);
    PERM(1, 256);
    PERM(1, 4096);
    PERM(1, 65536);
    PERM(4, 16);
    PERM(4, 256);
    {
        const size_t nch = nvec / (256 * 4);
        double ms = time_ms([&] { hipLaunchKernelGGL((fill_linear<4, true>), dim3(cus * 8), dim3(256), 0, 0, (u32x4 *)dst, nch); });
        report("fill U=4 nt, 8/CU persistent", ms, 1.0 * bytes);
        ms = time_ms([&] { hipLaunchKernelGGL((fill_linear<4, false>), dim3(cus * 8), dim3(256), 0, 0, (u32x4 *)dst, nch); });
        report("fill U=4 plain, 8/CU persistent", ms, 1.0 * bytes);
        ms = time_ms([&] { hipLaunchKernelGGL((fill_linear<4, true>), dim3((unsigned)nch), dim3(256), 0, 0, (u32x4 *)dst, nch); });
        report("fill U=4 nt, one-shot", ms, 1.0 * bytes);
        ms = time_ms([&] { hipLaunchKernelGGL((read_linear<4>), dim3(cus * 8), dim3(256), 0, 0, (const u32x4 *)src, nch, sink); });
        report("read U=4, 8/CU persistent", ms, 1.0 * bytes);
        ms = time_ms([&] { hipLaunchKernelGGL((read_linear<8>), dim3((unsigned)(nvec / 2048)), dim3(256), 0, 0, (const u32x4 *)src, nvec / 2048, sink); });
        report("read U=8, one-shot", ms, 1.0 * bytes);
        ms = time_ms([&] { CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)); });
        report("hipMemcpyAsync D2D", ms, 2.0 * bytes);
        ms = time_ms([&] { CK(hipMemsetAsync(dst, 0, bytes, 0)); });
        report("hipMemsetAsync", ms, 1.0 * bytes);
    }
#define ZINT(U, Z)                                                                                                   \
    do {                                                                                                             \
        const size_t zs = nvec / Z, nch = zs / (256 * U);                                                            \
        double ms = time_ms([&] { hipLaunchKernelGGL((copy_zinter<U, Z>), dim3(cus * 8), dim3(256), 0, 0, (const u32x4 *)src, (u32x4 *)dst, nch, zs); }); \
        snprintf(name, sizeof name, "copy of %d interleaved streams %.2f GiB apart, U=%d, 8/CU persistent", Z, zs * 16.0 / GiB, U); \
        report(name, ms, 2.0 * nch * 256 * U * 16 * Z);                                                              \
    } while (0)
    ZINT(4, 1);
    ZINT(4, 2);
    ZINT(4, 4);
    ZINT(2, 8);
    ZINT(1, 8);
    ZINT(1, 4);
    // rows: one plane = 16x16 tiles of 2048^2 (2 GiB) -> canvas 28864 rows; Z planes when the buffers hold them
    {
        const int G = 16, T = 2048, CR = 122;
        const int rows = T - 2 * CR;
        const size_t plane_src = (size_t)G * G * T * T * 2;
        for (int pitch_kind = 0; pitch_kind < 2; ++pitch_kind) {
            const size_t pitch = pitch_kind == 0 ? (size_t)29138 * 2 : (((size_t)G * rows * 2 + 127) / 128) * 128;
            const size_t plane_dst = (((size_t)G * rows * pitch) + 127) / 128 * 128;
            const int nplanes = (int)std::min(bytes / plane_src, bytes / plane_dst);
            const size_t n_items = (size_t)G * G * ((rows + 7) / 8);
            const double moved = 2.0 * G * G * (double)rows * rows * 2;
            // MODE 0: persistent 8/CU grid-stride; 1: one-shot (a block per item); 2: persistent, items from one atomic counter
#define ROWS(ALIGN, Z, ORDER, MODE)                                                                                  \
    do {                                                                                                             \
        if (nplanes >= Z) {                                                                                          \
            double ms = time_ms([&] {                                                                                \
                for (int p = 0; p + Z <= nplanes; p += Z) {                                                          \
                    if (MODE == 2) CK(hipMemsetAsync(counter, 0, 4, 0));                                             \
                    hipLaunchKernelGGL((copy_rows<ALIGN, Z>), dim3(MODE == 1 ? (unsigned)n_items : cus * 8), dim3(256), 0, 0, src + p * plane_src, dst + p * plane_dst, G, T, CR, pitch, ORDER, n_items, plane_src, plane_dst, MODE == 2 ? counter : nullptr); \
                }                                                                                                    \
            });                                                                                                      \
            snprintf(name, sizeof name, "rows 3608 B, pitch %zu, align %d, Z=%d, order %d, mode %d, %d planes", pitch, ALIGN, Z, ORDER, MODE, nplanes / Z * Z); \
            report(name, ms, moved * (nplanes / Z * Z));                                                             \
        }                                                                                                            \
    } while (0)
            ROWS(0, 1, 0, 0);
            ROWS(0, 1, 0, 1);
            ROWS(0, 1, 0, 2);
            ROWS(0, 1, 1, 0);
            ROWS(0, 1, 1, 1);
            ROWS(0, 1, 1, 2);
            ROWS(0, 1, 2, 0);
            ROWS(0, 1, 2, 1);
            ROWS(0, 1, 2, 2);
            ROWS(1, 1, 2, 1);
            ROWS(1, 1, 2, 2);
            ROWS(0, 2, 2, 1);
            ROWS(0, 2, 2, 2);
            ROWS(0, 3, 2, 2);
        }
    }
    return 0;
}
