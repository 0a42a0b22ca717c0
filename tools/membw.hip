// membw: what this MI355X box moves per second with plain streaming kernels -- the yardstick the fusion
// kernel's rates are read against (tools/, not product code).   hipcc --offload-arch=gfx950 -O3 membw.hip -o membw
//   1. linear 16-byte-per-lane copy / fill / read, one-shot and persistent grids, nt or plain policy
//   2. the same copy dealt over Z (source, destination) pairs 2 GiB apart, visited round-robin per chunk
//      (does interleaving the streams of several planes cost anything?)
//   3. a row-segment copy shaped like config 3 (3608-byte segments, source pitch 4096, canvas pitch 58276)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
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
                                                 size_t n_items, size_t src_z, size_t dst_z) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rows = T - 2 * CR, seg = rows * 2;   // bytes per row segment
    const int nblk = (rows + 7) / 8;
    for (size_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        int tile, blk;
        if (order == 0) { tile = (int)(it / nblk); blk = (int)(it % nblk); }
        else { blk = (int)(it / (G * G)); tile = (int)(it % (G * G)); }
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
    char name[160];
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
#define ROWS(ALIGN, Z, ORDER)                                                                                        \
    do {                                                                                                             \
        if (nplanes >= Z) {                                                                                          \
            double ms = time_ms([&] {                                                                                \
                for (int p = 0; p + Z <= nplanes; p += Z)                                                            \
                    hipLaunchKernelGGL((copy_rows<ALIGN, Z>), dim3(cus * 8), dim3(256), 0, 0, src + p * plane_src, dst + p * plane_dst, G, T, CR, pitch, ORDER, n_items, plane_src, plane_dst); \
            });                                                                                                      \
            snprintf(name, sizeof name, "rows 3608 B, pitch %zu, align %d, Z=%d, order %d, %d planes", pitch, ALIGN, Z, ORDER, nplanes / Z * Z); \
            report(name, ms, moved * (nplanes / Z * Z));                                                             \
        }                                                                                                            \
    } while (0)
            ROWS(0, 1, 0);
            ROWS(1, 1, 0);
            ROWS(1, 1, 1);
            ROWS(1, 2, 1);
            ROWS(1, 3, 1);
        }
    }
    return 0;
}
