// sq_arena: a device-memory arena whose every stretch lies over ALL of the card's memory classes.
//
// Why.  On MI355X (288 GiB of HBM3E in 12-high stacks, memory partition mode NPS1) device memory falls into three
// classes of a third of the card each (profiles/r04_exp_placement_classes.log: every GiB of a 256 GiB allocation classified
// -> 77 / 89 / 84 GiB).  A write stream made of short row segments at a large pitch -- what fusing cropped tiles into a canvas
// is -- runs at 0.55 of the 8 TB/s peak while everything it writes at one time lies in ONE class and at 0.73-0.76 when it
// is spread over two or three; a plain linear fill shows it too (0.85 against 0.91).  hipMalloc hands memory out in runs of
// tens of GiB of one class, so what a canvas gets is luck (rounds 2 and 3 reported it as a box-to-box spread of 5-10 %, and
// round 3 worked around it by dealing the planes of a group over the allocation).  The class of an address cannot be read
// from user space, but it can be MEASURED -- two planes in one class fill at one plane's rate, two planes in different
// classes 1.3x faster -- and HIP's virtual memory management lets the caller decide which physical memory backs which
// virtual address.  So:
//   1. take the memory in SLICES (hipMemCreate; 64 MiB each by default), map them in creation order into one reserved
//      virtual range (hipMemAddressReserve / hipMemMap);
//   2. classify UNITS (512 MiB of consecutively created slices: physical memory is handed out in long runs) by the pair
//      fill against reference units: class 0 = whatever collides with unit 0, class 1 = whatever collides with the first unit
//      outside class 0, ...;
//   3. unmap, and map the slices again ROUND-ROBIN over the classes.
// Every 200 MB of such an arena then holds all classes, wherever a plane starts and however many planes a launch writes:
// one plane alone fills at 0.72-0.75, five consecutive planes at 0.75-0.76, the fusion kernel's structure goes from
// 0.61-0.62 to 0.71 (tools/membw_gains 700, profiles/r04_exp_mixed_arena.log).  Reads do not care (tiles may live anywhere).
//
// The arena is the one place where this library allocates: an explicit allocator the caller asks for and owns
// (sq_arena_create / sq_arena_destroy); every other entry point still takes caller-owned pointers and neither allocates
// nor frees.  The reference has no counterpart -- its canvas is a dask array (stitcher.py:356-362); this is where the
// MI355X-native version of that canvas lives.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <ctime>
#include <vector>

#include "common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define SQ_G1 __attribute__((address_space(1)))

// The probe's write pattern: a G x G grid of tiles, every tile `rows` row segments of SEG bytes at `pitch`, one workgroup per
// (tile, block of 8 rows), wave w the rows w and w + 4, 16 bytes per lane -- the fusion kernel's access pattern without its
// reads -- into TWO planes at once (the same offsets in both).
constexpr int PROBE_SEG = 3600, PROBE_ROWS = 1800;

__global__ __launch_bounds__(256) void arena_pair_fill_kernel(char *d0, char *d1, int G, size_t pitch) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int nblk = (PROBE_ROWS + 7) / 8, nvec = PROBE_SEG / 16;
    const int tile = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int ty = tile / G, tx = tile % G;
    const u32x4 v = {0x53514152u, (uint32_t)blockIdx.x, (uint32_t)threadIdx.x, 0u};
    for (int j = 0; j < 2; ++j) {
        const int r = blk * 8 + wave + 4 * j;
        if (r >= PROBE_ROWS) break;
        const size_t off = ((size_t)ty * PROBE_ROWS + r) * pitch + (size_t)tx * PROBE_SEG;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k;
            if (i < nvec) {
                __builtin_nontemporal_store(v, (SQ_G1 u32x4 *)(d0 + off + (size_t)i * 16));
                __builtin_nontemporal_store(v, (SQ_G1 u32x4 *)(d1 + off + (size_t)i * 16));
            }
        }
    }
}

bool trace_on() { return getenv("SQ_ARENA_TRACE") != nullptr; }

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

}  // namespace

struct sq_arena {
    char *base = nullptr;
    size_t bytes = 0, slice = 0;
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    bool mapped = false;
    sq_arena_info info{};
};

namespace {

void release(sq_arena *a) {
    if (!a) return;
    if (a->base) {
        if (a->mapped) (void)hipMemUnmap(a->base, a->bytes);
        (void)hipMemAddressFree(a->base, a->bytes);
    }
    for (auto h : a->handles) (void)hipMemRelease(h);
    (void)hipGetLastError();
    delete a;
}

int map_in_order(sq_arena *a, const std::vector<int> &order) {
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = a->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t i = 0; i < order.size(); ++i) {
        hipError_t e = hipMemMap(a->base + i * a->slice, a->slice, 0, a->handles[order[i]], 0);
        if (e != hipSuccess) return sq::fail(SQ_ERR_HIP, "sq_arena: hipMemMap of slice %zu failed: %s", i, hipGetErrorString(e));
    }
    a->mapped = true;
    hipError_t e = hipMemSetAccess(a->base, a->bytes, &acc, 1);
    if (e != hipSuccess) return sq::fail(SQ_ERR_HIP, "sq_arena: hipMemSetAccess failed: %s", hipGetErrorString(e));
    return SQ_OK;
}

}  // namespace

extern "C" sq_arena *sq_arena_create(int64_t bytes, int64_t candidate_bytes, int64_t slice_bytes, int64_t unit_bytes, int32_t flags,
                                     void *stream, sq_arena_info *info) {
    if (bytes <= 0) {
        sq::fail(SQ_ERR_INVALID, "sq_arena_create: bytes must be positive");
        return nullptr;
    }
    const size_t slice = slice_bytes > 0 ? (size_t)slice_bytes : ((size_t)64 << 20);
    size_t unit = unit_bytes > 0 ? (size_t)unit_bytes : ((size_t)512 << 20);
    if (slice % ((size_t)2 << 20) || unit % slice) {
        sq::fail(SQ_ERR_INVALID, "sq_arena_create: the slice must be a multiple of 2 MiB and the unit a multiple of the slice");
        return nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    sq_arena *a = new sq_arena;
    if (hipGetDevice(&a->device) != hipSuccess) {
        sq::fail(SQ_ERR_HIP, "sq_arena_create: no device");
        delete a;
        return nullptr;
    }
    // n slices make the arena; up to ncand are taken and classified so that the arena can be BALANCED over the classes (a third
    // each): what hipMalloc-order memory offers is runs of tens of GiB of one class, and an arena of 50 GiB taken as it comes
    // was 65 % one class.  The slices not chosen go back to the driver before the call returns.
    const size_t n = ((size_t)bytes + slice - 1) / slice;
    size_t ncand = std::max(n, candidate_bytes > 0 ? (size_t)candidate_bytes / slice : n);
    if (flags & SQ_ARENA_NATURAL_ORDER) ncand = n;
    a->slice = slice;
    a->bytes = n * slice;
    const double t0 = now_s();
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = a->device;
    a->handles.reserve(ncand);
    for (size_t i = 0; i < ncand; ++i) {
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, slice, &prop, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (i >= n && e == hipErrorOutOfMemory) {      // fewer candidates than asked for: the card is full, go on with these
                ncand = i;
                break;
            }
            sq::fail(e == hipErrorNotSupported ? SQ_ERR_UNSUPPORTED : SQ_ERR_HIP,
                     "sq_arena_create: %shipMemCreate of slice %zu of %zu (%zu MiB each) failed: %s",
                     e == hipErrorNotSupported ? "virtual memory management unsupported: " : "", i, n, slice >> 20, hipGetErrorString(e));
            release(a);
            return nullptr;
        }
        a->handles.push_back(h);
    }
    if (trace_on()) fprintf(stderr, "[sq_arena] hipMemCreate x %zu (%zu MiB each): %.1f ms\n", ncand, slice >> 20, (now_s() - t0) * 1e3);
    double tp = now_s();
    // the candidates, in creation order, in a virtual range of their own for the probe
    const size_t cand_bytes = ncand * slice;
    a->bytes = cand_bytes;      // (release() unmaps / frees what `bytes` says while the candidates are mapped)
    hipError_t e = hipMemAddressReserve((void **)&a->base, cand_bytes, (size_t)1 << 30, nullptr, 0);
    if (e != hipSuccess) {
        a->base = nullptr;
        sq::fail(e == hipErrorNotSupported ? SQ_ERR_UNSUPPORTED : SQ_ERR_HIP, "sq_arena_create: %shipMemAddressReserve of %zu bytes failed: %s",
                 e == hipErrorNotSupported ? "virtual memory management unsupported: " : "", cand_bytes, hipGetErrorString(e));
        release(a);
        return nullptr;
    }
    std::vector<int> order(ncand);
    for (size_t i = 0; i < ncand; ++i) order[i] = (int)i;
    if (map_in_order(a, order) != SQ_OK) {
        release(a);
        return nullptr;
    }
    if (trace_on()) fprintf(stderr, "[sq_arena] reserve + map + set access of the candidates: %.1f ms\n", (now_s() - tp) * 1e3);
    tp = now_s();
    sq_arena_info &I = a->info;
    I.slice_bytes = (int64_t)slice;
    I.n_slices = (int32_t)n;
    I.n_candidates = (int32_t)ncand;
    I.n_classes = 1;
    I.interleaved = 0;

    // ---- classify the units ------------------------------------------------------------------------------------------
    const size_t spu = unit / slice;
    const int nu = (int)(ncand / spu);      // whole units; a tail shorter than a unit takes the class of the unit before it
    std::vector<int> cls(std::max(nu, 1), -1);
    int ncls = 1;
    float probe_ms = 0;
    double glo = 1e30, ghi = 0;
    if (!(flags & SQ_ARENA_NATURAL_ORDER) && nu >= 2) {
        int G = 16;
        size_t pitch = 0;
        for (; G > 1; --G) {
            pitch = (size_t)G * PROBE_SEG + 560;
            if ((size_t)G * PROBE_ROWS * pitch <= unit) break;
        }
        pitch = (size_t)G * PROBE_SEG + 560;
        if ((size_t)G * PROBE_ROWS * pitch > unit) {
            sq::fail(SQ_ERR_INVALID, "sq_arena_create: a unit of %zu bytes is too small for the probe (>= 16 MiB)", unit);
            release(a);
            return nullptr;
        }
        const unsigned grid = (unsigned)(G * G * ((PROBE_ROWS + 7) / 8));
        const double moved = 2.0 * G * G * (double)PROBE_ROWS * PROBE_SEG;      // bytes one pair fill writes
        hipEvent_t e0, e1, p0, p1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventCreate(&p0);
        (void)hipEventCreate(&p1);
        (void)hipEventRecord(p0, st);
        std::vector<double> rate(nu);
        ncls = 0;
        bool bad = false;
        for (; ncls < SQ_ARENA_MAX_CLASSES && !bad; ++ncls) {
            int ref = -1;
            for (int k = 0; k < nu; ++k)
                if (cls[k] < 0) { ref = k; break; }
            if (ref < 0) break;
            cls[ref] = ncls;
            double lo = 1e30, hi = 0;
            for (int k = 0; k < nu; ++k) {
                if (cls[k] >= 0) continue;
                char *d0 = a->base + (size_t)ref * unit, *d1 = a->base + (size_t)k * unit;
                double best = 1e30;
                for (int rep = 0; rep < 3; ++rep) {      // the first launch warms up (page tables, clocks); best of the other two
                    (void)hipEventRecord(e0, st);
                    hipLaunchKernelGGL(arena_pair_fill_kernel, dim3(grid), dim3(256), 0, st, d0, d1, G, pitch);
                    (void)hipEventRecord(e1, st);
                    if (hipEventSynchronize(e1) != hipSuccess) { bad = true; break; }
                    float ms = 0;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    if (rep && ms < best) best = ms;
                }
                if (bad) break;
                rate[k] = moved / best / 1e6;      // GB/s
                lo = std::min(lo, rate[k]);
                hi = std::max(hi, rate[k]);
            }
            if (bad) break;
            if (hi == 0) {      // nothing left to compare with: the reference is the last class, alone in it
                ++ncls;
                break;
            }
            glo = std::min(glo, lo);
            ghi = std::max(ghi, hi);
            // two populations (collides with the reference / does not) are ~25 % apart; inside one the spread is ~3 %
            double cut;
            if (hi - lo > 0.10 * hi) cut = 0.5 * (lo + hi);
            else if (ncls == 0) cut = 1e30;                              // nothing stands out: one class
            else cut = lo < 0.5 * (glo + ghi) ? 1e30 : -1;               // all collide with this reference / none does
            for (int k = 0; k < nu; ++k)
                if (cls[k] < 0 && rate[k] < cut) cls[k] = ncls;
        }
        (void)hipEventRecord(p1, st);
        (void)hipEventSynchronize(p1);
        (void)hipEventElapsedTime(&probe_ms, p0, p1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipEventDestroy(p0);
        (void)hipEventDestroy(p1);
        if (bad || hipGetLastError() != hipSuccess) {
            sq::fail(SQ_ERR_HIP, "sq_arena_create: the classification probe failed");
            release(a);
            return nullptr;
        }
        for (int k = 0; k < nu; ++k)
            if (cls[k] < 0) cls[k] = ncls - 1;      // more populations than SQ_ARENA_MAX_CLASSES: the rest joins the last one
        ncls = std::max(ncls, 1);
    } else {
        std::fill(cls.begin(), cls.end(), 0);
    }

    if (trace_on()) {
        fprintf(stderr, "[sq_arena] probe: %.1f ms host, %.1f ms device, %d classes; units in creation order: ", (now_s() - tp) * 1e3, probe_ms, ncls);
        for (int k = 0; k < nu; ++k) fputc('A' + cls[k], stderr);
        fputc('\n', stderr);
    }
    tp = now_s();
    // ---- choose n slices round-robin over the classes, give the rest back, map the chosen ones -------------------------------
    auto class_of_slice = [&](size_t s) { return cls[std::min((size_t)std::max(nu, 1) - 1, s / spu)]; };
    I.n_classes = ncls;
    I.probe_ms = probe_ms;
    I.min_pair_gbs = ghi > 0 ? (float)glo : 0.f;
    I.max_pair_gbs = (float)ghi;
    for (int c = 0; c < SQ_ARENA_MAX_CLASSES; ++c) I.class_slices[c] = I.class_candidates[c] = 0;
    for (size_t s = 0; s < ncand; ++s) I.class_candidates[class_of_slice(s)]++;
    if (hipStreamSynchronize(st) != hipSuccess || hipMemUnmap(a->base, cand_bytes) != hipSuccess) {
        sq::fail(SQ_ERR_HIP, "sq_arena_create: hipMemUnmap failed");
        a->mapped = false;
        release(a);
        return nullptr;
    }
    a->mapped = false;
    (void)hipMemAddressFree(a->base, cand_bytes);
    a->base = nullptr;
    if (trace_on()) fprintf(stderr, "[sq_arena] unmap + free of the candidates' range: %.1f ms\n", (now_s() - tp) * 1e3);
    tp = now_s();
    std::vector<int> chosen;
    chosen.reserve(n);
    {
        std::vector<size_t> next(ncls, 0);
        while (chosen.size() < n) {
            for (int c = 0; c < ncls && chosen.size() < n; ++c) {
                size_t &p = next[c];
                while (p < ncand && class_of_slice(p) != c) ++p;
                if (p < ncand) {
                    chosen.push_back((int)p++);
                    I.class_slices[c]++;
                }
            }
        }
    }
    if (ncls == 1) std::sort(chosen.begin(), chosen.end());
    {
        std::vector<char> keep(ncand, 0);
        for (int c : chosen) keep[c] = 1;
        std::vector<hipMemGenericAllocationHandle_t> kept;
        std::vector<int> remap(ncand, -1);
        for (size_t i = 0; i < ncand; ++i) {
            if (keep[i]) {
                remap[i] = (int)kept.size();
                kept.push_back(a->handles[i]);
            } else {
                (void)hipMemRelease(a->handles[i]);
            }
        }
        a->handles.swap(kept);
        for (int &c : chosen) c = remap[c];
    }
    if (trace_on()) fprintf(stderr, "[sq_arena] release of %zu slices: %.1f ms\n", ncand - n, (now_s() - tp) * 1e3);
    tp = now_s();
    a->bytes = n * slice;
    e = hipMemAddressReserve((void **)&a->base, a->bytes, (size_t)1 << 30, nullptr, 0);
    if (e != hipSuccess) {
        a->base = nullptr;
        sq::fail(SQ_ERR_HIP, "sq_arena_create: hipMemAddressReserve of %zu bytes failed: %s", a->bytes, hipGetErrorString(e));
        release(a);
        return nullptr;
    }
    if (map_in_order(a, chosen) != SQ_OK) {
        release(a);
        return nullptr;
    }
    if (trace_on()) fprintf(stderr, "[sq_arena] reserve + map + set access of the arena: %.1f ms\n", (now_s() - tp) * 1e3);
    I.base_dev = a->base;
    I.bytes = (int64_t)a->bytes;
    I.interleaved = ncls > 1 ? 1 : 0;
    I.create_ms = (float)((now_s() - t0) * 1e3);
    if (info) *info = I;
    return a;
}

extern "C" int sq_arena_info_get(const sq_arena *arena, sq_arena_info *info) {
    if (!arena || !info) return sq::fail(SQ_ERR_INVALID, "sq_arena_info_get: NULL argument");
    *info = arena->info;
    return SQ_OK;
}

extern "C" int sq_arena_destroy(sq_arena *arena) {
    if (!arena) return SQ_OK;
    (void)hipDeviceSynchronize();      // nothing may still be writing into memory that is about to be unmapped
    release(arena);
    return SQ_OK;
}
