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
//   1. take candidate memory in SLICES (hipMemCreate; 64 MiB each by default), chunk by chunk, mapped in creation order into
//      one reserved virtual range (hipMemAddressReserve / hipMemMap);
//   2. classify UNITS (512 MiB of consecutively created slices: physical memory is handed out in long runs) as they come: a
//      unit joins the first class whose reference unit it collides with -- the pair fills less than 1.17 times faster than the
//      reference alone -- or becomes the reference of a new class; stop taking memory once the three largest classes each
//      hold a third of the arena -- or, after 2.5 x the arena, the two largest half of it each -- (or the caller's cap is
//      reached, or the card is full);
//   3. choose the arena's slices round-robin over the classes, give the others back, and map the chosen ones in that order.
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
#include <atomic>
#include <cstring>
#include <ctime>
#include <map>
#include <utility>
#include <chrono>
#include <thread>
#include <vector>

#include "common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define SQ_G1 __attribute__((address_space(1)))

// The probe's write pattern: a G x G grid of tiles, every tile `rows` row segments of SEG bytes at `pitch`, one workgroup per
// (tile, block of 8 rows), wave w the rows w and w + 4, 16 bytes per lane -- the fusion kernel's access pattern without its
// reads -- into TWO planes at once (the same offsets in both), or into one (d1 == NULL).
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
                if (d1) __builtin_nontemporal_store(v, (SQ_G1 u32x4 *)(d1 + off + (size_t)i * 16));
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
    size_t bytes = 0, slice = 0;      // bytes: the RESERVED virtual range at base
    size_t mapped_slices = 0;         // slices [0, mapped_slices) of the range are mapped, one hipMemMap each
    bool ever_mapped = false;         // the range at base has held a mapping (it is retired, not freed: retire_range)
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    sq_arena_info info{};
};

namespace {

// Every mapping is undone by a hipMemUnmap of exactly ITS range: one call over a range that holds many mappings undoes the first
// and leaves the runtime's view of the others in place -- memcpys into a range reserved later at the same address then went to
// the slices that USED to be there while kernels (page tables) saw the new ones (found by the plain-C caller of the test suite,
// whose arena came back at the candidates' address).
// A virtual range that has held mappings is RETIRED, never freed: its reservation stays for the life of the process (address
// space only -- the physical slices go back to the driver).  An address that has been mapped once must not be mapped again, by a
// later arena or by anyone the driver might hand the range to: the copy engines kept translating such addresses to the slices
// that USED to be there (kernels saw the new ones) -- a 130-KB device-to-host copy of a freshly fused canvas came back as the
// zeros of an earlier arena's memory (tests/test_arena_gpu.py after the full-size tests in one process; the plain-C caller
// under ROCm 7.2's runtime).  A create reserves at most the candidates' cap plus the arena itself (a few hundred GiB of the
// 128 TiB of address space).
std::atomic<long long> g_retired_bytes{0};
void retire_range(char *base, size_t bytes, bool was_mapped) {
    if (!base) return;
    if (was_mapped) g_retired_bytes.fetch_add((long long)bytes, std::memory_order_relaxed);
    else (void)hipMemAddressFree(base, bytes);
}

void unmap_all(sq_arena *a) {
    for (size_t i = 0; i < a->mapped_slices; ++i) (void)hipMemUnmap(a->base + i * a->slice, a->slice);
    a->mapped_slices = 0;
}

void release(sq_arena *a) {
    if (!a) return;
    if (a->base) {
        unmap_all(a);
        retire_range(a->base, a->bytes, a->ever_mapped);
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
        a->mapped_slices = i + 1;
        a->ever_mapped = true;
        e = hipMemSetAccess(a->base + i * a->slice, a->slice, &acc, 1);
        if (e != hipSuccess) return sq::fail(SQ_ERR_HIP, "sq_arena: hipMemSetAccess of slice %zu failed: %s", i, hipGetErrorString(e));
    }
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
    const size_t unit = unit_bytes > 0 ? (size_t)unit_bytes : ((size_t)512 << 20);
    if (slice % ((size_t)2 << 20) || unit % slice) {
        sq::fail(SQ_ERR_INVALID, "sq_arena_create: the slice must be a multiple of 2 MiB and the unit a multiple of the slice");
        return nullptr;
    }
    // the probe's geometry: the largest G x G grid of row segments that fits a unit
    int G = 16;
    for (; G > 1; --G)
        if ((size_t)G * PROBE_ROWS * ((size_t)G * PROBE_SEG + 560) <= unit) break;
    const size_t pitch = (size_t)G * PROBE_SEG + 560;
    if ((size_t)G * PROBE_ROWS * pitch > unit) {
        sq::fail(SQ_ERR_INVALID, "sq_arena_create: a unit of %zu bytes is too small for the probe (>= 16 MiB)", unit);
        return nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    sq_arena *a = new sq_arena;
    if (hipGetDevice(&a->device) != hipSuccess) {
        sq::fail(SQ_ERR_HIP, "sq_arena_create: no device");
        delete a;
        return nullptr;
    }
    // n slices make the arena.  Memory comes in runs of tens of GiB of one class, so candidates are taken chunk by chunk and
    // classified as they come, until the three largest classes each hold a third of the arena (or `cap` slices are taken, or
    // the card is full); the slices not chosen go back to the driver before the call returns.
    const bool natural = (flags & SQ_ARENA_NATURAL_ORDER) != 0;
    const size_t n = ((size_t)bytes + slice - 1) / slice;
    const size_t spu = unit / slice;
    const size_t cap = natural ? n : std::max(n, candidate_bytes > 0 ? (size_t)candidate_bytes / slice : n);
    const size_t chunk = std::max<size_t>(8 * spu, (n / 4 + spu - 1) / spu * spu);      // >= 4 GiB at the default sizes
    a->slice = slice;
    const double t0 = now_s();
    size_t free_before = 0, total_mem = 0;
    (void)hipMemGetInfo(&free_before, &total_mem);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = a->device;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t cand_bytes = cap * slice;
    a->bytes = cand_bytes;      // (release() unmaps / frees what `bytes` says while the candidates' range exists)
    hipError_t e = hipMemAddressReserve((void **)&a->base, cand_bytes, 0, nullptr, 0);
    if (e != hipSuccess) {
        a->base = nullptr;
        sq::fail(e == hipErrorNotSupported ? SQ_ERR_UNSUPPORTED : SQ_ERR_HIP, "sq_arena_create: %shipMemAddressReserve of %zu bytes failed: %s",
                 e == hipErrorNotSupported ? "virtual memory management unsupported: " : "", cand_bytes, hipGetErrorString(e));
        release(a);
        return nullptr;
    }
    a->handles.reserve(cap);
    sq_arena_info &I = a->info;
    std::vector<int> cls;                 // class of every whole unit taken so far
    std::vector<int> ref_unit;            // per class: its reference unit ...
    std::vector<size_t> class_slices_taken;
    float probe_ms = 0;
    double glo = 1e30, ghi = 0, t_create = 0, t_map = 0;
    const unsigned grid = (unsigned)(G * G * ((PROBE_ROWS + 7) / 8));
    const double plane_bytes = (double)G * G * PROBE_ROWS * PROBE_SEG;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    bool bad = false;
    // GB/s written by the fill of unit u alone (v < 0) or of the pair (u, v): the first launch warms up, best of the other two
    auto fill_rate = [&](int u, int v) {
        char *d0 = a->base + (size_t)u * unit, *d1 = v >= 0 ? a->base + (size_t)v * unit : nullptr;
        double best = 1e30;
        for (int rep = 0; rep < 3 && !bad; ++rep) {
            (void)hipEventRecord(e0, st);
            hipLaunchKernelGGL(arena_pair_fill_kernel, dim3(grid), dim3(256), 0, st, d0, d1, G, pitch);
            (void)hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess) { bad = true; break; }
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            probe_ms += ms;
            if (rep && ms < best) best = ms;
        }
        return (v >= 0 ? 2.0 : 1.0) * plane_bytes / best / 1e6;
    };
    // Two planes in ONE class fill at about the rate of one plane alone, two planes in different classes 1.2-1.3 times faster
    // (profiles/r04_exp_placement_counters.log; with this probe's small geometry: alone 4.2, same class 4.3-4.7, different
    // classes 5.1-5.35 TB/s).  A unit COLLIDES with a class when its pair with the class' reference unit is on the slow side of
    // what that reference has been seen to do -- below the midpoint of its slowest and fastest pair once those are 12 % apart;
    // before a reference has shown both behaviours the rate of the reference alone decides (a pair less than 1.17 times
    // faster collides).  Every measured pair is kept, and the whole assignment is re-derived from the kept rates after every
    // chunk of candidates, so a decision taken before a reference's two populations were known does not stick.
    std::map<std::pair<int, int>, double> pair_rate;
    std::map<int, double> alone_rate;
    auto pair_of = [&](int r, int u) {
        auto it = pair_rate.find({r, u});
        if (it != pair_rate.end()) return it->second;
        const double v = fill_rate(r, u);
        glo = std::min(glo, v);
        ghi = std::max(ghi, v);
        pair_rate[{r, u}] = v;
        return v;
    };
    auto alone_of = [&](int r) {
        auto it = alone_rate.find(r);
        if (it != alone_rate.end()) return it->second;
        return alone_rate[r] = fill_rate(r, -1);
    };
    auto collides = [&](int r, double v) {
        double lo = 1e30, hi = 0;
        for (auto it = pair_rate.lower_bound({r, -1}); it != pair_rate.end() && it->first.first == r; ++it) {
            lo = std::min(lo, it->second);
            hi = std::max(hi, it->second);
        }
        if (hi > 1.12 * lo) return v < 0.5 * (lo + hi);
        return v < 1.17 * alone_of(r);
    };
    // classes of the units taken so far, from scratch: a unit joins the first class (in order of creation) it collides with,
    // else it founds one; twice, so that the second round decides with every rate of the first on the table
    auto classify_all = [&](size_t nunits) {
        for (int round = 0; round < 2 && !bad; ++round) {
            ref_unit.clear();
            cls.assign(nunits, -1);
            for (size_t u = 0; u < nunits && !bad; ++u) {
                int got = -1;
                for (size_t c = 0; c < ref_unit.size() && got < 0; ++c)
                    if (collides(ref_unit[c], pair_of(ref_unit[c], (int)u))) got = (int)c;
                if (got < 0) {
                    if ((int)ref_unit.size() < SQ_ARENA_MAX_CLASSES) {
                        ref_unit.push_back((int)u);
                        got = (int)ref_unit.size() - 1;
                    } else {
                        got = SQ_ARENA_MAX_CLASSES - 1;      // more populations than slots: the last takes the rest
                    }
                }
                cls[u] = got;
            }
        }
        class_slices_taken.assign(ref_unit.size(), 0);
        for (size_t u = 0; u < nunits; ++u) class_slices_taken[cls[u]] += spu;
    };
    // enough taken: a third of the arena in each of the three largest classes -- or, once 2.5 x the arena has been looked at, half
    // of it in each of the two largest (some cards show two classes only; going on to the end of such a card cost 8.6 s).  Three
    // thirds are worth waiting for: over the round's bench runs arenas of three or more classes gave 0.725-0.736 on config 3
    // and 0.732-0.741 on the headline job's batches, arenas of two halves 0.721-0.726 and 0.711-0.730 (profiles/r04_bench_cfg3_v36
    // ... v44.json; the one-process A/B on 20 planes, profiles/r04_exp_arena_two_classes.log, had seen no difference).
    auto balanced = [&](size_t have_now) {
        std::vector<size_t> c(class_slices_taken);
        std::sort(c.rbegin(), c.rend());
        if (c.size() >= 3 && c[2] >= (n + 2) / 3) return true;
        return c.size() >= 2 && 2 * have_now >= 5 * n && c[1] >= (n + 1) / 2;
    };
    size_t have = 0;
    bool full = false;
    while (have < cap && !full && !bad) {
        const size_t want = std::min(have == 0 ? std::max(n, chunk) : chunk, cap - have);
        double tc = now_s();
        size_t got = 0;
        for (; got < want; ++got) {
            hipMemGenericAllocationHandle_t h;
            e = hipMemCreate(&h, slice, &prop, 0);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                if (have + got >= n && e == hipErrorOutOfMemory) {      // the card is full: go on with what there is
                    full = true;
                    break;
                }
                sq::fail(e == hipErrorNotSupported ? SQ_ERR_UNSUPPORTED : SQ_ERR_HIP,
                         "sq_arena_create: %shipMemCreate of slice %zu of %zu (%zu MiB each) failed: %s",
                         e == hipErrorNotSupported ? "virtual memory management unsupported: " : "", have + got, n, slice >> 20, hipGetErrorString(e));
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
                release(a);      // unmaps the slices mapped so far, frees the range, releases every handle taken
                return nullptr;
            }
            a->handles.push_back(h);
        }
        t_create += now_s() - tc;
        tc = now_s();
        for (size_t i = have; i < have + got && !bad; ++i) {
            if (hipMemMap(a->base + i * slice, slice, 0, a->handles[i], 0) != hipSuccess) {
                bad = true;
                break;
            }
            a->mapped_slices = i + 1;
            a->ever_mapped = true;
            if (hipMemSetAccess(a->base + i * slice, slice, &acc, 1) != hipSuccess) bad = true;
        }
        t_map += now_s() - tc;
        have += got;
        if (natural || bad) break;
        classify_all(have / spu);
        if (balanced(have)) break;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const size_t ncand = have;
    auto unmap_candidates = [&]() {
        (void)hipStreamSynchronize(st);
        unmap_all(a);
        retire_range(a->base, cand_bytes, a->ever_mapped);
        a->base = nullptr;
        a->ever_mapped = false;
    };
    if (bad || hipGetLastError() != hipSuccess || ncand < n) {
        sq::fail(SQ_ERR_HIP, "sq_arena_create: mapping or probing the candidate slices failed");
        unmap_candidates();
        release(a);
        return nullptr;
    }
    const int nu = (int)cls.size();
    const int ncls = std::max<int>(1, (int)ref_unit.size());
    if (trace_on()) {
        fprintf(stderr, "[sq_arena] %zu candidate slices of %zu MiB for an arena of %zu: hipMemCreate %.1f ms, map %.1f ms, probe %.1f ms on the device, %d classes\n"
                        "[sq_arena] units in creation order: ", ncand, slice >> 20, n, t_create * 1e3, t_map * 1e3, probe_ms, ncls);
        for (int k = 0; k < nu; ++k) fputc('A' + cls[k], stderr);
        fputc('\n', stderr);
    }
    double tp = now_s();

    // ---- choose n slices round-robin over the classes, give the rest back, map the chosen ones -------------------------------
    auto class_of_slice = [&](size_t s) { return nu ? cls[std::min((size_t)nu - 1, s / spu)] : 0; };
    I.slice_bytes = (int64_t)slice;
    I.n_slices = (int32_t)n;
    I.n_candidates = (int32_t)ncand;
    I.n_classes = ncls;
    I.probe_ms = probe_ms;
    I.min_pair_gbs = ghi > 0 ? (float)glo : 0.f;
    I.max_pair_gbs = (float)ghi;
    for (int c = 0; c < SQ_ARENA_MAX_CLASSES; ++c) I.class_slices[c] = I.class_candidates[c] = 0;
    for (size_t s = 0; s < ncand; ++s) I.class_candidates[class_of_slice(s)]++;
    // The candidates are unmapped here; their RANGE is retired (retire_range), so the arena -- reserved below -- cannot come back
    // at an address the candidates were mapped at.
    (void)hipStreamSynchronize(st);
    unmap_all(a);
    char *cand_base = a->base;
    const bool cand_mapped = a->ever_mapped;
    a->base = nullptr;
    a->ever_mapped = false;
    if (trace_on()) fprintf(stderr, "[sq_arena] unmap of the candidates: %.1f ms\n", (now_s() - tp) * 1e3);
    tp = now_s();
    std::vector<int> chosen;
    chosen.reserve(n);
    {
        // SQ_ARENA_TWO_CLASSES (a measurement aid): the two largest classes only, as long as they have slices
        std::vector<char> skip(ncls, 0);
        if ((flags & SQ_ARENA_TWO_CLASSES) && ncls > 2) {
            std::vector<int> by_size(ncls);
            for (int c = 0; c < ncls; ++c) by_size[c] = c;
            std::sort(by_size.begin(), by_size.end(), [&](int x, int y) { return I.class_candidates[x] > I.class_candidates[y]; });
            if ((size_t)I.class_candidates[by_size[0]] + (size_t)I.class_candidates[by_size[1]] >= n)
                for (int k = 2; k < ncls; ++k) skip[by_size[k]] = 1;
        }
        std::vector<size_t> next(ncls, 0);
        while (chosen.size() < n) {
            for (int c = 0; c < ncls && chosen.size() < n; ++c) {
                if (skip[c]) continue;
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
    // hipMemRelease returns before the driver has the memory back (it clears what it takes back): a caller that allocates the
    // rest of the card right after this call -- the tiles after the canvas -- met "out of memory" with 6 of 288 GiB reported free
    // (the bench's counter pass, arena of all 4 501 candidates).  Wait until the card reports what it reported before, less the arena.
    if (ncand > n && free_before > 0) {
        const size_t want = free_before > n * slice + ((size_t)1 << 30) ? free_before - n * slice - ((size_t)1 << 30) : 0;
        size_t now_free = 0;
        for (int tries = 0; tries < 5000; ++tries) {      // at most ~10 s
            if (hipMemGetInfo(&now_free, &total_mem) != hipSuccess || now_free >= want) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        if (trace_on()) fprintf(stderr, "[sq_arena] the released memory is back after %.1f ms (%.1f of %.1f GiB free)\n", (now_s() - tp) * 1e3,
                                now_free / 1073741824.0, total_mem / 1073741824.0);
        tp = now_s();
    }
    a->bytes = n * slice;
    e = hipMemAddressReserve((void **)&a->base, a->bytes, 0, nullptr, 0);
    retire_range(cand_base, cand_bytes, cand_mapped);
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
    // nothing may still be writing into memory that is about to be unmapped: wait for the ARENA's device, whichever device is
    // current on the calling thread (a process that drives several GPUs), and put the caller's device back
    int current = -1;
    (void)hipGetDevice(&current);
    const int device = arena->device;
    if (current != device) (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    size_t free_before = 0, total_mem = 0, now_free = 0;
    const size_t held = arena->handles.size() * arena->slice;
    (void)hipMemGetInfo(&free_before, &total_mem);
    release(arena);
    // the driver has the memory back a little after hipMemRelease returns (sq_arena_create): the next allocation of the caller may
    // need it -- wait for it, at most ~3 s
    for (int tries = 0; held >= ((size_t)4 << 30) && tries < 1500; ++tries) {      // (small arenas: nothing anybody waits for)
        if (hipMemGetInfo(&now_free, &total_mem) != hipSuccess || now_free + ((size_t)1 << 30) >= free_before + held) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    if (current >= 0 && current != device) (void)hipSetDevice(current);
    return SQ_OK;
}
