// Host-side span planner for fusion.
//
// The reference places tiles one by one into the canvas (stitcher.py:652-681, :583-598); where
// rectangles overlap the later file wins.  Here the same result is described as a partition of
// the canvas: sweep the distinct y edges into bands, inside a band sweep the distinct x edges
// into intervals, give every interval its owner (overwrite: the last rect that covers it;
// feather: all rects that cover it, in write order), merge equal neighbours horizontally and
// then vertically.  The device then writes every canvas voxel exactly once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace sq {
std::string &last_error_ref() {
    static thread_local std::string msg;
    return msg;
}
}  // namespace sq

using namespace sq;

namespace {

// fn(t, lo, hi) over [0, n) cut into `nthreads` contiguous ranges; thread 0 is the caller
template <typename F>
void parallel_ranges(int nthreads, size_t n, F fn) {
    if (nthreads <= 1 || n == 0) {
        fn(0, (size_t)0, n);
        return;
    }
    // An exception in a worker (std::bad_alloc from a reserve, say) must not reach the top of its thread -- that is
    // std::terminate for the whole process: the first one is kept and re-thrown on the calling thread after the join, where
    // the entry points turn it into a status (guarded(), below).
    std::vector<std::thread> pool;
    pool.reserve(nthreads - 1);
    std::exception_ptr failed;
    std::mutex failed_mutex;
    auto bound = [&](int t) { return n * (size_t)t / (size_t)nthreads; };
    auto run = [&](int t) {
        try {
            fn(t, bound(t), bound(t + 1));
        } catch (...) {
            std::lock_guard<std::mutex> lock(failed_mutex);
            if (!failed) failed = std::current_exception();
        }
    };
    size_t started = 0;
    try {
        for (int t = 1; t < nthreads; ++t, ++started) pool.emplace_back(run, t);
    } catch (...) {      // no more threads to be had: the ranges not started run here, below
        std::lock_guard<std::mutex> lock(failed_mutex);
        if (!failed) failed = std::current_exception();
    }
    run(0);
    for (auto &th : pool) th.join();
    if (failed) std::rethrow_exception(failed);
}

// the planner's entry points are extern "C": nothing may propagate out of them
template <typename F>
sq_fuse_plan *guarded(const char *what, F make) {
    try {
        return make();
    } catch (const std::bad_alloc &) {
        sq::fail(SQ_ERR_INVALID, "%s: out of host memory while planning", what);
    } catch (const std::exception &e) {
        sq::fail(SQ_ERR_INVALID, "%s: %s", what, e.what());
    } catch (...) {
        sq::fail(SQ_ERR_INVALID, "%s: unknown failure while planning", what);
    }
    return nullptr;
}

struct Clipped {
    int y0, y1, x0, x1;  // canvas extent [y0,y1) x [x0,x1)
    int src_y, src_x;    // tile pixel at (y0, x0)
    int tile;
};

}  // namespace

// ---- storage pool of plan tables (see sq_table in common.h) ---------------------------------------
namespace {
struct TableSlab {
    void *ptr;
    size_t capacity;
};
std::mutex g_pool_mutex;
std::vector<TableSlab> g_pool;          // page-locked slabs waiting for reuse
constexpr size_t POOL_MAX_SLABS = 4;
bool g_pinning_failed = false;          // no device / out of lockable memory: stop trying
}   // namespace

void *sq_table_acquire(size_t bytes, size_t *capacity, bool *pinned) {
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].capacity >= bytes && g_pool[i].capacity <= 2 * bytes + (1 << 20)) {
                const TableSlab slab = g_pool[i];
                g_pool.erase(g_pool.begin() + (long)i);
                *capacity = slab.capacity;
                *pinned = true;
                return slab.ptr;
            }
    }
    if (!g_pinning_failed) {
        const size_t cap = (bytes + (bytes >> 2) + 4095) & ~size_t(4095);   // 25 % headroom: the next region's plan fits too
        void *p = nullptr;
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) == hipSuccess && p) {
            *capacity = cap;
            *pinned = true;
            return p;
        }
        (void)hipGetLastError();
        g_pinning_failed = true;
    }
    *capacity = bytes;
    *pinned = false;
    return malloc(bytes);
}

void sq_table_release(void *ptr, size_t capacity, bool pinned) {
    if (!pinned) {
        free(ptr);
        return;
    }
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        if (g_pool.size() < POOL_MAX_SLABS) {
            g_pool.push_back({ptr, capacity});
            return;
        }
    }
    (void)hipHostFree(ptr);
}

extern "C" {

int sq_version(void) { return SQ_VERSION; }
const char *sq_last_error(void) { return last_error_ref().c_str(); }

}   // extern "C"

namespace {
// The first half of a plan: the partition of the canvas into spans (and their refs).
struct SpanStage {
    std::vector<Span> spans;
    std::vector<Ref> refs;
    int max_refs = 0;
    int64_t covered = 0;
    std::chrono::steady_clock::time_point t_begin, t_sweep;
};

bool make_spans(const sq_rect *rects, int32_t n_rects, int32_t tile_h, int32_t tile_w, int32_t canvas_h, int32_t canvas_w,
                int32_t mode, const char *who, SpanStage &S) {
    std::vector<Span> &spans = S.spans;
    std::vector<Ref> &refs = S.refs;
    int &max_refs = S.max_refs;
    int64_t &covered = S.covered;
    if ((!rects && n_rects > 0) || n_rects < 0 || tile_h <= 0 || tile_w <= 0 || canvas_h <= 0 || canvas_w <= 0) {
        fail(SQ_ERR_INVALID, "%s: bad sizes (n_rects=%d tile=%dx%d canvas=%dx%d)", who, n_rects, tile_h,
             tile_w, canvas_h, canvas_w);
        return false;
    }
    if (mode != SQ_FUSE_OVERWRITE && mode != SQ_FUSE_FEATHER) {
        fail(SQ_ERR_INVALID, "%s: unknown mode %d", who, mode);
        return false;
    }
    S.t_begin = std::chrono::steady_clock::now();
    // Canvas clip of stitcher.py:590-594 (python slice semantics) + validation of the source side.
    std::vector<Clipped> cl;
    cl.reserve(n_rects);
    for (int i = 0; i < n_rects; ++i) {
        const sq_rect &r = rects[i];
        if (r.h < 0 || r.w < 0 || r.src_y0 < 0 || r.src_x0 < 0 || r.src_y0 + (int64_t)r.h > tile_h ||
            r.src_x0 + (int64_t)r.w > tile_w) {
            fail(SQ_ERR_INVALID, "%s: rect %d reads outside its %dx%d tile (src %d,%d size %dx%d)", who, i,
                 tile_h, tile_w, r.src_y0, r.src_x0, r.h, r.w);
            return false;
        }
        if (r.dst_y < 0 || r.dst_x < 0) {
            // The reference would wrap a negative index (python slicing); its placements are
            // never negative on this path, so refuse instead of guessing.
            fail(SQ_ERR_INVALID, "%s: rect %d has a negative canvas offset (%d,%d)", who, i, r.dst_y,
                 r.dst_x);
            return false;
        }
        int h = std::min<int64_t>(r.h, (int64_t)canvas_h - r.dst_y);
        int w = std::min<int64_t>(r.w, (int64_t)canvas_w - r.dst_x);
        if (h <= 0 || w <= 0) continue;
        cl.push_back({r.dst_y, r.dst_y + h, r.dst_x, r.dst_x + w, r.src_y0, r.src_x0, i});
    }

    std::vector<int> ys = {0, canvas_h};
    for (auto &c : cl) {
        ys.push_back(c.y0);
        ys.push_back(c.y1);
    }
    std::sort(ys.begin(), ys.end());
    ys.erase(std::unique(ys.begin(), ys.end()), ys.end());

    // rects sorted by y0 for the sweep's activation; active set kept in write order
    std::vector<int> by_y0(cl.size());
    for (size_t i = 0; i < cl.size(); ++i) by_y0[i] = (int)i;
    std::stable_sort(by_y0.begin(), by_y0.end(), [&](int a, int b) { return cl[a].y0 < cl[b].y0; });

    // Owner lists are small fixed arrays: no heap traffic per interval.
    struct Owners {
        int n = 0;
        int v[MAX_REFS];
        bool operator==(const Owners &o) const {
            if (n != o.n) return false;
            for (int i = 0; i < n; ++i)
                if (v[i] != o.v[i]) return false;
            return true;
        }
    };
    // Phase 1, band by band and independent per band (a few threads take contiguous ranges of bands): the band's RUNS --
    // maximal x intervals with one owner list, left to right.  A 32x32 grid has 2 049 bands of ~65 runs.
    struct Run {
        int xa, xb;
        Owners own;
    };
    const size_t n_bands = ys.size() - 1;
    const int hw_sweep = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const int sweep_threads = n_bands >= 512 ? hw_sweep : 1;
    std::vector<std::vector<Run>> runs_of(sweep_threads);           // per thread: the runs of its bands, band after band
    std::vector<std::vector<uint32_t>> run_end(sweep_threads);      // per thread: end of each of its bands in runs_of
    std::vector<size_t> first_band(sweep_threads + 1, 0);
    std::vector<int> overflow_at(sweep_threads, -1);                // feather: a band where more than MAX_REFS tiles overlap
    std::vector<int> overflow_x(sweep_threads, 0);
    parallel_ranges(sweep_threads, n_bands, [&](int t, size_t b_lo, size_t b_hi) {
        first_band[t] = b_lo;
        if (t + 1 == sweep_threads) first_band[t + 1] = n_bands;
        std::vector<Run> &out = runs_of[t];
        std::vector<uint32_t> &ends = run_end[t];
        out.reserve((b_hi - b_lo) * 72);
        ends.reserve(b_hi - b_lo);
        if (b_lo >= b_hi) return;
        // the active set at the range's first band: rects with y0 <= ya < y1, ascending (= write order) -- what the
        // band-by-band updates below keep
        std::vector<int> active;
        const int y_first = ys[b_lo];
        for (size_t i = 0; i < cl.size(); ++i)
            if (cl[i].y0 <= y_first && cl[i].y1 > y_first) active.push_back((int)i);
        size_t next = (size_t)(std::upper_bound(by_y0.begin(), by_y0.end(), y_first, [&](int y, int i) { return y < cl[i].y0; }) - by_y0.begin());
        std::vector<int> xs, owner_of, skip;
        for (size_t b = b_lo; b < b_hi; ++b) {
            const int ya = ys[b];
            while (next < by_y0.size() && cl[by_y0[next]].y0 <= ya) {
                active.insert(std::upper_bound(active.begin(), active.end(), by_y0[next]), by_y0[next]);
                ++next;
            }
            active.erase(std::remove_if(active.begin(), active.end(), [&](int i) { return cl[i].y1 <= ya; }), active.end());

            xs.clear();
            xs.push_back(0);
            xs.push_back(canvas_w);
            for (int i : active) {
                xs.push_back(cl[i].x0);
                xs.push_back(cl[i].x1);
            }
            std::sort(xs.begin(), xs.end());
            xs.erase(std::unique(xs.begin(), xs.end()), xs.end());
            const size_t m = xs.size() - 1;   // elementary x intervals of this band
            if (mode == SQ_FUSE_OVERWRITE) {
                // last writer wins: paint the intervals from the last active rect to the first, each interval
                // once (skip pointers jump over what is already painted)
                owner_of.assign(m, -1);
                skip.resize(m + 1);
                for (size_t k = 0; k <= m; ++k) skip[k] = (int)k;
                auto find = [&](int k) {
                    while (skip[k] != k) k = skip[k] = skip[skip[k]];
                    return k;
                };
                for (size_t a = active.size(); a-- > 0;) {
                    const Clipped &c = cl[active[a]];
                    const int lo = (int)(std::lower_bound(xs.begin(), xs.end(), c.x0) - xs.begin());
                    const int hi = (int)(std::lower_bound(xs.begin(), xs.end(), c.x1) - xs.begin());
                    for (int k = find(lo); k < hi; k = find(k + 1)) {
                        owner_of[k] = active[a];
                        skip[k] = k + 1;
                    }
                }
            }
            bool have = false;
            Run cur{};
            for (size_t k = 0; k < m; ++k) {
                const int xa = xs[k], xb = xs[k + 1];
                Owners own;
                if (mode == SQ_FUSE_OVERWRITE) {
                    if (owner_of[k] >= 0) own.v[own.n++] = owner_of[k];
                } else {
                    for (int i : active)
                        if (cl[i].x0 <= xa && cl[i].x1 >= xb) {
                            if (own.n == MAX_REFS) {
                                if (overflow_at[t] < 0) {
                                    overflow_at[t] = ya;
                                    overflow_x[t] = xa;
                                }
                                break;
                            }
                            own.v[own.n++] = i;
                        }
                }
                if (have && cur.own == own && cur.xb == xa) {
                    cur.xb = xb;
                } else {
                    if (have) out.push_back(cur);
                    cur.xa = xa;
                    cur.xb = xb;
                    cur.own = own;
                    have = true;
                }
            }
            if (have) out.push_back(cur);
            ends.push_back((uint32_t)out.size());
        }
    });
    for (int t = 0; t < sweep_threads; ++t)
        if (overflow_at[t] >= 0) {
            fail(SQ_ERR_UNSUPPORTED, "%s: more than %d tiles overlap at canvas (%d,%d)", who, MAX_REFS, overflow_at[t],
                 overflow_x[t]);
            return false;
        }
    const auto t_runs = std::chrono::steady_clock::now();
    // Phase 2, in band order: a run continues the span above it when interval and owners are the same (spans of the
    // previous band sorted by xa, the band's runs too: a two-pointer walk), else it opens a new span.
    struct Open {
        int xa, xb;
        int64_t index;   // into spans
        const Run *run;  // the run of the band above (its owner list; runs_of outlives the walk)
    };
    std::vector<Open> open, open_next;
    for (int t = 0; t < sweep_threads; ++t) {
        const std::vector<Run> &rs = runs_of[t];
        size_t at = 0;
        for (size_t k = 0; k < run_end[t].size(); ++k) {
            const size_t b = first_band[t] + k;
            const int ya = ys[b], yb = ys[b + 1];
            open_next.clear();
            size_t op = 0;   // pointer into `open`
            for (; at < run_end[t][k]; ++at) {
                const Run &cur = rs[at];
                while (op < open.size() && open[op].xa < cur.xa) ++op;
                int64_t index;
                if (op < open.size() && open[op].xa == cur.xa && open[op].xb == cur.xb && open[op].run->own == cur.own) {
                    index = open[op].index;
                    spans[index].h += yb - ya;
                } else {
                    Span sp{};
                    sp.dst_y = ya;
                    sp.dst_x = cur.xa;
                    sp.h = yb - ya;
                    sp.w = cur.xb - cur.xa;
                    sp.nref = cur.own.n;
                    sp.ref0 = (int)refs.size();
                    for (int q = 0; q < cur.own.n; ++q) {
                        const Clipped &c = cl[cur.own.v[q]];
                        refs.push_back({c.tile, c.src_y + (ya - c.y0), c.src_x + (cur.xa - c.x0), 0});
                    }
                    index = (int64_t)spans.size();
                    spans.push_back(sp);
                }
                open_next.push_back({cur.xa, cur.xb, index, &cur});
                if (cur.own.n) covered += (int64_t)(yb - ya) * (cur.xb - cur.xa);
                max_refs = std::max(max_refs, cur.own.n);
            }
            open.swap(open_next);
        }
    }

    S.t_sweep = std::chrono::steady_clock::now();
#ifdef SQ_EXPERIMENTS
    if (getenv("SQ_PLAN_TIMING")) {
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[spans] band runs %.3f ms (%d threads, %zu bands), merge %.3f ms (%zu spans)\n", ms(S.t_begin, t_runs), sweep_threads,
                n_bands, ms(t_runs, S.t_sweep), spans.size());
    }
#else
    (void)t_runs;
#endif
    return true;
}
}   // namespace

extern "C" {

static sq_fuse_plan *plan_create_impl(const sq_rect *rects, int32_t n_rects, int32_t tile_h, int32_t tile_w,
                                      int32_t canvas_h, int32_t canvas_w, int32_t mode) {
    SpanStage S;
    if (!make_spans(rects, n_rects, tile_h, tile_w, canvas_h, canvas_w, mode, "sq_fuse_plan_create", S)) return nullptr;
    std::vector<Span> &spans = S.spans;
    std::vector<Ref> &refs = S.refs;
    const int max_refs = S.max_refs;
    const int64_t covered = S.covered;
    const auto t_begin = S.t_begin, t_sweep = S.t_sweep;

    // ---- work items, generated straight into their final place in the table -----------------------
    // Order (overwrite mode): "one tile-row block per XCD".  Items are grouped by the block of
    // BLOCK_ROWS tile rows they read (all tiles, in tile order; a stable counting sort), and row block r
    // is laid out on list positions == r (mod 8): lane r mod 8.  The fusion kernel's workgroups pull
    // chunks of consecutive positions of the lane of the XCD they run on (fuse.hip; with the static
    // fallback, workgroup b takes b, b + G, ... and workgroups are dealt to the XCDs round-robin, which
    // comes to the same lanes), so the workgroups resident on one XCD all read the same few rows of the
    // flatfield at the same time, fetched into that XCD's L2 once instead of once per tile.  Positions
    // [0, 8 * lane_items) are interleaved this way (header field); the lanes' leftovers and the zero-fill
    // items follow.  Every item still moves whole row segments, so HBM sees the same contiguous runs, in
    // a different order.
    //   (experiment builds only, -DSQ_EXPERIMENTS: SQ_PLAN_ORDER=0 keeps span order, 1 = row blocks
    //   without the XCD interleave; the product library reads no environment variable)
#ifdef SQ_EXPERIMENTS
    const char *order_env = getenv("SQ_PLAN_ORDER");
#else
    const char *order_env = nullptr;
#endif
    // Feather plans take the overwrite plans' order (2): buckets of BLOCK_ROWS TILE rows (the first reference's) dealt to the XCD
    // lanes, so that an XCD works on a few rows of the gain image at a time -- the gains of a feather launch were fetched into
    // the L2s 60 times over in span order (FETCH_SIZE 142 GB per 40-plane launch against 87 GB of pixels, profiles/
    // r04_feather_counters.log).  Measured on 40 planes of config 3, uint16 canvas with gains: span order (0) 0.608, canvas
    // raster (3: bands of 16 rows left to right, x-neighbours in one workgroup's chunk) 0.611, this 0.620 of the HBM peak
    // (profiles/r04_exp_feather_order.log; experiment builds: SQ_FEATHER_ORDER).
#ifdef SQ_EXPERIMENTS
    const char *forder_env = getenv("SQ_FEATHER_ORDER");
#else
    const char *forder_env = nullptr;
#endif
    const int order_mode = mode == SQ_FUSE_OVERWRITE ? (order_env ? atoi(order_env) : 2) : (forder_env ? atoi(forder_env) : 2);
    constexpr int NX = 8;
    // order 4 ("canvas bands"): bucket = the band of BLOCK_ROWS canvas rows an item starts in, zero-fill items
    // included; spans are visited left to right so that a band's items come out in ascending x
    const bool bands = order_mode == 4;
    const int nblk = (bands ? canvas_h : tile_h) / BLOCK_ROWS + 2;   // buckets, +1 slack, +1 for the zero-fill bucket
    std::vector<size_t> span_order(spans.size());
    for (size_t i = 0; i < spans.size(); ++i) span_order[i] = i;
    if (bands)
        std::stable_sort(span_order.begin(), span_order.end(),
                         [&](size_t a, size_t b) { return spans[a].dst_x < spans[b].dst_x; });
    // Overwrite plans cut rows on the canvas' own grid of BLOCK_ROWS rows (not from the top of each span), so that
    // the items either side of a vertical seam cover the same rows and the seam can be given one owner (Seam in
    // common.h); a column remainder narrower than a line is widened at the expense of the piece before it.
    // Items are generated span by span; a 32x32 grid has 2 172 spans but 303 744 items, so the spans' item counts are
    // worked out first (the same cutting rules, counted) and the spans then filled in by a few threads, each span
    // straight into its final range of the list -- the list is the one a single thread would produce.
    const bool ow_mode = mode == SQ_FUSE_OVERWRITE;
    auto rows_of = [&](const Span &sp) { return (!ow_mode && sp.nref >= 2) ? FEATHER_BLEND_ROWS : BLOCK_ROWS; };
    auto col_pieces = [&](const Span &sp) {
        int n = 0;
        for (int c0 = 0; c0 < sp.w; ++n) {
            int cols = std::min(BLOCK_COLS, sp.w - c0);
            const int rest = sp.w - c0 - cols;
            if (ow_mode && rest > 0 && rest < SEAM_MIN_COLS) cols -= SEAM_MIN_COLS;
            c0 += cols;
        }
        return n;
    };
    std::vector<size_t> span_first(spans.size() + 1, 0);
    for (size_t so = 0; so < spans.size(); ++so) {
        const Span &sp = spans[span_order[so]];
        const int item_rows = rows_of(sp);
        const int64_t row_steps = sp.h <= 0 ? 0
                                  : (ow_mode ? ((int64_t)(sp.dst_y + sp.h - 1) / item_rows - sp.dst_y / item_rows + 1)
                                             : ((int64_t)sp.h + item_rows - 1) / item_rows);
        span_first[so + 1] = span_first[so] + (size_t)(row_steps * col_pieces(sp));
    }
    // The big work lists live in per-thread scratch that keeps its capacity between calls: a 32x32 plan needs ~30 MB
    // of them, and fresh pages for that (mmap, first touch, munmap) cost more than filling them.
    // (the worker threads below must see THIS thread's lists: plain references, not the thread_local names)
    static thread_local std::vector<Item> items_tls;
    static thread_local std::vector<Seam> seams_tls;
    static thread_local std::vector<int32_t> by_band_tls;
    std::vector<Item> &items = items_tls;
    std::vector<Seam> &seams = seams_tls;
    std::vector<int32_t> &by_band = by_band_tls;
    items.resize(span_first.back());
    const int hw_threads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
#ifdef SQ_EXPERIMENTS
    const char *threads_env = getenv("SQ_PLAN_THREADS");
    const int nthreads = threads_env ? std::max(1, atoi(threads_env)) : (items.size() >= 32768 ? hw_threads : 1);
#else
    const int nthreads = items.size() >= 32768 ? hw_threads : 1;
#endif
    parallel_ranges(nthreads, spans.size(), [&](int, size_t so_lo, size_t so_hi) {
        for (size_t so = so_lo; so < so_hi; ++so) {
            const size_t si = span_order[so];
            const Span &sp = spans[si];
            const bool ow = ow_mode;
            const Ref *rf = (ow && sp.nref) ? &refs[sp.ref0] : nullptr;
            // feather: spans that several tiles cover are blended (row, 8-pixel group) pair by pair by all threads of a
            // workgroup (fuse.hip blend_item): taller items there, two pairs per thread on a 244-pixel strip
            const int item_rows = rows_of(sp);
            Item *out = items.data() + span_first[so];
            for (int r0 = 0; r0 < sp.h;) {
                const int rows = std::min(ow ? item_rows - (sp.dst_y + r0) % item_rows : item_rows, sp.h - r0);
                for (int c0 = 0; c0 < sp.w;) {
                    int cols = std::min(BLOCK_COLS, sp.w - c0);
                    const int rest = sp.w - c0 - cols;
                    if (ow && rest > 0 && rest < SEAM_MIN_COLS) cols -= SEAM_MIN_COLS;
                    Item it;
                    it.dst_y = sp.dst_y + r0;
                    it.dst_x = sp.dst_x + c0;
                    it.hw = (rows << 16) | cols;
                    it.nref = sp.nref;
                    it.span = (int32_t)si;
                    if (ow) {
                        it.a = rf ? rf->tile : -1;
                        it.b = rf ? rf->src_y + r0 : 0;
                        it.c = rf ? rf->src_x + c0 : 0;
                    } else {
                        it.a = sp.ref0;
                        it.b = r0;
                        it.c = c0;
                    }
                    *out++ = it;
                    c0 += cols;
                }
                r0 += rows;
            }
        }
    });
    const int64_t n_items = (int64_t)items.size();
    const auto t_gen = std::chrono::steady_clock::now();

    // seam owners: item J takes the seam on its left when the item I that ends where J begins covers the same rows
    // and both are at least a line wide (tile or zero fill, either side)
    auto t_band = t_gen, t_sort = t_gen;
    seams.clear();
    if (mode == SQ_FUSE_OVERWRITE) {
        seams.assign(items.size(), Seam{-1, 0, 0, 0});
        const int nb = canvas_h / BLOCK_ROWS + 1;
        std::vector<int32_t> first(nb + 1, 0);
        by_band.resize(items.size());
        // (a counting sort by one thread: splitting it over the planner's threads was measured on the GPU box and did not
        // pay -- two more fork / joins for 0.5 ms of work)
        for (const Item &it : items) ++first[it.dst_y / BLOCK_ROWS + 1];
        for (int k = 0; k < nb; ++k) first[k + 1] += first[k];
        {
            std::vector<int32_t> at(first.begin(), first.end() - 1);
            for (size_t i = 0; i < items.size(); ++i) by_band[at[items[i].dst_y / BLOCK_ROWS]++] = (int32_t)i;
        }
        t_band = std::chrono::steady_clock::now();
        // inside a band: by (first row, END column), so that the item ending where J begins is a binary search away
        // (a 100 x 100 grid has 100+ items per band and millions of items)
        auto end_key = [&](int32_t i) { return ((int64_t)items[i].dst_y << 32) | (uint32_t)(items[i].dst_x + (items[i].hw & 0xFFFF)); };
        parallel_ranges(nthreads, (size_t)nb, [&](int, size_t k_lo, size_t k_hi) {
            for (size_t k = k_lo; k < k_hi; ++k)
                std::sort(by_band.begin() + first[k], by_band.begin() + first[k + 1],
                          [&](int32_t a, int32_t b) { return end_key(a) < end_key(b); });
        });
        t_sort = std::chrono::steady_clock::now();
        // J's record is written by the thread that owns j; the LEAVE_TAIL bit of its left neighbour I belongs to another
        // item's record (which its own thread may be flagging HAS_LEFT at this moment): both go in with atomic ORs
        parallel_ranges(nthreads, items.size(), [&](int, size_t j_lo, size_t j_hi) {
            for (size_t j = j_lo; j < j_hi; ++j) {
                const Item &J = items[j];
                if ((J.hw & 0xFFFF) < SEAM_MIN_COLS || J.dst_x == 0) continue;
                const int k = J.dst_y / BLOCK_ROWS;
                const int64_t want = ((int64_t)J.dst_y << 32) | (uint32_t)J.dst_x;
                auto lo = std::lower_bound(by_band.begin() + first[k], by_band.begin() + first[k + 1], want,
                                           [&](int32_t a, int64_t key) { return end_key(a) < key; });
                if (lo == by_band.begin() + first[k + 1] || end_key(*lo) != want) continue;
                const Item &I = items[*lo];      // the canvas is partitioned: at most one item ends at (row, column)
                const int in = I.hw & 0xFFFF;
                if ((I.hw >> 16) != (J.hw >> 16) || in < SEAM_MIN_COLS) continue;
                seams[j].a = I.a;
                seams[j].b = I.b;
                seams[j].c = I.c + in;
                __atomic_fetch_or(&seams[j].flags, SEAM_HAS_LEFT | (I.nref ? 0 : SEAM_LEFT_ZERO), __ATOMIC_RELAXED);
                __atomic_fetch_or(&seams[*lo].flags, SEAM_LEAVE_TAIL, __ATOMIC_RELAXED);
            }
        });
    }

    auto key_of = [&](const Item &it) {
        if (bands) return std::min(it.dst_y / BLOCK_ROWS, nblk - 2);
        // (experiment builds, SQ_PLAN_ORDER=5: the zero-fill items dealt into the lanes by their canvas row block instead
        // of following them -- fills running beside copies instead of after them)
        if (order_mode == 5 && !it.nref) return (it.dst_y / BLOCK_ROWS) % (nblk - 1);
        // feather items carry their row inside the SPAN (b); the tile row is the first reference's
        if (!ow_mode) return it.nref ? std::min((refs[it.a].src_y + it.b) / BLOCK_ROWS, nblk - 2) : nblk - 1;
        return it.nref ? std::min(it.b / BLOCK_ROWS, nblk - 2) : nblk - 1;
    };
    const bool bucketed = order_mode == 1 || order_mode == 2 || order_mode == 4 || order_mode == 5;
    // bucket sizes, kept per thread range: the k-th item of a bucket (in list order) then knows its rank without a
    // serial pass -- rank = items of the bucket in earlier ranges + its rank inside its own range
    std::vector<std::vector<int64_t>> hist(nthreads, std::vector<int64_t>(nblk, 0));
    parallel_ranges(nthreads, items.size(), [&](int t, size_t lo, size_t hi) {
        std::vector<int64_t> &h = hist[t];
        for (size_t i = lo; i < hi; ++i) ++h[bucketed ? key_of(items[i]) : 0];
    });
    std::vector<int64_t> count(nblk, 0);
    for (int t = 0; t < nthreads; ++t)
        for (int k = 0; k < nblk; ++k) {
            const int64_t c = hist[t][k];
            hist[t][k] = count[k];      // -> rank of the range's first item of bucket k
            count[k] += c;
        }
    const auto t_items = std::chrono::steady_clock::now();

    auto *plan = new sq_fuse_plan;
    TableHeader hd{};
    hd.magic = TABLE_MAGIC;
    hd.mode = mode;
    hd.canvas_h = canvas_h;
    hd.canvas_w = canvas_w;
    hd.tile_h = tile_h;
    hd.tile_w = tile_w;
    hd.n_tiles = n_rects;
    hd.max_refs = max_refs;
    hd.n_spans = (int64_t)spans.size();
    hd.n_refs = (int64_t)refs.size();
    hd.n_items = n_items;
    hd.off_spans = sizeof(TableHeader);
    hd.off_refs = hd.off_spans + hd.n_spans * (int64_t)sizeof(Span);
    hd.off_items = hd.off_refs + hd.n_refs * (int64_t)sizeof(Ref);
    hd.covered_voxels = covered;
    hd.off_seams = seams.empty() ? 0 : hd.off_items + hd.n_items * (int64_t)sizeof(Item);
    const int64_t total = hd.off_items + hd.n_items * (int64_t)sizeof(Item) + (int64_t)(seams.size() * sizeof(Seam));
    if (!plan->table.allocate((size_t)total)) {
        delete plan;
        fail(SQ_ERR_INVALID, "sq_fuse_plan_create: out of host memory for a %lld-byte table", (long long)total);
        return nullptr;
    }
    char *p = plan->table.ptr;
    std::memcpy(p, &hd, sizeof hd);
    if (!spans.empty()) std::memcpy(p + hd.off_spans, spans.data(), spans.size() * sizeof(Span));
    if (!refs.empty()) std::memcpy(p + hd.off_refs, refs.data(), refs.size() * sizeof(Ref));
    Item *dst = reinterpret_cast<Item *>(p + hd.off_items);
    Seam *dst_seam = seams.empty() ? nullptr : reinterpret_cast<Seam *>(p + hd.off_seams);
    auto place = [&](int64_t pos, size_t i) {
        dst[pos] = items[i];
        if (dst_seam) dst_seam[pos] = seams[i];
    };
    if (order_mode == 0) {
        for (size_t i = 0; i < items.size(); ++i) place((int64_t)i, i);
    } else if (order_mode == 3) {
        // canvas raster order: bands of BLOCK_ROWS canvas rows, left to right; zero-fill items in place
        std::vector<size_t> idx(items.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
        const int band_rows = mode == SQ_FUSE_OVERWRITE ? BLOCK_ROWS : FEATHER_BLEND_ROWS;
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
            const int ba = items[a].dst_y / band_rows, bb = items[b].dst_y / band_rows;
            return ba != bb ? ba < bb : items[a].dst_x < items[b].dst_x;
        });
        for (size_t i = 0; i < idx.size(); ++i) place((int64_t)i, idx[i]);
    } else {
        // position of the k-th item of bucket `key` in the row-block-sorted list ...
        std::vector<int64_t> start(nblk + 1, 0);
        for (int k = 0; k < nblk; ++k) start[k + 1] = start[k] + count[k];
        // ... and, for the XCD interleave, its rank inside its lane (blocks of the lane in order)
        std::vector<int64_t> lane_base(nblk, 0), lane_len(NX, 0);
        for (int k = 0; k < nblk - 1; ++k) {
            lane_base[k] = lane_len[k % NX];
            lane_len[k % NX] += count[k];
        }
        int64_t common = lane_len[0];
        for (int x = 1; x < NX; ++x) common = std::min(common, lane_len[x]);
        std::vector<int64_t> tail_at(NX, 0);
        int64_t tail = common * NX;   // the lanes' leftovers follow, lane by lane
        for (int x = 0; x < NX; ++x) {
            tail_at[x] = tail;
            tail += lane_len[x] - common;
        }
        if (order_mode == 2 || order_mode == 4 || order_mode == 5) {   // the header is already in the table: patch the field
            hd.lane_items = common;
            std::memcpy(plan->table.ptr, &hd, sizeof hd);
        }
        parallel_ranges(nthreads, items.size(), [&](int th, size_t lo, size_t hi) {
            std::vector<int64_t> seen = hist[th];
            for (size_t i = lo; i < hi; ++i) {
                const int k = key_of(items[i]);
                const int64_t j = seen[k]++;
                int64_t pos;
                if (k == nblk - 1 || order_mode == 1) {
                    pos = start[k] + j;
                } else {
                    const int x = k % NX;
                    const int64_t t = lane_base[k] + j;
                    pos = t < common ? t * NX + x : tail_at[x] + (t - common);
                }
                place(pos, i);
            }
        });
    }
    const auto t_order = std::chrono::steady_clock::now();
    const size_t n_spans_dbg = spans.size();
#ifdef SQ_EXPERIMENTS
    if (getenv("SQ_PLAN_TIMING")) {
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[plan] sweep %.3f ms, items %.3f, seam bands %.3f, seam sort %.3f, seam lookup + counts %.3f, emit %.3f ms (%zu spans, %lld items, %d threads)\n",
                ms(t_begin, t_sweep), ms(t_sweep, t_gen), ms(t_gen, t_band), ms(t_band, t_sort), ms(t_sort, t_items), ms(t_items, t_order), n_spans_dbg,
                (long long)n_items, nthreads);
    }
#else
    (void)t_begin; (void)t_sweep; (void)t_items; (void)t_order; (void)n_spans_dbg; (void)t_gen; (void)t_band; (void)t_sort;
#endif
    // The per-thread work lists keep their capacity for the next plan of this thread (a 32 x 32 grid's ~30 MB are cheaper to keep
    // than to fault in again), but not without bound: past SCRATCH_KEEP_BYTES they go back, so a thread that once planned a
    // 100 x 100 grid does not hold a third of a GB for the rest of its life.
    constexpr size_t SCRATCH_KEEP_BYTES = (size_t)64 << 20;
    if (items.capacity() * sizeof(Item) + seams.capacity() * sizeof(Seam) + by_band.capacity() * sizeof(int32_t) > SCRATCH_KEEP_BYTES) {
        std::vector<Item>().swap(items);
        std::vector<Seam>().swap(seams);
        std::vector<int32_t>().swap(by_band);
    }
    return plan;
}

// The plan up to its spans; items, seam owners and their order follow on the device (sq_fuse_plan_expand).  The table
// layout is the complete plan's: header | spans | refs | items | seams -- the host copy ends after the refs (plus the
// spans' first item numbers, which the expansion kernels read from their scratch).
static sq_fuse_plan *plan_create_spans_impl(const sq_rect *rects, int32_t n_rects, int32_t tile_h, int32_t tile_w,
                                        int32_t canvas_h, int32_t canvas_w, int32_t mode) {
    if (mode != SQ_FUSE_OVERWRITE) {
        fail(SQ_ERR_UNSUPPORTED, "sq_fuse_plan_create_spans: overwrite plans only (mode %d)", mode);
        return nullptr;
    }
    SpanStage S;
    if (!make_spans(rects, n_rects, tile_h, tile_w, canvas_h, canvas_w, mode, "sq_fuse_plan_create_spans", S)) return nullptr;
    const size_t ns = S.spans.size();
    std::vector<int64_t> span_first(ns + 1, 0);
    for (size_t i = 0; i < ns; ++i) {      // the cutting rules of sq_fuse_plan_create, counted
        const Span &sp = S.spans[i];
        int pieces = 0;
        for (int c0 = 0; c0 < sp.w; ++pieces) {
            int cols = std::min(BLOCK_COLS, sp.w - c0);
            const int rest = sp.w - c0 - cols;
            if (rest > 0 && rest < SEAM_MIN_COLS) cols -= SEAM_MIN_COLS;
            c0 += cols;
        }
        const int64_t row_steps = sp.h <= 0 ? 0 : ((int64_t)(sp.dst_y + sp.h - 1) / BLOCK_ROWS - sp.dst_y / BLOCK_ROWS + 1);
        span_first[i + 1] = span_first[i] + row_steps * pieces;
    }
    const int64_t n_items = span_first[ns];
    if (n_items >= (int64_t(1) << 30)) {
        fail(SQ_ERR_UNSUPPORTED, "sq_fuse_plan_create_spans: %lld items", (long long)n_items);
        return nullptr;
    }
    auto *plan = new sq_fuse_plan;
    TableHeader hd{};
    hd.magic = TABLE_MAGIC;
    hd.mode = mode;
    hd.canvas_h = canvas_h;
    hd.canvas_w = canvas_w;
    hd.tile_h = tile_h;
    hd.tile_w = tile_w;
    hd.n_tiles = n_rects;
    hd.max_refs = S.max_refs;
    hd.n_spans = (int64_t)ns;
    hd.n_refs = (int64_t)S.refs.size();
    hd.n_items = n_items;
    hd.off_spans = sizeof(TableHeader);
    hd.off_refs = hd.off_spans + hd.n_spans * (int64_t)sizeof(Span);
    hd.off_items = hd.off_refs + hd.n_refs * (int64_t)sizeof(Ref);
    hd.covered_voxels = S.covered;
    hd.lane_items = 0;              // known after the expansion
    hd.off_seams = n_items ? hd.off_items + n_items * (int64_t)sizeof(Item) : 0;   // (an overwrite plan without items has no seam records)
    plan->spans_only = true;
    plan->full_bytes = hd.off_items + n_items * (int64_t)(sizeof(Item) + sizeof(Seam));
    plan->off_span_first = (hd.off_items + 7) & ~int64_t(7);
    const int64_t host_bytes = plan->off_span_first + (int64_t)(ns + 1) * 8;
    if (!plan->table.allocate((size_t)host_bytes)) {
        delete plan;
        fail(SQ_ERR_INVALID, "sq_fuse_plan_create_spans: out of host memory for a %lld-byte table", (long long)host_bytes);
        return nullptr;
    }
    char *p = plan->table.ptr;
    std::memset(p, 0, (size_t)host_bytes);
    std::memcpy(p, &hd, sizeof hd);
    if (ns) std::memcpy(p + hd.off_spans, S.spans.data(), ns * sizeof(Span));
    if (!S.refs.empty()) std::memcpy(p + hd.off_refs, S.refs.data(), S.refs.size() * sizeof(Ref));
    std::memcpy(p + plan->off_span_first, span_first.data(), (ns + 1) * 8);
    return plan;
}


sq_fuse_plan *sq_fuse_plan_create(const sq_rect *rects, int32_t n_rects, int32_t tile_h, int32_t tile_w, int32_t canvas_h,
                                  int32_t canvas_w, int32_t mode) {
    return guarded("sq_fuse_plan_create", [&] { return plan_create_impl(rects, n_rects, tile_h, tile_w, canvas_h, canvas_w, mode); });
}

sq_fuse_plan *sq_fuse_plan_create_spans(const sq_rect *rects, int32_t n_rects, int32_t tile_h, int32_t tile_w, int32_t canvas_h,
                                        int32_t canvas_w, int32_t mode) {
    return guarded("sq_fuse_plan_create_spans",
                   [&] { return plan_create_spans_impl(rects, n_rects, tile_h, tile_w, canvas_h, canvas_w, mode); });
}

void sq_fuse_plan_destroy(sq_fuse_plan *plan) { delete plan; }

int sq_fuse_plan_upload(const sq_fuse_plan *plan, void *table_dev, int64_t table_bytes, void *stream_) {
    if (!plan || !table_dev) return fail(SQ_ERR_INVALID, "sq_fuse_plan_upload: NULL argument");
    if (plan->spans_only) return fail(SQ_ERR_INVALID, "sq_fuse_plan_upload: a plan of sq_fuse_plan_create_spans is completed on the device by sq_fuse_plan_expand");
    if (table_bytes < (int64_t)plan->table.size())
        return fail(SQ_ERR_INVALID, "sq_fuse_plan_upload: buffer %lld < table %zu bytes", (long long)table_bytes,
                    plan->table.size());
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // returns when the copy is done: the plan may be destroyed (its storage recycled) right after
    hipError_t e = hipMemcpyAsync(table_dev, plan->table.data(), plan->table.size(), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_fuse_plan_upload: %s", hipGetErrorString(e));
    return SQ_OK;
}

int64_t sq_fuse_plan_table_bytes(const sq_fuse_plan *plan) {
    if (!plan) return fail(SQ_ERR_INVALID, "sq_fuse_plan_table_bytes: NULL plan");
    return plan->device_bytes();
}

int sq_fuse_plan_export(const sq_fuse_plan *plan, void *host_buf, int64_t host_bytes) {
    if (!plan || !host_buf) return fail(SQ_ERR_INVALID, "sq_fuse_plan_export: NULL argument");
    if (plan->spans_only) return fail(SQ_ERR_INVALID, "sq_fuse_plan_export: the items of a plan of sq_fuse_plan_create_spans exist on the device only (read the table back from there)");
    if (host_bytes < (int64_t)plan->table.size())
        return fail(SQ_ERR_INVALID, "sq_fuse_plan_export: buffer %lld < table %zu bytes", (long long)host_bytes,
                    plan->table.size());
    std::memcpy(host_buf, plan->table.data(), plan->table.size());
    return SQ_OK;
}

int sq_fuse_plan_stats(const sq_fuse_plan *plan, int64_t *n_spans, int64_t *n_items, int64_t *covered_voxels,
                       int32_t *max_refs) {
    if (!plan) return fail(SQ_ERR_INVALID, "sq_fuse_plan_stats: NULL plan");
    const TableHeader &h = plan->header();
    if (n_spans) *n_spans = h.n_spans;
    if (n_items) *n_items = h.n_items;
    if (covered_voxels) *covered_voxels = h.covered_voxels;
    if (max_refs) *max_refs = h.max_refs;
    return SQ_OK;
}

}  // extern "C"
