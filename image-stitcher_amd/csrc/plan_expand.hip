// Device-side second half of an overwrite plan: work items, seam owners and the order of the work list.
//
// The host planner (plan.cpp) sweeps the rectangles into spans -- a few thousand for a 32x32 grid, 1.2 ms -- and then
// cuts the spans into ~300 000 items, finds each item's left neighbour, and sorts the list into the per-XCD lanes
// the fusion kernel's queues walk: 4 ms more on a few host threads plus a 14.7 MB upload, paid by the first job of a
// run (and by every region of a per-region-registration run).  Everything after the sweep is per-item work with no
// dependence between items beyond two lookups, so here it runs as a handful of kernels on the spans the host uploads
// (~100 KB); the table they leave in device memory is the host planner's BYTE FOR BYTE (tests compare the two).
//
//   expand_items_kernel   item i -> its span (binary search in the spans' first item numbers), row step and column
//                         piece under plan.cpp's cutting rules; its bucket (the block of BLOCK_ROWS tile rows it reads,
//                         or the zero-fill bucket); per-workgroup bucket histogram; (first row, END column) -> i into a
//                         hash table (the canvas is partitioned: keys are unique)
//   seam_owners_kernel    item J looks up the item I that ends where J begins on the same first row; same height and
//                         both a line wide: J owns the seam (Seam in common.h), I leaves its tail line
//   bucket_scan_kernel    exclusive scan of the histograms over the workgroups (the STABLE rank of a workgroup's first
//                         item of each bucket); bucket_tables_kernel: bucket starts, the lanes' lengths, header.lane_items
//   place_items_kernel    rank of every item inside its bucket in list order (wave by wave, ballots), its final position
//                         (lane-interleaved like plan.cpp's order 2), item and seam record written there
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>

#include "common.h"

using namespace sq;

namespace {

constexpr int EXPAND_THREADS = 256;
constexpr int NX = 8;                       // XCD lanes of the work list (plan.cpp)
constexpr int MAX_BUCKETS = 1024;           // tile_h / BLOCK_ROWS + 2 (a tile up to 8 176 rows)
constexpr unsigned long long EMPTY = ~0ull;

struct ExpandParams {
    const Span *spans;
    const Ref *refs;
    const int64_t *span_first;
    int32_t n_spans;
    int64_t n_items;
    Item *items_tmp;
    Seam *seams_tmp;
    int32_t *key;
    unsigned long long *hkeys;
    int32_t *hvals;
    uint32_t hmask;
    int32_t *block_hist;     // [n_blocks][nblk]: counts, then (bucket_tables_kernel) ranks of the block's first items
    int64_t *tables;         // count[nblk] | start[nblk + 1] | lane_base[nblk] | tail_at[NX] | common
    Item *dst;
    Seam *dst_seam;
    TableHeader *header_dev;
    int32_t nblk, n_blocks;
};

__device__ __forceinline__ uint32_t hash64(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    return (uint32_t)k;
}

__device__ __forceinline__ int piece_cols(int w, int c0) {   // plan.cpp's column cutting rule (overwrite plans)
    int cols = min(BLOCK_COLS, w - c0);
    const int rest = w - c0 - cols;
    if (rest > 0 && rest < SEAM_MIN_COLS) cols -= SEAM_MIN_COLS;
    return cols;
}

__global__ __launch_bounds__(EXPAND_THREADS) void expand_items_kernel(ExpandParams P) {
    __shared__ int32_t hist[MAX_BUCKETS];
    const int tid = threadIdx.x;
    for (int k = tid; k < P.nblk; k += EXPAND_THREADS) hist[k] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * EXPAND_THREADS + tid;
    if (i < P.n_items) {
        int lo = 0, hi = P.n_spans;      // the last span s with span_first[s] <= i (spans without items share their number with the next)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (P.span_first[mid] <= i) lo = mid;
            else hi = mid;
        }
        const Span sp = P.spans[lo];
        int pieces = 0;
        for (int c0 = 0; c0 < sp.w; ++pieces) c0 += piece_cols(sp.w, c0);
        const int q = (int)(i - P.span_first[lo]);
        const int step = q / pieces, piece = q - step * pieces;
        const int r0 = step ? (BLOCK_ROWS - sp.dst_y % BLOCK_ROWS) + (step - 1) * BLOCK_ROWS : 0;
        const int rows = min(BLOCK_ROWS - (sp.dst_y + r0) % BLOCK_ROWS, sp.h - r0);
        int c0 = 0;
        for (int j = 0; j < piece; ++j) c0 += piece_cols(sp.w, c0);
        const int cols = piece_cols(sp.w, c0);
        Item it;
        it.dst_y = sp.dst_y + r0;
        it.dst_x = sp.dst_x + c0;
        it.hw = (rows << 16) | cols;
        it.nref = sp.nref;
        it.span = lo;
        if (sp.nref) {
            const Ref rf = P.refs[sp.ref0];
            it.a = rf.tile;
            it.b = rf.src_y + r0;
            it.c = rf.src_x + c0;
        } else {
            it.a = -1;
            it.b = 0;
            it.c = 0;
        }
        P.items_tmp[i] = it;
        P.seams_tmp[i] = Seam{-1, 0, 0, 0};
        const int k = it.nref ? min(it.b / BLOCK_ROWS, P.nblk - 2) : P.nblk - 1;
        P.key[i] = k;
        atomicAdd(&hist[k], 1);
        const unsigned long long hk = ((unsigned long long)(uint32_t)it.dst_y << 32) | (uint32_t)(it.dst_x + cols);
        for (uint32_t slot = hash64(hk) & P.hmask;; slot = (slot + 1) & P.hmask)
            if (atomicCAS(&P.hkeys[slot], EMPTY, hk) == EMPTY) {
                P.hvals[slot] = (int32_t)i;
                break;
            }
    }
    __syncthreads();
    for (int k = tid; k < P.nblk; k += EXPAND_THREADS) P.block_hist[(int64_t)blockIdx.x * P.nblk + k] = hist[k];
}

__global__ __launch_bounds__(EXPAND_THREADS) void seam_owners_kernel(ExpandParams P) {
    const int64_t j = (int64_t)blockIdx.x * EXPAND_THREADS + threadIdx.x;
    if (j >= P.n_items) return;
    const Item J = P.items_tmp[j];
    if ((J.hw & 0xFFFF) < SEAM_MIN_COLS || J.dst_x == 0) return;
    const unsigned long long want = ((unsigned long long)(uint32_t)J.dst_y << 32) | (uint32_t)J.dst_x;
    for (uint32_t slot = hash64(want) & P.hmask;; slot = (slot + 1) & P.hmask) {
        const unsigned long long k = P.hkeys[slot];
        if (k == EMPTY) return;
        if (k != want) continue;
        const int32_t i = P.hvals[slot];
        const Item I = P.items_tmp[i];      // the canvas is partitioned: at most one item ends at (row, column)
        const int in = I.hw & 0xFFFF;
        if ((I.hw >> 16) != (J.hw >> 16) || in < SEAM_MIN_COLS) return;
        P.seams_tmp[j].a = I.a;
        P.seams_tmp[j].b = I.b;
        P.seams_tmp[j].c = I.c + in;
        atomicOr(&P.seams_tmp[j].flags, SEAM_HAS_LEFT | (I.nref ? 0 : SEAM_LEFT_ZERO));
        atomicOr(&P.seams_tmp[i].flags, SEAM_LEAVE_TAIL);
        return;
    }
}

// one workgroup per bucket: exclusive scan of the bucket's counts over the item blocks, in place (a block's entry becomes
// the stable rank of its first item of the bucket), and the bucket's total.  (One THREAD per bucket walking the ~1 200
// blocks was 0.30 of the expansion's 0.46 ms: a chain of dependent loads.)
__global__ __launch_bounds__(EXPAND_THREADS) void bucket_scan_kernel(ExpandParams P) {
    __shared__ int64_t part[EXPAND_THREADS];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int per = (P.n_blocks + EXPAND_THREADS - 1) / EXPAND_THREADS;
    const int b0 = min(tid * per, P.n_blocks), b1 = min(b0 + per, P.n_blocks);
    int64_t sum = 0;
    for (int b = b0; b < b1; ++b) sum += P.block_hist[(int64_t)b * P.nblk + k];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < EXPAND_THREADS; off <<= 1) {      // inclusive scan of the threads' sums
        const int64_t add = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    int64_t run = part[tid] - sum;
    for (int b = b0; b < b1; ++b) {
        int32_t *h = &P.block_hist[(int64_t)b * P.nblk + k];
        const int32_t c = *h;
        *h = (int32_t)run;
        run += c;
    }
    if (tid == EXPAND_THREADS - 1) P.tables[k] = part[tid];      // count[k]
}

// a few hundred buckets: one thread, plan.cpp's arithmetic line by line -- on copies in LDS (the same loops straight on
// global memory took 54 us: a chain of dependent round trips)
__global__ __launch_bounds__(EXPAND_THREADS) void bucket_tables_kernel(ExpandParams P) {
    __shared__ int64_t count[MAX_BUCKETS], start[MAX_BUCKETS + 1], lane_base[MAX_BUCKETS], tail_at[NX + 1];
    const int nblk = P.nblk, tid = threadIdx.x;
    for (int b = tid; b < nblk; b += EXPAND_THREADS) count[b] = P.tables[b];
    __syncthreads();
    if (tid == 0) {
        start[0] = 0;
        for (int b = 0; b < nblk; ++b) start[b + 1] = start[b] + count[b];
        int64_t lane_len[NX] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < nblk - 1; ++b) {
            lane_base[b] = lane_len[b % NX];
            lane_len[b % NX] += count[b];
        }
        lane_base[nblk - 1] = 0;
        int64_t common = lane_len[0];
        for (int x = 1; x < NX; ++x) common = min(common, lane_len[x]);
        int64_t tail = common * NX;   // the lanes' leftovers follow, lane by lane
        for (int x = 0; x < NX; ++x) {
            tail_at[x] = tail;
            tail += lane_len[x] - common;
        }
        tail_at[NX] = common;
        P.header_dev->lane_items = common;
    }
    __syncthreads();
    int64_t *g_start = P.tables + nblk, *g_lane_base = g_start + nblk + 1, *g_tail_at = g_lane_base + nblk;   // tail_at[NX] = common
    for (int b = tid; b <= nblk; b += EXPAND_THREADS) g_start[b] = start[b];
    for (int b = tid; b < nblk; b += EXPAND_THREADS) g_lane_base[b] = lane_base[b];
    for (int x = tid; x <= NX; x += EXPAND_THREADS) g_tail_at[x] = tail_at[x];
}

__global__ __launch_bounds__(EXPAND_THREADS) void place_items_kernel(ExpandParams P) {
    __shared__ int32_t next[MAX_BUCKETS];     // rank the next item of bucket k gets
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < P.nblk; k += EXPAND_THREADS) next[k] = P.block_hist[(int64_t)blockIdx.x * P.nblk + k];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * EXPAND_THREADS + tid;
    const bool live = i < P.n_items;
    const int k = live ? P.key[i] : -1;
    int rank = 0;
    for (int w = 0; w < EXPAND_THREADS / 64; ++w) {      // list order: wave after wave, inside a wave by lane
        if (wave == w) {
            bool todo = live;
            unsigned long long pending;
            while ((pending = __ballot(todo)) != 0) {
                const int first = __ffsll((long long)pending) - 1;
                const int leader = __shfl(k, first);
                const bool same = todo && k == leader;
                const unsigned long long mask = __ballot(same);
                const int base = next[leader];
                if (same) rank = base + __popcll(mask & ((1ull << lane) - 1ull));
                if (lane == first) next[leader] = base + __popcll(mask);
                todo = todo && !same;
            }
        }
        __syncthreads();
    }
    if (!live) return;
    const int nblk = P.nblk;
    const int64_t *start = P.tables + nblk, *lane_base = start + nblk + 1, *tail_at = lane_base + nblk;
    const int64_t common = tail_at[NX];
    int64_t pos;
    if (k == nblk - 1) {
        pos = start[k] + rank;
    } else {
        const int x = k % NX;
        const int64_t t = lane_base[k] + rank;
        pos = t < common ? t * NX + x : tail_at[x] + (t - common);
    }
    P.dst[pos] = P.items_tmp[i];
    P.dst_seam[pos] = P.seams_tmp[i];
}

struct ExpandScratch {
    int64_t span_first, items, seams, key, hkeys, hvals, hist, tables, total;
    uint32_t hslots;
    int32_t n_blocks, nblk;
};

ExpandScratch expand_scratch(const sq_fuse_plan *plan) {
    const TableHeader &h = plan->header();
    ExpandScratch S{};
    S.nblk = h.tile_h / BLOCK_ROWS + 2;
    S.n_blocks = (int32_t)((h.n_items + EXPAND_THREADS - 1) / EXPAND_THREADS);
    uint32_t slots = 1024;
    while ((int64_t)slots < 2 * h.n_items) slots <<= 1;
    S.hslots = slots;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        const int64_t at = off;
        off += (bytes + 255) & ~int64_t(255);
        return at;
    };
    S.span_first = take((h.n_spans + 1) * 8);
    S.items = take(h.n_items * (int64_t)sizeof(Item));
    S.seams = take(h.n_items * (int64_t)sizeof(Seam));
    S.key = take(h.n_items * 4);
    S.hkeys = take((int64_t)slots * 8);
    S.hvals = take((int64_t)slots * 4);
    S.hist = take((int64_t)S.n_blocks * S.nblk * 4);
    S.tables = take((int64_t)(3 * S.nblk + 1 + NX + 1) * 8);
    S.total = off;
    return S;
}

}  // namespace

extern "C" int64_t sq_fuse_plan_expand_scratch_bytes(const sq_fuse_plan *plan) {
    if (!plan || !plan->spans_only) return fail(SQ_ERR_INVALID, "sq_fuse_plan_expand_scratch_bytes: not a plan of sq_fuse_plan_create_spans");
    if (plan->header().tile_h / BLOCK_ROWS + 2 > MAX_BUCKETS)
        return fail(SQ_ERR_UNSUPPORTED, "sq_fuse_plan_expand: tiles of %d rows (at most %d)", plan->header().tile_h, (MAX_BUCKETS - 2) * BLOCK_ROWS);
    return expand_scratch(plan).total;
}

extern "C" int sq_fuse_plan_expand(sq_fuse_plan *plan, void *table_dev, int64_t table_bytes, void *scratch_dev, int64_t scratch_bytes,
                                   void *stream_) {
    if (!plan || !plan->spans_only) return fail(SQ_ERR_INVALID, "sq_fuse_plan_expand: not a plan of sq_fuse_plan_create_spans");
    if (!table_dev || table_bytes < plan->full_bytes)
        return fail(SQ_ERR_INVALID, "sq_fuse_plan_expand: table buffer %lld < %lld bytes", (long long)table_bytes, (long long)plan->full_bytes);
    const TableHeader h = plan->header();
    if (h.tile_h / BLOCK_ROWS + 2 > MAX_BUCKETS)
        return fail(SQ_ERR_UNSUPPORTED, "sq_fuse_plan_expand: tiles of %d rows (at most %d)", h.tile_h, (MAX_BUCKETS - 2) * BLOCK_ROWS);
    const ExpandScratch S = expand_scratch(plan);
    if (h.n_items > 0 && (!scratch_dev || scratch_bytes < S.total))
        return fail(SQ_ERR_WORKSPACE, "sq_fuse_plan_expand: scratch %lld < %lld bytes", (long long)scratch_bytes, (long long)S.total);
    if (reinterpret_cast<uintptr_t>(table_dev) % 16 || reinterpret_cast<uintptr_t>(scratch_dev) % 16)
        return fail(SQ_ERR_INVALID, "sq_fuse_plan_expand: buffers not 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char *table = static_cast<char *>(table_dev), *sc = static_cast<char *>(scratch_dev);
    hipError_t e = hipMemcpyAsync(table, plan->table.data(), (size_t)h.off_items, hipMemcpyHostToDevice, stream);   // header | spans | refs
    if (e == hipSuccess && h.n_items > 0) {
        e = hipMemcpyAsync(sc + S.span_first, plan->table.data() + plan->off_span_first, (size_t)(h.n_spans + 1) * 8, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipMemsetAsync(sc + S.hkeys, 0xFF, (size_t)S.hslots * 8, stream);
        if (e == hipSuccess) {
            ExpandParams P{};
            P.spans = reinterpret_cast<const Span *>(table + h.off_spans);
            P.refs = reinterpret_cast<const Ref *>(table + h.off_refs);
            P.span_first = reinterpret_cast<const int64_t *>(sc + S.span_first);
            P.n_spans = (int32_t)h.n_spans;
            P.n_items = h.n_items;
            P.items_tmp = reinterpret_cast<Item *>(sc + S.items);
            P.seams_tmp = reinterpret_cast<Seam *>(sc + S.seams);
            P.key = reinterpret_cast<int32_t *>(sc + S.key);
            P.hkeys = reinterpret_cast<unsigned long long *>(sc + S.hkeys);
            P.hvals = reinterpret_cast<int32_t *>(sc + S.hvals);
            P.hmask = S.hslots - 1;
            P.block_hist = reinterpret_cast<int32_t *>(sc + S.hist);
            P.tables = reinterpret_cast<int64_t *>(sc + S.tables);
            P.dst = reinterpret_cast<Item *>(table + h.off_items);
            P.dst_seam = reinterpret_cast<Seam *>(table + h.off_seams);
            P.header_dev = reinterpret_cast<TableHeader *>(table);
            P.nblk = S.nblk;
            P.n_blocks = S.n_blocks;
            hipLaunchKernelGGL(expand_items_kernel, dim3(S.n_blocks), dim3(EXPAND_THREADS), 0, stream, P);
            hipLaunchKernelGGL(seam_owners_kernel, dim3(S.n_blocks), dim3(EXPAND_THREADS), 0, stream, P);
            hipLaunchKernelGGL(bucket_scan_kernel, dim3(S.nblk), dim3(EXPAND_THREADS), 0, stream, P);
            hipLaunchKernelGGL(bucket_tables_kernel, dim3(1), dim3(EXPAND_THREADS), 0, stream, P);
            hipLaunchKernelGGL(place_items_kernel, dim3(S.n_blocks), dim3(EXPAND_THREADS), 0, stream, P);
            e = hipGetLastError();
            // the one number the host needs back: how much of the list is lane-interleaved (sq_fuse_planes reads it from
            // the host header)
            if (e == hipSuccess)
                e = hipMemcpyAsync(plan->table.ptr + offsetof(TableHeader, lane_items), table + offsetof(TableHeader, lane_items), 8,
                                   hipMemcpyDeviceToHost, stream);
        }
    }
    // returns when the table is complete: the scratch may be dropped, the plan used at once
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_fuse_plan_expand: %s", hipGetErrorString(e));
    plan->expanded = true;
    return SQ_OK;
}
