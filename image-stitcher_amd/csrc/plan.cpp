// Host-side span planner for fusion.
//
// The reference places tiles one by one into the canvas (stitcher.py:652-681, :583-598); where
// rectangles overlap the later file wins.  Here the same result is described as a partition of
// the canvas: sweep the distinct y edges into bands, inside a band sweep the distinct x edges
// into intervals, give every interval its owner (overwrite: the last rect that covers it;
// feather: all rects that cover it, in write order), merge equal neighbours horizontally and
// then vertically.  The device then writes every canvas voxel exactly once.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
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

struct Clipped {
    int y0, y1, x0, x1;  // canvas extent [y0,y1) x [x0,x1)
    int src_y, src_x;    // tile pixel at (y0, x0)
    int tile;
};

struct OpenSpan {
    int64_t index;  // into spans
    int y_end;
};

struct Key {
    int xa, xb;
    std::vector<int> owners;
    bool operator<(const Key &o) const {
        if (xa != o.xa) return xa < o.xa;
        if (xb != o.xb) return xb < o.xb;
        return owners < o.owners;
    }
};

}  // namespace

extern "C" {

int sq_version(void) { return SQ_VERSION; }
const char *sq_last_error(void) { return last_error_ref().c_str(); }

sq_fuse_plan *sq_fuse_plan_create(const sq_rect *rects, int32_t n_rects, int32_t tile_h, int32_t tile_w,
                                  int32_t canvas_h, int32_t canvas_w, int32_t mode) {
    if ((!rects && n_rects > 0) || n_rects < 0 || tile_h <= 0 || tile_w <= 0 || canvas_h <= 0 || canvas_w <= 0) {
        fail(SQ_ERR_INVALID, "sq_fuse_plan_create: bad sizes (n_rects=%d tile=%dx%d canvas=%dx%d)", n_rects, tile_h,
             tile_w, canvas_h, canvas_w);
        return nullptr;
    }
    if (mode != SQ_FUSE_OVERWRITE && mode != SQ_FUSE_FEATHER) {
        fail(SQ_ERR_INVALID, "sq_fuse_plan_create: unknown mode %d", mode);
        return nullptr;
    }
    // Canvas clip of stitcher.py:590-594 (python slice semantics) + validation of the source side.
    std::vector<Clipped> cl;
    cl.reserve(n_rects);
    for (int i = 0; i < n_rects; ++i) {
        const sq_rect &r = rects[i];
        if (r.h < 0 || r.w < 0 || r.src_y0 < 0 || r.src_x0 < 0 || r.src_y0 + (int64_t)r.h > tile_h ||
            r.src_x0 + (int64_t)r.w > tile_w) {
            fail(SQ_ERR_INVALID, "sq_fuse_plan_create: rect %d reads outside its %dx%d tile (src %d,%d size %dx%d)", i,
                 tile_h, tile_w, r.src_y0, r.src_x0, r.h, r.w);
            return nullptr;
        }
        if (r.dst_y < 0 || r.dst_x < 0) {
            // The reference would wrap a negative index (python slicing); its placements are
            // never negative on this path, so refuse instead of guessing.
            fail(SQ_ERR_INVALID, "sq_fuse_plan_create: rect %d has a negative canvas offset (%d,%d)", i, r.dst_y,
                 r.dst_x);
            return nullptr;
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
    size_t next = 0;
    std::vector<int> active;  // indices into cl, ascending (= write order)

    std::vector<Span> spans;
    std::vector<Ref> refs;
    std::map<Key, OpenSpan> open, open_next;
    int max_refs = 0;
    int64_t covered = 0;

    std::vector<int> xs;
    std::vector<int> owners;
    for (size_t b = 0; b + 1 < ys.size(); ++b) {
        const int ya = ys[b], yb = ys[b + 1];
        while (next < by_y0.size() && cl[by_y0[next]].y0 <= ya) {
            active.insert(std::upper_bound(active.begin(), active.end(), by_y0[next]), by_y0[next]);
            ++next;
        }
        active.erase(std::remove_if(active.begin(), active.end(), [&](int i) { return cl[i].y1 <= ya; }), active.end());

        xs.assign({0, canvas_w});
        for (int i : active) {
            xs.push_back(cl[i].x0);
            xs.push_back(cl[i].x1);
        }
        std::sort(xs.begin(), xs.end());
        xs.erase(std::unique(xs.begin(), xs.end()), xs.end());

        open_next.clear();
        Key cur;
        bool have = false;
        auto flush = [&]() {
            if (!have) return;
            auto it = open.find(cur);
            if (it != open.end() && it->second.y_end == ya) {
                Span &s = spans[it->second.index];
                s.h += yb - ya;
                open_next[cur] = {it->second.index, yb};
            } else {
                Span s{};
                s.dst_y = ya;
                s.dst_x = cur.xa;
                s.h = yb - ya;
                s.w = cur.xb - cur.xa;
                s.nref = (int)cur.owners.size();
                s.ref0 = (int)refs.size();
                for (int o : cur.owners) {
                    const Clipped &c = cl[o];
                    refs.push_back({c.tile, c.src_y + (ya - c.y0), c.src_x + (cur.xa - c.x0), 0});
                }
                open_next[cur] = {(int64_t)spans.size(), yb};
                spans.push_back(s);
            }
            if (!cur.owners.empty()) covered += (int64_t)(yb - ya) * (cur.xb - cur.xa);
            max_refs = std::max(max_refs, (int)cur.owners.size());
            have = false;
        };
        for (size_t k = 0; k + 1 < xs.size(); ++k) {
            const int xa = xs[k], xb = xs[k + 1];
            owners.clear();
            for (int i : active)
                if (cl[i].x0 <= xa && cl[i].x1 >= xb) owners.push_back(i);
            if (mode == SQ_FUSE_OVERWRITE && owners.size() > 1) owners.erase(owners.begin(), owners.end() - 1);
            if ((int)owners.size() > MAX_REFS) {
                fail(SQ_ERR_UNSUPPORTED, "sq_fuse_plan_create: %zu tiles overlap at canvas (%d,%d); feather supports %d",
                     owners.size(), ya, xa, MAX_REFS);
                return nullptr;
            }
            if (have && cur.owners == owners && cur.xb == xa) {
                cur.xb = xb;
            } else {
                flush();
                cur.xa = xa;
                cur.xb = xb;
                cur.owners = owners;
                have = true;
            }
        }
        flush();
        open.swap(open_next);
    }

    std::vector<Item> items;
    for (size_t s = 0; s < spans.size(); ++s) {
        const Span &sp = spans[s];
        for (int r0 = 0; r0 < sp.h; r0 += BLOCK_ROWS)
            for (int c0 = 0; c0 < sp.w; c0 += BLOCK_COLS) {
                Item it{};
                it.dst_y = sp.dst_y + r0;
                it.dst_x = sp.dst_x + c0;
                it.hw = (std::min(BLOCK_ROWS, sp.h - r0) << 16) | std::min(BLOCK_COLS, sp.w - c0);
                it.nref = sp.nref;
                it.span = (int32_t)s;
                if (mode == SQ_FUSE_OVERWRITE) {
                    if (sp.nref) {
                        const Ref &rf = refs[sp.ref0];
                        it.a = rf.tile;
                        it.b = rf.src_y + r0;
                        it.c = rf.src_x + c0;
                    } else {
                        it.a = -1;
                    }
                } else {
                    it.a = sp.ref0;
                    it.b = r0;
                    it.c = c0;
                }
                items.push_back(it);
            }
    }

    // Overwrite mode: walk the tiles "row-synchronously", one tile-row block per XCD.
    // Items are grouped by the block of BLOCK_ROWS tile rows they read (all tiles, in tile order).
    // Workgroup b of the persistent grid takes items b, b + G, ... and workgroups are dealt to the
    // 8 XCDs round-robin, so list position i is served by XCD i % 8 (observed placement: a speed
    // matter only).  Row block r is therefore laid out on positions == r (mod 8): the ~128
    // workgroups resident on one XCD all read the same 8 rows of the flatfield at the same time,
    // which are fetched into that XCD's L2 once instead of once per tile.  Every item still moves
    // whole row segments, so HBM sees the same contiguous runs, in a different order.
    //   SQ_PLAN_ORDER=0 keeps span order, 1 = row blocks without the XCD interleave (experiments).
    const char *order_env = getenv("SQ_PLAN_ORDER");
    const int order_mode = order_env ? atoi(order_env) : 2;
    if (mode == SQ_FUSE_OVERWRITE && order_mode > 0) {
        std::stable_sort(items.begin(), items.end(), [](const Item &x, const Item &y) {
            const bool zx = x.nref == 0, zy = y.nref == 0;
            if (zx != zy) return zy;                       // covered items first, zero-fill last
            if (zx) return false;
            const int bx = x.b / BLOCK_ROWS, by = y.b / BLOCK_ROWS;
            if (bx != by) return bx < by;
            return false;                                  // stable: keeps tile / span order inside a row block
        });
        if (order_mode > 1) {
            constexpr int NX = 8;
            std::vector<Item> lane[NX], rest;
            for (const Item &it : items) {
                if (it.nref) lane[(it.b / BLOCK_ROWS) % NX].push_back(it);
                else rest.push_back(it);
            }
            std::vector<Item> out;
            out.reserve(items.size());
            size_t common = lane[0].size();
            for (int x = 1; x < NX; ++x) common = std::min(common, lane[x].size());
            for (size_t k = 0; k < common; ++k)
                for (int x = 0; x < NX; ++x) out.push_back(lane[x][k]);
            for (int x = 0; x < NX; ++x) out.insert(out.end(), lane[x].begin() + common, lane[x].end());
            out.insert(out.end(), rest.begin(), rest.end());
            items.swap(out);
        }
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
    hd.max_refs = max_refs;
    hd.n_spans = (int64_t)spans.size();
    hd.n_refs = (int64_t)refs.size();
    hd.n_items = (int64_t)items.size();
    hd.off_spans = sizeof(TableHeader);
    hd.off_refs = hd.off_spans + hd.n_spans * (int64_t)sizeof(Span);
    hd.off_items = hd.off_refs + hd.n_refs * (int64_t)sizeof(Ref);
    hd.covered_voxels = covered;
    const int64_t total = hd.off_items + hd.n_items * (int64_t)sizeof(Item);
    plan->table.resize((size_t)total);
    char *p = &plan->table[0];
    std::memcpy(p, &hd, sizeof hd);
    if (!spans.empty()) std::memcpy(p + hd.off_spans, spans.data(), spans.size() * sizeof(Span));
    if (!refs.empty()) std::memcpy(p + hd.off_refs, refs.data(), refs.size() * sizeof(Ref));
    if (!items.empty()) std::memcpy(p + hd.off_items, items.data(), items.size() * sizeof(Item));
    return plan;
}

void sq_fuse_plan_destroy(sq_fuse_plan *plan) { delete plan; }

int64_t sq_fuse_plan_table_bytes(const sq_fuse_plan *plan) {
    if (!plan) return fail(SQ_ERR_INVALID, "sq_fuse_plan_table_bytes: NULL plan");
    return (int64_t)plan->table.size();
}

int sq_fuse_plan_export(const sq_fuse_plan *plan, void *host_buf, int64_t host_bytes) {
    if (!plan || !host_buf) return fail(SQ_ERR_INVALID, "sq_fuse_plan_export: NULL argument");
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
