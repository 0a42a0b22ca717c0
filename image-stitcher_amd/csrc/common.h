// Shared internals of libsquidstitch: error reporting and the device table layout.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/squidstitch.h"

namespace sq {

std::string &last_error_ref();

inline int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

// ---- fusion table (host builds it, device reads it) -------------------------------------
// One plane of the canvas is cut into disjoint spans.  A span is a rectangle of the canvas
// whose every voxel is produced from the same ordered list of tiles ("refs"): none (zero
// fill), one (overwrite mode: the last writer), or several (feather mode).
constexpr uint32_t TABLE_MAGIC = 0x53514654u;  // "SQFT"
#ifndef SQ_BLOCK_ROWS
#define SQ_BLOCK_ROWS 8
#endif
#ifndef SQ_BLOCK_COLS
#define SQ_BLOCK_COLS 2048
#endif
constexpr int BLOCK_ROWS = SQ_BLOCK_ROWS;      // rows of a span one workgroup takes (multiple of 4: one wave per row)
constexpr int BLOCK_COLS = SQ_BLOCK_COLS;      // columns of a span one workgroup takes
constexpr int MAX_REFS = 8;                    // feather: most tiles blended in one span
#ifndef SQ_FEATHER_BLEND_ROWS
#define SQ_FEATHER_BLEND_ROWS 16
#endif
constexpr int FEATHER_BLEND_ROWS = SQ_FEATHER_BLEND_ROWS;   // feather: rows of an item of a span several tiles cover: a 244-pixel
                                               // strip then gives every thread of the workgroup two (row, group) pairs
                                               // (measured 8 / 16 / 32 rows: 0.495 / 0.519 / 0.49 of peak, r02_exp9_feather.log)

struct TableHeader {
    uint32_t magic;
    int32_t mode;
    int32_t canvas_h, canvas_w;
    int32_t tile_h, tile_w;
    int32_t n_tiles, max_refs;
    int64_t n_spans, n_refs, n_items;
    int64_t off_spans, off_refs, off_items;  // byte offsets from the table start
    int64_t covered_voxels;
    int64_t lane_items;   // list positions [0, 8 * lane_items) are lane-interleaved: position = rank * 8 + lane (else 0)
    int64_t off_seams;    // n_items Seam records, item order (overwrite plans; 0: none)
};
static_assert(sizeof(TableHeader) == 104, "TableHeader layout");

struct Span {
    int32_t dst_y, dst_x, h, w;
    int32_t nref, ref0;
    int32_t pad[2];
};
static_assert(sizeof(Span) == 32, "Span layout");

struct Ref {
    int32_t tile;          // index into the plane's tile table
    int32_t src_y, src_x;  // tile pixel that lands on the span's (0, 0)
    int32_t pad;
};
static_assert(sizeof(Ref) == 16, "Ref layout");

// One unit of device work: up to BLOCK_ROWS x BLOCK_COLS of one span, self-contained so that a
// workgroup needs ONE descriptor load before it can start streaming (no span/ref indirection).
//   overwrite: nref in {0,1}; a = tile index, b/c = tile pixel (y, x) that lands on (dst_y, dst_x)
//   feather  : a = first ref of the span, b/c = (row0, col0) of this block inside the span
struct Item {
    int32_t dst_y, dst_x;
    int32_t hw;  // rows << 16 | cols
    int32_t nref;
    int32_t a, b, c;
    int32_t span;
};
static_assert(sizeof(Item) == 32, "Item layout");

// Who writes the 128-byte canvas line a vertical seam falls in (overwrite plans).  Two items that meet at a seam
// (same canvas rows, both at least SEAM_MIN_COLS wide) would each write their part of that line: two partial-line
// writes, measured 4 % of the whole launch (tools/membw.hip "2d", profiles/r02_membw_2d.log).  Instead the RIGHT
// item writes the whole line -- its own pixels and, from the left item's source, the pixels before its first --
// and the left item stops at the line boundary.  Which pixels that is depends on the address of the row (canvas
// pitch and base), so it is worked out per row on the device; the plan only says who owns the seam.
//   flags & SEAM_HAS_LEFT  : this item also writes the pixels between the line boundary and its dst_x, taken from
//                            tile `a` (row b + r, columns ending at c: the pixel at dst_x - j is column c - j), or
//                            zeros with SEAM_LEFT_ZERO
//   flags & SEAM_LEAVE_TAIL: the line this item's last pixel falls in belongs to its right neighbour, unless the
//                            item ends exactly on a line boundary
// A kernel either honours the flags for every item of a plane or for none (both are a partition of the canvas).
constexpr int SEAM_MIN_COLS = 64;      // one line of uint16: head line and tail line of an item are then distinct
enum : int32_t { SEAM_HAS_LEFT = 1, SEAM_LEAVE_TAIL = 2, SEAM_LEFT_ZERO = 4 };
struct Seam {
    int32_t a, b, c;
    int32_t flags;
};
static_assert(sizeof(Seam) == 16, "Seam layout");

}  // namespace sq

// TableHeader + spans + refs + items, exactly what goes to the device.  The builder writes every byte
// itself, so nothing is zero-filled first.  Storage is page-locked host memory when a device is
// present (the upload is then one DMA at link speed instead of a staged pageable copy), taken from a
// small pool so that building a plan per region does not page-lock memory each time; plain malloc
// without a device (the CPU tests).
void *sq_table_acquire(size_t bytes, size_t *capacity, bool *pinned);   // plan.cpp
void sq_table_release(void *ptr, size_t capacity, bool pinned);

struct sq_table {
    char *ptr = nullptr;
    size_t bytes = 0, capacity = 0;
    bool pinned = false;
    sq_table() = default;
    sq_table(const sq_table &) = delete;
    sq_table &operator=(const sq_table &) = delete;
    ~sq_table() { reset(); }
    void reset() {
        if (ptr) sq_table_release(ptr, capacity, pinned);
        ptr = nullptr;
        bytes = capacity = 0;
    }
    bool allocate(size_t n) {
        reset();
        ptr = static_cast<char *>(sq_table_acquire(n ? n : 1, &capacity, &pinned));
        bytes = ptr ? n : 0;
        return ptr != nullptr;
    }
    size_t size() const { return bytes; }
    const char *data() const { return ptr; }
};

struct sq_fuse_plan {
    sq_table table;
    const sq::TableHeader &header() const { return *reinterpret_cast<const sq::TableHeader *>(table.data()); }
    // A plan made by sq_fuse_plan_create_spans holds header + spans + refs (+ the spans' first item numbers) on the host;
    // its items, seam records and their order are produced on the device by sq_fuse_plan_expand (plan_expand.hip).
    bool spans_only = false;       // made by sq_fuse_plan_create_spans
    bool expanded = false;         // sq_fuse_plan_expand has run: header().lane_items is valid, the device table complete
    int64_t full_bytes = 0;        // size of the complete table (what the device buffer holds)
    int64_t off_span_first = 0;    // host table: (n_spans + 1) int64 item numbers, after the refs
    int64_t device_bytes() const { return spans_only ? full_bytes : (int64_t)table.size(); }
};
