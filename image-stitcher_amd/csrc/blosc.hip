// OME-Zarr chunks leave the GPU compressed: byte shuffle + LZ4 in Blosc-1 frames, the codec the reference's store
// gets by default (zarr.storage.default_compressor = Blosc(cname='lz4', clevel=5, shuffle=SHUFFLE), stitcher.py:814-818).
// Round 1 compressed on host threads (zlib, 0.75 GB/s on 16 cores): the wall of a run from files to the store.
//
// Frame (c-blosc 1.x README_HEADER / README_CHUNK_FORMAT; decoded in tests by an independent pure-Python reader):
//   16-byte header: version 2, versionlz 1, flags = shuffle (0x01, typesize > 1) | don't-split (0x10) | LZ4 (1 << 5),
//                   typesize, nbytes, blocksize, cbytes (little endian)
//   bstarts: int32 per block, offset of the block from the start of the chunk
//   block:   int32 cbytes, then an LZ4 block of the byte-shuffled 16 KiB (cbytes == raw size: stored raw)
// A chunk is cy x cx elements of a (t, c, z) plane, zero-padded past the plane's edge like zarr pads edge chunks;
// all-zero chunks get size 0 (the store's fill_value stands for them: emit_chunk's rule).
//
// Kernels:
//   lz4_blocks_kernel   one WAVE per 16 KiB block: gather the block's pixels (coalesced 2-byte loads), shuffle them into
//                       LDS, then a wave-parallel LZ4: every round the 64 lanes hash the 4 bytes at 64 consecutive
//                       positions against a 2048-entry LDS table, ballot the verified matches, extend them 64 bytes at a
//                       time and emit the sequences (literal runs copied 64 bytes per instruction).
//   chunk_size_kernel / scan_kernel / assemble_kernel
//                       chunk sizes -> exclusive scan -> headers, bstarts and blocks packed densely, so that only the
//                       COMPRESSED bytes cross PCIe.
// HBM-bound integer/byte work; nothing here is GEMM-shaped.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"

using namespace sq;

namespace {

constexpr int BLK = 16384;                       // Blosc block size (bytes)
constexpr int HBITS = 11, HSIZE = 1 << HBITS;
constexpr int SLOT = BLK + BLK / 255 + 96;       // worst-case LZ4 output of one block + slack for the cooperative writers

struct BloscParams {
    const void *planes;
    int64_t plane_stride, pitch;      // elements
    int32_t n_planes, h, w, esz;      // element size 1 or 2
    int32_t cy, cx, ncy, ncx;         // chunk shape, chunks per plane
    int32_t chunk_bytes, nb;          // uncompressed chunk size, blocks per chunk
    int64_t n_chunks, n_blocks;
    uint8_t *slots;                   // [n_blocks][SLOT]
    uint32_t *blk_size;               // [n_blocks]
    uint32_t *chunk_any;              // [n_chunks]
    uint64_t *offsets;                // [n_chunks + 1]: chunk c lives at out[offsets[c] .. offsets[c + 1])
    uint8_t *out;
    int64_t out_capacity;
    uint32_t *status;                 // [0] != 0: out too small
};

__device__ __forceinline__ uint32_t load4(const uint32_t *w, int p) {   // 4 bytes at any byte offset of an LDS word array
    const uint32_t a = w[p >> 2], b = w[(p >> 2) + 1];
    const int sh = (p & 3) * 8;
    return sh ? (a >> sh) | (b << (32 - sh)) : a;
}
__device__ __forceinline__ uint32_t load1(const uint32_t *w, int p) { return (w[p >> 2] >> ((p & 3) * 8)) & 0xFFu; }

// lanes write `count` length-extension bytes of value x (x >= 15 already checked by the caller): 255, 255, ..., rest
__device__ __forceinline__ int put_length(uint8_t *out, int op, int x, int lane) {
    const int rest = x - 15, count = rest / 255 + 1;
    for (int k = lane; k < count; k += 64) out[op + k] = (uint8_t)(k + 1 < count ? 255 : rest % 255);
    return op + count;
}

__global__ __launch_bounds__(256) void lz4_blocks_kernel(const BloscParams P) {
    __shared__ uint32_t src_all[4][BLK / 4 + 8];
    __shared__ uint16_t ht_all[4][HSIZE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t gb = (int64_t)blockIdx.x * 4 + wave;
    if (gb >= P.n_blocks) return;     // whole wave
    uint32_t *src = src_all[wave];
    uint16_t *ht = ht_all[wave];
    uint8_t *srcb = reinterpret_cast<uint8_t *>(src);
    const int64_t chunk = gb / P.nb;
    const int b = (int)(gb - chunk * P.nb);
    const int per_plane = P.ncy * P.ncx;
    const int plane = (int)(chunk / per_plane);
    const int ci = (int)(chunk - (int64_t)plane * per_plane);
    const int cyi = ci / P.ncx, cxi = ci - cyi * P.ncx;
    const int n = min(BLK, P.chunk_bytes - b * BLK);      // bytes of this block
    const int nel = n / P.esz;
    const int e0 = b * (BLK / P.esz);                     // first chunk element of the block
    // ---- gather + byte shuffle into LDS --------------------------------------------------------------------------
    uint32_t any = 0;
    for (int e = lane; e < nel; e += 64) {
        const int ce = e0 + e, row = ce / P.cx, col = ce - row * P.cx;
        const int y = cyi * P.cy + row, x = cxi * P.cx + col;
        uint32_t v = 0;
        if (y < P.h && x < P.w) {
            const int64_t at = (int64_t)plane * P.plane_stride + (int64_t)y * P.pitch + x;
            v = P.esz == 2 ? (uint32_t) static_cast<const uint16_t *>(P.planes)[at] : (uint32_t) static_cast<const uint8_t *>(P.planes)[at];
        }
        any |= v;
        if (P.esz == 2) {
            srcb[e] = (uint8_t)(v & 0xFF);
            srcb[nel + e] = (uint8_t)(v >> 8);
        } else {
            srcb[e] = (uint8_t)v;
        }
    }
    if (lane < 32) src[(n + 3) / 4 + (lane & 7)] = 0;    // defined bytes past the end for load4
    for (int i = lane; i < HSIZE; i += 64) ht[i] = 0xFFFF;
    if (__builtin_amdgcn_ballot_w64(any != 0) && lane == 0) atomicOr(&P.chunk_any[chunk], 1u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // the tail write above must not clobber data: (n + 3) / 4 is the first word wholly past the block
    // ---- LZ4 ----------------------------------------------------------------------------------------------------
    uint8_t *out = P.slots + gb * SLOT;
    const int mflimit = n - 12, matchlimit = n - 5;
    int op = 0, anchor = 0, pos = 0;
    while (pos <= mflimit) {
        const int p = pos + lane;
        const bool valid = p <= mflimit;
        const uint32_t v = valid ? load4(src, p) : 0u;
        const uint32_t hsh = (v * 2654435761u) >> (32 - HBITS);
        const int cand = valid ? (int)ht[hsh] : 0xFFFF;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (valid) ht[hsh] = (uint16_t)p;
        const bool m = valid && cand != 0xFFFF && load4(src, cand) == v;
        uint64_t mask = __builtin_amdgcn_ballot_w64(m);
        int cur = pos;
        while (mask) {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int mp = pos + l;
            if (mp < cur) continue;        // inside the previous match
            const int mc = __shfl(cand, l);
            int len = 4;
            while (true) {                 // extend 64 bytes at a time
                const int a = mp + len + lane;
                const bool eq = a < matchlimit && load1(src, a) == load1(src, mc + len + lane);
                const uint64_t ne = __builtin_amdgcn_ballot_w64(!eq);
                if (ne == 0) {
                    len += 64;
                    continue;
                }
                len += __builtin_ctzll(ne);
                break;
            }
            // sequence: literals [anchor, mp), then the match
            const int lit = mp - anchor, ml = len - 4;
            if (lane == 0) out[op] = (uint8_t)((min(lit, 15) << 4) | min(ml, 15));
            ++op;
            if (lit >= 15) op = put_length(out, op, lit, lane);
            for (int k = lane; k < lit; k += 64) out[op + k] = (uint8_t)load1(src, anchor + k);
            op += lit;
            if (lane == 0) {
                const int off = mp - mc;
                out[op] = (uint8_t)(off & 0xFF);
                out[op + 1] = (uint8_t)(off >> 8);
            }
            op += 2;
            if (ml >= 15) op = put_length(out, op, ml, lane);
            anchor = cur = mp + len;
        }
        pos = max(pos + 64, cur);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    {   // last literals (the format wants at least the last 5 bytes as literals; matches stop at matchlimit)
        const int lit = n - anchor;
        if (lane == 0) out[op] = (uint8_t)(min(lit, 15) << 4);
        ++op;
        if (lit >= 15) op = put_length(out, op, lit, lane);
        for (int k = lane; k < lit; k += 64) out[op + k] = (uint8_t)load1(src, anchor + k);
        op += lit;
    }
    if (op >= n) {   // no gain: the shuffled bytes go out raw (cbytes == raw size tells the decoder)
        for (int k = lane; k < n; k += 64) out[k] = (uint8_t)load1(src, k);
        op = n;
    }
    if (lane == 0) P.blk_size[gb] = (uint32_t)op;
}

__global__ __launch_bounds__(64) void chunk_size_kernel(const BloscParams P, uint64_t *sizes) {
    const int64_t chunk = blockIdx.x;
    uint64_t s = 0;
    for (int b = threadIdx.x; b < P.nb; b += 64) s += 4u + P.blk_size[chunk * P.nb + b];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) sizes[chunk] = P.chunk_any[chunk] ? 16u + 4u * (uint64_t)P.nb + s : 0u;
}

// exclusive scan of n sizes in place -> offsets[0..n], one workgroup (n is a few hundred thousand at most)
__global__ __launch_bounds__(1024) void scan_kernel(uint64_t *v, int64_t n, int64_t capacity, uint32_t *status) {
    __shared__ uint64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024, lo = min(n, tid * per), hi = min(n, lo + per);
    uint64_t s = 0;
    for (int64_t i = lo; i < hi; ++i) s += v[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; ++i) {
            const uint64_t t = part[i];
            part[i] = run;
            run += t;
        }
        v[n] = run;
        if ((int64_t)run > capacity) *status = 1u;
    }
    __syncthreads();
    uint64_t run = part[tid];
    for (int64_t i = lo; i < hi; ++i) {
        const uint64_t t = v[i];
        v[i] = run;
        run += t;
    }
}

__global__ __launch_bounds__(256) void assemble_kernel(const BloscParams P) {
    if (*P.status) return;
    const int64_t chunk = blockIdx.x;
    const uint64_t at = P.offsets[chunk], size = P.offsets[chunk + 1] - at;
    if (size == 0) return;
    uint8_t *dst = P.out + at;
    __shared__ uint32_t bstart[2048];     // chunk_bytes <= 2048 * BLK = 32 MiB
    const int tid = threadIdx.x;
    if (tid == 0) {
        uint32_t run = 16u + 4u * (uint32_t)P.nb;
        for (int b = 0; b < P.nb; ++b) {
            bstart[b] = run;
            run += 4u + P.blk_size[chunk * P.nb + b];
        }
        const uint32_t blocksize = (uint32_t)min(BLK, P.chunk_bytes);
        dst[0] = 2;                                                   // BLOSC_VERSION_FORMAT
        dst[1] = 1;                                                   // LZ4 format version
        dst[2] = (uint8_t)((P.esz > 1 ? 0x01 : 0x00) | 0x10 | (1 << 5));   // shuffle | don't split | LZ4
        dst[3] = (uint8_t)P.esz;
        const uint32_t words[3] = {(uint32_t)P.chunk_bytes, blocksize, (uint32_t)size};
        for (int k = 0; k < 3; ++k)
            for (int j = 0; j < 4; ++j) dst[4 + 4 * k + j] = (uint8_t)(words[k] >> (8 * j));
    }
    __syncthreads();
    for (int b = tid; b < P.nb; b += 256)
        for (int j = 0; j < 4; ++j) dst[16 + 4 * b + j] = (uint8_t)(bstart[b] >> (8 * j));
    for (int b = 0; b < P.nb; ++b) {
        const uint32_t cb = P.blk_size[chunk * P.nb + b];
        uint8_t *o = dst + bstart[b];
        if (tid < 4) o[tid] = (uint8_t)(cb >> (8 * tid));
        const uint8_t *s = P.slots + (chunk * P.nb + b) * SLOT;
        for (uint32_t k = tid; k < cb; k += 256) o[4 + k] = s[k];
    }
}

int geometry(BloscParams &P, int32_t h, int32_t w, int32_t n_planes, int32_t dtype, int32_t cy, int32_t cx) {
    if (h < 1 || w < 1 || n_planes < 0 || cy < 1 || cx < 1) return fail(SQ_ERR_INVALID, "sq_blosc: bad sizes (%dx%d planes %d chunks %dx%d)", h, w, n_planes, cy, cx);
    if (dtype != SQ_U8 && dtype != SQ_U16) return fail(SQ_ERR_UNSUPPORTED, "sq_blosc: dtype %d (uint8 / uint16 planes)", dtype);
    P.esz = dtype == SQ_U16 ? 2 : 1;
    P.h = h;
    P.w = w;
    P.n_planes = n_planes;
    P.cy = cy;                   // as given: the caller clamps a chunk dimension to the ARRAY's (zarr's rule) -- a row band
    P.cx = cx;                   // of a plane is shorter than the array and still has full-height, zero-padded chunks
    P.ncy = (h + P.cy - 1) / P.cy;
    P.ncx = (w + P.cx - 1) / P.cx;
    const int64_t cb = (int64_t)P.cy * P.cx * P.esz;
    if (cb > (int64_t)2048 * BLK) return fail(SQ_ERR_UNSUPPORTED, "sq_blosc: chunks of %lld bytes (at most 32 MiB)", (long long)cb);
    P.chunk_bytes = (int32_t)cb;
    P.nb = (int32_t)((cb + BLK - 1) / BLK);
    P.n_chunks = (int64_t)n_planes * P.ncy * P.ncx;
    P.n_blocks = P.n_chunks * P.nb;
    return SQ_OK;
}
int64_t up256(int64_t v) { return (v + 255) & ~int64_t(255); }

}  // namespace

extern "C" int64_t sq_blosc_chunk_count(int32_t n_planes, int32_t h, int32_t w, int32_t cy, int32_t cx) {
    BloscParams P{};
    if (geometry(P, h, w, n_planes, SQ_U16, cy, cx) != SQ_OK) return SQ_ERR_INVALID;
    return P.n_chunks;
}

extern "C" int64_t sq_blosc_out_bound(int32_t n_planes, int32_t h, int32_t w, int32_t dtype, int32_t cy, int32_t cx) {
    BloscParams P{};
    const int rc = geometry(P, h, w, n_planes, dtype, cy, cx);
    if (rc != SQ_OK) return rc;
    return P.n_chunks * (16 + 8 * (int64_t)P.nb + P.chunk_bytes);
}

extern "C" int64_t sq_blosc_scratch_bytes(int32_t n_planes, int32_t h, int32_t w, int32_t dtype, int32_t cy, int32_t cx) {
    BloscParams P{};
    const int rc = geometry(P, h, w, n_planes, dtype, cy, cx);
    if (rc != SQ_OK) return rc;
    return up256(P.n_blocks * SLOT) + up256(P.n_blocks * 4) + up256(P.n_chunks * 4) + 256;
}

extern "C" int sq_blosc_encode_planes(const void *planes_dev, int64_t plane_stride, int64_t pitch, int32_t n_planes, int32_t h,
                                      int32_t w, int32_t dtype, int32_t chunk_h, int32_t chunk_w, void *scratch_dev,
                                      int64_t scratch_bytes, uint64_t *offsets_dev, void *out_dev, int64_t out_capacity,
                                      uint32_t *status_dev, void *stream_) {
    if (!planes_dev || !scratch_dev || !offsets_dev || !out_dev || !status_dev) return fail(SQ_ERR_INVALID, "sq_blosc_encode_planes: NULL argument");
    BloscParams P{};
    int rc = geometry(P, h, w, n_planes, dtype, chunk_h, chunk_w);
    if (rc != SQ_OK) return rc;
    if (pitch < w || (n_planes > 1 && plane_stride < (int64_t)(h - 1) * pitch + w))
        return fail(SQ_ERR_INVALID, "sq_blosc_encode_planes: pitch / plane stride smaller than the plane");
    if (scratch_bytes < sq_blosc_scratch_bytes(n_planes, h, w, dtype, chunk_h, chunk_w))
        return fail(SQ_ERR_WORKSPACE, "sq_blosc_encode_planes: scratch %lld < %lld bytes", (long long)scratch_bytes,
                    (long long)sq_blosc_scratch_bytes(n_planes, h, w, dtype, chunk_h, chunk_w));
    if (reinterpret_cast<uintptr_t>(scratch_dev) % 256 || reinterpret_cast<uintptr_t>(offsets_dev) % 8)
        return fail(SQ_ERR_INVALID, "sq_blosc_encode_planes: scratch must be 256-byte, offsets 8-byte aligned");
    if (P.n_blocks > (int64_t)4 * 2147483647 || P.n_chunks > 2147483647) return fail(SQ_ERR_UNSUPPORTED, "sq_blosc_encode_planes: too many chunks");
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (P.n_chunks == 0) {
        if (hipMemsetAsync(offsets_dev, 0, 8, s) != hipSuccess) return fail(SQ_ERR_HIP, "sq_blosc_encode_planes: memset");
        return SQ_OK;
    }
    char *sc = static_cast<char *>(scratch_dev);
    P.planes = planes_dev;
    P.plane_stride = plane_stride;
    P.pitch = pitch;
    P.slots = reinterpret_cast<uint8_t *>(sc);
    P.blk_size = reinterpret_cast<uint32_t *>(sc + up256(P.n_blocks * SLOT));
    P.chunk_any = reinterpret_cast<uint32_t *>(sc + up256(P.n_blocks * SLOT) + up256(P.n_blocks * 4));
    P.offsets = offsets_dev;
    P.out = static_cast<uint8_t *>(out_dev);
    P.out_capacity = out_capacity;
    P.status = status_dev;
    if (hipMemsetAsync(P.chunk_any, 0, (size_t)P.n_chunks * 4, s) != hipSuccess || hipMemsetAsync(status_dev, 0, 4, s) != hipSuccess)
        return fail(SQ_ERR_HIP, "sq_blosc_encode_planes: memset");
    hipLaunchKernelGGL(lz4_blocks_kernel, dim3((unsigned)((P.n_blocks + 3) / 4)), dim3(256), 0, s, P);
    hipLaunchKernelGGL(chunk_size_kernel, dim3((unsigned)P.n_chunks), dim3(64), 0, s, P, offsets_dev);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, offsets_dev, P.n_chunks, out_capacity, status_dev);
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)P.n_chunks), dim3(256), 0, s, P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_blosc_encode_planes: launch failed: %s", hipGetErrorString(e));
    return SQ_OK;
}
