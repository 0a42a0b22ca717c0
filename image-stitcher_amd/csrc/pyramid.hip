// Pyramid level kernel for gfx950: level l+1 of the OME-Zarr multiscale image from level l.
//
// Replaces ome_zarr.scale.Scaler(max_layer=n-1).nearest(image) as called by the reference at
// stitcher.py:797-798: every level is skimage.transform.resize(plane, (Y//2, X//2), order=0,
// preserve_range=True, anti_aliasing=False) of the level before.  With order 0 that is a pure
// gather; the sampled source coordinate is (o + 0.5) * in/out - 0.5, rounded half up, which is
// 2*o + 1 for every o whether the size is even or odd, so
//
//     dst[y][x] = src[2*y + 1][2*x + 1],   dst is (src_h / 2) x (src_w / 2)   (floor).
//
// Roofline: HBM.  Algorithmic traffic per output voxel: the odd source rows are read whole
// (cache-line granularity: the even columns come along), 2*sizeof(T) B, + sizeof(T) B written
// = 6 B per uint16 output voxel = 1.5 B per source voxel.
//
// Mapping: one wave per output row, grid-stride over (plane, row).  A wave step reads 2 KiB of the
// source row as two contiguous 1 KiB loads (16 bytes per lane, at whatever 2-byte phase the row
// has), keeps the odd elements with v_perm_b32 and writes two contiguous 512-byte runs.  Stores
// start on a 16-byte boundary of the destination row (its pitch is arbitrary); the few elements
// before and after the aligned body go out one by one.
#include <hip/hip_runtime.h>

#include "common.h"

using namespace sq;

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed)) U32x4U {
    u32x4 v;
};
#define SQ_GLOBAL __attribute__((address_space(1)))

template <typename T>
struct OddSel;
template <>
struct OddSel<uint16_t> {   // bytes {2,3} of the low word, {2,3} of the high word
    static constexpr uint32_t sel = 0x07060302u;
};
template <>
struct OddSel<uint8_t> {    // bytes {1,3} of the low word, {1,3} of the high word
    static constexpr uint32_t sel = 0x07050301u;
};

template <typename T>
__global__ __launch_bounds__(256) void downsample2_kernel(const T *src, int64_t src_plane_stride, int64_t src_pitch,
                                                          T *dst, int64_t dst_plane_stride, int64_t dst_pitch, int32_t dst_h,
                                                          int32_t dst_w, int64_t n_rows) {
    constexpr int VEC = 16 / (int)sizeof(T);   // outputs per lane per step
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t plane = r / dst_h;
        const int y = (int)(r - plane * dst_h);
        const T *srow = src + plane * src_plane_stride + (int64_t)(2 * y + 1) * src_pitch;
        T *drow = dst + plane * dst_plane_stride + (int64_t)y * dst_pitch;
        // elements before the first 16-byte boundary of the destination row
        const int mis = (int)((reinterpret_cast<uintptr_t>(drow) / sizeof(T)) & (VEC - 1));
        const int lead = min(dst_w, (VEC - mis) & (VEC - 1));
        const int n_vec = (dst_w - lead) / VEC;
        const int tail0 = lead + n_vec * VEC;
        if (lane < lead) drow[lane] = *(const SQ_GLOBAL T *)(srow + 2 * lane + 1);
        if (lane < dst_w - tail0) drow[tail0 + lane] = *(const SQ_GLOBAL T *)(srow + 2 * (tail0 + lane) + 1);
        // body: a wave step covers 2 * 64 * VEC source elements with two fully contiguous 1 KiB loads
        // and two contiguous 512-byte stores (lane l keeps the odd elements of its 16 bytes: 8 bytes out)
        const int n_step = n_vec / 64;
        for (int s = 0; s < n_step; ++s) {
            const int x0 = lead + s * 64 * VEC;                 // first output of this step
            const T *sp = srow + 2 * x0 + lane * VEC;
            const u32x4 a = ((const SQ_GLOBAL U32x4U *)sp)->v;
            const u32x4 b = ((const SQ_GLOBAL U32x4U *)(sp + 64 * VEC))->v;
            u32x2 oa, ob;
            oa[0] = __builtin_amdgcn_perm(a[1], a[0], OddSel<T>::sel);
            oa[1] = __builtin_amdgcn_perm(a[3], a[2], OddSel<T>::sel);
            ob[0] = __builtin_amdgcn_perm(b[1], b[0], OddSel<T>::sel);
            ob[1] = __builtin_amdgcn_perm(b[3], b[2], OddSel<T>::sel);
            T *dp = drow + x0 + lane * (VEC / 2);
            __builtin_nontemporal_store(oa, (SQ_GLOBAL u32x2 *)dp);
            __builtin_nontemporal_store(ob, (SQ_GLOBAL u32x2 *)(dp + 32 * VEC));
        }
        // remainder (< 64 vectors): a lane reads 32 contiguous bytes and stores 16
        {
            const int v = n_step * 64 + lane;
            if (v < n_vec) {
                const int x0 = lead + v * VEC;
                const u32x4 a = ((const SQ_GLOBAL U32x4U *)(srow + 2 * x0))->v;
                const u32x4 b = ((const SQ_GLOBAL U32x4U *)(srow + 2 * x0 + VEC))->v;
                u32x4 o;
                o[0] = __builtin_amdgcn_perm(a[1], a[0], OddSel<T>::sel);
                o[1] = __builtin_amdgcn_perm(a[3], a[2], OddSel<T>::sel);
                o[2] = __builtin_amdgcn_perm(b[1], b[0], OddSel<T>::sel);
                o[3] = __builtin_amdgcn_perm(b[3], b[2], OddSel<T>::sel);
                __builtin_nontemporal_store(o, (SQ_GLOBAL u32x4 *)(drow + x0));
            }
        }
    }
}

}   // namespace

extern "C" int sq_downsample2(const void *src_dev, int64_t src_plane_stride, int32_t src_h, int32_t src_w, int64_t src_pitch,
                              void *dst_dev, int64_t dst_plane_stride, int64_t dst_pitch, int32_t n_planes, int32_t dtype,
                              void *stream_) {
    if (n_planes < 0 || src_h < 0 || src_w < 0 || src_pitch < src_w || dst_pitch < src_w / 2)
        return fail(SQ_ERR_INVALID, "sq_downsample2: bad sizes (planes=%d src=%dx%d pitch %lld dst pitch %lld)", n_planes, src_h,
                    src_w, (long long)src_pitch, (long long)dst_pitch);
    if (dtype != SQ_U8 && dtype != SQ_U16) return fail(SQ_ERR_UNSUPPORTED, "sq_downsample2: dtype %d", dtype);
    const int32_t dst_h = src_h / 2, dst_w = src_w / 2;
    const int64_t n_rows = (int64_t)n_planes * dst_h;
    if (n_rows == 0 || dst_w == 0) return SQ_OK;
    if (!src_dev || !dst_dev) return fail(SQ_ERR_INVALID, "sq_downsample2: NULL buffer");
    if (n_planes > 1 && (src_plane_stride < (int64_t)src_h * src_pitch || dst_plane_stride < (int64_t)dst_h * dst_pitch))
        return fail(SQ_ERR_INVALID, "sq_downsample2: plane strides smaller than a plane");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // enough waves to fill 256 CUs several times over; the grid stride covers the rest
    const int64_t blocks = std::min<int64_t>((n_rows + 3) / 4, 256 * 32);
    if (dtype == SQ_U16)
        downsample2_kernel<uint16_t><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(
            static_cast<const uint16_t *>(src_dev), src_plane_stride, src_pitch, static_cast<uint16_t *>(dst_dev),
            dst_plane_stride, dst_pitch, dst_h, dst_w, n_rows);
    else
        downsample2_kernel<uint8_t><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(
            static_cast<const uint8_t *>(src_dev), src_plane_stride, src_pitch, static_cast<uint8_t *>(dst_dev),
            dst_plane_stride, dst_pitch, dst_h, dst_w, n_rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_downsample2: launch failed: %s", hipGetErrorString(e));
    return SQ_OK;
}
