// Device twin of image-stitcher_amd/synth.py: the seeded scene + per-tile noise generator,
// in wrapping uint64 arithmetic, so bench/test tiles can be produced in HBM without a host
// round trip and still equal the numpy generator bit for bit.  Bench/test support only.
#include <hip/hip_runtime.h>

#include "common.h"

using namespace sq;

namespace {

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t hash2d(uint64_t seed, int64_t y, int64_t x) {
    const uint64_t a = mix(seed + 0x9E3779B97F4A7C15ull * (uint64_t)y);
    return mix(a ^ (0xD1B54A32D192ED03ull * (uint64_t)x));
}
__device__ __forceinline__ int64_t cell(uint64_t seed, int64_t y, int64_t x) {
    return (int64_t)((hash2d(seed, y, x) >> 33) % 10000ull);
}

template <typename T>
__global__ __launch_bounds__(256) void synth_kernel(const sq_synth_tile *tiles, int tile_h, int tile_w, int noise_amp,
                                                    T *out) {
    const sq_synth_tile t = tiles[blockIdx.z];
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= tile_w) return;
    const int64_t Y = t.oy + y, X = t.ox + x;
    int64_t v = 1000 + cell(t.scene_seed, Y, X) + cell(t.scene_seed, Y + 1, X) + cell(t.scene_seed, Y, X + 1) +
                cell(t.scene_seed, Y + 1, X + 1);
    if (noise_amp > 0) v += (int64_t)((hash2d(t.noise_seed, y, x) >> 33) % (uint64_t)(2 * noise_amp + 1)) - noise_amp;
    if (sizeof(T) == 1) v >>= 8;
    out[((int64_t)blockIdx.z * tile_h + y) * tile_w + x] = (T)v;
}

}  // namespace

extern "C" int sq_synth_tiles(const sq_synth_tile *tiles_dev, int32_t n_tiles, int32_t tile_h, int32_t tile_w,
                              int32_t noise_amp, int32_t tile_dtype, void *out_dev, void *stream) {
    if (!tiles_dev || !out_dev || n_tiles < 0 || tile_h <= 0 || tile_w <= 0 || tile_h > 65535 || n_tiles > 65535)
        return fail(SQ_ERR_INVALID, "sq_synth_tiles: bad arguments (n_tiles=%d %dx%d)", n_tiles, tile_h, tile_w);
    if (n_tiles == 0) return SQ_OK;
    dim3 grid((tile_w + 255) / 256, tile_h, n_tiles);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (tile_dtype == SQ_U16)
        hipLaunchKernelGGL(synth_kernel<uint16_t>, grid, dim3(256), 0, s, tiles_dev, tile_h, tile_w, noise_amp,
                           static_cast<uint16_t *>(out_dev));
    else if (tile_dtype == SQ_U8)
        hipLaunchKernelGGL(synth_kernel<uint8_t>, grid, dim3(256), 0, s, tiles_dev, tile_h, tile_w, noise_amp,
                           static_cast<uint8_t *>(out_dev));
    else
        return fail(SQ_ERR_UNSUPPORTED, "sq_synth_tiles: dtype %d", tile_dtype);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_synth_tiles: launch failed: %s", hipGetErrorString(e));
    return SQ_OK;
}
