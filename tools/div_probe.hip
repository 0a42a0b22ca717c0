// Which shortened float32 divide still yields the IEEE quotient bit for bit, for the operand classes of the feather blend?
//   class A: integer numerator 0..65535 / gain (every mantissa of a binade)           -- v_k = n_k / g_k
//   class B: acc (every mantissa of a binade) / weight sum (integer 2..16384)          -- acc / wsum
// Candidates: reciprocal R1 = v_rcp_f32 + one Newton step, R2 = + two; quotient Q1 = n r, ONE exact-residual correction (3 slots),
// Q2 = two corrections (5 slots, what ships).      hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o div_probe div_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __forceinline__ float R1(float g) { float r = __builtin_amdgcn_rcpf(g); return fmaf(fmaf(-g, r, 1.0f), r, r); }
__device__ __forceinline__ float R2(float g) { float r = R1(g); return fmaf(fmaf(-g, r, 1.0f), r, r); }
__device__ __forceinline__ float Q1(float n, float g, float r) { float q = n * r; return fmaf(fmaf(-g, q, n), r, q); }
__device__ __forceinline__ float Q2(float n, float g, float r) { float q = Q1(n, g, r); return fmaf(fmaf(-g, q, n), r, q); }
// class A; counts[0..3] = mismatches of (R1,Q1) (R2,Q1) (R1,Q2) (R2,Q2); counts[4] = mantissas where R1 != RN(1/g), [5] R2 != RN(1/g)
__global__ __launch_bounds__(256) void class_a(int exponent, int negative, unsigned long long *counts, uint32_t *examples) {
    const uint32_t mant = blockIdx.x * 256u + threadIdx.x;
    const float g = __uint_as_float(((uint32_t)(exponent + 127) << 23) | mant | (negative ? 0x80000000u : 0u));
    const float r1 = R1(g), r2 = R2(g), rn = __fdiv_rn(1.0f, g);
    unsigned long long c[4] = {0, 0, 0, 0}, c0 = 0;
    const float r0 = __builtin_amdgcn_rcpf(g);
    for (int v = 0; v < 65536; ++v) {
        const float n = (float)v, want = __fdiv_rn(n, g);
        c0 += __float_as_uint(Q1(n, g, r0)) != __float_as_uint(want) && want != 0.0f;
        const float a = Q1(n, g, r1), b = Q1(n, g, r2), cc = Q2(n, g, r1), d = Q2(n, g, r2);
        const bool ba = __float_as_uint(a) != __float_as_uint(want) && !(a == 0.0f && want == 0.0f);
        c[0] += ba;
        c[1] += __float_as_uint(b) != __float_as_uint(want) && !(b == 0.0f && want == 0.0f);
        c[2] += __float_as_uint(cc) != __float_as_uint(want) && !(cc == 0.0f && want == 0.0f);
        c[3] += __float_as_uint(d) != __float_as_uint(want) && !(d == 0.0f && want == 0.0f);
        if (ba) {
            const unsigned long long k = atomicAdd(counts + 6, 1ull);
            if (k < 8) examples[2 * k] = (uint32_t)v, examples[2 * k + 1] = __float_as_uint(g);
        }
    }
    for (int i = 0; i < 4; ++i) if (c[i]) atomicAdd(counts + i, c[i]);
    if (c0) atomicAdd(counts + 7, c0);
    if (r1 != rn) atomicAdd(counts + 4, 1ull);
    if (r2 != rn) atomicAdd(counts + 5, 1ull);
}
// class B; counts[0..3] as above for acc / wsum
__global__ __launch_bounds__(256) void class_b(int exponent, int negative, unsigned long long *counts, uint32_t *examples) {
    const uint32_t mant = blockIdx.x * 256u + threadIdx.x;
    const float acc = __uint_as_float(((uint32_t)(exponent + 127) << 23) | mant | (negative ? 0x80000000u : 0u));
    unsigned long long c[4] = {0, 0, 0, 0};
    for (int ws = 2; ws <= 16384; ++ws) {
        const float d = (float)ws, want = __fdiv_rn(acc, d), r1 = R1(d), r2 = R2(d);
        const float a = Q1(acc, d, r1);
        const bool ba = __float_as_uint(a) != __float_as_uint(want);
        c[0] += ba;
        c[1] += __float_as_uint(Q1(acc, d, r2)) != __float_as_uint(want);
        c[2] += __float_as_uint(Q2(acc, d, r1)) != __float_as_uint(want);
        c[3] += __float_as_uint(Q2(acc, d, r2)) != __float_as_uint(want);
        if (ba) {
            const unsigned long long k = atomicAdd(counts + 6, 1ull);
            if (k < 8) examples[2 * k] = (uint32_t)ws, examples[2 * k + 1] = __float_as_uint(acc);
        }
    }
    for (int i = 0; i < 4; ++i) if (c[i]) atomicAdd(counts + i, c[i]);
}
int main() {
    unsigned long long *counts, h[8];
    uint32_t *ex, hex[16];
    hipMalloc(&counts, 64), hipMalloc(&ex, 64);
    const int exps_a[] = {0, -1, 1, -7, 9, -20, 19}, exps_b[] = {0, 1, 13, 30, -44, 52};
    for (int neg = 0; neg < 2; ++neg)
        for (int e : exps_a) {
            hipMemset(counts, 0, 64), hipMemset(ex, 0, 64);
            hipLaunchKernelGGL(class_a, dim3(1u << 15), dim3(256), 0, 0, e, neg, counts, ex);
            { hipError_t err = hipDeviceSynchronize(); if (err != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(err)); return 1; } }
            hipMemcpy(h, counts, 64, hipMemcpyDeviceToHost), hipMemcpy(hex, ex, 64, hipMemcpyDeviceToHost);
            printf("A n/g  binade 2^%-3d %s: mismatches R1Q1 %llu  R2Q1 %llu  R1Q2 %llu  R2Q2 %llu   (R1 != RN(1/g) for %llu mantissas, R2 for %llu; sanity: raw v_rcp_f32 + Q1 %llu)",
                   e, neg ? "neg" : "pos", h[0], h[1], h[2], h[3], h[4], h[5], h[7]);
            for (unsigned long long k = 0; k < (h[6] < 3 ? h[6] : 3); ++k) printf("  e.g. %u / %a", hex[2 * k], *(float *)&hex[2 * k + 1]);
            printf("\n");
            fflush(stdout);
            if (neg && e != 0) break;
        }
    for (int e : exps_b) {
        hipMemset(counts, 0, 64), hipMemset(ex, 0, 64);
        hipLaunchKernelGGL(class_b, dim3(1u << 15), dim3(256), 0, 0, e, 0, counts, ex);
        { hipError_t err = hipDeviceSynchronize(); if (err != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(err)); return 1; } }
        hipMemcpy(h, counts, 64, hipMemcpyDeviceToHost), hipMemcpy(hex, ex, 64, hipMemcpyDeviceToHost);
        printf("B acc/ws binade 2^%-3d: mismatches R1Q1 %llu  R2Q1 %llu  R1Q2 %llu  R2Q2 %llu", e, h[0], h[1], h[2], h[3]);
        for (unsigned long long k = 0; k < (h[6] < 3 ? h[6] : 3); ++k) printf("  e.g. %a / %u", *(float *)&hex[2 * k + 1], hex[2 * k]);
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
