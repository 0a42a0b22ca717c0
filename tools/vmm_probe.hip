// vmm_probe: what the HIP runtime's copies see after a virtual range is unmapped, freed, reserved again and mapped differently.
//   hipcc --offload-arch=gfx950 -O2 vmm_probe.hip -o vmm_probe ; ./vmm_probe [0|1: free + re-reserve the range (1) or keep it (0)]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <dlfcn.h>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void fill(uint32_t *p, size_t n, uint32_t v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// mode 2: an arena from libsquidstitch (sq_arena_create through dlopen), then a kernel write and the runtime's copies, at once
static int arena_mode(const char *lib_path, int wait_us) {
    void *lib = dlopen(lib_path, RTLD_NOW);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
    typedef void *(*create_t)(int64_t, int64_t, int64_t, int64_t, int32_t, void *, void *);
    create_t create = (create_t)dlsym(lib, "sq_arena_create");
    struct { void *base; int64_t bytes, slice; int32_t n_slices, n_cand, n_classes, cs[8], cc[8], inter; float probe_ms, create_ms, lo, hi; } info;
    // small plain allocations made BEFORE the arena, as a caller has them (tables, tiles, results)
    const size_t sizes[4] = {4096, 100000, (size_t)1 << 20, (size_t)3 << 20};
    uint32_t *small[4];
    for (int i = 0; i < 4; ++i) {
        CK(hipMalloc(&small[i], sizes[i]));
        hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, small[i], sizes[i] / 4, 0xC0C0C000u + i);
    }
    CK(hipDeviceSynchronize());
    hipStream_t st;
    CK(hipStreamCreate(&st));
    if (!create((int64_t)64 << 20, (int64_t)192 << 20, (int64_t)8 << 20, (int64_t)32 << 20, 0, st, &info)) { printf("sq_arena_create failed\n"); return 1; }
    char *base = (char *)info.base;
    for (int i = 0; i < 4; ++i) {
        std::vector<uint32_t> back(sizes[i] / 4);
        CK(hipMemcpy(back.data(), small[i], sizes[i], hipMemcpyDeviceToHost));
        size_t wrong = 0;
        for (uint32_t v : back) wrong += v != 0xC0C0C000u + i;
        printf("plain allocation %d at %p (%zu bytes), made before the arena: %zu of %zu words changed (first %08x)\n", i, (void *)small[i], sizes[i], wrong, back.size(), back[0]);
    }
    if (wait_us) { CK(hipDeviceSynchronize()); usleep(wait_us); }
    uint32_t w[4];
    CK(hipMemcpy(w, base + 4096, 16, hipMemcpyDeviceToHost));
    printf("arena at %p, %d slices of %d candidates; before: %08x %08x\n", (void *)base, info.n_slices, info.n_cand, w[0], w[1]);
    CK(hipMemset(base, 0x33, 1 << 20));
    CK(hipMemcpy(w, base + 4096, 16, hipMemcpyDeviceToHost));
    printf("after hipMemset 0x33: %08x %08x\n", w[0], w[1]);
    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, st, (uint32_t *)base, (size_t)(16 << 20) / 4, 0xB0B0B0B0u);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(w, base + 4096, 16, hipMemcpyDeviceToHost));
    printf("after a kernel wrote b0b0b0b0: %08x %08x\n", w[0], w[1]);
    for (size_t nbytes : {(size_t)16, (size_t)4096, (size_t)65536, (size_t)119808, (size_t)1 << 20, (size_t)9 << 20}) {      // pageable host target, several sizes
        std::vector<uint32_t> big(nbytes / 4, 0);
        CK(hipMemcpy(big.data(), base + 4096, nbytes, hipMemcpyDeviceToHost));
        size_t wrong = 0;
        for (uint32_t v : big) wrong += v != 0xB0B0B0B0u;
        printf("D2H of %zu bytes from base + 4096: %zu of %zu words wrong (first %08x)\n", nbytes, wrong, big.size(), big[0]);
    }
    uint32_t *d_w;
    CK(hipMalloc(&d_w, 16));
    CK(hipMemcpy(d_w, base + 4096, 16, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(w, d_w, 16, hipMemcpyDeviceToHost));
    printf("the same through a D2D copy: %08x %08x\n", w[0], w[1]);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 2 && atoi(argv[1]) == 2) return arena_mode(argv[2], argc > 3 ? atoi(argv[3]) : 0);
    const int refree = argc > 1 ? atoi(argv[1]) : 1;
    const size_t S = (size_t)8 << 20;
    const int N = 6;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemGenericAllocationHandle_t h[N];
    for (int i = 0; i < N; ++i) CK(hipMemCreate(&h[i], S, &prop, 0));
    char *va = nullptr;
    CK(hipMemAddressReserve((void **)&va, N * S, 0, nullptr, 0));
    printf("first range at %p\n", (void *)va);
    for (int i = 0; i < N; ++i) { CK(hipMemMap(va + i * S, S, 0, h[i], 0)); CK(hipMemSetAccess(va + i * S, S, &acc, 1)); }
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, (uint32_t *)(va + i * S), S / 4, 0xA0000000u + i);   // slice i holds A000000i
    CK(hipDeviceSynchronize());
    for (int i = 0; i < N; ++i) CK(hipMemUnmap(va + i * S, S));
    char *vb = va;
    const int M = 3;
    if (refree) {
        CK(hipMemAddressFree(va, N * S));
        CK(hipMemAddressReserve((void **)&vb, M * S, 0, nullptr, 0));
    }
    printf("second range at %p (%s)\n", (void *)vb, vb == va ? "the same address" : "another address");
    const int order[M] = {4, 2, 5};      // handle order[i] at slice i
    for (int i = 0; i < M; ++i) { CK(hipMemMap(vb + i * S, S, 0, h[order[i]], 0)); CK(hipMemSetAccess(vb + i * S, S, &acc, 1)); }
    // what do a kernel and the runtime's copies see at slice i?
    uint32_t *d_out, host[M], viacopy[M], viaasync[M];
    CK(hipMalloc(&d_out, 64));
    for (int i = 0; i < M; ++i) CK(hipMemcpy(d_out + i, vb + i * S + 4096, 4, hipMemcpyDeviceToDevice));      // D2D by the runtime
    CK(hipMemcpy(host, d_out, sizeof host, hipMemcpyDeviceToHost));
    for (int i = 0; i < M; ++i) CK(hipMemcpy(&viacopy[i], vb + i * S + 4096, 4, hipMemcpyDeviceToHost));     // D2H by the runtime
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int i = 0; i < M; ++i) CK(hipMemcpyAsync(&viaasync[i], vb + i * S + 4096, 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    for (int i = 0; i < M; ++i)
        printf("slice %d (handle %d): D2D copy sees %08x, D2H copy sees %08x, async D2H sees %08x, expected %08x\n", i, order[i], host[i], viacopy[i], viaasync[i], 0xA0000000u + order[i]);
    // a kernel writes through the page tables; read back by copy
    for (int i = 0; i < M; ++i) hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, (uint32_t *)(vb + i * S), S / 4, 0xB0000000u + i);
    CK(hipDeviceSynchronize());
    for (int i = 0; i < M; ++i) CK(hipMemcpy(&viacopy[i], vb + i * S + 4096, 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < M; ++i) printf("slice %d after a kernel wrote B000000%d: D2H copy sees %08x\n", i, i, viacopy[i]);
    CK(hipMemset(vb, 0x33, 1 << 20));
    CK(hipMemcpy(&viacopy[0], vb + 4096, 4, hipMemcpyDeviceToHost));
    printf("after hipMemset 0x33: D2H copy sees %08x\n", viacopy[0]);
    return 0;
}
