// tools/placement_vmm.hip -- measurement aid (not part of the product): can a FAST placement be REQUESTED?  A 726 MiB
// virtual window is assembled with the HIP virtual-memory API (hipMemCreate / hipMemMap) from physical granules taken
// from a pool in different patterns -- one contiguous handle, sequential granules, granules spread over the pool -- and
// the configs[2] store stream (8192 waves x 92928 B) is timed into each.
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement_vmm.hip -o tools/placement_vmm
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void fill(f32x4* __restrict__ out, int per_block, int nblk) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    const int per_xcd = nblk >> 3;
    const int b = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    f32x4* o = out + (size_t)b * per_block;
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) o[i] = v;
}
static const int CHUNK = 92928;
static const size_t WIN = (size_t)8192 * CHUNK;  // 726 MiB
static hipEvent_t ea, eb;
static float t_us(void* base, int reps = 10) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fill, dim3(8192), dim3(64), 0, 0, (f32x4*)base, CHUNK / 16, 8192);
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fill, dim3(8192), dim3(64), 0, 0, (f32x4*)base, CHUNK / 16, 8192);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    return ms / reps * 1e3f;
}
static hipMemAllocationProp prop;
struct Window { void* va; size_t size; };
// map `n` granules of `g` bytes: granule i <- pool[pick(i)]
template <typename Pick>
static Window map_window(const std::vector<hipMemGenericAllocationHandle_t>& pool, size_t g, Pick pick) {
    const size_t n = (WIN + g - 1) / g;
    Window w{nullptr, n * g};
    CK(hipMemAddressReserve(&w.va, w.size, (size_t)2 << 20, nullptr, 0));
    for (size_t i = 0; i < n; ++i) CK(hipMemMap((char*)w.va + i * g, g, 0, pool[pick(i)], 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(w.va, w.size, &acc, 1));
    return w;
}
static void unmap_window(Window w) { CK(hipMemUnmap(w.va, w.size)); CK(hipMemAddressFree(w.va, w.size)); }

int main(int argc, char** argv) {
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    const size_t pool_gib = argc > 1 ? atoi(argv[1]) : 24;
    // A. one contiguous handle per window, 6 windows
    printf("# A. one hipMemCreate(726 MiB) handle per window:");
    std::vector<hipMemGenericAllocationHandle_t> keepA;
    for (int k = 0; k < 6; ++k) {
        hipMemGenericAllocationHandle_t h; const size_t sz = (WIN + ((size_t)2 << 20) - 1) >> 21 << 21;
        CK(hipMemCreate(&h, sz, &prop, 0));
        std::vector<hipMemGenericAllocationHandle_t> one{h};
        Window w = map_window(one, sz, [](size_t) { return 0; });
        printf(" %.1f", t_us(w.va)); fflush(stdout);
        unmap_window(w); keepA.push_back(h);
    }
    printf(" us\n");
    for (auto h : keepA) CK(hipMemRelease(h));
    for (size_t g : {(size_t)2 << 20, (size_t)32 << 20}) {
        const size_t npool = (pool_gib << 30) / g;
        auto t0 = std::chrono::steady_clock::now();
        std::vector<hipMemGenericAllocationHandle_t> pool(npool);
        for (size_t i = 0; i < npool; ++i) CK(hipMemCreate(&pool[i], g, &prop, 0));
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const size_t n = (WIN + g - 1) / g;
        printf("# pool: %zu handles of %zu MiB (%zu GiB) created in %.2f s; a window = %zu granules\n", npool, g >> 20, pool_gib, secs, n);
        // B. sequential granules at several pool positions
        printf("  B. sequential granules starting at pool fraction 0, 1/4, 1/2, 3/4:");
        for (int q = 0; q < 4; ++q) {
            const size_t start = (npool - n) * q / 4;
            Window w = map_window(pool, g, [&](size_t i) { return start + i; });
            printf(" %.1f", t_us(w.va)); fflush(stdout);
            unmap_window(w);
        }
        printf(" us\n");
        // C. granules spread evenly over the whole pool (stride), 3 phases
        printf("  C. granules spread evenly over the WHOLE pool (granule i <- pool[i * npool / n + phase]):");
        for (size_t phase : {(size_t)0, (size_t)3, (size_t)7}) {
            Window w = map_window(pool, g, [&](size_t i) { return (i * npool / n + phase) % npool; });
            printf(" %.1f", t_us(w.va)); fflush(stdout);
            unmap_window(w);
        }
        printf(" us\n");
        // D. spread over a part of the pool only: 1/2, 1/4, 1/8, 1/16 of it
        printf("  D. spread over the first 1/2, 1/4, 1/8, 1/16 of the pool:");
        for (size_t frac : {(size_t)2, (size_t)4, (size_t)8, (size_t)16}) {
            const size_t span = npool / frac;
            if (span < n) { printf(" -"); continue; }
            Window w = map_window(pool, g, [&](size_t i) { return i * span / n; });
            printf(" %.1f", t_us(w.va)); fflush(stdout);
            unmap_window(w);
        }
        printf(" us\n");
        // E. two halves: even granules from the pool's start, odd granules from its middle
        printf("  E. even granules from the start of the pool, odd granules from its middle:");
        {
            Window w = map_window(pool, g, [&](size_t i) { return (i & 1) ? npool / 2 + i / 2 : i / 2; });
            printf(" %.1f", t_us(w.va)); unmap_window(w);
        }
        printf(" us\n");
        // F. pseudo-random granules
        printf("  F. pseudo-random granules from the whole pool:");
        for (unsigned seed : {1u, 2u}) {
            std::vector<size_t> perm(npool);
            for (size_t i = 0; i < npool; ++i) perm[i] = i;
            unsigned s = seed * 2654435761u;
            for (size_t i = npool - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; std::swap(perm[i], perm[s % (i + 1)]); }
            Window w = map_window(pool, g, [&](size_t i) { return perm[i]; });
            printf(" %.1f", t_us(w.va)); fflush(stdout);
            unmap_window(w);
        }
        printf(" us\n");
        for (auto h : pool) CK(hipMemRelease(h));
    }
    return 0;
}
