// tools/placement_lab.hip -- measurement aid (not part of the product): why do equal hipMalloc'd 761 MB buffers fall
// into speed tiers for the same store stream?  Experiments:
//   1. N buffers allocated one after another: time vs allocation order / virtual address
//   2. XCD rotation: XCD x writes eighth (x + k) % 8 of a buffer -- does a tier depend on WHICH XCD writes WHERE?
//   3. one big slab, 761 MB windows at 761 MB stride and 95 MB sub-windows (one eighth) written by all XCDs
//   4. hipMemCreate/hipMemMap (VMM) allocations with the recommended granularity
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement_lab.hip -o tools/placement_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// 8192 waves, each streams one contiguous chunk; XCD x (= block % 8) owns eighth (x + rot) % 8
__global__ void fill_xcd(f32x4* __restrict__ out, int per_block, int nblk, int rot) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    const int per_xcd = nblk >> 3;
    const int x = ((blockIdx.x & 7) + rot) & 7;
    const int b = x * per_xcd + (blockIdx.x >> 3);
    f32x4* o = out + (size_t)b * per_block;
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) o[i] = v;
}
// every XCD writes into ONE eighth-sized window (chunks interleaved b -> b): which physical region is slow?
__global__ void fill_flat(f32x4* __restrict__ out, int per_block) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    f32x4* o = out + (size_t)blockIdx.x * per_block;
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) o[i] = v;
}
static const size_t BYTES = (size_t)8192 * 64 * 1452;
static float time_fill(void* p, int rot, int reps = 12) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int per = (int)(BYTES / 16 / 8192);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fill_xcd, dim3(8192), dim3(64), 0, 0, (f32x4*)p, per, 8192, rot);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fill_xcd, dim3(8192), dim3(64), 0, 0, (f32x4*)p, per, 8192, rot);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps * 1e3f;
}
static float time_flat(void* p, size_t bytes, int reps = 12) {  // 1024 waves into a window of `bytes`
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int nblk = 2048, per = (int)(bytes / 16 / nblk);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fill_flat, dim3(nblk), dim3(256), 0, 0, (f32x4*)p, per);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fill_flat, dim3(nblk), dim3(256), 0, 0, (f32x4*)p, per);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps * 1e3f;
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 32;
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    printf("HBM free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
    // ---- 1. allocation order ----
    std::vector<void*> bufs(N);
    for (int i = 0; i < N; ++i) { CK(hipMalloc(&bufs[i], BYTES)); CK(hipMemset(bufs[i], 0, BYTES)); }
    std::vector<float> t(N);
    printf("# 1. %d x hipMalloc(761 MB), in allocation order: index va us(rot 0)\n", N);
    for (int i = 0; i < N; ++i) { t[i] = time_fill(bufs[i], 0); printf("%2d %p %7.2f\n", i, bufs[i], t[i]); }
    // ---- 2. XCD rotation on the fastest and the slowest ----
    int fast = (int)(std::min_element(t.begin(), t.end()) - t.begin()), slow = (int)(std::max_element(t.begin(), t.end()) - t.begin());
    printf("# 2. XCD rotation k (XCD x writes eighth (x+k)%%8): fastest buffer %d, slowest buffer %d\n", fast, slow);
    for (int k = 0; k < 8; ++k) printf("rot %d: fast %7.2f  slow %7.2f\n", k, time_fill(bufs[fast], k), time_fill(bufs[slow], k));
    // ---- 2b. eighths of the fastest / slowest written by the whole chip ----
    printf("# 2b. each 95 MB eighth alone, written by all XCDs (2048 x 256 threads): fast buffer | slow buffer\n");
    for (int e = 0; e < 8; ++e)
        printf("eighth %d: %7.2f | %7.2f us\n", e, time_flat((char*)bufs[fast] + e * (BYTES / 8), BYTES / 8), time_flat((char*)bufs[slow] + e * (BYTES / 8), BYTES / 8));
    for (int i = 0; i < N; ++i) CK(hipFree(bufs[i]));
    // ---- 3. one slab ----
    void* slab; const size_t slab_bytes = (size_t)N * BYTES;
    CK(hipMalloc(&slab, slab_bytes)); CK(hipMemset(slab, 0, slab_bytes));
    printf("# 3. one hipMalloc(%.1f GB) slab at %p: 761 MB windows at 761 MB stride\n", slab_bytes / 1e9, slab);
    for (int i = 0; i < N; ++i) printf("win %2d +%6.0f MB %7.2f\n", i, i * (BYTES / 1048576.0), time_fill((char*)slab + (size_t)i * BYTES, 0));
    printf("# 3b. same slab, windows shifted by half a window\n");
    for (int i = 0; i + 1 < N; i += 2) printf("win %2d.5 %7.2f\n", i, time_fill((char*)slab + (size_t)i * BYTES + BYTES / 2, 0));
    CK(hipFree(slab));
    // ---- 4. VMM ----
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran_min = 0, gran_rec = 0;
    hipError_t e1 = hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum);
    hipError_t e2 = hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended);
    printf("# 4. VMM: granularity min %zu (%s) recommended %zu (%s)\n", gran_min, hipGetErrorString(e1), gran_rec, hipGetErrorString(e2));
    if (e1 == hipSuccess && gran_min) {
        const size_t g = gran_rec ? gran_rec : gran_min;
        const size_t sz = (BYTES + g - 1) / g * g;
        for (int i = 0; i < 12; ++i) {
            hipMemGenericAllocationHandle_t h; void* va = nullptr;
            if (hipMemCreate(&h, sz, &prop, 0) != hipSuccess) { printf("hipMemCreate failed\n"); break; }
            CK(hipMemAddressReserve(&va, sz, g, nullptr, 0));
            CK(hipMemMap(va, sz, 0, h, 0));
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, sz, &acc, 1));
            CK(hipMemset(va, 0, BYTES));
            printf("vmm %2d %p %7.2f\n", i, va, time_fill(va, 0));
        }
    }
    return 0;
}
