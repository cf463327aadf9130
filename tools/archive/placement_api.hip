// tools/placement_api.hip -- measurement aid (not part of the product): the configs[2] store stream (and a streaming
// read) into 761 MB buffers obtained through different allocation APIs, interleaved in one process.
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement_api.hip -o tools/placement_api
#include <hip/hip_runtime.h>
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
__global__ void readk(const f32x4* __restrict__ in, int per_block, int nblk, float* sink) {
    const int per_xcd = nblk >> 3;
    const int b = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const f32x4* o = in + (size_t)b * per_block;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) acc += o[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) sink[0] = 1.f;
}
static const int CHUNK = 92928, NBLK = 8192;
static const size_t BYTES = (size_t)NBLK * CHUNK;
static hipEvent_t ea, eb;
static float* sink;
template <typename F> static float t_us(F go, int reps = 8) {
    go(); go();
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) go();
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    return ms / reps * 1e3f;
}
static void report(const char* what, void* p) {
    CK(hipMemset(p, 0, BYTES));
    const float w = t_us([&] { hipLaunchKernelGGL(fill, dim3(NBLK), dim3(64), 0, 0, (f32x4*)p, CHUNK / 16, NBLK); });
    const float r = t_us([&] { hipLaunchKernelGGL(readk, dim3(NBLK), dim3(64), 0, 0, (const f32x4*)p, CHUNK / 16, NBLK, sink); });
    hipPointerAttribute_t at = {};
    hipError_t e = hipPointerGetAttributes(&at, p);
    printf("  %-44s %p  write %6.1f us  read %6.1f us  (attr: %s type %d managed %d)\n", what, p, w, r, hipGetErrorString(e), (int)at.type, (int)at.isManaged);
    (void)hipGetLastError();
}
int main() {
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    CK(hipMalloc((void**)&sink, 4));
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    const size_t G = (size_t)2 << 20, SZ = (BYTES + G - 1) / G * G;
    for (int round = 0; round < 3; ++round) {
        printf("# round %d\n", round);
        void* p;
        CK(hipMalloc(&p, BYTES)); report("hipMalloc", p); void* keep1 = p;
        {
            hipMemGenericAllocationHandle_t h; void* va;
            CK(hipMemCreate(&h, SZ, &prop, 0)); CK(hipMemAddressReserve(&va, SZ, G, nullptr, 0)); CK(hipMemMap(va, SZ, 0, h, 0));
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, SZ, &acc, 1));
            report("hipMemCreate + hipMemMap (VMM), 1 handle", va);
        }
        CK(hipMalloc(&p, BYTES)); report("hipMalloc (again)", p);
        if (hipExtMallocWithFlags(&p, BYTES, hipDeviceMallocFinegrained) == hipSuccess) report("hipExtMallocWithFlags(Finegrained)", p); else printf("  finegrained: failed\n");
        if (hipExtMallocWithFlags(&p, BYTES, hipDeviceMallocUncached) == hipSuccess) report("hipExtMallocWithFlags(Uncached)", p); else printf("  uncached: failed\n");
        if (hipExtMallocWithFlags(&p, BYTES, hipDeviceMallocContiguous) == hipSuccess) report("hipExtMallocWithFlags(Contiguous)", p); else { printf("  contiguous: failed\n"); (void)hipGetLastError(); }
        if (hipMallocAsync(&p, BYTES, 0) == hipSuccess) { CK(hipStreamSynchronize(0)); report("hipMallocAsync (stream-ordered pool)", p); } else printf("  mallocAsync: failed\n");
        CK(hipMalloc(&p, BYTES)); report("hipMalloc (third)", p);
        (void)keep1;
    }
    return 0;
}
