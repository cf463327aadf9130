// tools/placement_map.hip -- measurement aid (not part of the product): speed map of one big hipMalloc slab.  For every
// 64 MiB step the configs[2] store stream (8192 waves x 92928 B, XCD-contiguous) is timed into the 726 MiB window
// starting there; a second pass times a 16-B/lane streaming READ of the same window.
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement_map.hip -o tools/placement_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
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
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) acc += __builtin_nontemporal_load(&o[i]);
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) sink[0] = 1.f;
}
static const int CHUNK = 92928;
static hipEvent_t ea, eb;
template <typename F> static float t_us(F f, int reps = 5) {
    f();
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    return ms / reps * 1e3f;
}
int main(int argc, char** argv) {
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t SLAB = (size_t)(argc > 1 ? atoi(argv[1]) : 64) << 30, WIN = (size_t)8192 * CHUNK;
    char* slab; CK(hipMalloc((void**)&slab, SLAB)); CK(hipMemset(slab, 0, SLAB));
    float* sink; CK(hipMalloc((void**)&sink, 4));
    printf("slab %zu GiB at %p; one value per 64 MiB step, 16 per line (= 1 GiB per line)\n# store stream [us]\n", SLAB >> 30, slab);
    int n = 0;
    for (size_t off = 0; off + WIN <= SLAB; off += (size_t)64 << 20, ++n) {
        char* p = slab + off;
        printf("%4.0f%s", t_us([&] { hipLaunchKernelGGL(fill, dim3(8192), dim3(64), 0, 0, (f32x4*)p, CHUNK / 16, 8192); }), (n & 15) == 15 ? "\n" : " ");
    }
    printf("\n# streaming read [us]\n");
    n = 0;
    for (size_t off = 0; off + WIN <= SLAB; off += (size_t)64 << 20, ++n) {
        char* p = slab + off;
        printf("%4.0f%s", t_us([&] { hipLaunchKernelGGL(readk, dim3(8192), dim3(64), 0, 0, (const f32x4*)p, CHUNK / 16, 8192, sink); }), (n & 15) == 15 ? "\n" : " ");
    }
    printf("\n");
    return 0;
}
