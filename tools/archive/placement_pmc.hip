// tools/placement_pmc.hip -- measurement aid (not part of the product): finds a FAST and a SLOW 726 MiB window inside
// one hipMalloc slab (tools/placement_scan.hip), then runs the same store stream into each under distinct kernel names
// (fill_tag<1> = fast window, fill_tag<2> = slow window) so that `rocprofv3 --pmc ...` attributes counters to them.
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement_pmc.hip -o tools/placement_pmc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ void body(f32x4* out, int per_block, int nblk) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    const int per_xcd = nblk >> 3;
    const int b = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    f32x4* o = out + (size_t)b * per_block;
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) o[i] = v;
}
__global__ void fill_scan(f32x4* out, int per_block, int nblk) { body(out, per_block, nblk); }
template <int TAG> __global__ void fill_tag(f32x4* out, int per_block, int nblk) { body(out, per_block, nblk); }
static const int CHUNK = 92928;
static hipEvent_t ea, eb;
static float t_us(char* base, int reps = 6) {
    hipLaunchKernelGGL(fill_scan, dim3(8192), dim3(64), 0, 0, (f32x4*)base, CHUNK / 16, 8192);
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fill_scan, dim3(8192), dim3(64), 0, 0, (f32x4*)base, CHUNK / 16, 8192);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    return ms / reps * 1e3f;
}
int main(int argc, char** argv) {
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t SLAB = (size_t)(argc > 1 ? atoi(argv[1]) : 40) << 30, WIN = (size_t)8192 * CHUNK;
    char* slab; CK(hipMalloc((void**)&slab, SLAB)); CK(hipMemset(slab, 0, SLAB));
    size_t best = 0, worst = 0; float tb = 1e9f, tw = 0.f;
    for (size_t off = 0; off + WIN <= SLAB; off += (size_t)64 << 20) {
        const float t = t_us(slab + off);
        if (t < tb) { tb = t; best = off; }
        if (t > tw) { tw = t; worst = off; }
    }
    printf("slab %zu GiB: fastest window +%zu MiB %.1f us, slowest +%zu MiB %.1f us (timed under whatever tool is attached)\n",
           SLAB >> 30, best >> 20, tb, worst >> 20, tw);
    for (int i = 0; i < 12; ++i) {
        hipLaunchKernelGGL(fill_tag<1>, dim3(8192), dim3(64), 0, 0, (f32x4*)(slab + best), CHUNK / 16, 8192);
        hipLaunchKernelGGL(fill_tag<2>, dim3(8192), dim3(64), 0, 0, (f32x4*)(slab + worst), CHUNK / 16, 8192);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
