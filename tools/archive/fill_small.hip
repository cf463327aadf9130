// tools/fill_small.hip -- measurement aid (not part of the product): pure 16-B store streams at the sizes of the
// SHORT launches (configs[3] shard: 8192 envs x 23232 B = 190 MB; configs[1]: 1024 x 11616 B = 11.9 MB), one and two
// alternating buffers, so that the step kernel's time can be compared with what the memory system gives a bare fill.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fill_small.hip -o tools/fill_small
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void fill_chunk(f32x4* __restrict__ out, int per_block, int nblk) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    int b = blockIdx.x;
    const int per_xcd = nblk >> 3;
    if (b < (per_xcd << 3)) b = (b & 7) * per_xcd + (b >> 3);
    f32x4* o = out + (size_t)b * per_block;
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) o[i] = v;
}
int main() {
    struct { const char* name; int chunks; int chunk_bytes; } cases[] = {
        {"cfg3 8192 x 23232 B", 8192, 23232}, {"cfg1 1024 x 11616 B", 1024, 11616}, {"cfg2 8192 x 92928 B", 8192, 92928}};
    for (auto& c : cases) {
        const size_t bytes = (size_t)c.chunks * c.chunk_bytes;
        f32x4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
        for (int two = 0; two < 2; ++two)
            for (int thr : {64, 256}) {
                const int per = c.chunk_bytes / 16;
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(fill_chunk, dim3(c.chunks), dim3(thr), 0, 0, (i & 1) && two ? b : a, per, c.chunks);
                CK(hipDeviceSynchronize());
                const int reps = 200;
                CK(hipEventRecord(e0));
                for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fill_chunk, dim3(c.chunks), dim3(thr), 0, 0, (i & 1) && two ? b : a, per, c.chunks);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const float us = ms / reps * 1e3f;
                printf("%-22s %s, %3d thr/chunk : %7.1f GB/s (%.2f us per launch, launch-to-launch)\n", c.name,
                       two ? "2 alternating buffers" : "1 buffer             ", thr, bytes / us / 1e3, us);
            }
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
