// tools/placement_gap.hip -- measurement aid (not part of the product).  tools/placement_pairs shows: a window made of
// two separately allocated 364 MiB halves is always fast (110 us), one physically contiguous 726 MiB window slow
// (138 us).  Here the configs[2] store stream writes into ONE contiguous slab but leaves a gap of D bytes after every
// 1/P of the chunks (P parts): which relative displacement of the parts makes the stream fast?
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement_gap.hip -o tools/placement_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
// chunk b lives at b * per_block + (b / chunks_per_part) * gap16 float4s
__global__ void fill(f32x4* __restrict__ out, int per_block, int nblk, int chunks_per_part, size_t gap16) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    const int per_xcd = nblk >> 3;
    const int b = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    f32x4* o = out + (size_t)b * per_block + (size_t)(b / chunks_per_part) * gap16;
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) o[i] = v;
}
static const int CHUNK = 92928;
static hipEvent_t ea, eb;
static float t_us(char* base, int parts, size_t gap, int reps = 8) {
    auto go = [&] { hipLaunchKernelGGL(fill, dim3(8192), dim3(64), 0, 0, (f32x4*)base, CHUNK / 16, 8192, 8192 / parts, gap / 16); };
    go(); go();
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) go();
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    return ms / reps * 1e3f;
}
int main() {
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t SLAB = (size_t)100 << 30;
    char* slab; CK(hipMalloc((void**)&slab, SLAB)); CK(hipMemset(slab, 0, SLAB));
    const size_t MiB = (size_t)1 << 20;
    for (size_t base : {(size_t)0, 3 * 1024 * MiB, 9 * 1024 * MiB + 512 * MiB}) {
        printf("# window base +%zu MiB\n", base >> 20);
        for (int parts : {2, 4, 8}) {
            printf("  %d parts, gap D after each part [MiB -> us]:", parts);
            for (size_t d : {(size_t)0, 1024 * MiB, 2048 * MiB, 4096 * MiB, 6144 * MiB, 8192 * MiB, 10240 * MiB, 12288 * MiB, 14336 * MiB, 16384 * MiB - 363 * MiB, 16384 * MiB, 20480 * MiB, 24576 * MiB, 32768 * MiB, 49152 * MiB}) {
                if (base + (size_t)8192 * CHUNK + (size_t)(parts - 1) * d > SLAB) continue;
                printf(" %g:%.0f", d / 1048576.0, t_us(slab + base, parts, d));
            }
            printf("\n");
        }
    }
    return 0;
}
