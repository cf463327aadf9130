// tools/placement_zones.hip -- measurement aid (not part of the product).  tools/placement_gap shows that the configs[2]
// store stream is ~25 % faster when part of the window lies in a DIFFERENT physical zone of HBM (e.g. beyond +48 GiB
// of a big slab).  This maps the zones: the first half of the window stays at slab offset R, the second half is
// placed at offset X, X scanned in 1 GiB steps; fast <=> X is in another zone than R.
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement_zones.hip -o tools/placement_zones
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
// chunks [0, half) at out + off0, chunks [half, nblk) at out + off1 (float4 units)
__global__ void fill2(f32x4* __restrict__ out, int per_block, int nblk, size_t off0, size_t off1) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    const int per_xcd = nblk >> 3;
    const int b = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int half = nblk >> 1;
    f32x4* o = out + (b < half ? off0 + (size_t)b * per_block : off1 + (size_t)(b - half) * per_block);
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) o[i] = v;
}
static const int CHUNK = 92928;
static hipEvent_t ea, eb;
static float t_us(char* slab, size_t r, size_t x, int reps = 6) {
    auto go = [&] { hipLaunchKernelGGL(fill2, dim3(8192), dim3(64), 0, 0, (f32x4*)slab, CHUNK / 16, 8192, r / 16, x / 16); };
    go(); go();
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) go();
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    return ms / reps * 1e3f;
}
int main(int argc, char** argv) {
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t GiB = (size_t)1 << 30;
    const int G = argc > 1 ? atoi(argv[1]) : 160;
    char* slab; CK(hipMalloc((void**)&slab, (size_t)G * GiB)); CK(hipMemset(slab, 0, (size_t)G * GiB));
    const size_t half = (size_t)4096 * CHUNK;
    printf("slab %d GiB at %p\n", G, slab);
    std::vector<int> refs = {0};
    for (size_t ri = 0; ri < refs.size() && ri < 5; ++ri) {
        const int R = refs[ri];
        printf("# first half at +%d GiB; second half at +X GiB, X = 0.5, 1.5, ... (16 per line) [us]\n", R);
        int first_fast = -1;
        std::vector<float> row;
        for (int X = 0; X + 1 < G; ++X) {
            const size_t x = (size_t)X * GiB + GiB / 2;
            if ((size_t)R * GiB < x + half && x < (size_t)R * GiB + half) { printf("   -%s", (X & 15) == 15 ? "\n" : " "); row.push_back(0); continue; }
            const float t = t_us(slab, (size_t)R * GiB, x);
            row.push_back(t);
            if (t < 120.f && first_fast < 0) first_fast = X;
            printf("%4.0f%s", t, (X & 15) == 15 ? "\n" : " ");
        }
        printf("\n");
        // next reference: the middle of the first fast run seen from this reference that is not yet a reference zone
        if (first_fast >= 0) {
            int end = first_fast;
            while (end + 1 < (int)row.size() && row[end + 1] > 0 && row[end + 1] < 120.f) ++end;
            const int mid = (first_fast + end) / 2;
            bool have = false;
            for (int r : refs) if (r == mid) have = true;
            if (!have) refs.push_back(mid);
        }
    }
    return 0;
}
