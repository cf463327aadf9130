// tools/placement_pairs.hip -- measurement aid (not part of the product).  tools/placement_map shows that the configs[2]
// store stream runs ~25 % faster into a window that straddles certain boundaries of a big allocation.  Is that a
// property of PAIRS of physical regions?  N chunks of 364 MiB are created and committed in order (hipMemCreate +
// hipMemMap); windows are then assembled from two arbitrary chunks (first half <- chunk a, second half <- chunk b) and
// timed.
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement_pairs.hip -o tools/placement_pairs
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
static const int CHUNK = 92928;
static const size_t HALF = (size_t)364 << 20;
static hipEvent_t ea, eb;
static hipMemAllocationProp prop;
static float t_us(void* base, int reps = 6) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fill, dim3(8192), dim3(64), 0, 0, (f32x4*)base, CHUNK / 16, 8192);
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fill, dim3(8192), dim3(64), 0, 0, (f32x4*)base, CHUNK / 16, 8192);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    return ms / reps * 1e3f;
}
static std::vector<hipMemGenericAllocationHandle_t> H;
static void* win_va;
static float pair(int a, int b) {  // window = [chunk a | chunk b]
    CK(hipMemMap(win_va, HALF, 0, H[a], 0));
    CK(hipMemMap((char*)win_va + HALF, HALF, 0, H[b], 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(win_va, 2 * HALF, &acc, 1));
    const float t = t_us(win_va);
    CK(hipMemUnmap(win_va, 2 * HALF));
    return t;
}
int main(int argc, char** argv) {
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    const int N = argc > 1 ? atoi(argv[1]) : 200;  // 200 x 364 MiB = 71 GiB
    H.resize(N);
    // commit in order: map every chunk into one long VA range (a slab assembled by hand), keep it mapped
    void* slab; CK(hipMemAddressReserve(&slab, (size_t)N * HALF, (size_t)2 << 20, nullptr, 0));
    for (int i = 0; i < N; ++i) {
        CK(hipMemCreate(&H[i], HALF, &prop, 0));
        CK(hipMemMap((char*)slab + (size_t)i * HALF, HALF, 0, H[i], 0));
    }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(slab, (size_t)N * HALF, &acc, 1));
    CK(hipMemset(slab, 0, (size_t)N * HALF));
    CK(hipMemAddressReserve(&win_va, 2 * HALF, (size_t)2 << 20, nullptr, 0));
    printf("%d chunks of 364 MiB committed in order (%.1f GiB)\n# adjacent pairs [i | i+1] (16 per line):\n", N, N * 364.0 / 1024);
    std::vector<float> adj(N - 1);
    for (int i = 0; i + 1 < N; ++i) { adj[i] = pair(i, i + 1); printf("%4.0f%s", adj[i], (i & 15) == 15 ? "\n" : " "); }
    printf("\n");
    std::vector<int> fastb;
    for (int i = 0; i + 1 < N; ++i) if (adj[i] < 120.f) fastb.push_back(i);
    printf("# fast adjacent pairs (boundaries) at i =");
    for (int i : fastb) printf(" %d", i);
    printf("\n");
    if (fastb.empty()) return 0;
    const int b = fastb[0];  // chunks <= b are 'before', chunks > b are 'after' the first boundary
    printf("# around boundary %d: [a | c] for a before, c after the boundary at growing distance:\n", b);
    for (int d : {0, 1, 2, 4, 8, 16, 32}) {
        const int a = b - d, c = b + 1 + d;
        if (a < 0 || c >= N) break;
        printf("  d=%2d: [%d|%d] %.1f   reversed [%d|%d] %.1f   same side before [%d|%d] %.1f   same side after [%d|%d] %.1f\n", d, a, c,
               pair(a, c), c, a, pair(c, a), a, a > 0 ? a - 1 : a + 1, pair(a, a > 0 ? a - 1 : a + 1), c, c + 1 < N ? c + 1 : c - 1,
               pair(c, c + 1 < N ? c + 1 : c - 1));
    }
    // full row: chunk 0 paired with every chunk -> classes?
    printf("# row: [0 | j] for every j (16 per line):\n");
    for (int j = 1; j < N; ++j) printf("%4.0f%s", pair(0, j), (j & 15) == 0 ? "\n" : " ");
    printf("\n# row: [N/2 | j] for every j:\n");
    for (int j = 0; j < N; ++j) { if (j == N / 2) { printf("   -%s", (j & 15) == 15 ? "\n" : " "); continue; } printf("%4.0f%s", pair(N / 2, j), (j & 15) == 15 ? "\n" : " "); }
    printf("\n");
    return 0;
}
