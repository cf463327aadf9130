// tools/placement_layouts.hip -- measurement aid (not part of the product).  HBM on this device falls into physical
// zones (tools/placement_zones): the configs[2] store stream runs at ~5.5 TB/s when all of its 761 MB lie in one zone
// and at ~6.9 TB/s when they are split over two.  This finds one offset in each of up to three zones of a big slab and
// times different ways of distributing the 8192 chunks over them (chunk b -> zone pattern(b)).
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement_layouts.hip -o tools/placement_layouts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void fill_tab(f32x4* __restrict__ out, int per_block, int nblk, const size_t* __restrict__ off16) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    const int per_xcd = nblk >> 3;
    const int b = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    f32x4* o = out + off16[b];
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) o[i] = v;
}
__global__ void read_tab(const f32x4* __restrict__ in, int per_block, int nblk, const size_t* __restrict__ off16, float* sink) {
    const int per_xcd = nblk >> 3;
    const int b = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const f32x4* o = in + off16[b];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) acc += o[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) sink[0] = 1.f;
}
static int CHUNK = 92928, NBLK = 8192, THR = 64;
static hipEvent_t ea, eb;
static char* slab;
static size_t* d_off;
static float* sink;
// pattern(b) -> zone index; zone z lives at slab + zone_off[z]; chunks of a zone are packed densely in chunk order
static std::pair<float, float> run(const std::vector<size_t>& zone_off, std::function<int(int)> pattern) {
    std::vector<size_t> off(NBLK), used(zone_off.size(), 0);
    for (int b = 0; b < NBLK; ++b) { const int z = pattern(b); off[b] = (zone_off[z] + used[z]) / 16; used[z] += CHUNK; }
    CK(hipMemcpy(d_off, off.data(), NBLK * sizeof(size_t), hipMemcpyHostToDevice));
    auto time = [&](bool wr) {
        auto go = [&] { if (wr) hipLaunchKernelGGL(fill_tab, dim3(NBLK), dim3(THR), 0, 0, (f32x4*)slab, CHUNK / 16, NBLK, d_off);
                        else hipLaunchKernelGGL(read_tab, dim3(NBLK), dim3(THR), 0, 0, (const f32x4*)slab, CHUNK / 16, NBLK, d_off, sink); };
        go(); go();
        CK(hipEventRecord(ea));
        for (int i = 0; i < 8; ++i) go();
        CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb));
        return ms / 8 * 1e3f;
    };
    return {time(true), time(false)};
}
int main(int argc, char** argv) {
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t GiB = (size_t)1 << 30;
    const int G = argc > 1 ? atoi(argv[1]) : 200;
    CK(hipMalloc((void**)&slab, (size_t)G * GiB)); CK(hipMemset(slab, 0, (size_t)G * GiB));
    CK(hipMalloc((void**)&d_off, 8192 * sizeof(size_t))); CK(hipMalloc((void**)&sink, 4));
    // find zones: zone 0 = offset 0; zone 1 = first X that is fast against 0; zone 2 = first X fast against both
    std::vector<size_t> Z = {0};
    auto half = [&](size_t a, size_t b) { return run({a, b}, [](int c) { return c >= 4096; }).first; };
    const float same = half(0, 2 * GiB);
    for (int X = 4; X + 3 < G && Z.size() < 3; X += 4) {
        bool diff_all = true;
        for (size_t z : Z) if (half(z, (size_t)X * GiB) > 0.9f * same) { diff_all = false; break; }
        if (diff_all) Z.push_back((size_t)X * GiB);
    }
    printf("slab %d GiB; same-zone halves %.1f us; zones found at offsets [GiB]:", G, same);
    for (size_t z : Z) printf(" %zu", z >> 30);
    printf("\n");
    if (Z.size() < 2) return 0;
    struct { const char* name; int nblk, chunk, thr; } cases[] = {{"configs[2] 8192 x 92928 B", 8192, 92928, 64}, {"configs[3] 8192 x 23232 B", 8192, 23232, 64},
                                                                  {"configs[4] 4096 x 691200 B, 256 thr", 4096, 691200, 256}};
    for (auto& c : cases) {
        NBLK = c.nblk; CHUNK = c.chunk; THR = c.thr;
        printf("# %s: layout -> write us / read us\n", c.name);
        auto show = [&](const char* what, std::pair<float, float> t) { printf("  %-58s %7.1f / %7.1f\n", what, t.first, t.second); };
        const int N = NBLK;
        show("all in zone A", run(Z, [](int) { return 0; }));
        show("halves A | B", run(Z, [N](int b) { return b >= N / 2; }));
        show("5/8 A | 3/8 B", run(Z, [N](int b) { return b >= N * 5 / 8; }));
        show("3/4 A | 1/4 B", run(Z, [N](int b) { return b >= N * 3 / 4; }));
        show("alternate A/B every 1 chunk", run(Z, [](int b) { return b & 1; }));
        show("alternate A/B every 32 chunks", run(Z, [](int b) { return (b >> 5) & 1; }));
        show("alternate A/B every 512 chunks (each XCD eighth split)", run(Z, [](int b) { return (b >> 9) & 1; }));
        if (Z.size() >= 3) {
            show("thirds A | B | C", run(Z, [N](int b) { return b * 3 / N; }));
            show("round robin A/B/C every 32 chunks", run(Z, [](int b) { return (b >> 5) % 3; }));
        }
    }
    return 0;
}
