// tools/placement_handles.hip -- measurement aid (not part of the product).  One virtual window, assembled with the HIP
// VMM API from H separately created physical handles of equal size (each committed right after creation); the store
// stream of configs[2] / [3] / [4] is timed into it for H = 1, 2, 3, 4, 8, ...  (tools/placement_pairs: H = 2 is fast,
// H = 1 slow -- how general is that?)
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement_handles.hip -o tools/placement_handles
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
static hipEvent_t ea, eb;
static hipMemAllocationProp prop;
static float t_us(void* base, int nblk, int chunk, int thr, int reps = 8) {
    auto go = [&] { hipLaunchKernelGGL(fill, dim3(nblk), dim3(thr), 0, 0, (f32x4*)base, chunk / 16, nblk); };
    go(); go();
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) go();
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    return ms / reps * 1e3f;
}
struct Win { void* va; size_t size; std::vector<hipMemGenericAllocationHandle_t> h; };
static Win make(size_t bytes, int H) {
    const size_t G = (size_t)2 << 20;
    const size_t part = ((bytes + H - 1) / H + G - 1) / G * G;
    Win w; w.size = part * H;
    CK(hipMemAddressReserve(&w.va, w.size, G, nullptr, 0));
    for (int i = 0; i < H; ++i) {
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, part, &prop, 0));
        CK(hipMemMap((char*)w.va + (size_t)i * part, part, 0, h, 0));
        w.h.push_back(h);
    }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(w.va, w.size, &acc, 1));
    CK(hipMemset(w.va, 0, w.size));
    return w;
}
static void drop(Win& w) { CK(hipMemUnmap(w.va, w.size)); for (auto h : w.h) CK(hipMemRelease(h)); CK(hipMemAddressFree(w.va, w.size)); }
int main() {
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    struct { const char* name; int nblk, chunk, thr; } cases[] = {
        {"configs[2] 8192 x 92928 B (761 MB)", 8192, 92928, 64}, {"configs[3] 8192 x 23232 B (190 MB)", 8192, 23232, 64},
        {"configs[4] 4096 x 691200 B (2831 MB), 256 thr", 4096, 691200, 256}, {"configs[1] 1024 x 11616 B (12 MB)", 1024, 11616, 64}};
    for (auto& c : cases) {
        const size_t bytes = (size_t)c.nblk * c.chunk;
        printf("# %s\n  hipMalloc x3:", c.name);
        for (int k = 0; k < 3; ++k) { void* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes)); printf(" %.1f", t_us(p, c.nblk, c.chunk, c.thr)); CK(hipFree(p)); }
        printf(" us\n");
        for (int rep = 0; rep < 2; ++rep) {
            printf("  VMM handles ->");
            for (int H : {1, 2, 3, 4, 8, 16, 32, 64}) {
                if ((bytes / H) < ((size_t)2 << 20)) break;
                Win w = make(bytes, H);
                printf(" H=%d:%.1f", H, t_us(w.va, c.nblk, c.chunk, c.thr)); fflush(stdout);
                drop(w);
            }
            printf(" us\n");
        }
    }
    // two windows alive at once (the double buffer) with H = 2 each, alternating launches
    {
        const int nblk = 8192, chunk = 92928; const size_t bytes = (size_t)nblk * chunk;
        Win a = make(bytes, 2), b = make(bytes, 2);
        auto go = [&](int i) { hipLaunchKernelGGL(fill, dim3(nblk), dim3(64), 0, 0, (f32x4*)((i & 1) ? b.va : a.va), chunk / 16, nblk); };
        for (int i = 0; i < 4; ++i) go(i);
        CK(hipEventRecord(ea));
        for (int i = 0; i < 40; ++i) go(i);
        CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb));
        printf("# configs[2], two H=2 windows alternating: %.1f us per launch\n", ms / 40 * 1e3f);
    }
    return 0;
}
