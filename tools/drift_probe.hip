// tools/drift_probe.hip -- measurement aid (not part of the product): does the WRITE ORDER of a drifting on-device loop
// cost HBM bandwidth?  Persistent waves, each owning one chunk of every slot of a ring, alternate a pseudo state phase
// (s_sleep with jitter) with a 16-byte sc1 store stream into their chunk of slot t % slots -- pgx_rollout's memory
// behaviour without its arithmetic.  TEAM > 1: waves of one workgroup own ADJACENT chunks and meet at a barrier once per
// iteration, so that a workgroup's chunks are written together (the page locality a single launch gets from its in-order
// dispatch).  Build: hipcc --offload-arch=gfx950 -O3 tools/drift_probe.hip -o tools/drift_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int TEAM>
__global__ __launch_bounds__(64 * TEAM) void drift(f32x4* ring, size_t slot_f4, int slots, int chunk_f4, int K, int sleep_base,
                                                   int jitter_mask, int nchunks, int init_spread) {
    extern __shared__ uint32_t pad[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int per_xcd = nchunks / 8;
    const int ci = k * TEAM + wave;
    if (ci >= per_xcd) return;
    const size_t c = (size_t)x * per_xcd + ci;  // chunks of one XCD contiguous, chunks of one workgroup adjacent
    uint32_t rng = (uint32_t)c * 2654435761u + 12345u;
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    int slot = 0;
    if (init_spread > 0) {  // de-synchronise: every wave (team) starts at its own phase of the iteration
        const uint32_t h = ((uint32_t)(TEAM > 1 ? blockIdx.x : c) * 2246822519u) >> 16;
        const int n0 = (int)((h * (uint32_t)init_spread) >> 16);
        for (int i = 0; i < n0; ++i) __builtin_amdgcn_s_sleep(16);
    }
    for (int t = 0; t < K; ++t) {
        rng = rng * 1664525u + 1013904223u;
        const int n = sleep_base + (int)((rng >> 24) & (uint32_t)jitter_mask);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);  // 16 x 64 cycles ~ 0.5 us
        f32x4* o = ring + (size_t)slot * slot_f4 + c * (size_t)chunk_f4;
        for (int i = lane; i < chunk_f4; i += 64) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&o[i]), "v"(v) : "memory");
        slot = slot + 1 == slots ? 0 : slot + 1;
        if (TEAM > 1) __builtin_amdgcn_s_barrier();
    }
    if (pad[0] == 0xdeadbeef) ring[0] = v;
}

template <int TEAM>
static float run(f32x4* ring, size_t slot_f4, int slots, int chunk_f4, int K, int sleep_base, int jitter_mask, int nchunks, int lds_per_wave, int init_spread = 0) {
    const int grid = 8 * ((nchunks / 8 + TEAM - 1) / TEAM);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void*)drift<TEAM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto go = [&] { hipLaunchKernelGGL(drift<TEAM>, dim3(grid), dim3(64 * TEAM), lds_per_wave * TEAM, 0, ring, slot_f4, slots, chunk_f4, K, sleep_base, jitter_mask, nchunks, init_spread); };
    go(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < 3; ++i) go(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 3 / K * 1e3f;
}

int main(int argc, char** argv) {
    const int K = 64;
    struct Cfg { const char* name; int nchunks; int chunk_bytes; int slots; } cfgs[] = {
        {"cfg3-like: 8192 chunks x 23 KB, 8 slots", 8192, 16 * 1452, 8}, {"cfg3-like, 2 slots", 8192, 16 * 1452, 2},
        {"cfg2-like: 8192 chunks x 93 KB, 2 slots", 8192, 64 * 1452, 2}, {"cfg1-like: 1024 chunks x 11.6 KB, 64 slots", 1024, 8 * 1452, 64}};
    for (auto& c : cfgs) {
        const size_t slot_bytes = (size_t)c.nchunks * c.chunk_bytes;
        const size_t slot_f4 = ((slot_bytes + (2 << 20) - 1) / (2 << 20)) * (2 << 20) / 16;
        f32x4* ring; CK(hipMalloc(&ring, slot_f4 * 16 * c.slots)); CK(hipMemset(ring, 0, slot_f4 * 16 * c.slots));
        printf("%s (slot %.0f MB): 8 TB/s = %.1f us per step\n", c.name, slot_bytes / 1e6, slot_bytes / 8e12 * 1e6);
        for (int sleep_base : {4, 0}) {
            const int jm = sleep_base ? 3 : 0, lds = 10 * 1024;  // 10 KB per wave: 16 waves per CU = 4 per SIMD, as rollout_kernel
            printf("  pseudo state phase %d..%d x 0.5 us:  TEAM 1: %6.2f us  2: %6.2f  4: %6.2f  8: %6.2f  16: %6.2f   (8 waves per SIMD, TEAM 1: %6.2f  4: %6.2f)\n",
                   sleep_base, sleep_base + jm,
                   run<1>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, lds),
                   run<2>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, lds),
                   run<4>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, lds),
                   run<8>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, lds),
                   run<16>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, lds),
                   run<1>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, 5 * 1024),
                   run<4>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, 5 * 1024));
            printf("     ... with a random start phase of up to 8 / 16 / 32 / 64 us per wave, TEAM 1, 4 waves per SIMD: %6.2f / %6.2f / %6.2f / %6.2f;  8 per SIMD: %6.2f / %6.2f / %6.2f / %6.2f;  jitter x4: %6.2f\n",
                   run<1>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, lds, 16),
                   run<1>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, lds, 32),
                   run<1>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, lds, 64),
                   run<1>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, lds, 128),
                   run<1>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, 5 * 1024, 16),
                   run<1>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, 5 * 1024, 32),
                   run<1>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, 5 * 1024, 64),
                   run<1>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base, jm, c.nchunks, 5 * 1024, 128),
                   run<1>(ring, slot_f4, c.slots, c.chunk_bytes / 16, K, sleep_base ? 1 : 0, sleep_base ? 15 : 0, c.nchunks, lds, 0));
        }
        CK(hipFree(ring));
    }
    return 0;
}
