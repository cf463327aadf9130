// tools/drift_probe2.hip -- follow-up of drift_probe.hip (measurement aid): WHY does a pause between a wave's store
// bursts cost HBM-sized rings 8 us per step (34 vs 26) when drift, start phases, teams and occupancy do not matter?
// Variants of the pause: s_sleep / ALU busy loop / stores spread THROUGH the pause (a store every few hundred ns) /
// wait for the burst's acknowledgements first.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// MODE 0: burst, then s_sleep.  1: burst, then ALU busy loop.  2: stores interleaved with the pause (pause split into
// chunk_f4 / 64 pieces).  3: burst, s_waitcnt vmcnt(0), then s_sleep.  4: no pause.  5: burst with plain stores + pause.
template <int MODE>
__global__ __launch_bounds__(64) void drift(f32x4* ring, size_t slot_f4, int slots, int chunk_f4, int K, int pause_units, int nchunks, float* sink) {
    extern __shared__ uint32_t pad[];
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int per_xcd = nchunks / 8;
    if (k >= per_xcd) return;
    const size_t c = (size_t)x * per_xcd + k;
    uint32_t rng = (uint32_t)c * 2654435761u + 12345u;
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    int slot = 0;
    float acc = (float)lane;
    const int nst = (chunk_f4 + 63) / 64;
    for (int t = 0; t < K; ++t) {
        rng = rng * 1664525u + 1013904223u;
        const int n = pause_units + (int)((rng >> 24) & 3u) * (pause_units ? 1 : 0);  // x 0.5 us
        f32x4* o = ring + (size_t)slot * slot_f4 + c * (size_t)chunk_f4;
        if constexpr (MODE == 2) {
            const int per = (n * 16) / nst;  // s_sleep(1) units (64 cycles) per store
            for (int i = lane, q = 0; i < chunk_f4; i += 64, ++q) {
                for (int s = 0; s < per; ++s) __builtin_amdgcn_s_sleep(1);
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&o[i]), "v"(v) : "memory");
            }
        } else {
            if constexpr (MODE == 5) { for (int i = lane; i < chunk_f4; i += 64) o[i] = v; }
            else for (int i = lane; i < chunk_f4; i += 64) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&o[i]), "v"(v) : "memory");
            if constexpr (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (MODE == 1) { for (int i = 0; i < n * 64; ++i) acc = __builtin_fmaf(acc, 1.0001f, 0.5f); }   // ~0.5 us per 64 dependent FMAs x 4 cycles... calibrated below
            else if constexpr (MODE != 4) for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
        }
        slot = slot + 1 == slots ? 0 : slot + 1;
    }
    if (pad[0] == 0xdeadbeef || acc == 12345.678f) sink[0] = acc;
}

template <int MODE>
static float run(f32x4* ring, size_t slot_f4, int slots, int chunk_f4, int K, int pause_units, int nchunks, float* sink, int lds = 10 * 1024) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto go = [&] { hipLaunchKernelGGL(drift<MODE>, dim3(nchunks), dim3(64), lds, 0, ring, slot_f4, slots, chunk_f4, K, pause_units, nchunks, sink); };
    go(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < 3; ++i) go(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 3 / K * 1e3f;
}

int main() {
    const int K = 64;
    float* sink; CK(hipMalloc(&sink, 64));
    struct Cfg { const char* name; int nchunks; int chunk_bytes; int slots; } cfgs[] = {
        {"8192 x 23 KB, 8 slots", 8192, 16 * 1452, 8}, {"8192 x 23 KB, 2 slots", 8192, 16 * 1452, 2}, {"8192 x 46 KB, 4 slots", 8192, 32 * 1452, 4},
        {"16384 x 11.6 KB, 8 slots", 16384, 8 * 1452, 8}, {"4096 x 46 KB, 8 slots", 4096, 32 * 1452, 8}};
    for (auto& c : cfgs) {
        const size_t slot_bytes = (size_t)c.nchunks * c.chunk_bytes;
        const size_t slot_f4 = ((slot_bytes + (2 << 20) - 1) / (2 << 20)) * (2 << 20) / 16;
        f32x4* ring; CK(hipMalloc(&ring, slot_f4 * 16 * c.slots)); CK(hipMemset(ring, 0, slot_f4 * 16 * c.slots));
        const int cf = c.chunk_bytes / 16;
        printf("%s (slot %.0f MB; 8 TB/s = %.1f us):\n", c.name, slot_bytes / 1e6, slot_bytes / 8e12 * 1e6);
        printf("   no pause %6.2f | pauses of 1 / 2 / 4 / 8 x 0.5 us (+ jitter): s_sleep %6.2f %6.2f %6.2f %6.2f | ALU loop %6.2f %6.2f %6.2f | stores spread through the pause %6.2f %6.2f %6.2f | drain then sleep %6.2f %6.2f | plain stores + sleep %6.2f %6.2f\n",
               run<4>(ring, slot_f4, c.slots, cf, K, 0, c.nchunks, sink),
               run<0>(ring, slot_f4, c.slots, cf, K, 1, c.nchunks, sink), run<0>(ring, slot_f4, c.slots, cf, K, 2, c.nchunks, sink),
               run<0>(ring, slot_f4, c.slots, cf, K, 4, c.nchunks, sink), run<0>(ring, slot_f4, c.slots, cf, K, 8, c.nchunks, sink),
               run<1>(ring, slot_f4, c.slots, cf, K, 2, c.nchunks, sink), run<1>(ring, slot_f4, c.slots, cf, K, 4, c.nchunks, sink), run<1>(ring, slot_f4, c.slots, cf, K, 8, c.nchunks, sink),
               run<2>(ring, slot_f4, c.slots, cf, K, 2, c.nchunks, sink), run<2>(ring, slot_f4, c.slots, cf, K, 4, c.nchunks, sink), run<2>(ring, slot_f4, c.slots, cf, K, 8, c.nchunks, sink),
               run<3>(ring, slot_f4, c.slots, cf, K, 4, c.nchunks, sink), run<3>(ring, slot_f4, c.slots, cf, K, 8, c.nchunks, sink),
               run<5>(ring, slot_f4, c.slots, cf, K, 4, c.nchunks, sink), run<5>(ring, slot_f4, c.slots, cf, K, 8, c.nchunks, sink));
        CK(hipFree(ring));
    }
    return 0;
}
