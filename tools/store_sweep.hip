// tools/store_sweep.hip -- measurement aid (not part of the product): which 16-B store flavour / chunking sustains
// the highest HBM write rate for a 761 MB tensor (= one configs[2] observation tensor).
// Build: hipcc --offload-arch=gfx950 -O3 tools/store_sweep.hip -o tools/store_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__device__ __forceinline__ void store16(f32x4* p, f32x4 v) {
    if constexpr (MODE == 0) *p = v;
    else if constexpr (MODE == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// each workgroup streams one contiguous chunk of `per_block` float4s
template <int MODE>
__global__ void fill_chunk(f32x4* __restrict__ out, size_t per_block) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    f32x4* o = out + (size_t)blockIdx.x * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += blockDim.x) store16<MODE>(&o[i], v);
}
// block b handles chunk perm(b): decorrelate launch order and address order (XCD b%8 gets every 8th chunk)
template <int MODE>
__global__ void fill_chunk_swz(f32x4* __restrict__ out, size_t per_block, int nblk) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    const int b = (blockIdx.x % 8) * (nblk / 8) + blockIdx.x / 8;  // chunks of one XCD contiguous in memory
    f32x4* o = out + (size_t)b * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += blockDim.x) store16<MODE>(&o[i], v);
}
// XCD x owns groups of `g` consecutive chunks, groups interleaved over the 8 XCDs (g = 1: identity mapping; g = nblk / 8:
// the XCD-contiguous mapping)
template <int MODE>
__global__ void fill_chunk_group(f32x4* __restrict__ out, size_t per_block, int g) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int b = ((i / g) * 8 + x) * g + (i % g);
    f32x4* o = out + (size_t)b * per_block;
    for (size_t k = threadIdx.x; k < per_block; k += blockDim.x) store16<MODE>(&o[k], v);
}
// chunks handed out per CU: the waves resident on one CU stream ADJACENT chunks (first-come dense CU rank + per-CU
// ticket), so a CU touches ~3 MB of contiguous address space instead of 32 pages spread over its XCD's eighth
__global__ void fill_chunk_cu(f32x4* __restrict__ out, size_t per_block, int nblk, unsigned* table, unsigned* tickets,
                              unsigned* next_rank, unsigned* overflow) {
    __shared__ unsigned s_chunk;
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID[3:0]
        const unsigned key = (xcc << 8) | ((hw >> 8) & 0xFF);                          // xcc | se,sh,cu bits [15:8]
        unsigned rank = atomicCAS(&table[key], 0u, 0xFFFFFFFFu);
        if (rank == 0u) {                      // first wave of this CU: take a dense rank
            rank = atomicAdd(next_rank, 1u) + 1u;
            atomicExch(&table[key], rank);
        } else {
            while (rank == 0xFFFFFFFFu) rank = atomicAdd(&table[key], 0u);
        }
        const unsigned per_cu = (unsigned)nblk / 256u;
        const unsigned t = atomicAdd(&tickets[rank - 1u], 1u);
        s_chunk = (rank - 1u < 256u && t < per_cu) ? (rank - 1u) * per_cu + t : 0x80000000u | atomicAdd(overflow, 1u);
    }
    __syncthreads();
    unsigned chunk = s_chunk;
    if (chunk & 0x80000000u) return;  // microbenchmark only: overflow chunks are dropped (reported by the host)
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    f32x4* o = out + (size_t)chunk * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += blockDim.x) o[i] = v;
}
template <typename F> static float time_us(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps * 1e3f;
}
int main() {
    const size_t bytes = (size_t)8192 * 64 * 1452;
    const size_t n = bytes / 16;
    f32x4* a; CK(hipMalloc(&a, bytes)); CK(hipMemset(a, 0, bytes));
    const int reps = 20;
    const char* names[] = {"plain", "nt", "sc1", "sc0 sc1", "sc0", "sc0 sc1 nt"};
#define RUN(MODE)                                                                                          \
    {                                                                                                      \
        float t = time_us([&] { hipLaunchKernelGGL(fill_chunk<MODE>, dim3(8192), dim3(64), 0, 0, a, n / 8192); }, reps); \
        printf("8192 waves x 93KB  %-10s : %7.1f GB/s (%.1f us)\n", names[MODE], bytes / t / 1e3, t);          \
    }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    for (int waves : {2048, 4096, 16384, 32768, 65536}) {
        float t = time_us([&] { hipLaunchKernelGGL(fill_chunk<0>, dim3(waves), dim3(64), 0, 0, a, n / waves); }, reps);
        printf("%5d waves x %4zuKB plain      : %7.1f GB/s (%.1f us)\n", waves, bytes / waves / 1024, bytes / t / 1e3, t);
    }
    for (int thr : {128, 256, 512, 1024}) {
        const int blocks = 8192 * 64 / thr;
        float t = time_us([&] { hipLaunchKernelGGL(fill_chunk<0>, dim3(blocks), dim3(thr), 0, 0, a, n / blocks); }, reps);
        printf("%5d blocks x %4d thr x %4zuKB : %7.1f GB/s (%.1f us)\n", blocks, thr, bytes / blocks / 1024, bytes / t / 1e3, t);
    }
    {
        float t = time_us([&] { hipLaunchKernelGGL(fill_chunk_swz<0>, dim3(8192), dim3(64), 0, 0, a, n / 8192, 8192); }, reps);
        printf("8192 waves x 93KB  XCD-contig  : %7.1f GB/s (%.1f us)\n", bytes / t / 1e3, t);
    }
    for (int g : {1, 4, 16, 64, 256, 1024}) {
        float t = time_us([&] { hipLaunchKernelGGL(fill_chunk_group<0>, dim3(8192), dim3(64), 0, 0, a, n / 8192, g); }, reps);
        printf("8192 waves x 93KB  XCD groups of %4d chunks : %7.1f GB/s (%.1f us)\n", g, bytes / t / 1e3, t);
    }
    {
        unsigned *table, *tickets, *misc;
        CK(hipMalloc(&table, 4096 * 4)); CK(hipMalloc(&tickets, 1024 * 4)); CK(hipMalloc(&misc, 8));
        auto run = [&] {
            hipMemsetAsync(table, 0, 4096 * 4, 0); hipMemsetAsync(tickets, 0, 1024 * 4, 0); hipMemsetAsync(misc, 0, 8, 0);
            hipLaunchKernelGGL(fill_chunk_cu, dim3(8192), dim3(64), 0, 0, a, n / 8192, 8192, table, tickets, misc, misc + 1);
        };
        float t = time_us(run, reps);
        unsigned h[2]; CK(hipMemcpy(h, misc, 8, hipMemcpyDeviceToHost));
        printf("8192 waves x 93KB  CU-contig   : %7.1f GB/s (%.1f us)  [CUs seen %u, overflow chunks %u]\n", bytes / t / 1e3, t, h[0], h[1]);
    }
    return 0;
}
