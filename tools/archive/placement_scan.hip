// tools/placement_scan.hip -- measurement aid (not part of the product): the speed tier of a 761 MB store stream as a
// function of the window's BASE OFFSET inside one big hipMalloc slab and of the per-wave chunk STRIDE.
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement_scan.hip -o tools/placement_scan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
// 8192 waves; wave b streams `per_block` float4 starting at b * stride16 float4 (XCD-contiguous mapping)
__global__ void fill(f32x4* __restrict__ out, int per_block, size_t stride16, int nblk) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    const int per_xcd = nblk >> 3;
    const int b = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    f32x4* o = out + (size_t)b * stride16;
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) o[i] = v;
}
static const int CHUNK = 92928;  // configs[2]: bytes of observations per env
static hipEvent_t ea, eb;
static float t_us(char* base, size_t stride = CHUNK, int reps = 8) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fill, dim3(8192), dim3(64), 0, 0, (f32x4*)base, CHUNK / 16, stride / 16, 8192);
    CK(hipEventRecord(ea));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fill, dim3(8192), dim3(64), 0, 0, (f32x4*)base, CHUNK / 16, stride / 16, 8192);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    return ms / reps * 1e3f;
}
int main() {
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t SLAB = (size_t)26 << 30, WIN = (size_t)8192 * CHUNK;
    char* slab; CK(hipMalloc((void**)&slab, SLAB)); CK(hipMemset(slab, 0, SLAB));
    printf("slab %p, %zu GiB\n", slab, SLAB >> 30);
    // a. coarse scan, 16 MiB steps
    const size_t step = (size_t)16 << 20;
    std::vector<size_t> fast;
    int n = 0, nfast = 0;
    printf("# a. coarse scan (16 MiB steps): offsets [MiB] of windows faster than 125 us\n");
    for (size_t off = 0; off + WIN + ((size_t)1 << 30) <= SLAB; off += step, ++n) {
        const float t = t_us(slab + off);
        if (t < 125.f) { printf("  +%zu MiB %.1f us\n", off >> 20, t); fast.push_back(off); ++nfast; }
    }
    printf("  %d of %d windows fast\n", nfast, n);
    if (fast.empty()) { printf("no fast window found\n"); return 0; }
    const size_t f0 = fast[fast.size() / 2];
    // b. fine scans around one fast offset
    for (size_t gran : {(size_t)1 << 20, (size_t)64 << 10, (size_t)4 << 10, (size_t)256}) {
        printf("# b. around +%zu MiB in steps of %zu B: ", f0 >> 20, gran);
        for (int k = -12; k <= 12; ++k) {
            const long long o = (long long)f0 + (long long)k * (long long)gran;
            if (o < 0) continue;
            printf("%.0f ", t_us(slab + o));
        }
        printf("\n");
    }
    // c. chunk stride on a fast and a slow base (same bytes written per wave, gaps between chunks)
    size_t slow = 0;
    for (size_t off = 0; off + WIN * 2 <= SLAB; off += step) { bool isf = false; for (size_t f : fast) if (f == off) isf = true; if (!isf) { slow = off; break; } }
    printf("# c. chunk stride (bytes between the starts of consecutive waves' chunks): fast base +%zu MiB | slow base +%zu MiB\n", f0 >> 20, slow >> 20);
    for (size_t stride : {92928, 92928 + 256, 92928 + 512, 92928 + 1024, 92928 + 2048, 92928 + 4096 - 2816 /* 94208 = 23 * 4096 */, 98304 /* 96 KiB */, 131072}) {
        printf("  stride %6zu: %7.1f | %7.1f us\n", stride, t_us(slab + f0, stride), t_us(slab + slow, stride));
    }
    // d. is the fast offset still fast later? (temporal stability)
    printf("# d. repeat: fast %.1f slow %.1f | fast %.1f slow %.1f\n", t_us(slab + f0), t_us(slab + slow), t_us(slab + f0), t_us(slab + slow));
    return 0;
}
