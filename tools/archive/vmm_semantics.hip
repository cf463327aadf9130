// tools/vmm_semantics.hip -- measurement aid (not part of the product): two facts about ROCm's virtual-memory API that
// pgx_buffers.hip depends on.  (1) Does hipMemUnmap give the physical memory of a still-live handle back (so that the
// next allocation can land on it)?  (2) Can one handle be mapped at two virtual addresses at once?
// Build: hipcc --offload-arch=gfx950 -O3 tools/vmm_semantics.hip -o tools/vmm_semantics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main() {
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t SZ = (size_t)384 << 20;
    size_t f0, f1, f2, f3, tot;
    CK(hipMemGetInfo(&f0, &tot));
    hipMemGenericAllocationHandle_t A; CK(hipMemCreate(&A, SZ, &prop, 0));
    CK(hipMemGetInfo(&f1, &tot));
    void* va; CK(hipMemAddressReserve(&va, SZ, 2 << 20, nullptr, 0)); CK(hipMemMap(va, SZ, 0, A, 0)); CK(hipMemSetAccess(va, SZ, &acc, 1));
    CK(hipMemGetInfo(&f2, &tot));
    CK(hipMemset(va, 0xAB, SZ)); CK(hipDeviceSynchronize());
    CK(hipMemUnmap(va, SZ));
    CK(hipMemGetInfo(&f3, &tot));
    printf("free memory [MiB]: start %zu, after hipMemCreate %zu, after map+access %zu, after unmap (handle alive) %zu\n", f0 >> 20, f1 >> 20, f2 >> 20, f3 >> 20);
    // fill other memory meanwhile, then map A again elsewhere and look at its contents
    std::vector<void*> junk;
    for (int i = 0; i < 8; ++i) { void* p; CK(hipMalloc(&p, SZ)); CK(hipMemset(p, 0x11, SZ)); junk.push_back(p); }
    void* vb; CK(hipMemAddressReserve(&vb, SZ, 2 << 20, nullptr, 0)); CK(hipMemMap(vb, SZ, 0, A, 0)); CK(hipMemSetAccess(vb, SZ, &acc, 1));
    std::vector<unsigned char> h(1 << 20);
    CK(hipMemcpy(h.data(), (char*)vb + (SZ / 2), h.size(), hipMemcpyDeviceToHost));
    size_t same = 0; for (unsigned char c : h) same += c == 0xAB;
    printf("(1) contents after unmap + other allocations + remap: %zu of %zu bytes still 0xAB -> unmap %s the physical memory\n", same, h.size(),
           same == h.size() ? "KEEPS" : "RELEASES");
    // (2) second mapping of the same handle while the first is alive
    void* vc; CK(hipMemAddressReserve(&vc, SZ, 2 << 20, nullptr, 0));
    hipError_t e = hipMemMap(vc, SZ, 0, A, 0);
    printf("(2) second simultaneous mapping of one handle: hipMemMap -> %s", hipGetErrorString(e));
    if (e == hipSuccess) {
        e = hipMemSetAccess(vc, SZ, &acc, 1);
        printf(", hipMemSetAccess -> %s", hipGetErrorString(e));
        if (e == hipSuccess) {
            CK(hipMemset(vb, 0x5C, 4096)); CK(hipDeviceSynchronize());
            unsigned char probe[16]; CK(hipMemcpy(probe, vc, 16, hipMemcpyDeviceToHost));
            printf(", write through mapping 1 visible through mapping 2: %s", probe[0] == 0x5C ? "yes" : "NO");
        }
    }
    printf("\n");
    return 0;
}
