// tools/hbm_peak.hip -- measurement aid (not part of the product): what a pure 16-B-per-lane store
// stream / copy stream sustains on this GPU, i.e. the practical ceiling for the observation write
// of pgx::step_kernel.  Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_peak.hip -o tools/hbm_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NT_STORE>
__global__ void fill_kernel(f32x4* __restrict__ out, size_t n) {
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT_STORE) __builtin_nontemporal_store(v, &out[i]); else out[i] = v;
    }
}
// contiguous chunk per workgroup (the access pattern of step_kernel: one wave streams ~93 KB)
template <int NT_STORE>
__global__ void fill_chunk_kernel(f32x4* __restrict__ out, size_t per_block) {
    extern __shared__ float lds_pad[];  // only to limit occupancy like the real kernel's LDS does
    if (per_block == 0) lds_pad[threadIdx.x] = 0.f;
    const f32x4 v = {1.f, 0.f, 1.f, 0.f};
    f32x4* o = out + (size_t)blockIdx.x * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += blockDim.x) {
        if (NT_STORE) __builtin_nontemporal_store(v, &o[i]); else o[i] = v;
    }
}
__global__ void copy_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
template <typename F> static float time_ms(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
    const size_t bytes = (size_t)8192 * 64 * 1452;  // = one configs[2] observation tensor (761 MB)
    const size_t n = bytes / 16;
    f32x4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    const int reps = 20;
    float t;
    t = time_ms([&] { hipLaunchKernelGGL(fill_kernel<0>, dim3(2048), dim3(256), 0, 0, a, n); }, reps);
    printf("fill  plain   grid-stride 2048x256 : %.1f GB/s (%.1f us)\n", bytes / t / 1e6, t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL(fill_kernel<1>, dim3(2048), dim3(256), 0, 0, a, n); }, reps);
    printf("fill  nt      grid-stride 2048x256 : %.1f GB/s (%.1f us)\n", bytes / t / 1e6, t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL(fill_chunk_kernel<0>, dim3(8192), dim3(64), 0, 0, a, n / 8192); }, reps);
    printf("fill  plain   8192 waves x 93KB    : %.1f GB/s (%.1f us)\n", bytes / t / 1e6, t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL(fill_chunk_kernel<1>, dim3(8192), dim3(64), 0, 0, a, n / 8192); }, reps);
    printf("fill  nt      8192 waves x 93KB    : %.1f GB/s (%.1f us)\n", bytes / t / 1e6, t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL(fill_chunk_kernel<1>, dim3(8192), dim3(256), 0, 0, a, n / 8192); }, reps);
    printf("fill  nt      8192 x 256thr x 93KB : %.1f GB/s (%.1f us)\n", bytes / t / 1e6, t * 1e3);
    for (int lds : {4600, 5100, 6600, 8200, 10900, 20000}) {
        t = time_ms([&] { hipLaunchKernelGGL(fill_chunk_kernel<0>, dim3(8192), dim3(64), lds, 0, a, n / 8192); }, reps);
        printf("fill  plain   8192 waves x 93KB, %5d B LDS/wave (%2d waves/CU): %.1f GB/s (%.1f us)\n", lds,
               (160 * 1024 / lds) > 32 ? 32 : (160 * 1024 / lds), bytes / t / 1e6, t * 1e3);
    }
    t = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, a, b, n); }, reps);
    printf("copy  float4  grid-stride 2048x256 : %.1f GB/s r+w (%.1f us)\n", 2.0 * bytes / t / 1e6, t * 1e3);
    return 0;
}
