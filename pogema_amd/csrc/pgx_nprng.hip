// pgx_nprng.hip -- C-ABI of the numpy-compatible random primitives (pgx_nprng.h): many independent
// `np.random.default_rng(seed)` streams advanced in parallel, one GPU thread (or one host loop iteration) per stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "../../include/pogema_amd.h"
#include "pgx_nprng.h"

namespace pgx {
int fail_msg(int code, const char* fmt, ...);

__host__ __device__ inline void np_stream_run(uint64_t seed, int op, uint64_t n, double p, double qn, int64_t draws, void* out_row) {
    pgxnp::Pcg64 g = pgxnp::default_rng(seed);
    switch (op) {
        case PGX_NP_UINT64: { uint64_t* o = (uint64_t*)out_row; for (int64_t k = 0; k < draws; ++k) o[k] = pgxnp::next_uint64(g); break; }
        case PGX_NP_RANDOM: { double* o = (double*)out_row; for (int64_t k = 0; k < draws; ++k) o[k] = pgxnp::next_double(g); break; }
        case PGX_NP_INTEGERS: { int64_t* o = (int64_t*)out_row; for (int64_t k = 0; k < draws; ++k) o[k] = (int64_t)pgxnp::integers_below(g, n); break; }
        case PGX_NP_BINOMIAL1: { int64_t* o = (int64_t*)out_row; for (int64_t k = 0; k < draws; ++k) o[k] = pgxnp::binomial1(g, p, qn); break; }
        case PGX_NP_PERMUTATION: {
            int64_t* o = (int64_t*)out_row;
            for (int64_t k = 0; k < draws; ++k) o[k] = k;
            pgxnp::shuffle(g, o, draws);
            break;
        }
        default: break;
    }
}

__global__ void np_streams_kernel(const uint64_t* __restrict__ seeds, int64_t streams, int op, uint64_t n, double p, double qn,
                                  int64_t draws, char* out) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= streams) return;
    np_stream_run(seeds[s], op, n, p, qn, draws, out + (size_t)s * (size_t)draws * 8);
}

static int check_args(const void* seeds, int64_t streams, int op, uint64_t n, double p, int64_t draws, const void* out) {
    if (!seeds || !out || streams < 1 || draws < 1) return fail_msg(PGX_E_INVALID, "pgx_np_streams: bad argument");
    if (op < PGX_NP_UINT64 || op > PGX_NP_PERMUTATION) return fail_msg(PGX_E_INVALID, "pgx_np_streams: unknown op %d", op);
    if (op == PGX_NP_INTEGERS && n < 1) return fail_msg(PGX_E_INVALID, "pgx_np_streams: integers(0, n) needs n >= 1");
    if (op == PGX_NP_BINOMIAL1 && !(p >= 0.0 && p <= 1.0)) return fail_msg(PGX_E_INVALID, "pgx_np_streams: p outside [0, 1]");
    return PGX_OK;
}

__global__ void np_generate_kernel(const uint64_t* __restrict__ seeds, int batch, int H, int W, int A, double density, double qn,
                                   const uint8_t* __restrict__ given_map,
                                   uint8_t* obstacles, int32_t* agent_xy, int32_t* target_xy, uint32_t* scratch, int32_t* status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const size_t cells = (size_t)H * W;
    status[b] = pgxnp::generate_instance(seeds[b], H, W, A, density, qn, given_map, obstacles + (size_t)b * cells, agent_xy + (size_t)b * A * 2,
                                         target_xy + (size_t)b * A * 2, scratch + (size_t)b * 4 * cells);
}

// exp(n * log(q)) of numpy's random_binomial_inversion for n = 1, with the q numpy would use for this p
static double binomial1_qn(double p) {
    const double P = p <= 0.5 ? p : 1.0 - p;
    const double Q = 1.0 - P;
    return std::exp(1.0 * std::log(Q));
}
}  // namespace pgx

extern "C" {

int pgx_np_streams(const uint64_t* seeds, int64_t streams, int32_t op, uint64_t n, double p, int64_t draws, void* out,
                   void* stream) {
    if (int rc = pgx::check_args(seeds, streams, op, n, p, draws, out)) return rc;
    const int bs = 64;
    hipLaunchKernelGGL(pgx::np_streams_kernel, dim3((unsigned)((streams + bs - 1) / bs)), dim3(bs), 0, (hipStream_t)stream, seeds,
                       streams, (int)op, n, p, pgx::binomial1_qn(p), draws, (char*)out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pgx::fail_msg(PGX_E_HIP, "pgx_np_streams: %s", hipGetErrorString(e));
    return PGX_OK;
}

static int check_gen_args(const void* seeds, int batch, int H, int W, int A, double density, const void* a, const void* b,
                          const void* c, const void* d, const void* e) {
    if (!seeds || !a || !b || !c || !d || !e || batch < 1 || H < 1 || W < 1 || A < 1)
        return pgx::fail_msg(PGX_E_INVALID, "pgx_np_generate: bad argument");
    if (!(density >= 0.0 && density <= 1.0)) return pgx::fail_msg(PGX_E_INVALID, "pgx_np_generate: density outside [0, 1]");
    return PGX_OK;
}

int pgx_np_generate(const uint64_t* seeds, int32_t batch, int32_t height, int32_t width, int32_t num_agents, double density,
                    const uint8_t* given_map, uint8_t* obstacles, int32_t* agent_xy, int32_t* target_xy, uint32_t* scratch, int32_t* status, void* stream) {
    if (int rc = check_gen_args(seeds, batch, height, width, num_agents, density, obstacles, agent_xy, target_xy, scratch, status)) return rc;
    const int bs = 64;
    hipLaunchKernelGGL(pgx::np_generate_kernel, dim3((batch + bs - 1) / bs), dim3(bs), 0, (hipStream_t)stream, seeds, batch, height,
                       width, num_agents, density, pgx::binomial1_qn(density), given_map, obstacles, agent_xy, target_xy, scratch, status);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pgx::fail_msg(PGX_E_HIP, "pgx_np_generate: %s", hipGetErrorString(e));
    return PGX_OK;
}

int pgx_np_generate_host(const uint64_t* seeds, int32_t batch, int32_t height, int32_t width, int32_t num_agents, double density,
                         const uint8_t* given_map, uint8_t* obstacles, int32_t* agent_xy, int32_t* target_xy, uint32_t* scratch, int32_t* status) {
    if (int rc = check_gen_args(seeds, batch, height, width, num_agents, density, obstacles, agent_xy, target_xy, scratch, status)) return rc;
    const size_t cells = (size_t)height * width;
    const double qn = pgx::binomial1_qn(density);
    for (int b = 0; b < batch; ++b)
        status[b] = pgxnp::generate_instance(seeds[b], height, width, num_agents, density, qn, given_map, obstacles + (size_t)b * cells,
                                             agent_xy + (size_t)b * num_agents * 2, target_xy + (size_t)b * num_agents * 2,
                                             scratch);  // one scratch block is enough on the host
    return PGX_OK;
}

int pgx_np_streams_host(const uint64_t* seeds, int64_t streams, int32_t op, uint64_t n, double p, int64_t draws, void* out) {
    if (int rc = pgx::check_args(seeds, streams, op, n, p, draws, out)) return rc;
    const double qn = pgx::binomial1_qn(p);
    for (int64_t s = 0; s < streams; ++s) pgx::np_stream_run(seeds[s], (int)op, n, p, qn, draws, (char*)out + (size_t)s * (size_t)draws * 8);
    return PGX_OK;
}

}  // extern "C"
