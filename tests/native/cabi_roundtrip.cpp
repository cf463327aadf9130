// tests/native/cabi_roundtrip.cpp -- a host program with NO Python and NO torch in it: it drives the engine purely
// through the C-ABI of include/pogema_amd.h (plain pointers from hipMalloc), the way a non-Python caller would bind
// it.  Prints one line of FNV-1a checksums that tests/test_cabi_native_gpu.py compares with the CPU oracle's.
//   usage: cabi_roundtrip <batch> <size> <agents> <obs_radius> <collision 0..2> <on_target 0..2> <steps> <seed>
// Build: make -C tests/native   (hipcc, links ../../pogema_amd/libpogema_amd.so)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/pogema_amd.h"

#define HIPCK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            return 2;                                                               \
        }                                                                           \
    } while (0)
#define PGXCK(x)                                                                    \
    do {                                                                            \
        if ((x) != PGX_OK) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, pgx_last_error());                      \
            return 3;                                                               \
        }                                                                           \
    } while (0)

static uint64_t fnv(uint64_t h, const void* data, size_t n) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}

int main(int argc, char** argv) {
    if (argc != 9) {
        fprintf(stderr, "usage: %s batch size agents obs_radius collision on_target steps seed\n", argv[0]);
        return 1;
    }
    const int B = atoi(argv[1]), S = atoi(argv[2]), A = atoi(argv[3]), r = atoi(argv[4]);
    const int collision = atoi(argv[5]), on_target = atoi(argv[6]), T = atoi(argv[7]);
    const uint64_t seed = strtoull(argv[8], nullptr, 10);
    const int W = 2 * r + 1;

    pgx_config cfg{};
    cfg.batch = B; cfg.height = S; cfg.width = S; cfg.num_agents = A; cfg.obs_radius = r;
    cfg.collision_system = collision; cfg.on_target = on_target; cfg.max_episode_steps = 7; cfg.auto_reset = 1;
    cfg.seed = seed; cfg.env_index_base = 5;
    cfg.abi_version = PGX_ABI_VERSION;
    if (pgx_abi_version() != PGX_ABI_VERSION) return 4;
    pgx_env* env = nullptr;
    PGXCK(pgx_create(&cfg, 0, &env));

    const size_t BA = (size_t)B * A, n_obs = (size_t)pgx_obs_elems(env);
    if (pgx_agent_elems(env) != (int64_t)BA || n_obs != BA * 3 * W * W) return 5;
    int8_t* d_act; float *d_obs, *d_rew; uint8_t *d_term, *d_trunc, *d_active, *d_map; int32_t *d_axy, *d_txy, *d_elapsed;
    HIPCK(hipMalloc(&d_act, BA)); HIPCK(hipMalloc(&d_obs, n_obs * 4)); HIPCK(hipMalloc(&d_rew, BA * 4));
    HIPCK(hipMalloc(&d_term, BA)); HIPCK(hipMalloc(&d_trunc, BA)); HIPCK(hipMalloc(&d_active, BA));
    HIPCK(hipMalloc(&d_map, (size_t)B * S * S)); HIPCK(hipMalloc(&d_axy, BA * 8)); HIPCK(hipMalloc(&d_txy, BA * 8));
    HIPCK(hipMalloc(&d_elapsed, (size_t)B * 4));
    hipStream_t stream;
    HIPCK(hipStreamCreate(&stream));

    // a step before any reset must be refused, not crash
    if (pgx_step(env, d_act, PGX_ACTION_I8, d_obs, d_rew, d_term, d_trunc, d_active, stream) != PGX_E_STATE) return 6;

    PGXCK(pgx_reset_random(env, 0.3f, seed, nullptr, nullptr, 10, stream));
    PGXCK(pgx_observe(env, d_obs, stream));
    std::vector<float> obs(n_obs), rew(BA);
    std::vector<uint8_t> term(BA), trunc(BA), active(BA), map((size_t)B * S * S);
    std::vector<int32_t> axy(BA * 2), txy(BA * 2), elapsed(B);
    std::vector<int8_t> act(BA);
    PGXCK(pgx_get_map(env, d_map, stream));
    PGXCK(pgx_get_state(env, d_axy, d_txy, nullptr, nullptr, nullptr, stream));
    HIPCK(hipMemcpyAsync(obs.data(), d_obs, n_obs * 4, hipMemcpyDeviceToHost, stream));
    HIPCK(hipMemcpyAsync(map.data(), d_map, map.size(), hipMemcpyDeviceToHost, stream));
    HIPCK(hipMemcpyAsync(axy.data(), d_axy, BA * 8, hipMemcpyDeviceToHost, stream));
    HIPCK(hipMemcpyAsync(txy.data(), d_txy, BA * 8, hipMemcpyDeviceToHost, stream));
    HIPCK(hipStreamSynchronize(stream));
    uint64_t h_reset = 0xCBF29CE484222325ull;
    h_reset = fnv(h_reset, map.data(), map.size());
    h_reset = fnv(h_reset, axy.data(), BA * 8);
    h_reset = fnv(h_reset, txy.data(), BA * 8);
    uint64_t h_obs = fnv(0xCBF29CE484222325ull, obs.data(), n_obs * 4);
    uint64_t h_flags = 0xCBF29CE484222325ull, h_state = 0xCBF29CE484222325ull;

    for (int t = 0; t < T; ++t) {
        for (size_t g = 0; g < BA; ++g) act[g] = (int8_t)((t * 7 + (g / A) * 3 + (g % A) * 5 + ((t + g) >> 2)) % 5);
        HIPCK(hipMemcpyAsync(d_act, act.data(), BA, hipMemcpyHostToDevice, stream));
        PGXCK(pgx_step(env, d_act, PGX_ACTION_I8, d_obs, d_rew, d_term, d_trunc, d_active, stream));
        PGXCK(pgx_get_state(env, d_axy, d_txy, nullptr, d_elapsed, nullptr, stream));
        HIPCK(hipMemcpyAsync(obs.data(), d_obs, n_obs * 4, hipMemcpyDeviceToHost, stream));
        HIPCK(hipMemcpyAsync(rew.data(), d_rew, BA * 4, hipMemcpyDeviceToHost, stream));
        HIPCK(hipMemcpyAsync(term.data(), d_term, BA, hipMemcpyDeviceToHost, stream));
        HIPCK(hipMemcpyAsync(trunc.data(), d_trunc, BA, hipMemcpyDeviceToHost, stream));
        HIPCK(hipMemcpyAsync(active.data(), d_active, BA, hipMemcpyDeviceToHost, stream));
        HIPCK(hipMemcpyAsync(axy.data(), d_axy, BA * 8, hipMemcpyDeviceToHost, stream));
        HIPCK(hipMemcpyAsync(txy.data(), d_txy, BA * 8, hipMemcpyDeviceToHost, stream));
        HIPCK(hipMemcpyAsync(elapsed.data(), d_elapsed, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        HIPCK(hipStreamSynchronize(stream));
        h_obs = fnv(h_obs, obs.data(), n_obs * 4);
        h_flags = fnv(h_flags, rew.data(), BA * 4);
        h_flags = fnv(h_flags, term.data(), BA);
        h_flags = fnv(h_flags, trunc.data(), BA);
        h_flags = fnv(h_flags, active.data(), BA);
        h_state = fnv(h_state, axy.data(), BA * 8);
        h_state = fnv(h_state, txy.data(), BA * 8);
        h_state = fnv(h_state, elapsed.data(), (size_t)B * 4);
    }
    printf("reset=%016llx obs=%016llx flags=%016llx state=%016llx\n", (unsigned long long)h_reset,
           (unsigned long long)h_obs, (unsigned long long)h_flags, (unsigned long long)h_state);
    PGXCK(pgx_destroy(env));
    return 0;
}
