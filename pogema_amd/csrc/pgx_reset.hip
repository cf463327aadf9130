// pgx_reset.hip -- gfx950 kernels of the on-device reset path:
//   instance generator "GEN v2" (Bernoulli obstacles -> connected components -> start/target pairs),
//   replacing upstream pogema/generator.py (`generate_obstacles`, `generate_positions_and_targets_fast`,
//   the BFS component labelling) and the component tables `PogemaLifeLong` draws new targets from.
// The generator is counter-based (normative statement: oracle/generator_oracle.py, test infrastructure),
// so host (pgx_generate) and device (pgx_reset_random) produce identical instances.
//
// Everything here is HBM/L2-latency-bound integer work on the reset path, not the step hot path:
//   gen_obstacles_kernel   one thread per cell, 3 splitmix64 rounds, 1 byte out          (streaming)
//   ccl_kernel             one workgroup per environment: union-find with L2 atomicMin, every root
//                          ends as the smallest row-major index of its component           (atomics)
//   place_kernel           one lane per environment walks the candidate stream              (latency)
//   tables_kernel          one wave per environment: component sizes (L2 atomics), exclusive scan
//                          (DPP-free shuffles), stable fill with ballot grouping            (latency)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgx_internal.h"

namespace pgx {

namespace {

__device__ __forceinline__ uint64_t sm64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint64_t instance_hash(uint64_t key, uint32_t epoch, uint32_t attempt) {
    return sm64(sm64(sm64(0ull) ^ key) ^ (((uint64_t)epoch << 32) | attempt));
}

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t TAKEN = 0x80000000u;

// relaxed agent-scope accesses: served by L2, never by a (possibly stale) L1 line
__device__ __forceinline__ uint32_t ld(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t uf_find(const uint32_t* parent, uint32_t x) {
    uint32_t p = ld(parent + x);
    while (p != x) {
        x = p;
        p = ld(parent + x);
    }
    return x;
}

// lock-free union: the larger root is hung under the smaller one, retried until it sticks
__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t a, uint32_t b) {
    bool done = false;
    while (!done) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) {
            done = true;
        } else {
            if (a < b) {
                const uint32_t t = a;
                a = b;
                b = t;
            }  // a > b
            const uint32_t old = atomicMin(parent + a, b);
            done = (old == a);
            a = old;
        }
    }
}

}  // namespace

// todo/regen flags and generation counters for one pgx_reset_random call
__global__ void reset_begin_kernel(const uint8_t* __restrict__ mask, uint8_t* __restrict__ todo,
                                   uint8_t* __restrict__ regen, uint32_t* __restrict__ epoch, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const uint8_t m = mask ? (mask[b] != 0 ? 1 : 0) : 1;
    todo[b] = m;
    regen[b] = m;
    epoch[b] = mask ? epoch[b] + m : 0u;
}

// obstacles of the envs still to do: Bernoulli(thr / 2^24) per cell, or a copy of the shared map
__global__ void gen_obstacles_kernel(uint8_t* __restrict__ obst, const uint8_t* __restrict__ shared_map,
                                     const uint8_t* __restrict__ todo, const uint32_t* __restrict__ epoch,
                                     int env_begin, int env_count, int cells, uint32_t thr, uint64_t key_base,
                                     uint32_t attempt) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)env_count * cells) return;
    const int env = env_begin + (int)(gid / cells);
    const uint32_t c = (uint32_t)(gid % cells);
    if (!todo[env]) return;
    uint8_t v;
    if (shared_map) {
        v = shared_map[c] != 0 ? 1 : 0;
    } else {
        const uint64_t h = instance_hash(key_base + (uint64_t)env, epoch[env], attempt);
        v = (sm64(h ^ (GEN_TAG_OBST | (uint64_t)c)) >> 40) < thr ? 1 : 0;
    }
    obst[(size_t)env * cells + c] = v;
}

// Connected components of the FREE cells of one environment per workgroup.
// labels[c] = smallest row-major index of c's component (NONE on obstacles); pending[] cleared.
//   shared != 0: one map for all envs (GridConfig.map): blockIdx 0 labels it once into slot 0.
__global__ __launch_bounds__(256) void ccl_kernel(const uint8_t* __restrict__ obst_all, uint32_t* __restrict__ labels,
                                                  uint32_t* __restrict__ pending, const uint8_t* __restrict__ todo,
                                                  int env_begin, int H, int Wd, int shared) {
    const int local = blockIdx.x;
    const int env = env_begin + local;
    const int cells = H * Wd;
    uint32_t* pend = pending + (size_t)local * cells;
    const bool mine = todo[env] != 0;
    if (mine)
        for (int c = threadIdx.x; c < cells; c += blockDim.x) pend[c] = 0u;
    if (shared ? (local != 0) : !mine) return;
    const uint8_t* obst = obst_all + (shared ? 0 : (size_t)env * cells);  // shared: obst_all IS the one map
    uint32_t* parent = labels + (shared ? 0 : (size_t)local * cells);
    for (int c = threadIdx.x; c < cells; c += blockDim.x) st(parent + c, obst[c] ? NONE : (uint32_t)c);
    __syncthreads();
    for (int c = threadIdx.x; c < cells; c += blockDim.x) {
        if (obst[c]) continue;
        const int x = c / Wd, y = c - x * Wd;
        if (y > 0 && !obst[c - 1]) uf_union(parent, (uint32_t)c, (uint32_t)(c - 1));
        if (x > 0 && !obst[c - Wd]) uf_union(parent, (uint32_t)c, (uint32_t)(c - Wd));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cells; c += blockDim.x)
        if (!obst[c]) st(parent + c, uf_find(parent, (uint32_t)c));  // in-place compression: still a valid ancestor
}

// One lane per environment walks the candidate stream of GEN v2 and closes start/target pairs.
// pending[c]: bit 31 = cell c already taken; low bits (root entries only) = open start cell + 1.
__global__ void place_kernel(const uint8_t* __restrict__ obst_all, const uint32_t* __restrict__ labels,
                             uint32_t* __restrict__ pending, uint8_t* __restrict__ todo,
                             const uint32_t* __restrict__ epoch, uint32_t* __restrict__ pos, uint32_t* __restrict__ tgt,
                             uint32_t* __restrict__ pos0, uint32_t* __restrict__ tgt0, uint8_t* __restrict__ active,
                             uint32_t* __restrict__ tcount, int32_t* __restrict__ elapsed, int4* __restrict__ macc,
                             uint32_t* __restrict__ fail_count, int env_begin, int env_count, int A, int Wd, int cells,
                             int r, int shared, uint64_t key_base, uint32_t attempt) {
    const int local = blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= env_count) return;
    const int env = env_begin + local;
    if (!todo[env]) return;
    const uint8_t* obst = obst_all + (size_t)env * cells;
    const uint32_t* lab = labels + (shared ? 0 : (size_t)local * cells);
    uint32_t* pend = pending + (size_t)local * cells;
    const uint64_t h = instance_hash(key_base + (uint64_t)env, epoch[env], attempt);
    const uint32_t budget = gen_candidate_budget((uint32_t)cells);
    const uint32_t pad = ((uint32_t)r << 16) | (uint32_t)r;
    int placed = 0;
    for (uint32_t t = 0; t < budget && placed < A; ++t) {
        const uint32_t c = (uint32_t)(((sm64(h ^ (GEN_TAG_PLACE | (uint64_t)t)) >> 32) * (uint64_t)cells) >> 32);
        if (obst[c]) continue;
        const uint32_t mark = pend[c];
        if (mark & TAKEN) continue;
        pend[c] = mark | TAKEN;
        const uint32_t root = lab[c];
        const uint32_t v = (root == c) ? (mark | TAKEN) : pend[root];
        const uint32_t open = v & ~TAKEN;
        if (open == 0u) {
            pend[root] = v | (c + 1u);
        } else {
            pend[root] = v & TAKEN;
            const uint32_t s = open - 1u;
            const size_t gi = (size_t)env * A + placed;
            const uint32_t pc = (((s / Wd) << 16) | (s % Wd)) + pad;
            const uint32_t tc = (((c / Wd) << 16) | (c % Wd)) + pad;
            pos[gi] = pc;
            pos0[gi] = pc;
            tgt[gi] = tc;
            tgt0[gi] = tc;
            active[gi] = 1;
            if (tcount) tcount[gi] = 0u;
            ++placed;
        }
    }
    if (placed == A) {
        todo[env] = 0;
        elapsed[env] = 0;
        macc[env] = make_int4(0, 0, 0, 0);
    } else {
        atomicAdd(fail_count, 1u);
    }
}

// Lifelong component tables of one environment per wave (stable counting sort of the free cells by
// component): comp_begin/comp_len per cell, comp_cells = unpadded (x << 16) | y grouped by component in
// order of the components' first cells, row-major inside.  `counter` (scratch, one word per cell) is used
// for the component sizes and then as the fill pointers.
__global__ __launch_bounds__(64) void tables_kernel(const uint8_t* __restrict__ obst_all,
                                                    const uint32_t* __restrict__ labels, uint32_t* __restrict__ counters,
                                                    const uint8_t* __restrict__ regen, uint32_t* __restrict__ comp_begin,
                                                    uint32_t* __restrict__ comp_len, uint32_t* __restrict__ comp_cells,
                                                    int env_begin, int Wd, int cells, int shared) {
    const int local = blockIdx.x;
    const int env = env_begin + local;
    if (!regen[env]) return;
    const int lane = threadIdx.x;
    const uint8_t* obst = obst_all + (size_t)env * cells;
    const uint32_t* lab = labels + (shared ? 0 : (size_t)local * cells);
    uint32_t* cnt = counters + (size_t)local * cells;
    uint32_t* cb = comp_begin + (size_t)env * cells;
    uint32_t* cl = comp_len + (size_t)env * cells;
    uint32_t* cc = comp_cells + (size_t)env * cells;
    for (int c = lane; c < cells; c += 64) st(cnt + c, 0u);
    __syncthreads();
    for (int c = lane; c < cells; c += 64)
        if (!obst[c]) atomicAdd(cnt + lab[c], 1u);
    __syncthreads();
    // exclusive scan of the component sizes in order of the roots' cell indices
    uint32_t carry = 0u;
    for (int base = 0; base < cells; base += 64) {
        const int c = base + lane;
        const uint32_t v = (c < cells) ? ld(cnt + c) : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (c < cells) {
            cb[c] = v ? (carry + incl - v) : 0u;
            cl[c] = v;
            if (v) st(cnt + c, carry + incl - v);  // fill pointer of this component
        }
        carry += (uint32_t)__shfl((int)incl, 63, 64);
    }
    __syncthreads();
    // every free cell copies its component's slice (roots already hold theirs)
    for (int c = lane; c < cells; c += 64) {
        if (obst[c]) continue;
        const uint32_t root = lab[c];
        if (root != (uint32_t)c) {
            cb[c] = cb[root];
            cl[c] = cl[root];
        }
    }
    // stable fill: 64 consecutive cells at a time, lanes of one component take consecutive slots
    for (int base = 0; base < cells; base += 64) {
        const int c = base + lane;
        const bool free_cell = (c < cells) && !obst[c];
        const uint32_t root = free_cell ? lab[c] : NONE;
        unsigned long long left = __ballot(free_cell);
        while (left) {
            const int leader = __ffsll((long long)left) - 1;
            const uint32_t lr = (uint32_t)__shfl((int)root, leader, 64);
            const unsigned long long m = __ballot(free_cell && root == lr);
            uint32_t slot0 = 0u;
            if (lane == leader) slot0 = atomicAdd(cnt + lr, (uint32_t)__popcll(m));
            slot0 = (uint32_t)__shfl((int)slot0, leader, 64);
            if (free_cell && root == lr) {
                const uint32_t below = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                cc[slot0 + below] = ((uint32_t)(c / Wd) << 16) | (uint32_t)(c % Wd);
            }
            left &= ~m;
        }
    }
}

// agent/target cells -> validity flags for explicitly given states are checked on the host (VecPogema); nothing here.

// ---- launchers --------------------------------------------------------------------------------------
hipError_t launch_reset_begin(const uint8_t* mask, uint8_t* todo, uint8_t* regen, uint32_t* epoch, int batch,
                              hipStream_t s) {
    hipLaunchKernelGGL(reset_begin_kernel, dim3((batch + 255) / 256), dim3(256), 0, s, mask, todo, regen, epoch, batch);
    return hipGetLastError();
}

hipError_t launch_gen_obstacles(uint8_t* obst, const uint8_t* shared_map, const uint8_t* todo, const uint32_t* epoch,
                                int env_begin, int env_count, int cells, uint32_t thr, uint64_t key_base,
                                uint32_t attempt, hipStream_t s) {
    const size_t total = (size_t)env_count * cells;
    hipLaunchKernelGGL(gen_obstacles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, obst, shared_map,
                       todo, epoch, env_begin, env_count, cells, thr, key_base, attempt);
    return hipGetLastError();
}

hipError_t launch_ccl(const uint8_t* obst, uint32_t* labels, uint32_t* pending, const uint8_t* todo, int env_begin,
                      int env_count, int H, int Wd, int shared, hipStream_t s) {
    hipLaunchKernelGGL(ccl_kernel, dim3(env_count), dim3(256), 0, s, obst, labels, pending, todo, env_begin, H, Wd,
                       shared);
    return hipGetLastError();
}

hipError_t launch_place(const uint8_t* obst, const uint32_t* labels, uint32_t* pending, uint8_t* todo,
                        const uint32_t* epoch, uint32_t* pos, uint32_t* tgt, uint32_t* pos0, uint32_t* tgt0,
                        uint8_t* active, uint32_t* tcount, int32_t* elapsed, int4* macc, uint32_t* fail_count,
                        int env_begin, int env_count, int A, int Wd, int cells, int r, int shared, uint64_t key_base,
                        uint32_t attempt, hipStream_t s) {
    hipLaunchKernelGGL(place_kernel, dim3((env_count + 63) / 64), dim3(64), 0, s, obst, labels, pending, todo, epoch,
                       pos, tgt, pos0, tgt0, active, tcount, elapsed, macc, fail_count, env_begin, env_count, A, Wd,
                       cells, r, shared, key_base, attempt);
    return hipGetLastError();
}

hipError_t launch_tables(const uint8_t* obst, const uint32_t* labels, uint32_t* counters, const uint8_t* regen,
                         uint32_t* comp_begin, uint32_t* comp_len, uint32_t* comp_cells, int env_begin, int env_count,
                         int Wd, int cells, int shared, hipStream_t s) {
    hipLaunchKernelGGL(tables_kernel, dim3(env_count), dim3(64), 0, s, obst, labels, counters, regen, comp_begin,
                       comp_len, comp_cells, env_begin, Wd, cells, shared);
    return hipGetLastError();
}

}  // namespace pgx
