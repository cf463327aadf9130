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
//   place_kernel           one wave per environment: 64 candidates fetched at a time, the order-dependent
//                          pairing replayed with wave-uniform registers                     (latency)
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

// Relaxed atomic accesses to the union-find forest.  HBM-resident forest: agent scope = served by L2, never by a
// (possibly stale) L1 line.  LDS-resident forest: workgroup scope.
template <bool LDS>
__device__ __forceinline__ uint32_t ld(const uint32_t* p) {
    if constexpr (LDS) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool LDS>
__device__ __forceinline__ void st(uint32_t* p, uint32_t v) {
    if constexpr (LDS) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool LDS>
__device__ __forceinline__ uint32_t uf_find(const uint32_t* parent, uint32_t x) {
    uint32_t p = ld<LDS>(parent + x);
    while (p != x) {
        x = p;
        p = ld<LDS>(parent + x);
    }
    return x;
}

// lock-free union: the larger root is hung under the smaller one, retried until it sticks
template <bool LDS>
__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t a, uint32_t b) {
    bool done = false;
    while (!done) {
        a = uf_find<LDS>(parent, a);
        b = uf_find<LDS>(parent, b);
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
//   1. horizontal runs: every free cell points at the first cell of its run (one wave per row, ballot + clz,
//      no atomics);
//   2. vertical merges: only where a run STARTS touching an upper run (the left neighbours are not both free) --
//      ~0.25 unions per cell at density 0.3 instead of ~1 -- lock-free union-find, larger root under smaller;
//   3. every cell resolves its root (in-place compression).
//   LDS = true: the forest lives in LDS (maps up to 128 x 128), only the final labels go to HBM.
//   shared != 0: one map for all envs (GridConfig.map): blockIdx 0 labels it once into slot 0.
template <bool LDS>
__global__ __launch_bounds__(256) void ccl_kernel(const uint8_t* __restrict__ obst_all, uint32_t* __restrict__ labels,
                                                  uint32_t* __restrict__ pending, const uint8_t* __restrict__ todo,
                                                  int env_begin, int H, int Wd, int shared) {
    extern __shared__ uint32_t s_forest[];
    const int local = blockIdx.x;
    const int env = env_begin + local;
    const int cells = H * Wd;
    uint32_t* pend = pending + (size_t)local * cells;
    const bool mine = todo[env] != 0;
    if (mine)
        for (int c = threadIdx.x; c < cells; c += blockDim.x) pend[c] = 0u;
    if (shared ? (local != 0) : !mine) return;
    const uint8_t* obst = obst_all + (shared ? 0 : (size_t)env * cells);  // shared: obst_all IS the one map
    uint32_t* out = labels + (shared ? 0 : (size_t)local * cells);
    uint32_t* parent = LDS ? s_forest : out;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int x = wave; x < H; x += nwaves) {
        uint32_t carry = 0u;
        bool carry_open = false;  // the run of the previous chunk's last cell continues into this chunk
        for (int y0 = 0; y0 < Wd; y0 += 64) {
            const int y = y0 + lane;
            const uint32_t c = (uint32_t)(x * Wd + y);
            const bool free_cell = y < Wd && !obst[c];
            const unsigned long long m = __ballot(free_cell);
            const unsigned long long zeros_below = ~m & ((1ull << lane) - 1ull);
            uint32_t start;
            if (zeros_below == 0ull) start = carry_open ? carry : (uint32_t)(x * Wd + y0);
            else start = (uint32_t)(x * Wd + y0 + (64 - __clzll((long long)zeros_below)));
            if (y < Wd) st<LDS>(parent + c, free_cell ? start : NONE);
            carry = (uint32_t)__shfl((int)start, 63, 64);
            carry_open = (m >> 63) & 1ull;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x + Wd; c < cells; c += blockDim.x) {
        if (obst[c] || obst[c - Wd]) continue;
        const int y = c % Wd;
        if (y == 0 || obst[c - 1] || obst[c - Wd - 1]) uf_union<LDS>(parent, (uint32_t)c, (uint32_t)(c - Wd));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cells; c += blockDim.x) {
        if (obst[c]) {
            if (LDS) out[c] = NONE;
        } else {
            const uint32_t root = uf_find<LDS>(parent, (uint32_t)c);
            if (LDS) out[c] = root;
            else st<false>(parent + c, root);  // in-place compression: still a valid ancestor for concurrent finds
        }
    }
}

// One WAVE per environment walks the candidate stream of GEN v2 and closes start/target pairs.
// pending[c]: bit 31 = cell c already taken; low bits (root entries only) = open start cell + 1.
// 64 candidates at a time: every lane hashes one candidate and fetches its obstacle byte, taken mark and label
// (one exposed L2 latency for 64 candidates); the order-dependent part -- "first visit of a component opens a
// pair, the next closes it" -- is then replayed in candidate order with wave-uniform registers.  The pending
// word of the component being visited is cached in registers (the giant component takes ~95 % of the visits at
// density 0.3), so most candidates cost a few scalar-like ALU instructions and no memory round trip.
// All `pending` traffic uses relaxed agent-scope accesses (L2-served), so batches see earlier batches' marks.
__global__ __launch_bounds__(64) void place_kernel(const uint8_t* __restrict__ obst_all, const uint32_t* __restrict__ labels,
                                                   uint32_t* __restrict__ pending, uint8_t* __restrict__ todo,
                                                   const uint32_t* __restrict__ epoch, uint32_t* __restrict__ pos,
                                                   uint32_t* __restrict__ tgt, uint32_t* __restrict__ pos0,
                                                   uint32_t* __restrict__ tgt0, uint8_t* __restrict__ active,
                                                   uint32_t* __restrict__ tcount, int32_t* __restrict__ elapsed,
                                                   int4* __restrict__ macc, uint32_t* __restrict__ fail_count,
                                                   int env_begin, int env_count, int A, int Wd, int cells, int r,
                                                   int shared, uint64_t key_base, uint32_t attempt) {
    const int local = blockIdx.x;
    const int env = env_begin + local;
    if (!todo[env]) return;
    const int lane = threadIdx.x;
    const uint8_t* obst = obst_all + (size_t)env * cells;
    const uint32_t* lab = labels + (shared ? 0 : (size_t)local * cells);
    uint32_t* pend = pending + (size_t)local * cells;
    const uint64_t h = instance_hash(key_base + (uint64_t)env, epoch[env], attempt);
    const uint32_t budget = gen_candidate_budget((uint32_t)cells);
    const uint32_t pad = ((uint32_t)r << 16) | (uint32_t)r;
    int placed = 0;
    uint32_t cur_root = NONE, cur_val = 0u;  // cached pending word (wave-uniform)
    for (uint32_t t0 = 0; t0 < budget && placed < A; t0 += 64) {
        const uint32_t t = t0 + (uint32_t)lane;
        const uint32_t c = (uint32_t)(((sm64(h ^ (GEN_TAG_PLACE | (uint64_t)t)) >> 32) * (uint64_t)cells) >> 32);
        const uint8_t blocked = obst[c];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the previous batch's marks have reached L2
        const uint32_t mark = ld<false>(pend + c);
        const uint32_t root = lab[c];
        const bool ok = t < budget && !blocked && !(mark & TAKEN);
        unsigned long long left = __ballot(ok);
        while (left && placed < A) {
            const int k = __ffsll((long long)left) - 1;
            const uint32_t ck = (uint32_t)__shfl((int)c, k, 64);
            const uint32_t rk = (uint32_t)__shfl((int)root, k, 64);
            left &= ~__ballot(ok && c == ck);  // this candidate and its duplicates later in the batch
            if (rk != cur_root) {              // switch the cached component (rare)
                if (lane == 0 && cur_root != NONE) st<false>(pend + cur_root, cur_val);
                cur_root = rk;
                cur_val = (uint32_t)__shfl((int)(lane == 0 ? ld<false>(pend + rk) : 0u), 0, 64);
            }
            if (ck == rk) {  // the candidate is the root cell itself: its taken bit lives in the cached word
                if (cur_val & TAKEN) continue;  // (the prefetched mark may predate the cached update)
                cur_val |= TAKEN;
            } else if (lane == k) {
                st<false>(pend + ck, mark | TAKEN);
            }
            const uint32_t open = cur_val & ~TAKEN;
            if (open == 0u) {
                cur_val |= ck + 1u;
            } else {
                cur_val &= TAKEN;
                if (lane == 0) {
                    const uint32_t s = open - 1u;
                    const size_t gi = (size_t)env * A + placed;
                    const uint32_t pc = (((s / Wd) << 16) | (s % Wd)) + pad;
                    const uint32_t tc = (((ck / Wd) << 16) | (ck % Wd)) + pad;
                    pos[gi] = pc;
                    pos0[gi] = pc;
                    tgt[gi] = tc;
                    tgt0[gi] = tc;
                    active[gi] = 1;
                    if (tcount) tcount[gi] = 0u;
                }
                ++placed;
            }
        }
    }
    if (lane == 0) {
        if (cur_root != NONE) st<false>(pend + cur_root, cur_val);
        if (placed == A) {
            todo[env] = 0;
            elapsed[env] = 0;
            macc[env] = make_int4(0, 0, 0, 0);
        } else {
            atomicAdd(fail_count, 1u);
        }
    }
}

// Lifelong component tables of one environment per workgroup (stable counting sort of the free cells by
// component): comp_begin/comp_len per cell, comp_cells = unpadded (x << 16) | y grouped by component in
// order of the components' first cells, row-major inside.  `counters` (scratch, one word per cell) holds the
// component sizes and then the fill pointers.  The parallel passes (clear, count, copy) use all four waves;
// the two order-dependent passes (exclusive scan over the roots, stable fill) are walked by wave 0, four
// 64-cell chunks per iteration so that four loads are in flight.  The LARGEST component (the giant one holds
// ~95 % of the free cells at density 0.3) keeps its fill pointer in a register: no L2 round trip per chunk.
__global__ __launch_bounds__(256) void tables_kernel(const uint8_t* __restrict__ obst_all,
                                                     const uint32_t* __restrict__ labels, uint32_t* __restrict__ counters,
                                                     const uint8_t* __restrict__ regen, uint32_t* __restrict__ comp_begin,
                                                     uint32_t* __restrict__ comp_len, uint32_t* __restrict__ comp_cells,
                                                     int env_begin, int Wd, int cells, int shared) {
    const int local = blockIdx.x;
    const int env = env_begin + local;
    if (!regen[env]) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint8_t* obst = obst_all + (size_t)env * cells;
    const uint32_t* lab = labels + (shared ? 0 : (size_t)local * cells);
    uint32_t* cnt = counters + (size_t)local * cells;
    uint32_t* cb = comp_begin + (size_t)env * cells;
    uint32_t* cl = comp_len + (size_t)env * cells;
    uint32_t* cc = comp_cells + (size_t)env * cells;
    __shared__ uint32_t s_big[2];  // root and first slot of the largest component
    for (int c = tid; c < cells; c += 256) st<false>(cnt + c, 0u);
    __syncthreads();
    for (int c = tid; c < cells; c += 256)
        if (!obst[c]) atomicAdd(cnt + lab[c], 1u);
    __syncthreads();
    if (tid < 64) {
        // exclusive scan of the component sizes in order of the roots' cell indices
        uint32_t carry = 0u, best = 0u, best_root = NONE, best_begin = 0u;
        for (int base = 0; base < cells; base += 256) {
            uint32_t v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = base + 64 * j + lane;
                v[j] = (c < cells) ? ld<false>(cnt + c) : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = base + 64 * j + lane;
                uint32_t incl = v[j];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
                    if (lane >= d) incl += up;
                }
                const uint32_t begin = carry + incl - v[j];
                if (c < cells) {
                    cb[c] = v[j] ? begin : 0u;
                    cl[c] = v[j];
                    if (v[j]) st<false>(cnt + c, begin);  // fill pointer of this component
                }
                if (v[j] > best) {
                    best = v[j];
                    best_root = (uint32_t)c;
                    best_begin = begin;
                }
                carry += (uint32_t)__shfl((int)incl, 63, 64);
            }
        }
        // wave arg-max (ties: smallest root) of the per-lane candidates
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t ob = (uint32_t)__shfl_xor((int)best, d, 64);
            const uint32_t orr = (uint32_t)__shfl_xor((int)best_root, d, 64);
            const uint32_t obg = (uint32_t)__shfl_xor((int)best_begin, d, 64);
            if (ob > best || (ob == best && orr < best_root)) {
                best = ob;
                best_root = orr;
                best_begin = obg;
            }
        }
        if (lane == 0) {
            s_big[0] = best_root;
            s_big[1] = best_begin;
        }
    }
    __syncthreads();
    // every free cell copies its component's slice (roots already hold theirs)
    for (int c = tid; c < cells; c += 256) {
        if (obst[c]) continue;
        const uint32_t root = lab[c];
        if (root != (uint32_t)c) {
            cb[c] = cb[root];
            cl[c] = cl[root];
        }
    }
    if (tid >= 64) return;
    // stable fill: 64 consecutive cells at a time, lanes of one component take consecutive slots
    const uint32_t big_root = s_big[0];
    uint32_t big_fill = s_big[1];
    for (int base = 0; base < cells; base += 256) {
        bool free_cell[4];
        uint32_t root[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = base + 64 * j + lane;
            free_cell[j] = (c < cells) && !obst[c];
            root[j] = free_cell[j] ? lab[c] : NONE;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = base + 64 * j + lane;
            const uint32_t packed = ((uint32_t)(c / Wd) << 16) | (uint32_t)(c % Wd);
            const unsigned long long mb = __ballot(free_cell[j] && root[j] == big_root);
            if (free_cell[j] && root[j] == big_root)
                cc[big_fill + (uint32_t)__popcll(mb & ((1ull << lane) - 1ull))] = packed;
            big_fill += (uint32_t)__popcll(mb);
            unsigned long long left = __ballot(free_cell[j]) & ~mb;
            while (left) {
                const int leader = __ffsll((long long)left) - 1;
                const uint32_t lr = (uint32_t)__shfl((int)root[j], leader, 64);
                const unsigned long long m = __ballot(free_cell[j] && root[j] == lr);
                uint32_t slot0 = 0u;
                if (lane == leader) slot0 = atomicAdd(cnt + lr, (uint32_t)__popcll(m));
                slot0 = (uint32_t)__shfl((int)slot0, leader, 64);
                if (free_cell[j] && root[j] == lr)
                    cc[slot0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = packed;
                left &= ~m;
            }
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
    const size_t forest_bytes = (size_t)H * Wd * sizeof(uint32_t);
    if (forest_bytes <= 64 * 1024) {  // forest in LDS
        if (forest_bytes > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ccl_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)forest_bytes);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(ccl_kernel<true>, dim3(env_count), dim3(256), forest_bytes, s, obst, labels, pending, todo,
                           env_begin, H, Wd, shared);
    } else {
        hipLaunchKernelGGL(ccl_kernel<false>, dim3(env_count), dim3(256), 0, s, obst, labels, pending, todo, env_begin,
                           H, Wd, shared);
    }
    return hipGetLastError();
}

hipError_t launch_place(const uint8_t* obst, const uint32_t* labels, uint32_t* pending, uint8_t* todo,
                        const uint32_t* epoch, uint32_t* pos, uint32_t* tgt, uint32_t* pos0, uint32_t* tgt0,
                        uint8_t* active, uint32_t* tcount, int32_t* elapsed, int4* macc, uint32_t* fail_count,
                        int env_begin, int env_count, int A, int Wd, int cells, int r, int shared, uint64_t key_base,
                        uint32_t attempt, hipStream_t s) {
    hipLaunchKernelGGL(place_kernel, dim3(env_count), dim3(64), 0, s, obst, labels, pending, todo, epoch,
                       pos, tgt, pos0, tgt0, active, tcount, elapsed, macc, fail_count, env_begin, env_count, A, Wd,
                       cells, r, shared, key_base, attempt);
    return hipGetLastError();
}

hipError_t launch_tables(const uint8_t* obst, const uint32_t* labels, uint32_t* counters, const uint8_t* regen,
                         uint32_t* comp_begin, uint32_t* comp_len, uint32_t* comp_cells, int env_begin, int env_count,
                         int Wd, int cells, int shared, hipStream_t s) {
    hipLaunchKernelGGL(tables_kernel, dim3(env_count), dim3(256), 0, s, obst, labels, counters, regen, comp_begin,
                       comp_len, comp_cells, env_begin, Wd, cells, shared);
    return hipGetLastError();
}

}  // namespace pgx
