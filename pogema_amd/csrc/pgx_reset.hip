// pgx_reset.hip -- gfx950 kernels of the on-device reset path:
//   instance generator "GEN v2" (Bernoulli obstacles -> connected components -> start/target pairs),
//   replacing upstream pogema/generator.py (`generate_obstacles`, `generate_positions_and_targets_fast`,
//   the BFS component labelling) and the component tables `PogemaLifeLong` draws new targets from.
// The generator is counter-based (normative statement: oracle/generator_oracle.py, test infrastructure),
// so host (pgx_generate) and device (pgx_reset_random) produce identical instances.
//
// Everything here is L2-latency-bound integer work on the reset path, not the step hot path.  ONE kernel
// (reset_env_kernel) builds one environment per 256-thread workgroup, retries included:
//   obstacles    one hash per cell                                                          (streaming)
//   ccl_phase    horizontal runs by ballot, vertical merges by lock-free union-find (LDS forest for maps up to
//                ~128 x 128, L2 atomics beyond): every root ends as the smallest row-major index   (atomics)
//   place_phase  wave 0: 64 candidates fetched at a time, the order-dependent pairing replayed with
//                wave-uniform registers                                                      (latency)
//   tables_phase lifelong only: component sizes, exclusive scan, stable fill with ballot grouping (latency)
//   pack_phase   padded 1-bit-per-cell bitmap with the artificial border
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "pgx_internal.h"

namespace pgx {

namespace {

__device__ __forceinline__ uint64_t sm64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// seed and global env index are separate key components: adjacent seeds share no instance
__device__ __forceinline__ uint64_t instance_hash(uint64_t seed, uint64_t env, uint32_t epoch, uint32_t attempt) {
    return sm64(sm64(sm64(seed) ^ env) ^ (((uint64_t)epoch << 32) | attempt));
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


// ---- phases of the per-environment reset (device functions; all threads of the 256-thread workgroup call them) ----

// Connected components of the FREE cells: labels[c] = smallest row-major index of c's component (NONE on
// obstacles).
//   1. horizontal runs: every free cell points at the first cell of its run (one wave per row, ballot + clz,
//      no atomics);
//   2. vertical merges: only where a run STARTS touching an upper run (the left neighbours are not both free) --
//      ~0.25 unions per cell at density 0.3 instead of ~1 -- lock-free union-find, larger root under smaller;
//   3. every cell resolves its root (in-place compression).
//   LDS = true: the forest lives in LDS (maps up to ~128 x 128), only the final labels go to HBM.
template <bool LDS>
__device__ void ccl_phase(const uint8_t* __restrict__ obst, uint32_t* __restrict__ out, uint32_t* forest, int H, int Wd) {
    const int cells = H * Wd;
    uint32_t* parent = LDS ? forest : out;
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
    __syncthreads();
}

// Wave 0 walks the candidate stream of GEN v2 and closes start/target pairs into `pairs` (LDS: start, target
// row-major cell per agent).  Returns the number of pairs closed (wave-uniform).
// pend[c]: bit 31 = cell c already taken; low bits (root entries only) = open start cell + 1.
// 64 candidates at a time: every lane hashes one candidate and fetches its obstacle byte, taken mark and label
// (one exposed L2 latency for 64 candidates); the order-dependent part -- "first visit of a component opens a
// pair, the next closes it" -- is then replayed in candidate order with wave-uniform registers.  The pending
// word of the component being visited is cached in registers (the giant component takes ~95 % of the visits at
// density 0.3), so most candidates cost a few ALU instructions and no memory round trip.
// All `pend` traffic uses relaxed agent-scope accesses (L2-served), so batches see earlier batches' marks.
__device__ int place_phase(const uint8_t* __restrict__ obst, const uint32_t* __restrict__ lab, uint32_t* __restrict__ pend,
                           uint32_t* pairs, int A, int cells, uint64_t h) {
    const int lane = threadIdx.x;
    const uint32_t budget = gen_candidate_budget((uint32_t)cells);
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
                    pairs[2 * placed] = open - 1u;
                    pairs[2 * placed + 1] = ck;
                }
                ++placed;
            }
        }
    }
    return placed;
}

// Lifelong component tables (stable counting sort of the free cells by component): comp_begin/comp_len per cell,
// comp_cells = unpadded (x << 16) | y grouped by component in order of the components' first cells, row-major
// inside.  `cnt` (scratch, one word per cell) holds the component sizes and then the fill pointers.  The parallel
// passes (clear, count, copy) use all four waves; the two order-dependent passes (exclusive scan over the roots,
// stable fill) are walked by wave 0, four 64-cell chunks per iteration so that four loads are in flight.  The
// LARGEST component (the giant one holds ~95 % of the free cells at density 0.3) keeps its fill pointer in a
// register: no L2 round trip per chunk.
__device__ void tables_phase(const uint8_t* __restrict__ obst, const uint32_t* __restrict__ lab, uint32_t* __restrict__ cnt,
                             uint32_t* __restrict__ cb, uint32_t* __restrict__ cl, uint32_t* __restrict__ cc, int Wd,
                             int cells, uint32_t* s_big) {
    const int tid = threadIdx.x, lane = tid & 63;
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

// u8 map [H, W] -> this env's padded 1-bit-per-cell bitmap with the artificial border of SURVEY A1
// (same layout as pack_obstacles_kernel in pgx_kernels.hip).
__device__ void pack_phase(const uint8_t* __restrict__ obst, uint32_t* __restrict__ bm, int H, int Wd, int r, int wpr, int bmw,
                           const OutsideParams& outside, int env) {
    const int PH = H + 2 * r, PW = Wd + 2 * r;
    uint64_t h_out = 0;
    if (outside.enabled)
        h_out = gen_outside_hash(outside.seed, (uint64_t)(outside.env_index_base + env), outside.epoch[env]);
    for (int w = threadIdx.x; w < bmw; w += blockDim.x) {
        const int x = w / wpr;
        const int y0 = (w - x * wpr) * 32;
        uint32_t bits = 0u;
        const bool ring_row = (x == r - 1) || (x == PH - r);
        const bool in_rows = (x >= r) && (x < PH - r);
        for (int b = 0; b < 32; ++b) {
            const int y = y0 + b;
            if (y >= PW) break;
            uint32_t v = 0u;
            const bool span = (y >= r - 1) && (y <= PW - r);
            if (ring_row && span) v = 1u;
            else if (in_rows && (y == r - 1 || y == PW - r)) v = 1u;
            else if (in_rows && y >= r && y < PW - r) v = obst[(x - r) * Wd + (y - r)] != 0 ? 1u : 0u;
            else if (outside.enabled && gen_is_outside(x, y, PH, PW, r)) v = gen_outside_bit(h_out, x, y, PW, outside.thr);
            bits |= v << b;
        }
        bm[w] = bits;
    }
}

}  // namespace

// todo/regen flags and generation counters for one reset call
__global__ void reset_begin_kernel(const uint8_t* __restrict__ mask, uint8_t* __restrict__ todo,
                                   uint8_t* __restrict__ regen, uint32_t* __restrict__ epoch, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const uint8_t m = mask ? (mask[b] != 0 ? 1 : 0) : 1;
    todo[b] = m;
    regen[b] = m;
    epoch[b] = mask ? epoch[b] + m : 0u;
}

// ------------------------------------------------------------------------------------------------
// The reset kernel: one 256-thread workgroup builds one environment's instance from nothing.
//   for attempt = 0 .. max_retries-1:   obstacles (hash per cell, or the given map) -> components -> placement
//   on success: commit map, agents, targets, step counter, metric accumulators; pack the padded bitmap;
//               lifelong: component tables.   on failure: the environment is left untouched and counted.
// Everything an attempt produces lives in per-slot scratch (map bytes, labels, pending) and LDS (the pairs), so
// a failed environment keeps its previous instance -- which is what lets pgx_regenerate run without a host sync.
//   given_state != 0 (pgx_reset_from_state, lifelong): the map is already installed; only components + tables.
// ------------------------------------------------------------------------------------------------
template <bool LDS>
__global__ __launch_bounds__(256) void reset_env_kernel(const ResetParams p) {
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint32_t s_flag[4];
    const int local = blockIdx.x;
    const int env = p.env_begin + local;
    if (!p.todo[env]) return;
    const int tid = threadIdx.x;
    const int cells = p.H * p.Wd;
    uint32_t* forest = s_dyn;                               // [cells] when LDS
    uint32_t* pairs = s_dyn + (LDS ? cells : 0);            // [2 * A]
    uint32_t* lab = p.labels + (size_t)local * cells;
    uint32_t* pend = p.pending + (size_t)local * cells;
    uint8_t* env_map = p.map_u8 + (size_t)env * cells;
    const uint8_t* obst = env_map;
    if (!p.given_state) {
        uint8_t* draft = p.scratch_map + (size_t)local * cells;
        obst = draft;
        const uint32_t epoch = p.epoch[env];
        bool ok = false;
        for (int attempt = 0; attempt < p.max_retries && !ok; ++attempt) {
            const uint64_t h = instance_hash(p.gen_seed, (uint64_t)(p.env_index_base + env), epoch, (uint32_t)attempt);
            const bool redraw = !p.shared_map || attempt == 0;
            for (int c = tid; c < cells; c += 256) {
                if (redraw)
                    draft[c] = p.shared_map ? (p.shared_map[c] != 0 ? 1 : 0)
                                            : ((sm64(h ^ (GEN_TAG_OBST | (uint64_t)c)) >> 40) < p.thr ? 1 : 0);
                st<false>(pend + c, 0u);
            }
            __syncthreads();
            if (redraw) ccl_phase<LDS>(draft, lab, forest, p.H, p.Wd);
            if (tid < 64) {
                const int placed = place_phase(draft, lab, pend, pairs, p.A, cells, h);
                if (tid == 0) s_flag[0] = placed == p.A ? 1u : 0u;
            }
            __syncthreads();
            ok = s_flag[0] != 0u;
            __syncthreads();
        }
        if (!ok) {
            if (tid == 0) atomicAdd(p.fail_count, 1u);
            return;
        }
        // ---- commit -------------------------------------------------------------------------------
        for (int c = tid; c < cells; c += 256) env_map[c] = draft[c];
        const uint32_t pad = ((uint32_t)p.r << 16) | (uint32_t)p.r;
        for (int i = tid; i < p.A; i += 256) {
            const uint32_t s = pairs[2 * i], t = pairs[2 * i + 1];
            const size_t gi = (size_t)env * p.A + i;
            const uint32_t pc = (((s / p.Wd) << 16) | (s % p.Wd)) + pad;
            const uint32_t tc = (((t / p.Wd) << 16) | (t % p.Wd)) + pad;
            p.pos[gi] = pc;
            p.pos0[gi] = pc;
            p.tgt[gi] = tc;
            p.tgt0[gi] = tc;
            p.active[gi] = 1;
            if (p.tcount) p.tcount[gi] = 0u;
            if (p.np_state) p.np_state[gi] = p.np_state0[gi];
        }
        if (tid == 0) {
            p.todo[env] = 0;
            p.elapsed[env] = 0;
            p.macc[env] = make_int4(0, 0, 0, 0);
        }
        pack_phase(draft, p.obst_bm + (size_t)env * p.bmw, p.H, p.Wd, p.r, p.wpr, p.bmw, p.outside, env);
    } else {
        ccl_phase<LDS>(obst, lab, forest, p.H, p.Wd);
    }
    if (p.lifelong)
        tables_phase(obst, lab, pend, p.comp_begin + (size_t)env * cells, p.comp_len + (size_t)env * cells,
                     p.comp_cells + (size_t)env * cells, p.Wd, cells, s_flag + 2);
}

// ---- launchers --------------------------------------------------------------------------------------
hipError_t launch_reset_begin(const uint8_t* mask, uint8_t* todo, uint8_t* regen, uint32_t* epoch, int batch,
                              hipStream_t s) {
    hipLaunchKernelGGL(reset_begin_kernel, dim3((batch + 255) / 256), dim3(256), 0, s, mask, todo, regen, epoch, batch);
    return hipGetLastError();
}

bool reset_forest_in_lds(int H, int Wd, int A) { return (size_t)H * Wd * 4 + (size_t)A * 8 <= 64 * 1024; }

hipError_t launch_reset_env(const ResetParams& p, hipStream_t s) {
    const bool lds = reset_forest_in_lds(p.H, p.Wd, p.A);
    const size_t dyn = (lds ? (size_t)p.H * p.Wd * 4 : 0) + (size_t)p.A * 8;
    if (lds) {
        if (dyn > 48 * 1024) {  // opt in once per process (and per device) for the 64 KiB maximum this kernel uses
            static std::atomic<uint64_t> opted_in{0};
            int dev = 0;
            (void)hipGetDevice(&dev);
            const uint64_t bit = 1ull << (dev & 63);
            if (!(opted_in.load(std::memory_order_relaxed) & bit)) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&reset_env_kernel<true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
                if (e != hipSuccess) return e;
                opted_in.fetch_or(bit, std::memory_order_relaxed);
            }
        }
        hipLaunchKernelGGL(reset_env_kernel<true>, dim3(p.env_count), dim3(256), dyn, s, p);
    } else {
        hipLaunchKernelGGL(reset_env_kernel<false>, dim3(p.env_count), dim3(256), dyn, s, p);
    }
    return hipGetLastError();
}

}  // namespace pgx
