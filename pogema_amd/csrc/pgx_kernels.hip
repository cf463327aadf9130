// pgx_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the vectorized POGEMA step engine.
//
// Hot path replaced (upstream names, see include/pogema_amd.h and SURVEY.md section 8a):
//   Pogema.step / move_agents / Grid.move / _obs() / MultiTimeLimit.step           rows A2..A13
//
// Design (DESIGN.md has the full story):
//   * one lane = one agent.  num_agents <= 64: a workgroup is ONE wave holding up to 64/G environments (G = next
//     power of two of num_agents); num_agents > 64: ceil(A/64) symmetric waves per environment, partners of other
//     waves and the transitive closure go through LDS.
//   * collision resolution is register-resident: an all-pairs DPP sweep (index-carrying words, 7 instructions per
//     partner) yields "who stands on my destination" and "which lower index claims it too"; the three collision
//     systems of the reference are closed forms of those plus a pointer-doubling closure.  No global atomics, no
//     per-cell tables in HBM.
//   * the padded obstacle bitmap (1 bit per cell) is staged HBM -> LDS once per step, the occupancy bitmap is
//     rebuilt in LDS from the agents' cells with LDS atomics, every (agent, channel, window-row) is reduced to one
//     row mask (16-bit, aliased over the dead state, when the window side is <= 16), and the observation tensor is
//     then produced as a flat, fully coalesced stream of 16-byte stores (float32: 12*(2r+1)^2 bytes per
//     agent-step, the only large HBM stream of the kernel; uint8 optional).
//   * integer indexing only -- no MFMA on purpose; the bound is HBM write bandwidth.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <array>
#include <type_traits>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <utility>

#include "pgx_internal.h"
#include "pgx_nprng.h"

namespace pgx {

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// docs/SPEC.md S5, lifelong stream: uniform index in [0, n) for (seed, global env, agent, counter).
__device__ __forceinline__ uint32_t lifelong_draw(uint64_t seed, uint64_t env_index, uint32_t agent,
                                                  uint32_t counter, uint32_t n) {
    uint64_t h = splitmix64(seed);
    h = splitmix64(h ^ env_index);
    h = splitmix64(h ^ (((uint64_t)agent << 32) | counter));
    return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32);
}

// pgx_rollout without an action tensor: the uniform random policy, action in 0..4 for (seed, global env, agent, step).
// Stated in oracle/generator_oracle.py (policy_action); tests/test_rollout_gpu.py.
__device__ __forceinline__ int policy_action(uint64_t seed, uint64_t env_index, uint32_t agent, uint64_t step) {
    uint64_t h = splitmix64(seed ^ 0x504F4C4943590000ull);  // 'POLICY'
    h = splitmix64(h ^ env_index);
    h = splitmix64(h ^ (step << 20 | (uint64_t)agent));
    return (int)(((h >> 32) * 5ull) >> 32);
}

__device__ __forceinline__ pgxnp::Pcg64 np_unpack(const NpGen& s) {
    pgxnp::Pcg64 g;
    g.state = ((pgxnp::u128)s.w[0] << 64) | s.w[1];
    g.inc = ((pgxnp::u128)s.w[2] << 64) | s.w[3];
    g.has_uint32 = (uint32_t)(s.w[4] >> 32);
    g.uinteger = (uint32_t)s.w[4];
    return g;
}
__host__ __device__ inline NpGen np_pack(const pgxnp::Pcg64& g) {
    NpGen s;
    s.w[0] = (uint64_t)(g.state >> 64); s.w[1] = (uint64_t)g.state;
    s.w[2] = (uint64_t)(g.inc >> 64); s.w[3] = (uint64_t)g.inc;
    s.w[4] = ((uint64_t)g.has_uint32 << 32) | g.uinteger;
    return s;
}

// packed cell in HBM/LDS: (x << 16) | y in PADDED coordinates.
__device__ __forceinline__ uint32_t bm_test(const uint32_t* bm, int wpr, uint32_t cell) {
    const uint32_t x = cell >> 16, y = cell & 0xFFFFu;
    return (bm[x * wpr + (y >> 5)] >> (y & 31)) & 1u;
}

// ---- 22-bit cell keys -----------------------------------------------------------------------------
// HBM/LDS hold cells as (x << 16) | y.  For the collision sweep they are re-packed to (x << 11) | y
// (x, y < 2048) so that ((a ^ b) << 10) | index fits one 32-bit word: a single v_min_u32 then keeps
// "the smallest index among exact matches" without any compare/select or scalar instruction.
constexpr uint32_t NOCELL_A = 0x3FFFFFu;  // "stands nowhere"   (hidden / invalid lane)
constexpr uint32_t NOCELL_B = 0x3FFFFEu;  // "claims nothing"
constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t to_c22(uint32_t packed) { return ((packed >> 16) << 11) | (packed & 0x7FFu); }
__device__ __forceinline__ uint32_t from_c22(uint32_t c) { return ((c >> 11) << 16) | (c & 0x7FFu); }
__device__ __forceinline__ uint32_t move_c22(uint32_t c, int a) {
    const uint32_t delta = (a == 1) ? (0u - 2048u) : (a == 2) ? 2048u : (a == 3) ? 0xFFFFFFFFu : (a == 4) ? 1u : 0u;
    return c + delta;
}

// Rotate a value by one lane inside the environment group (lane i receives lane i+1's value, the last
// lane of the group receives the first).  DPP where the hardware has a matching pattern, ds_bpermute
// otherwise; applied cumulatively it walks every partner of the group in G-1 steps.
template <int G>
__device__ __forceinline__ uint32_t rot1(uint32_t v, int src_lane) {
    if constexpr (G == 64) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x134 /* wave_rol:1 */, 0xF, 0xF, true);
    } else if constexpr (G == 16) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x12F /* row_ror:15 */, 0xF, 0xF, true);
    } else if constexpr (G == 4) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x39 /* quad_perm:[1,2,3,0] */, 0xF, 0xF, true);
    } else if constexpr (G == 2) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true);
    } else {
        return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
    }
}

// "does any lane of my environment group satisfy pred"
template <int G>
__device__ __forceinline__ bool group_any(bool pred, int gbase) {
    const unsigned long long m = __ballot(pred);
    if constexpr (G == 64) {
        return m != 0ull;
    } else {
        constexpr unsigned long long gm = (1ull << G) - 1ull;
        return ((m >> gbase) & gm) != 0ull;
    }
}

// All-pairs sweep, branch-free and VALU-only.  Every agent j publishes two words that carry its index in the low
// 10 bits, P_j = (vis_j << 10) | j and Q_j = (want_j << 10) | j; the lane of agent i holds WSH = want_i << 10.
// For every OTHER agent j of the environment (values arrive by DPP rotation):
//   x = P_j ^ WSH  -> x < 1024 <=> j stands on want_i, and then x == j               okey = min x
//   y = Q_j ^ WSH  -> y < 1024 <=> j claims want_i too, and then y == j               cany = min y
//   z = (i-1) - y  -> z < i    <=> that claimant has a LOWER index; the smallest z is the largest such index
//                     (unsigned wrap-around keeps every other y at z >= i)              zmin = min z
// 2 DPP moves + 2 xor + 1 sub + 3 min per partner (the first version recomputed j and the distance per step: 14).
struct SweepAcc {
    uint32_t okey, cany, zmin;
};
__device__ __forceinline__ void pair_min(uint32_t pj, uint32_t qj, uint32_t wsh, uint32_t im1, SweepAcc& a) {
    const uint32_t y = qj ^ wsh;
    a.okey = min(a.okey, pj ^ wsh);
    a.cany = min(a.cany, y);
    a.zmin = min(a.zmin, im1 - y);
}

// One slot of G partners whose (P, Q) words sit in the lanes of this wave's group: G-1 DPP rotations (plus the
// unrotated words when the slot belongs to another wave, `with_k0`).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}

template <int G>
__device__ __forceinline__ void sweep_slot(uint32_t pj, uint32_t qj, bool with_k0, int src, uint32_t wsh, uint32_t im1,
                                           SweepAcc& a) {
    if (with_k0) pair_min(pj, qj, wsh, im1, a);
    if constexpr (G == 8) {
        // No 8-lane rotation exists in DPP, but the seven partners of lane i are exactly i ^ 1 .. i ^ 7 and the order of
        // the visits does not matter (three running minima): i^1, i^2, i^3 are quad permutations, i^7 is
        // row_half_mirror, i^6 / i^5 / i^4 are quad permutations of the mirrored words.  14 independent DPP moves
        // instead of a dependent chain of 14 ds_bpermute round trips (configs[1]: -0.5 us on the step's critical path).
        constexpr int X1 = 0xB1, X2 = 0x4E, X3 = 0x1B, HM = 0x141;  // quad_perm [1,0,3,2] / [2,3,0,1] / [3,2,1,0], row_half_mirror
        const uint32_t ph = dpp<HM>(pj), qh = dpp<HM>(qj);
        pair_min(dpp<X1>(pj), dpp<X1>(qj), wsh, im1, a);
        pair_min(dpp<X2>(pj), dpp<X2>(qj), wsh, im1, a);
        pair_min(dpp<X3>(pj), dpp<X3>(qj), wsh, im1, a);
        pair_min(ph, qh, wsh, im1, a);
        pair_min(dpp<X1>(ph), dpp<X1>(qh), wsh, im1, a);
        pair_min(dpp<X2>(ph), dpp<X2>(qh), wsh, im1, a);
        pair_min(dpp<X3>(ph), dpp<X3>(qh), wsh, im1, a);
        return;
    }
    for (int k = 1; k < G; ++k) {
        pj = rot1<G>(pj, src);
        qj = rot1<G>(qj, src);
        pair_min(pj, qj, wsh, im1, a);
    }
}

// Workgroup synchronisation that does NOT drain the vector-memory queue.  `__syncthreads()` makes
// hipcc emit s_waitcnt vmcnt(0) first, i.e. the wave would sit on the acknowledgements of its own
// state/flag stores -- microseconds under a saturated HBM write stream.  Only LDS traffic has to be
// ordered here.  Single-wave blocks: LDS operations of a wave execute in order, so a compiler-level
// fence is enough (no s_barrier at all).
template <bool MW>
__device__ __forceinline__ void lds_sync() {
    if constexpr (!MW) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}


// 16-byte observation store (StepParams::store_policy).  0: plain (the line stays in the XCD's L2), 1: nontemporal,
// 2: sc1 (write-through: the line leaves the L2 at once, so the stream neither evicts the small per-step state from the
// L2s nor leaves megabytes of dirty lines to be flushed at the kernel boundary).  Measured, same buffers (ab_inproc):
// sc1 vs plain -1.3 % configs[2], -3.6 % configs[3], -10 % configs[1]; the multi-wave kernel (configs[4]) prefers
// nontemporal (-1 %), sc1 costs it +0.5 %.  step_geometry() picks; PGX_STORE overrides.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_obs16(f32x4_t* ptr, f32x4_t v, uint32_t policy) {
    if (policy == 0u) *ptr = v;
    else if (policy == 1u) __builtin_nontemporal_store(v, ptr);
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
}

// uint8 observation stream (obs_dtype = PGX_OBS_U8; NOT the drop-in dtype -- a 4x lighter mode for callers that
// cast on their side): the workgroup's n cells as one byte each, 16 per lane per store.  `row_bits(row)` returns
// the W-bit mask of flat window row `row` (rows past the end read as 0).  16 consecutive cells starting inside row
// `row` at column `col` are funnelled out of ceil(16/W)+1 row masks and spread 4 bits -> 4 bytes with one multiply
// ((x * 0x204081) & 0x01010101: bit i lands on bit 8i, no two partial products share a bit).
template <typename RowBits>
__device__ __forceinline__ void stream_obs_u8(uint8_t* out, size_t base, int n, int W, uint32_t magic, int tid, int NT,
                                              bool nontemporal, RowBits row_bits, int wave = 0, int nw = 1) {
    uint8_t* o = out + base;
    const int head = min(n, (int)((16 - (base & 15)) & 15));
    const int nvec = (n - head) >> 4;
    const int tail0 = head + (nvec << 4);
    if (tid < 32) {  // unaligned head / tail bytes
        const int e = (tid < 16) ? tid : tail0 + (tid - 16);
        const bool mine = (tid < 16) ? (tid < head) : (e < n);
        if (mine) {
            const int row = (int)__umulhi((uint32_t)e, magic);
            o[e] = (uint8_t)((row_bits(row) >> (e - row * W)) & 1u);
        }
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4* out4 = reinterpret_cast<u32x4*>(o + head);
    // nw > 1: every wave streams its own contiguous part (see wave_span)
    const int part = (nvec + nw - 1) / nw;
    const int q0 = nw > 1 ? wave * part + (tid & 63) : tid, q1 = nw > 1 ? min(nvec, (wave + 1) * part) : nvec;
    const int qs = nw > 1 ? 64 : NT;
    auto loop = [&](auto nt_tag) {  // the store flavour as a compile-time constant (no branch inside the loop)
        for (int q = q0; q < q1; q += qs) {
            const int e0 = head + (q << 4);
            int row = (int)__umulhi((uint32_t)e0, magic);
            const int col = e0 - row * W;
            uint32_t m = row_bits(row) >> col;
            for (int have = W - col; have < 16; have += W) m |= row_bits(++row) << have;
            u32x4 v;
            v.x = ((m & 15u) * 0x204081u) & 0x01010101u;
            v.y = (((m >> 4) & 15u) * 0x204081u) & 0x01010101u;
            v.z = (((m >> 8) & 15u) * 0x204081u) & 0x01010101u;
            v.w = (((m >> 12) & 15u) * 0x204081u) & 0x01010101u;
            if constexpr (decltype(nt_tag)::value) __builtin_nontemporal_store(v, &out4[q]);
            else out4[q] = v;
        }
    };
    if (nontemporal) loop(std::true_type{});
    else loop(std::false_type{});
}

// 16-bit float observation stream (obs_dtype = PGX_OBS_BF16 / PGX_OBS_F16; NOT the drop-in dtype): the same 0/1 planes as
// bfloat16 or float16, what a mixed-precision policy network consumes directly -- half the HBM bytes of float32 with no
// cast kernel on the consumer's side (uint8 + a cast to bf16 moves as many bytes as float32 did).  `one` is the bit
// pattern of 1.0 (0x3F80 / 0x3C00).  8 cells per lane per 16-byte store, funnelled out of ceil(8/W)+1 row masks; a pair
// of cells becomes one 32-bit word with ONE multiply: bits 0 and 16 times `one` (no carry: one < 2^16).
template <typename RowBits>
__device__ __forceinline__ void stream_obs_h16(uint16_t* out, size_t base, int n, int W, uint32_t magic, int tid, int NT,
                                               bool nontemporal, uint32_t one, RowBits row_bits, int wave = 0, int nw = 1) {
    uint16_t* o = out + base;
    const int head = min(n, (int)((8 - (base & 7)) & 7));
    const int nvec = (n - head) >> 3;
    const int tail0 = head + (nvec << 3);
    if (tid < 16) {  // unaligned head / tail cells
        const int e = (tid < 8) ? tid : tail0 + (tid - 8);
        const bool mine = (tid < 8) ? (tid < head) : (e < n);
        if (mine) {
            const int row = (int)__umulhi((uint32_t)e, magic);
            o[e] = (uint16_t)(((row_bits(row) >> (e - row * W)) & 1u) * one);
        }
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4* out4 = reinterpret_cast<u32x4*>(o + head);
    const int part = (nvec + nw - 1) / nw;  // nw > 1: every wave streams its own contiguous part (as stream_obs_u8)
    const int q0 = nw > 1 ? wave * part + (tid & 63) : tid, q1 = nw > 1 ? min(nvec, (wave + 1) * part) : nvec;
    const int qs = nw > 1 ? 64 : NT;
    auto loop = [&](auto nt_tag) {
        for (int q = q0; q < q1; q += qs) {
            const int e0 = head + (q << 3);
            int row = (int)__umulhi((uint32_t)e0, magic);
            const int col = e0 - row * W;
            uint32_t m = row_bits(row) >> col;
            for (int have = W - col; have < 8; have += W) m |= row_bits(++row) << have;
            u32x4 v;
            v.x = ((m & 1u) | ((m & 2u) << 15)) * one;
            v.y = (((m >> 2) & 1u) | (((m >> 2) & 2u) << 15)) * one;
            v.z = (((m >> 4) & 1u) | (((m >> 4) & 2u) << 15)) * one;
            v.w = (((m >> 6) & 1u) | (((m >> 6) & 2u) << 15)) * one;
            if constexpr (decltype(nt_tag)::value) __builtin_nontemporal_store(v, &out4[q]);
            else out4[q] = v;
        }
    };
    if (nontemporal) loop(std::true_type{});
    else loop(std::false_type{});
}

// ---- P16 float32 stream (window side <= 16): the slice's (agent, channel, window row) masks sit in LDS as packed u16
// (`rows16`, two per word, four zero halfwords behind the last one); a float4 at flat offset e needs the bits of rows
// e / W and e / W + 1 (one ds_read2_b32).  `stream_rows16_span` writes the float4s q0, q0 + qs, ... < q1 (index 0 =
// float offset `head`), keeping (row, col) incrementally; `stream_rows16_edges` the <= 3 + 3 floats in front of / behind
// the 16-byte aligned part (lanes 0..7 of one wave).
// store_obs16's policy is a compile-time constant here: with the switch inside the loop every 16-byte store dragged ~25
// scalar instructions and branches along -- configs[1] 8.3 -> 7.7 us, configs[2] 116.3 -> 115.2, configs[3] 38.75 -> 38.5,
// configs[4] 421.3 -> 419.0 (in-process A/B on shared buffers, profiles/r3/stream_loop_ab.txt)
template <int POLICY, bool PIPE>
__device__ __forceinline__ void stream_rows16_loop(f32x4_t* out4, const uint32_t* rows32, int W, int row, int col, int q0, int q1,
                                                   int qs, int last_word) {
    const int drow = (4 * qs) / W, dcol = 4 * qs - drow * W;
    if constexpr (PIPE) {
        // software pipeline: the row words of iteration k+1 are read from LDS while iteration k is converted and stored
        // (the compiler does not do it: one exposed LDS round trip per 1-KiB store otherwise; configs[2] 113.8 -> 112.9 us,
        // profiles/r3/stream_loop_ab.txt); the read past the last iteration is clamped to the zero pad behind the rows
        const int fw = min(row >> 1, last_word);  // (a lane with q0 >= q1 starts beyond the rows: clamped like the in-loop read)
        uint32_t w0 = rows32[fw], w1 = rows32[fw + 1];
        for (int q = q0; q < q1; q += qs) {
            int ncol = col + dcol, nrow = row + drow;
            if (ncol >= W) {
                ncol -= W;
                nrow += 1;
            }
            const int nw = min(nrow >> 1, last_word);
            const uint32_t n0 = rows32[nw], n1 = rows32[nw + 1];
            const uint32_t pair = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (16 * (row & 1)));  // row | row+1 << 16
            const uint32_t b = ((pair & 0xFFFFu) >> col) | ((pair >> 16) << (W - col));
            f32x4_t v;
            v.x = (float)(b & 1u);
            v.y = (float)((b >> 1) & 1u);
            v.z = (float)((b >> 2) & 1u);
            v.w = (float)((b >> 3) & 1u);
            store_obs16(&out4[q], v, (uint32_t)POLICY);
            w0 = n0;
            w1 = n1;
            row = nrow;
            col = ncol;
        }
    } else {  // the rollout kernels sit at their register limit: four live values fewer
        for (int q = q0; q < q1; q += qs) {
            const uint32_t w0 = rows32[row >> 1], w1 = rows32[(row >> 1) + 1];
            const uint32_t pair = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (16 * (row & 1)));
            const uint32_t b = ((pair & 0xFFFFu) >> col) | ((pair >> 16) << (W - col));
            f32x4_t v;
            v.x = (float)(b & 1u);
            v.y = (float)((b >> 1) & 1u);
            v.z = (float)((b >> 2) & 1u);
            v.w = (float)((b >> 3) & 1u);
            store_obs16(&out4[q], v, (uint32_t)POLICY);
            col += dcol;
            row += drow;
            if (col >= W) {
                col -= W;
                row += 1;
            }
        }
    }
}
// The same walk for the 16-bit float formats (8 cells = 16 bytes per lane-store; window side >= 7, so that 8 consecutive
// cells never span more than two rows): (row, col) kept incrementally, the row pair funnelled out of one ds_read2_b32.
template <int POLICY>
__device__ __forceinline__ void stream_rows16_loop_h16(f32x4_t* out4, const uint32_t* rows32, int W, int row, int col, int q0,
                                                       int q1, int qs, uint32_t one) {
    const int drow = (8 * qs) / W, dcol = 8 * qs - drow * W;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (int q = q0; q < q1; q += qs) {
        const uint32_t w0 = rows32[row >> 1], w1 = rows32[(row >> 1) + 1];
        const uint32_t pair = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (16 * (row & 1)));
        const uint32_t m = ((pair & 0xFFFFu) >> col) | ((pair >> 16) << (W - col));
        u32x4 v;
        v.x = ((m & 1u) | ((m & 2u) << 15)) * one;
        v.y = (((m >> 2) & 1u) | (((m >> 2) & 2u) << 15)) * one;
        v.z = (((m >> 4) & 1u) | (((m >> 4) & 2u) << 15)) * one;
        v.w = (((m >> 6) & 1u) | (((m >> 6) & 2u) << 15)) * one;
        store_obs16(&out4[q], __builtin_bit_cast(f32x4_t, v), (uint32_t)POLICY);
        col += dcol;
        row += drow;
        if (col >= W) {
            col -= W;
            row += 1;
        }
    }
}

template <bool PIPE>
__device__ __forceinline__ void stream_rows16_span(f32x4_t* out4, const uint32_t* rows32, int head, int W, uint32_t magic,
                                                   uint32_t spol, int q0, int q1, int qs, int last_word) {
    const int e0 = head + (q0 << 2);
    const int row = (int)__umulhi((uint32_t)e0, magic);
    const int col = e0 - row * W;
    if (spol == 0u) stream_rows16_loop<0, PIPE>(out4, rows32, W, row, col, q0, q1, qs, last_word);
    else if (spol == 1u) stream_rows16_loop<1, PIPE>(out4, rows32, W, row, col, q0, q1, qs, last_word);
    else stream_rows16_loop<2, PIPE>(out4, rows32, W, row, col, q0, q1, qs, last_word);
}
__device__ __forceinline__ void stream_rows16_edges(float* out, const uint16_t* rows16, int n, int head, int tail0, int W,
                                                    uint32_t magic, int t8) {
    if (t8 < 8) {
        const int e = (t8 < 4) ? t8 : tail0 + (t8 - 4);
        const bool mine = (t8 < 4) ? (t8 < head) : (e < n);
        if (mine) {
            const int row = (int)__umulhi((uint32_t)e, magic);
            const int col = e - row * W;
            out[e] = (float)((rows16[row] >> col) & 1u);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The step kernel.  One lane = one agent, always.
//   MW == false : block = 1 wave holding p.epw (<= 64/G) environments of G lanes each (G = next
//                 power of two >= num_agents <= 64).  Cross-agent traffic is DPP / ds_bpermute only.
//   MW == true  : num_agents > 64: block = ceil(A/64) waves, one environment per block, wave w owns
//                 agents [64w, 64w+64).  Every wave resolves its own agents against all A partners
//                 (partners of other waves come from LDS, then the same DPP rotation), the
//                 transitive closure and the env-wide reductions go through LDS.
//                 The same kernel also runs environments of <= 64 agents with HELPER waves (step_geometry(): small
//                 launches of large environments; three waves for large launches of 64-agent environments): wave 0
//                 holds the agents and runs the single-wave state phase, the other waves only share the row masks
//                 and the observation write.
//   P16         : window side <= 16: row masks are packed to 16 bits, staged through registers and
//                 written OVER the (by then dead) bitmaps and exchange arrays, so the LDS footprint is
//                 max(state, rows) and many waves stay resident per CU (residency).
// ------------------------------------------------------------------------------------------------
enum { MISC_FLAGS = 0, MISC_ARRIVED = 16, MISC_UNSOLVED = 17, MISC_OFFGOAL = 18, MISC_WORDS = 32 };

// rollout_kernel (round 6): what a lane keeps in REGISTERS from one iteration of the on-device loop to the next -- the
// agent's cell, target and is_active, its environment's time-limit counter and metric accumulators (leader lane), and
// the caller's actions of eight steps at a time (4 bits each, 15 = out of range; `anext` is the block behind `ablk`,
// fetched one block ahead).  HBM sees the agent / env state once before the first and once after the last step.
struct Carry {
    uint32_t pos = 0u, tgt = 1u;
    uint32_t ablk = 0u, anext = 0u;
    int elapsed = 0;
    int4 macc = {0, 0, 0, 0};
    bool active = false;
};
__device__ __forceinline__ uint32_t pack_action4(int a) { return (uint32_t)a <= 4u ? (uint32_t)a : 15u; }

// P / R: the argument blocks, either plain (`const StepParams`, one step per launch) or read through a laundered
// kernarg pointer (rollout_kernel).  ROLL: step `t` of a pgx_rollout launch -- the per-step I/O tensors are slices t of
// the caller's [K, ...] buffers (observations: ring slot `slot`), addressed where they are used so that nothing but `t`
// and `slot` lives across the loop.
//   BIG         : large maps (round 6, VERDICT r5 missing #4): two whole padded bitmaps of one environment no longer fit a CU's
//                 160 KB of LDS beyond ~800 x 800 cells.  Only the OCCUPANCY bitmap is kept in LDS (1054 x 1054 bits =
//                 136 KB fit alone); the obstacle bitmap is not staged at all -- the `blocked` bit of a move and the
//                 obstacle rows of the observation windows are read from the HBM bitmap through the L2 (0.8 % of the
//                 step's traffic).  Always the multi-wave form (one environment per workgroup).
//   PC          : rollout_kernel only, small single-wave environments (round 6): a PAIR of waves per environment group.
//                 Wave 0 -- the resolver -- runs the state phase of step t (registers, DPP, its own LDS slice) and publishes
//                 the agents' cells and the occupancy bitmap into one of two LDS buffers; wave 1 -- the streamer -- turns
//                 the buffer of step t - 1 into row masks and streams that step's observations meanwhile.  One barrier
//                 per iteration, K + 1 iterations.  A lone wave spends 1.6 us resolving and 2.4 us on row masks + stream
//                 per configs[1] step one after the other (profiles/r6/rollout_timeline_after.txt); the pair overlaps
//                 them, and the streamer's bursts follow each other without the resolver's pause in between
//                 (tools/drift_probe2.hip: what an HBM-sized configs[3] ring loses).
template <int G, bool MW, bool P16, bool ROLL, bool BIG, bool PC, typename P, typename R>
__device__ __forceinline__ void step_body(P& p, R& rp, const int t, const int slot, Carry& c) {
    static_assert(!MW || G == 64, "multi-wave environments use full waves");
    static_assert(!BIG || MW, "the large-map layout runs one environment per workgroup");
    static_assert(!PC || (ROLL && !MW && P16), "the resolver / streamer pair exists for single-wave rollouts with packed rows");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

    const int tid = PC ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;
    [[maybe_unused]] const int role = PC ? (int)(threadIdx.x >> 6) : 0;  // PC: 0 resolves step t, 1 streams step t - 1
    bool do_resolve = true;
    [[maybe_unused]] bool do_stream = true;
    if constexpr (PC) {
        do_resolve = role == 0 && t < rp.steps;
        do_stream = role == 1 && t > 0;
        if (!do_resolve && !do_stream) return;
    }
    const int NT = MW ? (int)blockDim.x : 64;
    const int lane = tid & 63;
    const int wave = MW ? (tid >> 6) : 0;
    const int nw = NT >> 6;
    const int epw = MW ? 1 : p.epw;
    // Workgroup -> environment slice.  Hardware places workgroup b on XCD b % 8; the slices of one XCD are made
    // contiguous in memory, so each XCD's L2 writes back one contiguous eighth of the observation tensor: -11.5 % per
    // configs[4] step (523 -> 465 us), -2 % on configs[2], neutral on short launches (buffer-controlled A/B,
    // profiles/r1/controlled_ab.txt).  PGX_FLAGS bit 3 switches back to the identity mapping for A/B.
    // The XCDs do not get through their streams equally fast (the odd ones lag 5-15 %, whatever they write to:
    // profiles/r2/xcd_rates.txt), so the shares need not be equal (pgx_xcd_tune): workgroup b is the (b >> 3)-th
    // of XCD b & 7 and takes slice xcd_base + (b >> 3) if that XCD still has one.
    // ASSUMPTION: blockIdx & 7 is the XCD -- true in SPX mode with 8 XCDs (round-robin dispatch of consecutive
    // workgroups), which is how this pool's MI355X run.  pgx_create checks it (256 CUs in one partition) and otherwise
    // sets bit 3 -- identity mapping, equal shares, no stagger (pgx_geometry.xcd_aware = 0): in a partitioned mode
    // (CPX/DPX ...) the remap would still cover every slice exactly once (results are unaffected), but the
    // contiguity-per-L2 and the tuned shares would mean nothing.
    int blk = blockIdx.x;
    if (!(p.flags & 8u)) {
        // PGX_FLAGS bits 10..12 (diagnostic): XCD x takes the share of XCD (x + rot) & 7 -- does a slow XCD stay slow?
        const int x = ((blk & 7) + (int)((p.flags >> 10) & 7u)) & 7, k = blk >> 3;
        if (k >= p.xcd_n[x]) return;
        blk = p.xcd_base[x] + k;
    } else if (blk >= (MW ? p.batch : (p.batch + epw - 1) / epw)) {
        return;
    }
    const int env0 = blk * epw;
    const int nenv = min(epw, p.batch - env0);
    if (p.only) {  // masked observe (after pgx_regenerate): untouched environments keep their observation
        bool any = false;
        for (int el = 0; el < nenv; ++el) any = any || p.only[env0 + el] != 0;
        if (!any) return;
    }
    const int A = p.num_agents;
    const int bmw = p.bm_words;
    const int wpr = p.wpr;
    const int r = p.r;
    const int W = 2 * r + 1;
    const int nag = nenv * A;             // agents handled by this workgroup
    const int AS = MW ? NT : epw * A;     // agent slots in LDS

    uint32_t* s_obst = smem;                         // (BIG: not staged, the region does not exist)
    uint32_t* s_occ = s_obst + (BIG ? 0 : epw * bmw);
    if constexpr (PC) s_occ += (((role == 0 ? t : t - 1) & 1) ? epw * bmw + 2 * AS : 0);  // {occupancy, cells, targets} x 2
    uint32_t* s_apos = s_occ + epw * bmw;  // [AS]
    uint32_t* s_atgt = s_apos + AS;        // [AS]
    uint32_t* s_vis = s_atgt + AS;         // MW only: [NT] cells as seen by others
    uint32_t* s_want = s_vis + (MW ? NT : 0);
    uint32_t* s_x0 = s_want + (MW ? NT : 0);   // closure exchange, double-buffered
    uint32_t* s_x1 = s_x0 + (MW ? NT : 0);
    uint32_t* s_misc = s_x1 + (MW ? NT : 0);   // MW only: round flags + env-wide reductions
    uint32_t* s_rows = s_misc + (MW ? MISC_WORDS : 0);  // generic path: [nag*3*W + 1] 32-bit row masks

    const bool dbg = (p.flags & 4u) && p.dbg;
    if (dbg && tid == 0) p.dbg[(size_t)blk * 4 + 0] = wall_clock64();
    // ---- phase 0: issue every global load of the step up front (one exposed HBM latency) ----------
    const int env_l = MW ? 0 : (lane / G);
    const int gbase = MW ? 0 : (lane & ~(G - 1));
    const int alane = MW ? lane : (lane & (G - 1));
    const int agent = MW ? tid : alane;
    const bool env_ok = env_l < nenv;
    const int env = env0 + env_l;
    const bool valid = env_ok && agent < A;
    const size_t gi = (size_t)env * A + agent;

    uint32_t pos = 0u, tgt = 1u;
    bool active = false;
    bool ghost_in = false;  // ACTIVE_GHOST of the stored byte (see `ghost` below)
    int act = 0, elapsed = 0;
    int4 macc = make_int4(0, 0, 0, 0);
    // (evaluated where each branch needs it: in the single-step form BEHIND the global loads -- `p.mode` is a scalar load, and
    // waiting for it first delays every load of the step by that round trip: 0.5-1 % per launch, profiles/r6/step_ab_r5_vs_r6.txt)
    bool env_leader;
    [[maybe_unused]] int araw[8];                 // ROLL: the actions of steps t+8 .. t+15 in flight (every eighth iteration)
    [[maybe_unused]] bool fetch_block = false;
    if constexpr (ROLL) {
      env_leader = env_ok && agent == 0 && p.mode == MODE_STEP;
      if (do_resolve) {
        // Register-resident loop: the state is loaded by the first iteration only and then carried in
        // `c`.  gfx9 retires loads and stores through ONE in-order counter (vmcnt), so a load issued behind the observation
        // stores of the previous iteration cannot be consumed before every one of those stores has been acknowledged
        // (1.5-2 us for a lone wave: profiles/r6/rollout_timeline_before.txt) -- hence no per-iteration loads at all: the
        // caller's actions arrive eight steps at a time, one block ahead, and are waited for once per block (below, behind
        // the state phase); the random policy needs no load.
        auto load_action = [&](size_t ai) -> int {
            if (p.action_dtype == 0) return ((const int8_t*)p.actions)[ai];
            if (p.action_dtype == 1) return ((const int32_t*)p.actions)[ai];
            return (int)((const int64_t*)p.actions)[ai];
        };
        const bool given = p.actions != nullptr;
        const int steps = rp.steps;
        const size_t stride = (size_t)rp.agents_stride;
        if (t == 0) {
            if (valid) {
                pos = p.pos[gi];
                tgt = p.tgt[gi];
                active = (p.active[gi] & ACTIVE_BIT) != 0;  // (ACTIVE_GHOST only matters to launches that LOOK at the state)
                if (given) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) araw[k] = k < steps ? load_action(gi + (size_t)k * stride) : 0;
                    uint32_t b = 0u;
#pragma unroll
                    for (int k = 0; k < 8; ++k) b |= pack_action4(araw[k]) << (4 * k);
                    c.ablk = b;
                }
            }
            if (env_ok) elapsed = p.elapsed[env];
            if (env_leader) macc = p.macc[env];
        } else {
            pos = c.pos;
            tgt = c.tgt;
            active = c.active;
            elapsed = c.elapsed;
            macc = c.macc;
        }
        if (valid) {
            if (!given) {  // the engine's own uniform random policy
                act = policy_action((uint64_t)rp.policy_seed, (uint64_t)(p.env_index_base + env), (uint32_t)agent,
                                    (uint64_t)(rp.policy_step0 + t));
                if (rp.actions_out) rp.actions_out[gi + (size_t)t * stride] = (int8_t)act;
            } else {
                act = (int)((c.ablk >> (4 * (t & 7))) & 15u);
                if ((t & 7) == 0 && t + 8 < steps) {
                    fetch_block = true;
#pragma unroll
                    for (int k = 0; k < 8; ++k) araw[k] = t + 8 + k < steps ? load_action(gi + (size_t)(t + 8 + k) * stride) : 0;
                }
            }
        }
      }
    } else {
    if (valid) {
        pos = p.pos[gi];
        tgt = p.tgt[gi];
        const uint32_t ab = p.active[gi];
        active = (ab & ACTIVE_BIT) != 0;
        ghost_in = (ab & ACTIVE_GHOST) != 0;
        if (p.mode == MODE_STEP) {
            if (p.action_dtype == 0) act = ((const int8_t*)p.actions)[gi];
            else if (p.action_dtype == 1) act = ((const int32_t*)p.actions)[gi];
            else act = (int)((const int64_t*)p.actions)[gi];
        }
    }
    if (env_ok && p.mode == MODE_STEP) elapsed = p.elapsed[env];
    env_leader = env_ok && agent == 0 && p.mode == MODE_STEP;
    if (env_leader) macc = p.macc[env];
    }

    // ---- phase 1: stage obstacle bitmaps HBM -> LDS, clear the occupancy bitmaps ---------------
    if (do_resolve) {
        const uint32_t* g = p.obst + (size_t)env0 * bmw;
        const int n = nenv * bmw;
        const int stagger = (ROLL && t > 0) ? 0 : p.stagger;  // only the first step of a rollout starts in lockstep
        // ROLL with a resident bitmap: the map does not change inside a launch, so the obstacle bitmap is staged by the
        // first iteration only and stays in its own LDS region (the P16 row masks alias what lies BEHIND it); later
        // iterations just clear the occupancy bitmap.
        bool stage = true;
        if constexpr (ROLL) stage = t == 0 || !rp.resident_bitmap;
        if constexpr (BIG) stage = false;  // nothing to stage: clear the occupancy bitmap
        if (!stage) {
#pragma unroll 4
            for (int i = tid; i < n; i += NT) s_occ[i] = 0u;
        } else if (!MW && stagger > 0 && n <= 16 * 64) {
            // Two cohorts.  Every wave issues ALL its loads at t = 0, while the memory system is idle; then the waves in
            // odd hardware slots sleep for `stagger` x 8128 cycles.  The even slots run their state phase with half the
            // SIMD/LDS contention and start streaming early; the odd slots compute under that stream (their loads are
            // already in registers, their small state stores are never waited for) and follow.  Without this every
            // wave runs the ~11 us state phase at the same time and HBM idles meanwhile.  Enabled by step_geometry()
            // for single-round launches of long streams (-2.3 % per configs[2] step, profiles/r1/controlled_ab.txt).
            uint32_t bmr[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) bmr[k] = (tid + 64 * k < n) ? g[tid + 64 * k] : 0u;
            const uint32_t slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));  // HW_REG_HW_ID.WAVE_ID
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (slot & 1u)
                for (int k = 0; k < stagger; ++k) __builtin_amdgcn_s_sleep(127);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (tid + 64 * k < n) {
                    s_obst[tid + 64 * k] = bmr[k];
                    s_occ[tid + 64 * k] = 0u;
                }
        } else
#pragma unroll 4
        for (int i = tid; i < n; i += NT) {
            s_obst[i] = g[i];
            s_occ[i] = 0u;
        }
        if (MW && tid < MISC_WORDS) s_misc[tid] = 0u;
    }
    lds_sync<MW>();
    const bool dbg2 = dbg && (p.flags & 64u);  // alternate stamps: [1] loads + staging done, [2] collisions resolved, [3] state phase done
    if (dbg2 && tid == 0) p.dbg[(size_t)blk * 4 + 1] = wall_clock64();

    // ---- phase 2: state update ---------------------------------------------------------------------
    // The per-step results leave the wave through emit_state(): ~10 small global stores (rewards, flags, agent state,
    // counters, metrics).  Under a saturated write stream a store INSTRUCTION can wait microseconds for a slot in the
    // CU's memory pipeline (profiles/r3/timeline_short_cfg3.txt: "collisions resolved" p90 8 us, "state phase done" p90
    // 25 us), so WHEN they are issued matters (StepParams::state_stores): 0 right after they are known, 1 after the
    // cells have been published to LDS and the barrier has been passed (helper waves no longer wait for them),
    // 2 after the wave's own observation stream (nobody waits for them at all).
    // docs/SPEC.md Q2, PGX_SOFT_OCCUPANCY_INDEX_ORDER (the recalled literal, value 0): upstream's per-agent clear-old /
    // set-new loop in index order leaves an agent that entered the cell a HIGHER-index agent is leaving out of the
    // occupancy array (that agent's turn clears the cell again) -- closed form of the literal loop: moved &&
    // occupant-of-destination index > own.  `Grid.positions` is persistent state upstream: the agent stays missing until a
    // later step's loop re-sets it, so the flag is kept with the agent (bit ACTIVE_GHOST of the `active` byte, written by
    // emit_state) and honoured by every launch that only LOOKS at the state (MODE_OBSERVE: pgx_observe, the masked observe
    // after pgx_regenerate; pgx_get_state's occupancy; snapshots carry the byte).  A step recomputes it from scratch: every
    // active agent takes its turn in the loop again.
    bool ghost = p.mode != MODE_STEP && ghost_in;
    struct StateOut {
        float rew = 0.0f;
        uint8_t term = 0;
        bool trunc = false, finished = false, do_reset = false, act = false;
        int n_arrived = 0;
    } so;
    int when_stores = p.mode == MODE_STEP ? p.state_stores : 0;
    if constexpr (PC) when_stores = 0;  // the resolver stores its results at once: nobody streams behind it in this wave
    const bool late_stores = when_stores != 0;
    [[maybe_unused]] bool last_step = true;
    if constexpr (ROLL) last_step = t == rp.steps - 1;
    auto emit_state = [&](uint32_t pos_, uint32_t tgt_, bool active_, int elapsed_, int4 macc_, const StateOut& o) {
        const bool fin = p.on_target == ON_TARGET_FINISH, coop = p.on_target == ON_TARGET_NOTHING;
        if (valid) {
            const size_t go = ROLL ? gi + (size_t)t * (size_t)rp.agents_stride : gi;
            p.rewards[go] = o.rew;
            p.terminated[go] = o.term;
            p.truncated[go] = o.trunc ? 1 : 0;
            if (p.act_out) p.act_out[go] = o.act ? 1 : 0;
            if (o.do_reset && p.np_state) p.np_state[gi] = p.np_state0[gi];  // upstream re-creates the generators in reset()
            if (!ROLL || last_step) {  // (rollout: the agents' state leaves the registers after the last step only)
                p.pos[gi] = pos_;
                p.tgt[gi] = tgt_;
                p.active[gi] = active_ ? (uint8_t)(ACTIVE_BIT | (ghost ? ACTIVE_GHOST : 0u)) : (uint8_t)0;
            }
        }
        if (env_leader) {
            if (!ROLL || last_step) p.elapsed[env] = o.do_reset ? 0 : elapsed_;
            // ---- metric wrappers, fused: per-env accumulators, emitted when the episode finishes ----
            const int n_arrived = o.n_arrived;
            const int step = elapsed_ - 1;
            if (fin) {
                macc_.x += n_arrived;
                macc_.y += n_arrived * step;
                if (n_arrived) macc_.z = max(macc_.z, step);
            } else if (p.on_target == ON_TARGET_RESTART) {
                macc_.w += n_arrived;
            }
            if (o.finished && p.metrics_out) {
                float* mo = p.metrics_out + ((size_t)env + (ROLL ? (size_t)t * (size_t)rp.envs_stride : 0)) * 6;
                const float fA = (float)A;
                if (fin) {
                    const int unsolved = A - macc_.x;
                    const int total = macc_.y + unsolved * step;
                    const int mx = unsolved ? step : macc_.z;
                    mo[0] = (float)macc_.x / fA; mo[1] = macc_.x == A ? 1.0f : 0.0f; mo[2] = (float)total / fA + 1.0f;
                    mo[3] = (float)(total + A); mo[4] = (float)(mx + 1); mo[5] = 0.0f;
                } else if (coop) {
                    mo[0] = (float)n_arrived / fA; mo[1] = n_arrived == A ? 1.0f : 0.0f; mo[2] = (float)(step + 1);
                    mo[3] = (float)(A * (step + 1)); mo[4] = (float)(step + 1); mo[5] = 0.0f;
                } else {
                    const int denom = p.max_steps > 0 ? p.max_steps : step + 1;
                    mo[0] = 0.0f; mo[1] = 0.0f; mo[2] = (float)(step + 1); mo[3] = 0.0f; mo[4] = 0.0f;
                    mo[5] = (float)macc_.w / (float)denom;
                }
            }
            if (p.episode_done) p.episode_done[(size_t)env + (ROLL ? (size_t)t * (size_t)rp.envs_stride : 0)] = o.finished ? 1 : 0;
            if constexpr (ROLL) {
                if (o.finished) macc_ = make_int4(0, 0, 0, 0);
                if (last_step) p.macc[env] = macc_;
                c.macc = macc_;
            } else {
                p.macc[env] = o.finished ? make_int4(0, 0, 0, 0) : macc_;
            }
        }
    };
    if (do_resolve) {
        const uint32_t* obm;
        if constexpr (BIG) obm = p.obst + (size_t)env0 * bmw;  // the HBM bitmap itself (one word per mover, L2)
        else obm = s_obst + env_l * bmw;
        if (act < 0 || act > 4) {  // docs/SPEC.md Q7: a noop either way; FLAG also counts it for the host's IndexError
            if (p.bad_action != 0 && active && p.mode == MODE_STEP) atomicAdd(p.bad_count, 1u);
            act = 0;
        }
        uint32_t cur = to_c22(pos);                     // 22-bit key of the own cell
        uint32_t vis = active ? cur : NOCELL_A;         // ... as seen by others (hidden agents stand nowhere)

        if (p.mode == MODE_STEP) {
            // ================= move + collision resolve =========================================
            // One all-pairs sweep gives, per agent: who stands on its destination (okey) and which
            // other agents claim the same destination (ckey).  The three collision systems of the
            // reference are closed-form functions of those two facts plus a transitive closure over
            // "the agent in front of me does not move"; tests/test_parity_gpu.py proves each form
            // equal to the literal sequential / dict-based algorithms of the oracle.
            const bool mover = active && act != 0;
            const uint32_t d = move_c22(cur, act);
            const bool blocked = mover && bm_test(obm, wpr, from_c22(d));
            // block_both registers a claim for every active agent (a noop claims its own cell)
            const bool claims = (p.collision == COLLISION_BLOCK_BOTH) ? active : mover;
            const uint32_t want = claims ? d : NOCELL_B;
            const int i = agent;
            SweepAcc acc = {KEY_NONE, KEY_NONE, KEY_NONE};
            const uint32_t wsh = want << 10, im1 = (uint32_t)(i - 1);
            const uint32_t pw = (vis << 10) | (uint32_t)i, qw = wsh | (uint32_t)i;
            const int src = gbase + ((alane + 1) & (G - 1));
            // `solo`: a multi-wave workgroup whose agents all sit in wave 0 (num_agents <= 64 with helper waves that only
            // share the observation write).  Its state phase is the single-wave one -- DPP / bpermute, no LDS exchange,
            // no barrier; the helper waves run the same code on invalid lanes and simply arrive early at the next barrier.
            const bool solo = MW && A <= 64;
            if (MW && !solo) {
                s_vis[tid] = pw;
                s_want[tid] = qw;
                lds_sync<true>();
                const int nwa = nw;  // every wave holds agents
                for (int sj = 0; sj < nwa; ++sj) {  // wave-uniform
                    const bool own = (sj == wave);
                    const uint32_t pj = own ? pw : s_vis[sj * 64 + lane];
                    const uint32_t qj = own ? qw : s_want[sj * 64 + lane];
                    sweep_slot<64>(pj, qj, !own, src, wsh, im1, acc);
                }
            } else {
                sweep_slot<G>(pw, qw, false, src, wsh, im1, acc);
            }
            const uint32_t okey = acc.okey;

            bool stay;
            if (p.collision == COLLISION_BLOCK_BOTH) {
                // blocked <=> destination is someone's current cell or claimed twice.
                stay = !mover || blocked || okey < 1024u || acc.cany < 1024u;
            } else {
                const int o = okey < 1024u ? (int)okey : -1;           // agent on my destination
                const bool lower = acc.zmin < (uint32_t)i;             // a lower index claims it too
                const int c1 = lower ? (i - 1 - (int)acc.zmin) : -1;   // the largest such index
                int nxt = mover ? o : -1;
                if (p.collision == COLLISION_PRIORITY) {
                    // SURVEY A3 (agents move one by one in index order).  Agent i ends up moving iff
                    // its destination is free AT ITS TURN: the occupant o (if any) has a lower index
                    // and moves away itself, and no claimant j with o < j < i got there first.
                    stay = !mover || blocked || o > i || c1 > o;
                } else {
                    // SURVEY A5 'soft' (net effect of the dict/recursion algorithm): the lowest-index
                    // claimant of a cell is the only candidate; edge swaps stay.  docs/SPEC.md Q1 alternative
                    // (PGX_SOFT_ALL_STAY): any other claimant blocks.
                    stay = !mover || blocked || (p.soft_rule != 0 ? acc.cany < 1024u : lower);
                    uint32_t want_of_o;
                    if (MW && !solo) {
                        want_of_o = nxt >= 0 ? (s_want[nxt] >> 10) : NOCELL_B;
                    } else {
                        const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((gbase + (nxt < 0 ? alane : nxt)) << 2, (int)want);
                        want_of_o = nxt >= 0 ? got : NOCELL_B;
                    }
                    stay = stay || (want_of_o == cur);  // edge swap
                }
                // transitive closure of "the agent on my destination stays": pointer doubling, at most
                // ceil(log2 A) rounds, usually 1-3 (exit as soon as every chain of the workgroup has ended).
                int rounds = 1;
                while ((1 << rounds) < A) ++rounds;
                for (int it = 0; it < rounds; ++it) {
                    const bool open = nxt >= 0 && !stay;
                    const uint32_t packed = (stay ? 0x80000000u : 0u) | (uint32_t)(nxt + 1);
                    uint32_t got;
                    if (MW && !solo) {
                        uint32_t* buf = (it & 1) ? s_x1 : s_x0;
                        buf[tid] = packed;
                        if (__ballot(open) != 0ull && lane == 0) s_misc[MISC_FLAGS + it] = 1u;
                        lds_sync<true>();
                        if (s_misc[MISC_FLAGS + it] == 0u) break;  // block-uniform
                        got = nxt >= 0 ? buf[nxt] : 0u;
                    } else {
                        if (__ballot(open) == 0ull) break;
                        got = (uint32_t)__builtin_amdgcn_ds_bpermute((gbase + (nxt < 0 ? alane : nxt)) << 2, (int)packed);
                    }
                    if (nxt >= 0) {
                        stay = stay || (got >> 31);
                        nxt = (int)(got & 0x7FFFFFFFu) - 1;
                    }
                }
            }
            if (dbg2 && tid == 0) p.dbg[(size_t)blk * 4 + 2] = wall_clock64();
            if (!stay) {
                cur = want;
                vis = want;
                pos = from_c22(want);
                ghost = p.soft_occupancy == 0 && p.collision == COLLISION_SOFT && okey < 1024u && (int)okey > i;
            }

            // ================= goals, rewards, done flags ==========
            const bool on_goal = valid && pos == tgt;
            const bool arrived = on_goal && active;  // `was_on_goal` of the reference
            bool solved, all_on_goal;
            if (MW && !solo) {
                const unsigned long long ma = __ballot(arrived);
                const unsigned long long mu = __ballot(valid && !arrived);
                const unsigned long long mo = __ballot(valid && !on_goal);
                if (lane == 0) {
                    if (ma) atomicAdd(&s_misc[MISC_ARRIVED], (uint32_t)__popcll(ma));
                    if (mu) s_misc[MISC_UNSOLVED] = 1u;
                    if (mo) s_misc[MISC_OFFGOAL] = 1u;
                }
                lds_sync<true>();
                so.n_arrived = (int)s_misc[MISC_ARRIVED];
                solved = s_misc[MISC_UNSOLVED] == 0u;
                all_on_goal = s_misc[MISC_OFFGOAL] == 0u;
            } else {
                const unsigned long long ma = __ballot(arrived);
                if constexpr (G == 64) so.n_arrived = __popcll(ma);
                else so.n_arrived = __popcll((ma >> gbase) & ((1ull << G) - 1ull));
                solved = !group_any<G>(valid && !arrived, gbase);
                all_on_goal = !group_any<G>(valid && !on_goal, gbase);
            }
            const bool coop = p.on_target == ON_TARGET_NOTHING;
            const bool fin = p.on_target == ON_TARGET_FINISH;
            so.rew = ((coop && p.coop_reward == 0) ? solved : arrived) ? 1.0f : 0.0f;  // Q4
            so.term = (coop ? solved : (fin && on_goal)) ? 1 : 0;
            const bool all_term = coop ? solved : (fin && all_on_goal);
            if (fin && on_goal) {  // hide_agent
                active = false;
                vis = NOCELL_A;
                ghost = false;
            }
            if (p.on_target == ON_TARGET_RESTART && on_goal) {
                const uint32_t x = (pos >> 16) - r, y = (pos & 0xFFFFu) - r;
                const size_t ci = (size_t)env * p.map_cells + (size_t)x * p.map_w + y;
                const uint32_t begin = p.comp_begin[ci];
                const uint32_t len = p.comp_len[ci];
                const uint32_t cnt = p.tcount[gi];
                uint32_t k;
                if (p.np_state) {  // PGX_LIFELONG_RNG_NUMPY: generator[agent].integers(0, len) == choice(component, 1)
                    pgxnp::Pcg64 g = np_unpack(p.np_state[gi]);
                    k = (uint32_t)pgxnp::integers_below(g, len);
                    p.np_state[gi] = np_pack(g);
                } else {
                    k = lifelong_draw(p.seed, (uint64_t)(p.env_index_base + env), (uint32_t)agent, cnt, len);
                }
                const uint32_t cell = p.comp_cells[(size_t)env * p.map_cells + begin + k];
                tgt = cell + (((uint32_t)r << 16) | (uint32_t)r);
                p.tcount[gi] = cnt + 1;
            }
            elapsed += 1;
            so.trunc = p.max_steps > 0 && elapsed >= p.max_steps;
            so.finished = all_term || so.trunc;
            so.do_reset = p.auto_reset && so.finished;
            so.act = active;  // is_active as reported for THIS step (before an auto-reset re-activates everybody)
            if (valid && so.do_reset) {  // auto-reset wrapper: observation comes from the reset state
                pos = p.pos0[gi];
                tgt = p.tgt0[gi];
                active = true;
                vis = to_c22(pos);
            }
            if (so.do_reset) ghost = false;
            if (!late_stores) emit_state(pos, tgt, active, elapsed, macc, so);
        }

        // ---- publish agent cells to LDS and rebuild the occupancy bitmap ------------------------
        if (p.obs && valid) {
            const int la = MW ? tid : (env_l * A + alane);
            s_apos[la] = pos;
            s_atgt[la] = tgt;
            if (vis != NOCELL_A && !ghost) {
                const uint32_t x = pos >> 16, y = pos & 0xFFFFu;
                atomicOr(&s_occ[env_l * bmw + x * wpr + (y >> 5)], 1u << (y & 31));
            }
        }
    }
    if constexpr (ROLL) {
      if (do_resolve) {
        c.pos = pos;
        c.tgt = tgt;
        c.active = active;
        c.elapsed = so.do_reset ? 0 : elapsed;
        if (fetch_block) {  // the one wait for loads in eight iterations: issued before the state phase, consumed behind it
            uint32_t b = 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) b |= pack_action4(araw[k]) << (4 * k);
            c.anext = b;
        }
      }
    }
    if (dbg && tid == 0) p.dbg[(size_t)blk * 4 + (dbg2 ? 3 : 1)] = wall_clock64();
    if (!p.obs) {
        if (late_stores) emit_state(pos, tgt, active, elapsed, macc, so);
        return;
    }
    if constexpr (PC) {
        if (!do_stream) return;  // the resolver's iteration ends here; the pair meets at the barrier in rollout_kernel
    }
    lds_sync<MW>();
    if (when_stores == 1) emit_state(pos, tgt, active, elapsed, macc, so);
    int out_slot = slot;
    if constexpr (PC) out_slot = slot == 0 ? rp.obs_slots - 1 : slot - 1;  // the streamer writes step t - 1
    float* const obs_out = ROLL ? reinterpret_cast<float*>(reinterpret_cast<char*>(p.obs) + (size_t)out_slot * (size_t)rp.obs_stride) : p.obs;

    if constexpr (P16) {
        // The window side is a compile-time constant for the radii that matter (W = 11: obs_radius 5, the default and
        // BASELINE configs[1..3]; 15: configs[4]; 7: configs[0]); other radii run the same code with W at run time.  With W
        // known the 2 x W LDS reads of an item are issued back to back and waited for once -- before, every row paid its
        // own LDS round trip, a 16-cycle multiply and two scalar branches, which a lone wave pays in latency.
        const int W_rt = W, r_rt = r;
        uint32_t* rows_base = smem;  // the packed rows go over the dead state -- in a rollout with a resident bitmap: behind it
        if constexpr (ROLL) rows_base = rp.resident_bitmap ? s_occ : smem;
        if constexpr (PC) rows_base = s_obst + epw * bmw + 2 * (epw * bmw + 2 * AS);  // behind both hand-over buffers
        auto p16_phases = [&](auto wt_tag) {
        constexpr int WT = decltype(wt_tag)::value;
        const int W = WT ? WT : W_rt;
        const int r = WT ? (WT - 1) / 2 : r_rt;
        // ---- phase 3 (P16): row masks -> registers -> (sync) -> packed u16 rows over the dead state -------
        const uint32_t wmask = (1u << W) - 1u;
        constexpr int RW = WT ? (WT + 1) / 2 : 8;  // words per item: two 16-bit rows each
        uint32_t rp[3][RW];  // W rows x 16 bit per item, 3 items per lane (nag * 3 <= 3 * NT)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int m = 0; m < RW; ++m) rp[t][m] = 0u;
            const int item = tid + t * NT;
            if (item < nag * 3) {
                const int la = item / 3;
                const int c = item - la * 3;
                const int el = MW ? 0 : (p.a_magic ? (int)__umulhi((uint32_t)la, p.a_magic) : la);  // la / A (la < 4096)
                const uint32_t cell = s_apos[la];
                const int x = (int)(cell >> 16), y = (int)(cell & 0xFFFFu);
                if (c < 2) {
                    const uint32_t* bm;
                    if constexpr (BIG) bm = c == 0 ? p.obst + (size_t)env0 * bmw : s_occ;  // flat loads: HBM (L2) or LDS
                    else bm = (c == 0 ? s_obst : s_occ) + el * bmw;
                    const int start = y - r;
                    const int w0 = start >> 5, sh = start & 31;
                    // Both words of the 64-bit funnel are read unconditionally: when w0 is the last word of the row the
                    // window ends inside it (sh + W <= 32), so whatever follows in LDS (the next row, the next array)
                    // never reaches the low W bits.
                    if constexpr (WT == 0) {  // window side at run time: one row at a time (two words in flight)
#pragma unroll
                        for (int wy = 0; wy < 16; ++wy) {
                            if (wy < W) {
                                const uint32_t* rowp = bm + (x - r + wy) * wpr + w0;
                                const uint32_t bits = (uint32_t)((((uint64_t)rowp[1] << 32) | rowp[0]) >> sh) & wmask;
                                rp[t][wy >> 1] |= bits << (16 * (wy & 1));
                            }
                        }
                    } else {
                    const uint32_t* row0 = bm + (x - r) * wpr + w0;
                        // RB rows (2 RB LDS words) in flight at a time: more would cost the 8th wave per SIMD its registers
                        constexpr int RB = WT == 11 ? 6 : WT == 15 ? 5 : 7;
    #pragma unroll
                        for (int w8 = 0; w8 < (WT ? WT : 16); w8 += RB) {
                            uint32_t lo[RB], hi[RB];
    #pragma unroll
                            for (int k = 0; k < RB; ++k) {
                                const int wy = w8 + k;
                                if (wy < (WT ? WT : 16) && (WT || wy < W)) {
                                    lo[k] = row0[wy * wpr];
                                    hi[k] = row0[wy * wpr + 1];
                                }
                            }
    #pragma unroll
                            for (int k = 0; k < RB; ++k) {
                                const int wy = w8 + k;
                                if (wy < (WT ? WT : 16) && (WT || wy < W)) {
                                    const uint32_t bits = (uint32_t)((((uint64_t)hi[k] << 32) | lo[k]) >> sh) & wmask;
                                    rp[t][wy >> 1] |= bits << (16 * (wy & 1));
                                }
                            }
                            asm volatile("" ::: "memory");  // keep the next batch of loads behind this batch's use
                        }
                    }
                } else {
                    // get_square_target: per-axis clamp of the offset to the window edge
                    const uint32_t tc = s_atgt[la];
                    int dx = x - (int)(tc >> 16), dy = y - (int)(tc & 0xFFFFu);
                    dx = max(-r, min(r, dx));
                    dy = max(-r, min(r, dy));
                    const int hit = r - dx;
                    const uint32_t val = (1u << (r - dy)) << (16 * (hit & 1));
#pragma unroll
                    for (int m = 0; m < RW; ++m) rp[t][m] = (m == (hit >> 1)) ? val : 0u;
                }
            }
        }
        lds_sync<MW>();  // every lane has read the bitmaps / agent cells: the region may be overwritten
        uint16_t* rows16 = reinterpret_cast<uint16_t*>(rows_base);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int item = tid + t * NT;
            if (item < nag * 3) {
#pragma unroll
                for (int wy = 0; wy < (WT ? WT : 16); ++wy)
                    if (WT || wy < W) rows16[item * W + wy] = (uint16_t)(rp[t][wy >> 1] >> (16 * (wy & 1)));
            }
        }
        if (tid < 4) rows16[nag * 3 * W + tid] = 0;
        lds_sync<MW>();

        // ---- phase 4 (P16): stream the observations --------------------------------------------------------
        const int n = nag * 3 * W * W;
        const size_t base = (size_t)env0 * A * 3 * W * W;
        if constexpr (WT == 0) {  // (the lighter formats run the run-time-W instance: keeps the float32 instances lean)
        if (p.obs_one && W >= 7) {  // bfloat16 / float16 planes, the fast walk (8 cells span at most two rows)
            uint16_t* o = reinterpret_cast<uint16_t*>(obs_out) + base;
            const int head = min(n, (int)((8 - (base & 7)) & 7));
            const int nvec = (n - head) >> 3;
            const int tail0 = head + (nvec << 3);
            if (tid < 16) {  // unaligned head / tail cells
                const int e = (tid < 8) ? tid : tail0 + (tid - 8);
                if ((tid < 8) ? (tid < head) : (e < n)) {
                    const int row = (int)__umulhi((uint32_t)e, p.w_magic);
                    o[e] = (uint16_t)((((uint32_t)rows16[row] >> (e - row * W)) & 1u) * p.obs_one);
                }
            }
            const bool span = MW && !(p.flags & 512u);
            const int part = (nvec + nw - 1) / nw;
            const int q0 = span ? wave * part + lane : tid, q1 = span ? min(nvec, (wave + 1) * part) : nvec;
            const int e0 = head + (q0 << 3);
            const int row0 = (int)__umulhi((uint32_t)e0, p.w_magic);
            f32x4_t* out4 = reinterpret_cast<f32x4_t*>(o + head);
            const uint32_t* rows32 = reinterpret_cast<const uint32_t*>(rows16);
            const int qs = span ? 64 : NT;
            if (q0 < q1) {
                if (p.store_policy == 0) stream_rows16_loop_h16<0>(out4, rows32, W, row0, e0 - row0 * W, q0, q1, qs, p.obs_one);
                else if (p.store_policy == 1) stream_rows16_loop_h16<1>(out4, rows32, W, row0, e0 - row0 * W, q0, q1, qs, p.obs_one);
                else stream_rows16_loop_h16<2>(out4, rows32, W, row0, e0 - row0 * W, q0, q1, qs, p.obs_one);
            }
            if (when_stores == 2) emit_state(pos, tgt, active, elapsed, macc, so);
            return;
        }
        if (p.obs_one) {  // ... window sides below 7: the generic funnel
            const int nrows = nag * 3 * W;
            stream_obs_h16(reinterpret_cast<uint16_t*>(obs_out), base, n, W, p.w_magic, tid, NT, p.store_policy == 1, p.obs_one,
                           [&](int row) -> uint32_t { return row < nrows ? (uint32_t)rows16[row] : 0u; }, wave,
                           (MW && !(p.flags & 512u)) ? nw : 1);
            if (when_stores == 2) emit_state(pos, tgt, active, elapsed, macc, so);
            return;
        }
        if (p.obs_u8) {
            if (dbg && !dbg2 && tid == 0) p.dbg[(size_t)blk * 4 + 2] = wall_clock64();
            const int nrows = nag * 3 * W;
            stream_obs_u8(reinterpret_cast<uint8_t*>(obs_out), base, n, W, p.w_magic, tid, NT, p.store_policy == 1,
                          [&](int row) -> uint32_t { return row < nrows ? (uint32_t)rows16[row] : 0u; }, wave,
                          (MW && !(p.flags & 512u)) ? nw : 1);
            if (when_stores == 2) emit_state(pos, tgt, active, elapsed, macc, so);
            if (dbg && !dbg2 && tid == 0) {
                __builtin_amdgcn_s_waitcnt(0);
                p.dbg[(size_t)blk * 4 + 3] = wall_clock64();
            }
            return;
        }
        }
        float* out = obs_out + base;
        const int head = min(n, (int)((4 - (base & 3)) & 3));
        const uint32_t magic = p.w_magic;
        const int nvec = (n - head) >> 2;
        const int tail0 = head + (nvec << 2);
        stream_rows16_edges(out, rows16, n, head, tail0, W, magic, tid);
        if (dbg && !dbg2 && tid == 0) p.dbg[(size_t)blk * 4 + 2] = wall_clock64();
        // Multi-wave workgroups: every wave streams its own CONTIGUOUS part of the slice, 1 KiB per store instruction,
        // like the single-wave kernel.  Interleaving the waves (q = tid, q += NT) makes each wave hop by NT * 16 bytes:
        // with 4 waves that is 4 KiB, with 8 waves 8 KiB -- every store of a wave then lands on the same memory channel
        // (configs[2] forced onto 3 / 4 / 8 waves with the interleaved order: 120 / 147 / 419 us per step, measured in
        // round 2 with tools/ab_inproc.py; the raw output was not kept -- PGX_WAVES=k with PGX_FLAGS=512 reproduces it).
        // PGX_FLAGS bit 9 restores the interleaved order for A/B.
        const bool span = MW && !(p.flags & 512u);
        const int part = (nvec + nw - 1) / nw;
        const int q0 = span ? wave * part + lane : tid, q1 = span ? min(nvec, (wave + 1) * part) : nvec;
        stream_rows16_span<true>(reinterpret_cast<f32x4_t*>(out + head), reinterpret_cast<const uint32_t*>(rows16), head, W, magic,
                           (uint32_t)p.store_policy, q0, q1, span ? 64 : NT, (nag * 3 * W) >> 1);
        if (when_stores == 2) emit_state(pos, tgt, active, elapsed, macc, so);
        if (dbg && !dbg2 && tid == 0) {
            __builtin_amdgcn_s_waitcnt(0);
            p.dbg[(size_t)blk * 4 + 3] = wall_clock64();
        }
        return;
        };
        if (p.obs_u8 || p.obs_one) p16_phases(std::integral_constant<int, 0>{});
        else if (W_rt == 11) p16_phases(std::integral_constant<int, 11>{});
        else if (W_rt == 15) p16_phases(std::integral_constant<int, 15>{});
        else if (W_rt == 7) p16_phases(std::integral_constant<int, 7>{});
        else p16_phases(std::integral_constant<int, 0>{});
        return;
    }

    // ---- phase 3: one 32-bit mask per (agent, channel, window row) --------------------------------
    {
        const uint32_t wmask = (W >= 32) ? 0xFFFFFFFFu : ((1u << W) - 1u);
        for (int item = tid; item < nag * 3; item += NT) {
            const int la = item / 3;
            const int c = item - la * 3;
            const int el = MW ? 0 : (p.a_magic ? (int)__umulhi((uint32_t)la, p.a_magic) : la);  // la / A
            const uint32_t cell = s_apos[la];
            const int x = (int)(cell >> 16), y = (int)(cell & 0xFFFFu);
            uint32_t* out = s_rows + item * W;
            if (c < 2) {
                const uint32_t* bm;
                if constexpr (BIG) bm = c == 0 ? p.obst + (size_t)env0 * bmw : s_occ;
                else bm = (c == 0 ? s_obst : s_occ) + el * bmw;
                const int start = y - r;  // >= 0: agents live inside the padded interior
                const int w0 = start >> 5, sh = start & 31;
                for (int wy = 0; wy < W; ++wy) {
                    const uint32_t* rowp = bm + (x - r + wy) * wpr + w0;
                    const uint32_t lo = rowp[0];
                    const uint32_t hi = (w0 + 1 < wpr) ? rowp[1] : 0u;
                    const uint64_t both = ((uint64_t)hi << 32) | lo;
                    out[wy] = (uint32_t)(both >> sh) & wmask;
                }
            } else {
                const uint32_t t = s_atgt[la];
                int dx = x - (int)(t >> 16), dy = y - (int)(t & 0xFFFFu);
                dx = max(-r, min(r, dx));
                dy = max(-r, min(r, dy));
                const int hit = r - dx;
                const uint32_t bit = 1u << (r - dy);
                for (int wy = 0; wy < W; ++wy) out[wy] = (wy == hit) ? bit : 0u;
            }
        }
        if (tid == 0) s_rows[nag * 3 * W] = 0u;
    }
    lds_sync<MW>();

    // ---- phase 4: stream the observations, 16 bytes per lane per store ---------------------------------
    if (p.obs_one) {
        const int nrows = nag * 3 * W;
        stream_obs_h16(reinterpret_cast<uint16_t*>(obs_out), (size_t)env0 * A * 3 * W * W, nrows * W, W, p.w_magic, tid, NT,
                       p.store_policy == 1, p.obs_one, [&](int row) -> uint32_t { return row < nrows ? s_rows[row] : 0u; }, wave,
                       (MW && !(p.flags & 512u)) ? nw : 1);
        if (when_stores == 2) emit_state(pos, tgt, active, elapsed, macc, so);
        return;
    }
    if (p.obs_u8) {
        const int nrows = nag * 3 * W;
        stream_obs_u8(reinterpret_cast<uint8_t*>(obs_out), (size_t)env0 * A * 3 * W * W, nrows * W, W, p.w_magic, tid, NT,
                      p.store_policy == 1, [&](int row) -> uint32_t { return row < nrows ? s_rows[row] : 0u; }, wave,
                      (MW && !(p.flags & 512u)) ? nw : 1);
        if (when_stores == 2) emit_state(pos, tgt, active, elapsed, macc, so);
        return;
    }
    {
        const int n = nag * 3 * W * W;  // floats written by this workgroup
        const size_t base = (size_t)env0 * A * 3 * W * W;
        float* out = obs_out + base;
        const int head = min(n, (int)((4 - (base & 3)) & 3));
        const uint32_t magic = p.w_magic;  // ceil(2^32 / W)
        // unaligned head / tail (only when the workgroup's float count is not a multiple of 4)
        const int nvec = (n - head) >> 2;
        const int tail0 = head + (nvec << 2);
        if (tid < 8) {
            const int e = (tid < 4) ? tid : tail0 + (tid - 4);
            const bool mine = (tid < 4) ? (tid < head) : (e < n);
            if (mine) {
                const int row = (int)__umulhi((uint32_t)e, magic);
                const int col = e - row * W;
                out[e] = (float)((s_rows[row] >> col) & 1u);
            }
        }
        if (dbg && !dbg2 && tid == 0) p.dbg[(size_t)blk * 4 + 2] = wall_clock64();
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4* out4 = reinterpret_cast<f32x4*>(out + head);
        const bool span = MW && !(p.flags & 512u);  // as in the P16 stream: contiguous part per wave
        const int part = (nvec + nw - 1) / nw;
        const int q0 = span ? wave * part + lane : tid, q1 = span ? min(nvec, (wave + 1) * part) : nvec;
        const int qs = span ? 64 : NT;
        auto stream32 = [&](auto policy_tag) {  // the store policy as a compile-time constant, as in stream_rows16_loop
            constexpr uint32_t POLICY = decltype(policy_tag)::value;
            for (int q = q0; q < q1; q += qs) {
                const int e0 = head + (q << 2);
                const int row = (int)__umulhi((uint32_t)e0, magic);
                const int col = e0 - row * W;
                const uint32_t b = (s_rows[row] >> col) | (s_rows[row + 1] << (W - col));
                f32x4 v;
                v.x = (float)(b & 1u);
                v.y = (float)((b >> 1) & 1u);
                v.z = (float)((b >> 2) & 1u);
                v.w = (float)((b >> 3) & 1u);
                store_obs16(reinterpret_cast<f32x4_t*>(&out4[q]), v, POLICY);
            }
        };
        if (p.store_policy == 0) stream32(std::integral_constant<uint32_t, 0u>{});
        else if (p.store_policy == 1) stream32(std::integral_constant<uint32_t, 1u>{});
        else stream32(std::integral_constant<uint32_t, 2u>{});
        if (when_stores == 2) emit_state(pos, tgt, active, elapsed, macc, so);
        if (dbg && !dbg2 && tid == 0) {
            __builtin_amdgcn_s_waitcnt(0);  // stores retired (vmcnt 0) before the end stamp
            p.dbg[(size_t)blk * 4 + 3] = wall_clock64();
        }
    }
}

template <int G, bool MW, bool P16, bool BIG = false>
__global__ __launch_bounds__(MW ? 1024 : 64, MW ? 1 : 8) void step_kernel(const StepParams p) {
    const RolloutParams none{};
    Carry unused;
    step_body<G, MW, P16, false, BIG, false>(p, none, 0, 0, unused);
}

// K steps in ONE launch (pgx_rollout).  Environments never interact, so a workgroup can run its own environments
// through all K steps without waiting for anybody else: step t of the launch reads actions[t] and writes the outputs
// of slice t (observations: slot t % obs_slots).  Unlike K launches there is no launch boundary at which all waves
// start their state phase together with HBM idle and finish together: after the first step the waves drift apart and
// one wave's state phase runs under the other waves' observation streams.
// Round 6: a real on-device loop.  Until round 5 every iteration re-loaded pos / tgt / active / elapsed / macc and the
// obstacle bitmap from HBM, stored them again and drained vmcnt(0) before the next one -- a K-step launch was no faster
// than K launches where it matters (configs[1] 8.2 us per step against 7.9; per-iteration stamps of that kernel:
// profiles/r6/rollout_timeline_before.txt -- 1.2 us loads + staging, 2 us run-time-W row masks, 2 us from the first
// observation store to the last acknowledgement).  Now the agent / env state lives in registers (`Carry`), the obstacle
// bitmap stays in LDS, the caller's actions are fetched eight steps at a time one block ahead, state goes back to HBM
// after the last step, and NOTHING waits for a store: a wave's observation stores of step t are still in flight while it
// resolves step t + 1 (the hardware's own limit of 64 outstanding vector-memory instructions per wave is the only
// back-pressure).  Stores and later loads of one lane to the same address (tcount, np_state: lifelong) are kept in
// order by the hardware.  Multi-wave workgroups keep one LDS barrier between iterations (the next iteration clears
// what the slower waves still stream from) -- a barrier that no longer drains the vector-memory queue.
// The launch runs at most PGX_ROLL_OCC (4) waves per SIMD: 128 VGPRs -- room for the carried state, the action block in
// flight and the specialised (compile-time window side, software-pipelined) row / stream code of step_kernel, which the
// 64-VGPR rollout kernels of rounds 3-5 had to do without; a loop that never waits does not need eight waves per SIMD
// to keep HBM busy (in-process A/B of both builds: profiles/r6/).
// Bit-identical with K pgx_step calls (tests/test_rollout_gpu.py).
// The argument blocks are read through a kernarg pointer that is laundered once per iteration: otherwise LICM hoists
// every argument load and every address computation of the (inlined) step out of the loop.
#ifndef PGX_ROLL_OCC
#define PGX_ROLL_OCC 4
#endif
template <int G, bool MW, bool P16, bool BIG = false, bool PC = false>
__global__ __launch_bounds__(MW ? 1024 : (PC ? 128 : 64), MW ? 1 : PGX_ROLL_OCC) void rollout_kernel(const StepParams p0, const RolloutParams rp0) {
    typedef const __attribute__((address_space(4))) char KC;
    typedef const __attribute__((address_space(4))) StepParams KP;
    typedef const __attribute__((address_space(4))) RolloutParams KR;
    // kernarg layout: explicit arguments in order, each at its natural alignment (code object v5)
    static_assert(alignof(StepParams) == 8 && alignof(RolloutParams) == 8, "kernarg layout of rollout_kernel");
    constexpr size_t rp_offset = (sizeof(StepParams) + alignof(RolloutParams) - 1) / alignof(RolloutParams) * alignof(RolloutParams);
    const int steps = rp0.steps, slots = rp0.obs_slots;
    int slot = 0;
    Carry c;
    for (int t = 0; t < steps + (PC ? 1 : 0); ++t) {  // (PC: the streamer is one step behind the resolver)
        KC* ka = (KC*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        KP& p = *reinterpret_cast<KP*>(ka);
        KR& rp = *reinterpret_cast<KR*>(ka + rp_offset);
        step_body<G, MW, P16, true, BIG, PC>(p, rp, t, slot, c);
        slot = slot + 1 == slots ? 0 : slot + 1;
        if ((t & 7) == 7) c.ablk = c.anext;
        if constexpr (MW || PC) lds_sync<true>();
    }
}

// ------------------------------------------------------------------------------------------------
// reset helpers
// ------------------------------------------------------------------------------------------------
// u8 [B,H,W] obstacles -> padded 1-bit-per-cell bitmap with the artificial border of SURVEY A1:
// padding r, OBSTACLE ring at offset r-1 (and at r+H / r+W), FREE outside.
__global__ void pack_obstacles_kernel(const uint8_t* __restrict__ obstacles, const uint8_t* __restrict__ only,
                                      uint32_t* __restrict__ bm, int batch, int H, int Wd, int r, int wpr, int bmw,
                                      const OutsideParams outside) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)batch * bmw;
    if (gid >= total) return;
    const int env = (int)(gid / bmw);
    if (only && !only[env]) return;
    const int w = (int)(gid - (size_t)env * bmw);
    const int x = w / wpr;
    const int y0 = (w - x * wpr) * 32;
    const int PH = H + 2 * r, PW = Wd + 2 * r;
    uint32_t bits = 0u;
    const bool ring_row = (x == r - 1) || (x == PH - r);
    const bool in_rows = (x >= r) && (x < PH - r);
    uint64_t h_out = 0;
    if (outside.enabled)
        h_out = gen_outside_hash(outside.seed, (uint64_t)(outside.env_index_base + env), outside.epoch[env]);
    for (int b = 0; b < 32; ++b) {
        const int y = y0 + b;
        if (y >= PW) break;
        uint32_t v = 0u;
        const bool span = (y >= r - 1) && (y <= PW - r);
        if (ring_row && span) v = 1u;
        else if (in_rows && (y == r - 1 || y == PW - r)) v = 1u;
        else if (in_rows && y >= r && y < PW - r)
            v = obstacles[((size_t)env * H + (x - r)) * Wd + (y - r)] != 0 ? 1u : 0u;
        else if (outside.enabled && gen_is_outside(x, y, PH, PW, r))
            v = gen_outside_bit(h_out, x, y, PW, outside.thr);
        bits |= v << b;
    }
    bm[gid] = bits;
}

__global__ void pack_agents_kernel(const int32_t* __restrict__ agent_xy, const int32_t* __restrict__ target_xy,
                                   uint32_t* __restrict__ pos, uint32_t* __restrict__ tgt,
                                   uint32_t* __restrict__ pos0, uint32_t* __restrict__ tgt0,
                                   uint8_t* __restrict__ active, uint32_t* __restrict__ tcount, size_t n, int r,
                                   NpGen* __restrict__ np_state, const NpGen* __restrict__ np_state0) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pc = ((uint32_t)(agent_xy[2 * i] + r) << 16) | (uint32_t)(agent_xy[2 * i + 1] + r);
    const uint32_t tc = ((uint32_t)(target_xy[2 * i] + r) << 16) | (uint32_t)(target_xy[2 * i + 1] + r);
    pos[i] = pc;
    pos0[i] = pc;
    tgt[i] = tc;
    tgt0[i] = tc;
    active[i] = 1;
    if (tcount) tcount[i] = 0u;
    if (np_state) np_state[i] = np_state0[i];
}

// lifelong_rng = NUMPY (upstream PogemaLifeLong._initialize_grid, recalled): per env
//   main = default_rng(seed + global env); seeds = main.integers(2^31 - 1, size=A); generator[a] = default_rng(seeds[a])
__global__ void init_np_lifelong_kernel(NpGen* __restrict__ np_state0, uint64_t seed, int64_t env_index_base, int batch, int A) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= batch) return;
    pgxnp::Pcg64 main_rng = pgxnp::default_rng(seed + (uint64_t)(env_index_base + env));
    for (int a = 0; a < A; ++a) {
        const uint64_t s = pgxnp::integers_below(main_rng, 2147483647ull);
        np_state0[(size_t)env * A + a] = np_pack(pgxnp::default_rng(s));
    }
}

// pgx_set_targets: overwrite the targets of the flagged agents (all when mask == null)
__global__ void set_targets_kernel(const int32_t* __restrict__ target_xy, const uint8_t* __restrict__ mask,
                                   uint32_t* __restrict__ tgt, size_t n, int r) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (mask && !mask[i])) return;
    tgt[i] = ((uint32_t)(target_xy[2 * i] + r) << 16) | (uint32_t)(target_xy[2 * i + 1] + r);
}

__global__ void zero_i32_kernel(int32_t* v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = 0;
}

__global__ void unpack_state_kernel(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ tgt,
                                    const uint8_t* __restrict__ active, int32_t* __restrict__ agent_xy,
                                    int32_t* __restrict__ target_xy, uint8_t* __restrict__ act_out, size_t n,
                                    int r) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (agent_xy) {
        agent_xy[2 * i] = (int32_t)(pos[i] >> 16) - r;
        agent_xy[2 * i + 1] = (int32_t)(pos[i] & 0xFFFFu) - r;
    }
    if (target_xy) {
        target_xy[2 * i] = (int32_t)(tgt[i] >> 16) - r;
        target_xy[2 * i + 1] = (int32_t)(tgt[i] & 0xFFFFu) - r;
    }
    if (act_out) act_out[i] = active[i] & ACTIVE_BIT;
}

// occupancy export: u8 [B, PH, PW]; must be zero-filled by the caller (hipMemsetAsync) first.
__global__ void occupancy_kernel(const uint32_t* __restrict__ pos, const uint8_t* __restrict__ active,
                                 uint8_t* __restrict__ occ, size_t n, int A, int PH, int PW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if ((active[i] & (ACTIVE_BIT | ACTIVE_GHOST)) != ACTIVE_BIT) return;  // hidden, or standing there unseen (Q2 literal)
    const size_t env = i / A;
    const uint32_t x = pos[i] >> 16, y = pos[i] & 0xFFFFu;
    occ[(env * PH + x) * PW + y] = 1;
}

// ------------------------------------------------------------------------------------------------
// host-side launch helpers (called from pgx_api.cpp through pgx_internal.h)
// ------------------------------------------------------------------------------------------------
template <int G, bool MW, bool P16>
static const void* step_fn() { return reinterpret_cast<const void*>(&step_kernel<G, MW, P16>); }

static const void* step_fn_for(const StepGeometry& g) {
    if (g.big) return g.p16 ? reinterpret_cast<const void*>(&step_kernel<64, true, true, true>)
                            : reinterpret_cast<const void*>(&step_kernel<64, true, false, true>);
    if (g.multi_wave) return g.p16 ? step_fn<64, true, true>() : step_fn<64, true, false>();
#define PGX_CASE(gg) case gg: return g.p16 ? step_fn<gg, false, true>() : step_fn<gg, false, false>();
    switch (g.G) {
        PGX_CASE(1) PGX_CASE(2) PGX_CASE(4) PGX_CASE(8) PGX_CASE(16) PGX_CASE(32) PGX_CASE(64)
        default: return nullptr;
    }
#undef PGX_CASE
}

// G, waves per block, envs per wave, P16 and the LDS footprint for one configuration.
//   epw_override > 0 forces the number of environments per wave (A <= 32 only; clamped to 64/G).
StepGeometry step_geometry(int batch, int A, int bmw, int W, bool allow_p16, int epw_override, int obs_elem_bytes,
                           int waves_override, bool for_rollout) {
    StepGeometry g{};
    g.multi_wave = A > 64;
    g.G = 64;
    g.waves = (A + 63) / 64;
    // Helper waves: a launch of few, large environments (fewer workgroups than 4 per CU, >= 16 KB of observations each)
    // leaves the chip idle while each lone wave streams its whole tensor slice; such environments run on the multi-wave
    // kernel with 2 or 4 waves -- wave 0 holds the agents, the others only share the observation write (64-agent 64x64
    // envs: 17.8 -> 11.7 us per step at batch 16, 19.6 -> 13.9 at batch 256, 21.9 -> 18.2 at batch 512).
    // waves_override (PGX_WAVES): 1 = never, k > 1 = force k waves (tests cover both paths).
    if (!g.multi_wave) {
        const size_t env_stream = (size_t)A * 3 * W * W * (size_t)obs_elem_bytes;
        int helpers = 1;
        if (batch < 1024 && env_stream >= 16 * 1024) {
            helpers = 2;
            while (helpers < 4 && batch * helpers < 1024) helpers <<= 1;  // 8 waves measured slower than 1
        }
        // Three waves per environment for LARGE launches of 64-agent-class environments (>= 64 KB of float32
        // observations each): 3 x batch waves no longer fit the chip at once, so the launch runs in overlapping rounds
        // -- later workgroups compute under earlier ones' streams and finish in finer grains -- while each workgroup
        // still idles only two waves during its state phase.  In-process A/B, same buffers (profiles/r2/helpers_ab.txt):
        // configs[2] 123.6 -> 118.8 us (2 waves 123.9, 4 waves 145), batch 4096: 84.8 -> 82.8, batch 2048: 49.0 -> 44.5;
        // worse for batch 1024 (26.0 -> 27.8), for 32- and 16-agent environments, for uint8 observations and inside a
        // rollout launch (whose waves drift apart anyway: 114.6 -> 118.9), which keep one wave.
        // (the 16-bit float formats likewise -- same cells, half the bytes: configs[2] bfloat16 68.4 -> 63.3 us,
        // gpurun r4r/bf16_waves_cfg2.txt; uint8 46.2 -> 52.7: stays on one wave)
        if (helpers == 1 && !for_rollout && A > 32 && batch >= 2048 && obs_elem_bytes >= 2 &&
            env_stream * (size_t)(4 / obs_elem_bytes) >= 64 * 1024)
            helpers = 3;
        if (waves_override == 1) helpers = 1;
        else if (waves_override > 1) helpers = waves_override > 16 ? 16 : waves_override;
        if (helpers > 1) {
            g.multi_wave = true;
            g.waves = helpers;
        }
    }
    if (!g.multi_wave) {
        g.G = 1;
        while (g.G < A) g.G <<= 1;
    }
    const int max_epw = g.multi_wave ? 1 : 64 / g.G;
    g.epw = max_epw;
    if (!g.multi_wave) {
        // Fewer environments per wave = more, shorter waves: every wave's prologue (loads, bitmap staging,
        // collision sweep) overlaps with other waves' observation streams, and small launches still cover
        // the 256 CUs.  Measured (profiles/r1/epw_sweep.txt): best while the launch stays within ~4 rounds
        // of the 8192 resident waves; beyond that the per-wave fixed cost shows (A = 8: +16 % at 65536 waves).
        while (g.epw > 1 && (batch + g.epw / 2 - 1) / (g.epw / 2) <= 32768) g.epw >>= 1;
        // The on-device loop prefers LONGER store bursts per wave: a wave that pauses (its state phase) between bursts of
        // less than ~32 KB costs an HBM-sized ring bandwidth that the same bytes in 46 KB bursts from half as many waves do
        // not (tools/drift_probe2.hip, profiles/r6/drift_probe_pause_kinds.txt: 23 KB x 8192 waves +3.1 us per step for a
        // 2.75 us pause, 46 KB x 4096 waves +0.3); configs[3] shard, in-process A/B on shared rings
        // (profiles/r6/rollout_cfg3_epw_ab.txt): 2 envs per wave 33.7 against 35.5 us into an 8-slot ring, 23.8 against
        // 23.9-24.1 into the two-slot one.  Only while more than 4096 waves remain to cover the chip.
        if (for_rollout)
            while (g.epw < max_epw && (size_t)g.epw * A * 3 * W * W * (size_t)obs_elem_bytes < 32 * 1024 &&
                   (batch + g.epw - 1) / g.epw > 4096)
                g.epw <<= 1;
        if (epw_override > 0) g.epw = epw_override < max_epw ? epw_override : max_epw;
    }
    g.p16 = allow_p16 && W <= 16;
    g.store_policy = g.multi_wave ? 1 : 2;  // see store_obs16()
    // The small per-step result stores go out after the LDS barrier (emit_state(), mode 1): in-process A/B on shared
    // buffers (profiles/r3/state_stores_ab.txt) configs[3] 39.9 -> 39.1 us, configs[2] 115.7 -> 115.1, configs[1]/[4]
    // unchanged; after the stream (mode 2) configs[3] 39.3, configs[2] 116.5.  PGX_STATE_STORES overrides.
    g.state_stores = 1;
    // Cohort stagger (step_kernel phase 1): pays when every wave of the launch is resident at once (one round of at
    // most 256 CUs x 32 waves, at least half of them used) and each wave streams for long (>= 32 KB of observations):
    // -2.3 % per configs[2] step in a buffer-controlled A/B on two boxes (profiles/r1/controlled_ab.txt); with several
    // rounds every odd-slot wave of every round would pay the delay.  PGX_STAGGER overrides (0 = off).
    {
        const long blocks = g.multi_wave ? batch : (batch + g.epw - 1) / g.epw;
        const size_t stream_bytes = (size_t)g.epw * A * 3 * W * W * (size_t)obs_elem_bytes;
        g.stagger = (!g.multi_wave && blocks <= 8192 && blocks >= 4096 && stream_bytes >= 32 * 1024 &&
                     (size_t)g.epw * bmw <= 16 * 64) ? 6 : 0;
    }
    g.big = false;
    int NT = 64 * g.waves;
    size_t agent_slots = g.multi_wave ? (size_t)NT : (size_t)g.epw * A;
    size_t state_words = (size_t)2 * g.epw * bmw + 2 * agent_slots;
    if (g.multi_wave) state_words += (size_t)4 * NT + MISC_WORDS;
    size_t nag = (size_t)g.epw * A;
    auto layout_bytes = [&]() -> size_t {
        if (g.p16) {
            const size_t rows_bytes = (nag * 3 * W + 4) * 2;  // u16 rows alias the whole state region
            return state_words * 4 > rows_bytes ? state_words * 4 : rows_bytes;
        }
        return (state_words + nag * 3 * W + 1) * 4;
    };
    size_t bytes = layout_bytes();
    // The large-map layout is also the FASTER one well before the staged layout stops fitting: once an environment's two
    // bitmaps take more than ~64 KB, at most two workgroups fit a CU and every step stages tens of KB that the windows
    // hardly touch (profiles/r6/big_vs_staged.txt, same instances and buffers: 640 x 640 x 256 agents 114.0 -> 83.6 us per
    // 1024-env step, 768 x 768 122.1 -> 105.6, 512 x 512 x 64 agents 119.3 -> 77.0; equal at 512 x 512 x 256 agents and below).
    const char* force_big = getenv("PGX_BIG");  // diagnostic / tests: "1" = on any map, "0" = only when nothing else fits
    const size_t big_from = (force_big && force_big[0] == '0') ? (size_t)160 * 1024 : (size_t)64 * 1024;
    if (bytes > big_from || (force_big && force_big[0] == '1')) {
        // Large maps: ONE environment per workgroup, only the occupancy bitmap in LDS, obstacles read from the HBM bitmap
        // through the L2 (step_body, BIG).  At least four waves: they share the clearing of the bitmap, the row masks and
        // the observation write; wave 0 holds the agents when there are at most 64.
        g.big = true;
        g.multi_wave = true;
        g.G = 64;
        g.epw = 1;
        g.waves = (A + 63) / 64;
        if (g.waves < 4) g.waves = 4;
        if (waves_override > 1 && waves_override >= (A + 63) / 64) g.waves = waves_override > 16 ? 16 : waves_override;
        g.store_policy = 1;
        g.stagger = 0;
        NT = 64 * g.waves;
        agent_slots = (size_t)NT;
        nag = (size_t)A;
        state_words = (size_t)bmw + 2 * agent_slots + (size_t)4 * NT + MISC_WORDS;
        bytes = layout_bytes();
    }
    g.resident_bitmap = false;
    g.pc = false;
    if (for_rollout && !g.big) {
        // rollout_kernel stages the obstacle bitmap once per launch: it keeps its own LDS region and the P16 rows alias
        // only what lies behind it.  Costs LDS when the rows are the larger part; if that no longer fits one CU the
        // launch re-stages every iteration (layout of step_kernel).
        const size_t bm_bytes = (size_t)g.epw * bmw * 4;
        const size_t rest = state_words * 4 - bm_bytes;
        size_t res;
        if (g.p16) {
            const size_t rows_bytes = (nag * 3 * W + 4) * 2;
            res = bm_bytes + (rest > rows_bytes ? rest : rows_bytes);
        } else {
            res = bytes;
        }
        const char* off = getenv("PGX_ROLL_RESIDENT");  // diagnostic / tests: "0" forces the re-staging layout
        if (res <= 160 * 1024 && !(off && off[0] == '0')) {
            g.resident_bitmap = true;
            bytes = res;
        }
        // Resolver / streamer pair (step_body, PC): small single-wave environments whose wave would otherwise alternate a
        // state phase with a short store burst.  float32 observations, packed rows, resident bitmap; not the 64-lane
        // groups (their 93 KB bursts already keep the memory pipeline full: configs[2] is stream-bound either way).
        const char* pc = getenv("PGX_ROLL_PC");  // diagnostic / tests: "0" off, "1" wherever the instance exists
        const bool pc_ok = !g.multi_wave && g.p16 && g.resident_bitmap && g.G <= 32 && obs_elem_bytes == 4;
        const size_t burst = nag * 3 * W * W * 4;
        // ... and only where BOTH waves of every pair are resident at once (<= 4096 waves at four per SIMD): on a launch that
        // fills the chip anyway the pair halves the environments in flight and is slower (configs[3] shard 24.2 -> 25.6 us,
        // configs[1] 4.95 -> 3.25: profiles/r6/rollout_pair_ab.txt)
        const long pairs = (batch + g.epw - 1) / g.epw;
        g.pc = pc_ok && ((pc && pc[0] == '1') || (!(pc && pc[0] == '0') && burst < 64 * 1024 && pairs * 2 <= 4096));
        if (g.pc) {
            const size_t bufw = (size_t)g.epw * bmw + 2 * agent_slots;
            bytes = ((size_t)g.epw * bmw + 2 * bufw) * 4 + (nag * 3 * W + 4) * 2 + 16;
            if (bytes > 160 * 1024) {
                g.pc = false;
                bytes = res;
            }
        }
    }
    g.lds_bytes = (bytes + 15) & ~(size_t)15;
    return g;
}

// once per handle (NOT per launch: keeps pgx_step capturable in a HIP graph).  The > 48 KB opt-in is an attribute of
// the kernel FUNCTION on a device, not of a handle: two live handles may share a template instance with different LDS
// needs, so the limit is only ever raised (largest request so far per function and device).
static hipError_t raise_lds_limit(const void* fn, size_t lds_bytes);
static const void* rollout_fn_for(const StepGeometry& g);

hipError_t prepare_step(const StepGeometry& g, const StepGeometry& roll) {
    const void* fn = step_fn_for(g);
    const void* rfn = rollout_fn_for(roll);
    if (!fn || !rfn) return hipErrorInvalidValue;
    if (hipError_t e = raise_lds_limit(fn, g.lds_bytes)) return e;
    return raise_lds_limit(rfn, roll.lds_bytes);
}

static hipError_t raise_lds_limit(const void* fn, size_t lds_bytes) {
    if (lds_bytes <= 48 * 1024) return hipSuccess;
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = granted[{fn, dev}];
    if (lds_bytes <= have) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) have = lds_bytes;
    return e;
}

template <int G, bool MW, bool P16>
static const void* rollout_fn() { return reinterpret_cast<const void*>(&rollout_kernel<G, MW, P16>); }

static const void* rollout_fn_for(const StepGeometry& g) {
    if (g.pc) {
#define PGX_CASE(gg) case gg: return reinterpret_cast<const void*>(&rollout_kernel<gg, false, true, false, true>);
        switch (g.G) {
            PGX_CASE(1) PGX_CASE(2) PGX_CASE(4) PGX_CASE(8) PGX_CASE(16) PGX_CASE(32)
            default: return nullptr;
        }
#undef PGX_CASE
    }
    if (g.big) return g.p16 ? reinterpret_cast<const void*>(&rollout_kernel<64, true, true, true>)
                            : reinterpret_cast<const void*>(&rollout_kernel<64, true, false, true>);
    if (g.multi_wave) return g.p16 ? rollout_fn<64, true, true>() : rollout_fn<64, true, false>();
#define PGX_CASE(gg) case gg: return g.p16 ? rollout_fn<gg, false, true>() : rollout_fn<gg, false, false>();
    switch (g.G) {
        PGX_CASE(1) PGX_CASE(2) PGX_CASE(4) PGX_CASE(8) PGX_CASE(16) PGX_CASE(32) PGX_CASE(64)
        default: return nullptr;
    }
#undef PGX_CASE
}

hipError_t launch_rollout(const StepParams& p, const RolloutParams& rp, const StepGeometry& g, hipStream_t stream) {
    const void* fn = rollout_fn_for(g);
    if (!fn) return hipErrorInvalidValue;
    StepParams args = p;
    RolloutParams rargs = rp;
    void* kargs[] = {&args, &rargs};
    return hipLaunchKernel(fn, dim3(g.grid), dim3(64 * g.waves * (g.pc ? 2 : 1)), kargs, g.lds_bytes, stream);
}

// ---- shares of a launch's workgroups per XCD -------------------------------------------------------------------------
int xcd_partition(int blocks, const float w[8], int32_t n[8], int32_t base[8]) {
    double acc = 0.0;
    int given = 0, grid = 0;
    for (int x = 0; x < 8; ++x) {  // cumulative rounding: the shares add up to `blocks` exactly
        acc += (double)w[x];
        int upto = x == 7 ? blocks : (int)(acc * blocks + 0.5);
        if (upto > blocks) upto = blocks;
        if (upto < given) upto = given;
        n[x] = upto - given;
        base[x] = given;
        given = upto;
        if (n[x] > grid) grid = n[x];
    }
    return grid * 8;
}

hipError_t launch_step(const StepParams& p, const StepGeometry& g, hipStream_t stream) {
    const void* fn = step_fn_for(g);
    if (!fn) return hipErrorInvalidValue;
    StepParams args = p;
    void* kargs[] = {&args};
    return hipLaunchKernel(fn, dim3(g.grid), dim3(64 * g.waves), kargs, g.lds_bytes, stream);
}


hipError_t launch_pack_obstacles(const uint8_t* obstacles, const uint8_t* only, uint32_t* bm, int batch, int H, int Wd,
                                 int r, int wpr, int bmw, const OutsideParams& outside, hipStream_t stream) {
    const size_t total = (size_t)batch * bmw;
    const int bs = 256;
    hipLaunchKernelGGL(pack_obstacles_kernel, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, stream,
                       obstacles, only, bm, batch, H, Wd, r, wpr, bmw, outside);
    return hipGetLastError();
}

hipError_t launch_pack_agents(const int32_t* agent_xy, const int32_t* target_xy, uint32_t* pos, uint32_t* tgt,
                              uint32_t* pos0, uint32_t* tgt0, uint8_t* active, uint32_t* tcount, size_t n,
                              int r, hipStream_t stream, NpGen* np_state, const NpGen* np_state0) {
    const int bs = 256;
    hipLaunchKernelGGL(pack_agents_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, stream, agent_xy,
                       target_xy, pos, tgt, pos0, tgt0, active, tcount, n, r, np_state, np_state0);
    return hipGetLastError();
}

hipError_t launch_init_np_lifelong(NpGen* np_state0, uint64_t seed, int64_t env_index_base, int batch, int A, hipStream_t stream) {
    hipLaunchKernelGGL(init_np_lifelong_kernel, dim3((batch + 63) / 64), dim3(64), 0, stream, np_state0, seed, env_index_base, batch, A);
    return hipGetLastError();
}

hipError_t launch_set_targets(const int32_t* target_xy, const uint8_t* mask, uint32_t* tgt, size_t n, int r,
                              hipStream_t stream) {
    const int bs = 256;
    hipLaunchKernelGGL(set_targets_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, stream, target_xy, mask, tgt, n, r);
    return hipGetLastError();
}

hipError_t launch_zero_i32(int32_t* v, size_t n, hipStream_t stream) {
    const int bs = 256;
    hipLaunchKernelGGL(zero_i32_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, stream, v, n);
    return hipGetLastError();
}

hipError_t launch_unpack_state(const uint32_t* pos, const uint32_t* tgt, const uint8_t* active, int32_t* agent_xy,
                               int32_t* target_xy, uint8_t* act_out, size_t n, int r, hipStream_t stream) {
    const int bs = 256;
    hipLaunchKernelGGL(unpack_state_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, stream, pos, tgt,
                       active, agent_xy, target_xy, act_out, n, r);
    return hipGetLastError();
}

hipError_t launch_occupancy(const uint32_t* pos, const uint8_t* active, uint8_t* occ, size_t n, int A, int PH,
                            int PW, hipStream_t stream) {
    const int bs = 256;
    hipLaunchKernelGGL(occupancy_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, stream, pos, active,
                       occ, n, A, PH, PW);
    return hipGetLastError();
}

}  // namespace pgx
