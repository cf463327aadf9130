// pgx_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the vectorized POGEMA step engine.
//
// Hot path replaced (upstream names, see include/pogema_amd.h and SURVEY.md section 8a):
//   Pogema.step / move_agents / Grid.move / _obs() / MultiTimeLimit.step           rows A2..A13
//
// Design (DESIGN.md has the full story):
//   * one WAVE owns one environment (or 64/G environments when num_agents <= 32, G = next pow2 of
//     num_agents); lane = agent.  For num_agents > 64 each lane carries K = ceil(A/64) agents in
//     registers and three more waves of the workgroup only help with the observation write.
//   * collision resolution is register-resident: the agent-index-ordered semantics of the
//     reference are reproduced with v_readlane / ds_bpermute broadcasts + 64-lane ballots; no
//     global atomics, no per-cell tables in HBM.
//   * the padded obstacle bitmap (1 bit per cell) is staged HBM -> LDS once per step, the occupancy
//     bitmap is rebuilt in LDS from the agents' cells with LDS atomics, every (agent, channel,
//     window-row) is reduced to one 32-bit row mask in LDS, and the float32 observation tensor is
//     then produced as a flat, fully coalesced stream of 16-byte stores (the only large HBM
//     stream of the kernel: 12*(2r+1)^2 bytes per agent-step).
//   * integer indexing only -- no MFMA on purpose; the bound is HBM write bandwidth.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgx_internal.h"

namespace pgx {

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// DESIGN.md "lifelong RNG": uniform index in [0, n) for (seed, global env, agent, counter).
__device__ __forceinline__ uint32_t lifelong_draw(uint64_t seed, uint64_t env_index, uint32_t agent,
                                                  uint32_t counter, uint32_t n) {
    uint64_t h = splitmix64(seed);
    h = splitmix64(h ^ env_index);
    h = splitmix64(h ^ (((uint64_t)agent << 32) | counter));
    return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32);
}

// packed cell: (x << 16) | y in PADDED coordinates; MOVES = noop, up, down, left, right with the
// first index the row.
__device__ __forceinline__ uint32_t apply_move(uint32_t cell, int a) {
    const uint32_t delta = (a == 1) ? 0xFFFF0000u : (a == 2) ? 0x00010000u : (a == 3) ? 0xFFFFFFFFu
                         : (a == 4) ? 0x00000001u : 0u;
    return cell + delta;
}

__device__ __forceinline__ uint32_t bm_test(const uint32_t* bm, int wpr, uint32_t cell) {
    const uint32_t x = cell >> 16, y = cell & 0xFFFFu;
    return (bm[x * wpr + (y >> 5)] >> (y & 31)) & 1u;
}

constexpr uint32_t NOCELL_A = 0xFFFFFFFFu;  // "stands nowhere"   (hidden / invalid lane)
constexpr uint32_t NOCELL_B = 0xFFFFFFFEu;  // "claims nothing"

// Broadcast of agent j's value to its environment group.
//   G == 64 (one env per wave, incl. K > 1): j's lane is wave-uniform -> v_readlane_b32.
//   G <  64: source lane differs per group -> ds_bpermute via __shfl.
template <int G>
__device__ __forceinline__ uint32_t group_bcast(uint32_t v, int lj, int gbase) {
    if constexpr (G == 64) {
        return (uint32_t)__builtin_amdgcn_readlane((int)v, lj);
    } else {
        return (uint32_t)__shfl((int)v, gbase + lj, 64);
    }
}

// "does any lane of my environment group satisfy pred"
template <int G>
__device__ __forceinline__ bool group_any(bool pred, int gbase) {
    const unsigned long long m = __ballot(pred);
    if constexpr (G == 64) {
        return m != 0ull;
    } else {
        constexpr unsigned long long gm = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
        return ((m >> gbase) & gm) != 0ull;
    }
}

template <int G>
__device__ __forceinline__ bool group_all(bool pred, int gbase, unsigned long long validmask) {
    // all over the lanes selected by validmask (bits relative to the group)
    const unsigned long long m = __ballot(pred);
    if constexpr (G == 64) {
        return (m & validmask) == validmask;
    } else {
        return ((m >> gbase) & validmask) == validmask;
    }
}

// ------------------------------------------------------------------------------------------------
// The step kernel.  K = agents per lane, G = lanes per environment group (power of two).
//   K == 1 : block = 1 wave, 64/G envs per block.
//   K  > 1 : block = 4 waves, 1 env per block, G == 64; wave 0 resolves the moves.
// ------------------------------------------------------------------------------------------------
template <int K, int G>
__global__ __launch_bounds__((K == 1) ? 64 : 256) void step_kernel(const StepParams p) {
    constexpr int NT = (K == 1) ? 64 : 256;
    constexpr int EPW = (K == 1) ? (64 / G) : 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int env0 = blockIdx.x * EPW;
    const int nenv = min(EPW, p.batch - env0);
    const int A = p.num_agents;
    const int bmw = p.bm_words;
    const int wpr = p.wpr;
    const int r = p.r;
    const int W = 2 * r + 1;
    const int nag = nenv * A;  // agents handled by this workgroup

    uint32_t* s_obst = smem;
    uint32_t* s_occ = s_obst + EPW * bmw;
    uint32_t* s_apos = s_occ + EPW * bmw;       // [EPW*A]
    uint32_t* s_atgt = s_apos + EPW * A;        // [EPW*A]
    uint32_t* s_rows = s_atgt + EPW * A;        // [EPW*A*3*W + 1]
    uint32_t* s_flag = s_rows + EPW * A * 3 * W + 1;  // [max(A,1)] soft-closure scratch (K > 1)

    // ---- phase 1: stage obstacle bitmaps HBM -> LDS, clear the occupancy bitmaps ---------------
    {
        const uint32_t* g = p.obst + (size_t)env0 * bmw;
        for (int i = tid; i < nenv * bmw; i += NT) {
            s_obst[i] = g[i];
            s_occ[i] = 0u;
        }
    }
    __syncthreads();

    // ---- phase 2: state update (wave 0) ---------------------------------------------------------
    if (K == 1 || tid < 64) {
        const int env_l = (K == 1) ? (lane / G) : 0;
        const int gbase = (K == 1) ? (lane & ~(G - 1)) : 0;
        const int alane = (K == 1) ? (lane & (G - 1)) : lane;  // agent index within slot
        const bool env_ok = env_l < nenv;
        const int env = env0 + env_l;
        const uint32_t* obm = s_obst + env_l * bmw;

        uint32_t pos[K], tgt[K], vis[K];
        bool valid[K], active[K];
        int act[K];
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const int agent = s * 64 + alane;
            valid[s] = env_ok && agent < A;
            const size_t gi = (size_t)env * A + agent;
            pos[s] = valid[s] ? p.pos[gi] : NOCELL_A;
            tgt[s] = valid[s] ? p.tgt[gi] : NOCELL_B;
            active[s] = valid[s] ? (p.active[gi] != 0) : false;
            int a = 0;
            if (valid[s] && p.mode == MODE_STEP) {
                if (p.action_dtype == 0) a = ((const int8_t*)p.actions)[gi];
                else if (p.action_dtype == 1) a = ((const int32_t*)p.actions)[gi];
                else a = (int)((const int64_t*)p.actions)[gi];
                if (a < 0 || a > 4) a = 0;
            }
            act[s] = a;
            vis[s] = active[s] ? pos[s] : NOCELL_A;
        }

        if (p.mode == MODE_STEP) {
            // ================= move + collision resolve =========================================
            if (p.collision == COLLISION_PRIORITY) {
                // Sequential semantics: agent j moves iff its destination is free of
                // obstacles and of agents AT ITS TURN (lower indices already moved).  The loop runs
                // in agent-index order; occupancy lives in the `vis` registers of the group.
                uint32_t dstm[K];  // bit31 = wants-to-move-legally, low bits = destination
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const uint32_t d = apply_move(pos[s], act[s]);
                    const bool mv = active[s] && act[s] != 0 && !bm_test(obm, wpr, d);
                    dstm[s] = mv ? (d | 0x80000000u) : 0u;
                }
#pragma unroll
                for (int sj = 0; sj < K; ++sj) {
                    const int jn = min(64, A - sj * 64);
                    const int jmax = (K == 1) ? min(G, A) : jn;
                    for (int lj = 0; lj < jmax; ++lj) {
                        const uint32_t dj = group_bcast<G>(dstm[sj], lj, gbase);
                        if (G == 64 && dj == 0u) continue;  // wave-uniform skip
                        const uint32_t d = dj & 0x7FFFFFFFu;
                        bool hit = false;
#pragma unroll
                        for (int s = 0; s < K; ++s) hit |= (vis[s] == d);
                        const bool occupied = group_any<G>(hit, gbase);
                        if (dj != 0u && !occupied && alane == lj) {
                            pos[sj] = d;
                            vis[sj] = d;
                        }
                    }
                }
            } else if (p.collision == COLLISION_BLOCK_BOTH) {
                // a destination is blocked if it is any active agent's current cell or is
                // claimed by two agents (claims are made regardless of obstacles).
                uint32_t raw[K];
                bool conflict[K];
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    raw[s] = active[s] ? apply_move(pos[s], act[s]) : NOCELL_B;
                    conflict[s] = false;
                }
#pragma unroll
                for (int sj = 0; sj < K; ++sj) {
                    const int jn = min(64, A - sj * 64);
                    const int jmax = (K == 1) ? min(G, A) : jn;
                    for (int lj = 0; lj < jmax; ++lj) {
                        const uint32_t cj = group_bcast<G>(vis[sj], lj, gbase);
                        const uint32_t dj = group_bcast<G>(raw[sj], lj, gbase);
#pragma unroll
                        for (int s = 0; s < K; ++s) {
                            const bool self = (s == sj) && (alane == lj);
                            conflict[s] |= !self && (raw[s] == cj || raw[s] == dj);
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    if (active[s] && act[s] != 0 && !conflict[s] && !bm_test(obm, wpr, raw[s])) {
                        pos[s] = raw[s];
                        vis[s] = raw[s];
                    }
                }
            } else {
                // SURVEY A5 'soft'.  Net effect of the reference's dict/recursion algorithm, proven
                // equal to it by tests/test_collision_equivalence.py (literal oracle vs this form):
                //   agent i stays  <=>  noop | destination is an obstacle | edge swap with another
                //   mover | a LOWER-index mover claims the same destination | the agent standing on
                //   the destination stays (transitively).
                uint32_t raw[K];
                bool mover[K], stay[K];
                int nxt[K];
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    mover[s] = active[s] && act[s] != 0;
                    raw[s] = mover[s] ? apply_move(pos[s], act[s]) : NOCELL_B;
                    stay[s] = !mover[s] || bm_test(obm, wpr, raw[s]);
                    nxt[s] = -1;
                }
#pragma unroll
                for (int sj = 0; sj < K; ++sj) {
                    const int jn = min(64, A - sj * 64);
                    const int jmax = (K == 1) ? min(G, A) : jn;
                    for (int lj = 0; lj < jmax; ++lj) {
                        const uint32_t cj = group_bcast<G>(vis[sj], lj, gbase);
                        const uint32_t dj = group_bcast<G>(raw[sj], lj, gbase);  // NOCELL_B if not a mover
                        const int j = sj * 64 + lj;
#pragma unroll
                        for (int s = 0; s < K; ++s) {
                            const int i = s * 64 + alane;
                            if (mover[s] && i != j) {
                                if (cj == raw[s]) nxt[s] = j;                      // j stands on my destination
                                if (dj == raw[s] && j < i) stay[s] = true;         // lower index wins the cell
                                if (dj == pos[s] && cj == raw[s]) stay[s] = true;  // edge swap
                            }
                        }
                    }
                }
                // transitive closure over "the agent on my destination stays": pointer doubling.
                int rounds = 1;
                while ((1 << rounds) < A) ++rounds;
                if constexpr (K == 1) {
                    int nx = nxt[0];
                    bool st = stay[0];
                    for (int it = 0; it < rounds; ++it) {
                        const int src = gbase + (nx < 0 ? alane : nx);
                        const int st_n = __shfl((int)st, src, 64);
                        const int nx_n = __shfl(nx, src, 64);
                        if (nx >= 0) {
                            st = st || (st_n != 0);
                            nx = nx_n;
                        }
                    }
                    stay[0] = st;
                } else {
                    // A > 64: exchange through LDS (s_flag holds {stay bit31 | next+1}).
                    for (int it = 0; it < rounds; ++it) {
#pragma unroll
                        for (int s = 0; s < K; ++s) {
                            const int i = s * 64 + alane;
                            if (i < A) s_flag[i] = (stay[s] ? 0x80000000u : 0u) | (uint32_t)(nxt[s] + 1);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        uint32_t got[K];
#pragma unroll
                        for (int s = 0; s < K; ++s) got[s] = (nxt[s] >= 0) ? s_flag[nxt[s]] : 0u;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                        for (int s = 0; s < K; ++s) {
                            if (nxt[s] >= 0) {
                                stay[s] = stay[s] || (got[s] >> 31);
                                nxt[s] = (int)(got[s] & 0x7FFFFFFFu) - 1;
                            }
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    if (mover[s] && !stay[s]) {
                        pos[s] = raw[s];
                        vis[s] = raw[s];
                    }
                }
            }

            // ================= goals, rewards, done flags ==========
            bool on_goal[K];
            bool all_goal_l = true, all_term_l = true;
            float rew[K];
            uint8_t term[K];
#pragma unroll
            for (int s = 0; s < K; ++s) {
                on_goal[s] = valid[s] && pos[s] == tgt[s];
                if (valid[s]) all_goal_l = all_goal_l && on_goal[s] && active[s];
            }
            // env-wide AND (all lanes of the group; invalid lanes contribute true)
            const bool solved = !group_any<G>(!all_goal_l, gbase);
#pragma unroll
            for (int s = 0; s < K; ++s) {
                if (p.on_target == ON_TARGET_FINISH) {
                    rew[s] = (on_goal[s] && active[s]) ? 1.0f : 0.0f;
                    term[s] = on_goal[s] ? 1 : 0;
                    if (on_goal[s]) {  // hide_agent
                        active[s] = false;
                        vis[s] = NOCELL_A;
                    }
                } else if (p.on_target == ON_TARGET_RESTART) {
                    rew[s] = (on_goal[s] && active[s]) ? 1.0f : 0.0f;
                    term[s] = 0;
                    if (on_goal[s]) {
                        const int agent = s * 64 + alane;
                        const size_t gi = (size_t)env * A + agent;
                        const uint32_t x = (pos[s] >> 16) - r, y = (pos[s] & 0xFFFFu) - r;
                        const size_t ci = (size_t)env * p.map_cells + (size_t)x * p.map_w + y;
                        const uint32_t begin = p.comp_begin[ci];
                        const uint32_t len = p.comp_len[ci];
                        const uint32_t cnt = p.tcount[gi];
                        const uint32_t k = lifelong_draw(p.seed, (uint64_t)(p.env_index_base + env),
                                                         (uint32_t)agent, cnt, len);
                        const uint32_t cell = p.comp_cells[(size_t)env * p.map_cells + begin + k];
                        tgt[s] = cell + (((uint32_t)r << 16) | (uint32_t)r);
                        p.tcount[gi] = cnt + 1;
                    }
                } else {
                    rew[s] = solved ? 1.0f : 0.0f;
                    term[s] = solved ? 1 : 0;
                }
                if (valid[s]) all_term_l = all_term_l && (term[s] != 0);
            }
            const bool all_term = !group_any<G>(!all_term_l, gbase);
            int elapsed = env_ok ? p.elapsed[env] : 0;
            elapsed += 1;
            const bool trunc = p.max_steps > 0 && elapsed >= p.max_steps;
            const bool do_reset = p.auto_reset && (all_term || trunc);
#pragma unroll
            for (int s = 0; s < K; ++s) {
                if (valid[s]) {
                    const int agent = s * 64 + alane;
                    const size_t gi = (size_t)env * A + agent;
                    p.rewards[gi] = rew[s];
                    p.terminated[gi] = term[s];
                    p.truncated[gi] = trunc ? 1 : 0;
                    if (p.act_out) p.act_out[gi] = active[s] ? 1 : 0;
                    if (do_reset) {  // auto-reset wrapper: observation comes from the reset state
                        pos[s] = p.pos0[gi];
                        tgt[s] = p.tgt0[gi];
                        active[s] = true;
                        vis[s] = pos[s];
                    }
                    p.pos[gi] = pos[s];
                    p.tgt[gi] = tgt[s];
                    p.active[gi] = active[s] ? 1 : 0;
                }
            }
            if (env_ok && alane == 0) p.elapsed[env] = do_reset ? 0 : elapsed;
        }

        // ---- publish agent cells to LDS and rebuild the occupancy bitmap ------------------------
        if (p.obs) {
#pragma unroll
            for (int s = 0; s < K; ++s) {
                if (valid[s]) {
                    const int la = env_l * A + s * 64 + alane;
                    s_apos[la] = pos[s];
                    s_atgt[la] = tgt[s];
                    if (vis[s] != NOCELL_A) {
                        const uint32_t x = vis[s] >> 16, y = vis[s] & 0xFFFFu;
                        atomicOr(&s_occ[env_l * bmw + x * wpr + (y >> 5)], 1u << (y & 31));
                    }
                }
            }
        }
    }
    if (!p.obs) return;
    __syncthreads();

    // ---- phase 3: one 32-bit mask per (agent, channel, window row) --------------------------------
    {
        const uint32_t wmask = (W >= 32) ? 0xFFFFFFFFu : ((1u << W) - 1u);
        for (int item = tid; item < nag * 3; item += NT) {
            const int la = item / 3;
            const int c = item - la * 3;
            const int env_l = (K == 1) ? (la / A) : 0;
            const uint32_t cell = s_apos[la];
            const int x = (int)(cell >> 16), y = (int)(cell & 0xFFFFu);
            uint32_t* out = s_rows + item * W;
            if (c < 2) {
                const uint32_t* bm = (c == 0 ? s_obst : s_occ) + env_l * bmw;
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
                // get_square_target: per-axis clamp of the offset to the window edge
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
    __syncthreads();

    // ---- phase 4: stream the float32 observations, 16 bytes per lane per store ---------------------
    {
        const int n = nag * 3 * W * W;  // floats written by this workgroup
        const size_t base = (size_t)env0 * A * 3 * W * W;
        float* out = p.obs + base;
        const int head = min(n, (int)((4 - (base & 3)) & 3));
        const uint32_t magic = p.w_magic;  // ceil(2^32 / W)
        // unaligned head / tail (only when A*3*W*W*EPW is not a multiple of 4)
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
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4* out4 = reinterpret_cast<f32x4*>(out + head);
        for (int q = tid; q < nvec; q += NT) {
            const int e0 = head + (q << 2);
            const int row = (int)__umulhi((uint32_t)e0, magic);
            const int col = e0 - row * W;
            const uint32_t b = (s_rows[row] >> col) | (s_rows[row + 1] << (W - col));
            f32x4 v;
            v.x = (float)(b & 1u);
            v.y = (float)((b >> 1) & 1u);
            v.z = (float)((b >> 2) & 1u);
            v.w = (float)((b >> 3) & 1u);
            __builtin_nontemporal_store(v, &out4[q]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// reset helpers
// ------------------------------------------------------------------------------------------------
// u8 [B,H,W] obstacles -> padded 1-bit-per-cell bitmap with the artificial border of SURVEY A1:
// padding r, OBSTACLE ring at offset r-1 (and at r+H / r+W), FREE outside.
__global__ void pack_obstacles_kernel(const uint8_t* __restrict__ obstacles, uint32_t* __restrict__ bm,
                                      int batch, int H, int Wd, int r, int wpr, int bmw) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)batch * bmw;
    if (gid >= total) return;
    const int env = (int)(gid / bmw);
    const int w = (int)(gid - (size_t)env * bmw);
    const int x = w / wpr;
    const int y0 = (w - x * wpr) * 32;
    const int PH = H + 2 * r, PW = Wd + 2 * r;
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
        else if (in_rows && y >= r && y < PW - r)
            v = obstacles[((size_t)env * H + (x - r)) * Wd + (y - r)] != 0 ? 1u : 0u;
        bits |= v << b;
    }
    bm[gid] = bits;
}

__global__ void pack_agents_kernel(const int32_t* __restrict__ agent_xy, const int32_t* __restrict__ target_xy,
                                   uint32_t* __restrict__ pos, uint32_t* __restrict__ tgt,
                                   uint32_t* __restrict__ pos0, uint32_t* __restrict__ tgt0,
                                   uint8_t* __restrict__ active, uint32_t* __restrict__ tcount, size_t n, int r) {
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
    if (act_out) act_out[i] = active[i];
}

// occupancy export: u8 [B, PH, PW]; must be zero-filled by the caller (hipMemsetAsync) first.
__global__ void occupancy_kernel(const uint32_t* __restrict__ pos, const uint8_t* __restrict__ active,
                                 uint8_t* __restrict__ occ, size_t n, int A, int PH, int PW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!active[i]) return;
    const size_t env = i / A;
    const uint32_t x = pos[i] >> 16, y = pos[i] & 0xFFFFu;
    occ[(env * PH + x) * PW + y] = 1;
}

// ------------------------------------------------------------------------------------------------
// host-side launch helpers (called from pgx_api.cpp through pgx_internal.h)
// ------------------------------------------------------------------------------------------------
template <int K, int G>
static hipError_t launch_step_t(const StepParams& p, size_t lds_bytes, hipStream_t stream) {
    constexpr int NT = (K == 1) ? 64 : 256;
    constexpr int EPW = (K == 1) ? (64 / G) : 1;
    const int blocks = (p.batch + EPW - 1) / EPW;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel<K, G>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((step_kernel<K, G>), dim3(blocks), dim3(NT), lds_bytes, stream, p);
    return hipGetLastError();
}

hipError_t launch_step(const StepParams& p, int K, int G, size_t lds_bytes, hipStream_t stream) {
    if (K == 1) {
        switch (G) {
            case 1: return launch_step_t<1, 1>(p, lds_bytes, stream);
            case 2: return launch_step_t<1, 2>(p, lds_bytes, stream);
            case 4: return launch_step_t<1, 4>(p, lds_bytes, stream);
            case 8: return launch_step_t<1, 8>(p, lds_bytes, stream);
            case 16: return launch_step_t<1, 16>(p, lds_bytes, stream);
            case 32: return launch_step_t<1, 32>(p, lds_bytes, stream);
            case 64: return launch_step_t<1, 64>(p, lds_bytes, stream);
            default: return hipErrorInvalidValue;
        }
    }
    switch (K) {
        case 2: return launch_step_t<2, 64>(p, lds_bytes, stream);
        case 4: return launch_step_t<4, 64>(p, lds_bytes, stream);
        case 8: return launch_step_t<8, 64>(p, lds_bytes, stream);
        case 16: return launch_step_t<16, 64>(p, lds_bytes, stream);
        default: return hipErrorInvalidValue;
    }
}

size_t step_lds_bytes(int K, int G, int A, int bmw, int W) {
    const int EPW = (K == 1) ? (64 / G) : 1;
    size_t words = (size_t)2 * EPW * bmw + (size_t)2 * EPW * A + (size_t)EPW * A * 3 * W + 1 + (size_t)(A > 0 ? A : 1);
    return ((words * 4) + 15) & ~(size_t)15;
}

hipError_t launch_pack_obstacles(const uint8_t* obstacles, uint32_t* bm, int batch, int H, int Wd, int r,
                                 int wpr, int bmw, hipStream_t stream) {
    const size_t total = (size_t)batch * bmw;
    const int bs = 256;
    hipLaunchKernelGGL(pack_obstacles_kernel, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, stream,
                       obstacles, bm, batch, H, Wd, r, wpr, bmw);
    return hipGetLastError();
}

hipError_t launch_pack_agents(const int32_t* agent_xy, const int32_t* target_xy, uint32_t* pos, uint32_t* tgt,
                              uint32_t* pos0, uint32_t* tgt0, uint8_t* active, uint32_t* tcount, size_t n,
                              int r, hipStream_t stream) {
    const int bs = 256;
    hipLaunchKernelGGL(pack_agents_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, stream, agent_xy,
                       target_xy, pos, tgt, pos0, tgt0, active, tcount, n, r);
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
