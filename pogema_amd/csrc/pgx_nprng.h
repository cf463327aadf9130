// pgx_nprng.h -- numpy-compatible random primitives, host and device (header-only, plain integer arithmetic).
//
// Upstream POGEMA draws everything random from numpy `Generator`s (`np.random.default_rng(seed)`: SeedSequence -> PCG64
// -> `integers` / `choice` / `shuffle` / `binomial` / `random`; upstream pogema/generator.py and the per-agent generators
// of `PogemaLifeLong`, names recalled -- the source is not mounted, /root/reference/README.md:3,5).  The CALL SEQUENCE
// upstream makes cannot be pinned without the source, but the primitive layer can be pinned today against numpy itself
// (numpy is in the build image): tools/gen_numpy_vectors.py writes tests/golden/numpy_rng_vectors.npz, and
// tests/test_nprng*.py check this header -- on the host and on the GPU -- against it, bit for bit.
//
// Restated from numpy's published algorithms (numpy 2.2: numpy/random/bit_generator.pyx `SeedSequence`,
// src/pcg64/pcg64.h, src/distributions/distributions.c), not copied:
//   SeedSequence(int)      pool of 4 uint32 mixed from the entropy words (hashmix / mix with the constants below)
//   PCG64                  128-bit LCG (multiplier 0x2360ED051FC65DA44385DF649FCCF645), output XSL-RR 128/64; seeded
//                          from generate_state(4, uint64): state = (w0 << 64 | w1), inc = ((w2 << 64 | w3) << 1) | 1
//   next_uint32            low half of a fresh 64-bit output, the high half is buffered for the next call
//   random()               (next64 >> 11) * 2^-53
//   integers(0, n)         Lemire's multiply-shift rejection on 32 bits for n <= 2^32, on 64 bits beyond (int64 path)
//   shuffle / permutation  Fisher-Yates from the top, index from masked rejection (`random_interval`)
//   binomial(1, p)         the inversion algorithm of `random_binomial` for n = 1 (qn = exp(log(q)) supplied by the host)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)  // hipcc (host + device); a plain C++ compiler sees host-only inlines
#define PGX_NP_HD __host__ __device__ inline
#else
#define PGX_NP_HD static inline
#endif

namespace pgxnp {

typedef unsigned __int128 u128;

struct Pcg64 {
    u128 state, inc;
    uint32_t has_uint32, uinteger;
};

// ---- SeedSequence ---------------------------------------------------------------------------------------------------
constexpr uint32_t SS_INIT_A = 0x43b0d7e5u, SS_MULT_A = 0x931e8875u, SS_INIT_B = 0x8b51f9ddu, SS_MULT_B = 0x58f38dedu;
constexpr uint32_t SS_MIX_L = 0xca01f9ddu, SS_MIX_R = 0x4973f715u;

PGX_NP_HD uint32_t ss_hashmix(uint32_t value, uint32_t& hash_const) {
    value ^= hash_const;
    hash_const *= SS_MULT_A;
    value *= hash_const;
    value ^= value >> 16;
    return value;
}
PGX_NP_HD uint32_t ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = SS_MIX_L * x - SS_MIX_R * y;
    r ^= r >> 16;
    return r;
}
// SeedSequence(seed).pool for a non-negative integer seed < 2^64 (numpy splits it into little-endian 32-bit words;
// 0 is the single word [0]).
PGX_NP_HD void seed_sequence_pool(uint64_t seed, uint32_t pool[4]) {
    uint32_t entropy[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const int n = (seed >> 32) ? 2 : 1;
    uint32_t hc = SS_INIT_A;
    for (int i = 0; i < 4; ++i) pool[i] = ss_hashmix(i < n ? entropy[i] : 0u, hc);
    for (int s = 0; s < 4; ++s)
        for (int d = 0; d < 4; ++d)
            if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], hc));
    // (entropy longer than the pool would be mixed in here; an integer seed < 2^64 never is)
}
// SeedSequence.generate_state(n_words, uint32)
PGX_NP_HD void seed_sequence_state32(const uint32_t pool[4], int n_words, uint32_t* out) {
    uint32_t hc = SS_INIT_B;
    for (int i = 0; i < n_words; ++i) {
        uint32_t v = pool[i & 3];
        v ^= hc;
        hc *= SS_MULT_B;
        v *= hc;
        v ^= v >> 16;
        out[i] = v;
    }
}

// ---- PCG64 ----------------------------------------------------------------------------------------------------------
PGX_NP_HD u128 pcg_mult() { return ((u128)0x2360ED051FC65DA4ull << 64) | (u128)0x4385DF649FCCF645ull; }
PGX_NP_HD void pcg_step(Pcg64& g) { g.state = g.state * pcg_mult() + g.inc; }

// np.random.default_rng(seed) / np.random.PCG64(seed)
PGX_NP_HD Pcg64 default_rng(uint64_t seed) {
    uint32_t pool[4], w[8];
    seed_sequence_pool(seed, pool);
    seed_sequence_state32(pool, 8, w);
    const uint64_t q0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32), q1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    const uint64_t q2 = (uint64_t)w[4] | ((uint64_t)w[5] << 32), q3 = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
    Pcg64 g;
    g.state = 0;
    g.inc = ((((u128)q2 << 64) | q3) << 1) | 1;
    pcg_step(g);
    g.state += ((u128)q0 << 64) | q1;
    pcg_step(g);
    g.has_uint32 = 0;
    g.uinteger = 0;
    return g;
}

PGX_NP_HD uint64_t next_uint64(Pcg64& g) {
    pcg_step(g);
    const uint64_t hi = (uint64_t)(g.state >> 64), lo = (uint64_t)g.state;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(g.state >> 122);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}
PGX_NP_HD uint32_t next_uint32(Pcg64& g) {
    if (g.has_uint32) {
        g.has_uint32 = 0;
        return g.uinteger;
    }
    const uint64_t v = next_uint64(g);
    g.has_uint32 = 1;
    g.uinteger = (uint32_t)(v >> 32);
    return (uint32_t)v;
}
// Generator.random()
PGX_NP_HD double next_double(Pcg64& g) { return (double)(next_uint64(g) >> 11) * (1.0 / 9007199254740992.0); }

// Generator.integers(0, n) (default int64 dtype, endpoint=False, scalar or array: the same draws), n >= 1;
// also Generator.choice(n) / choice(sequence of length n) without p
PGX_NP_HD uint64_t integers_below(Pcg64& g, uint64_t n) {
    const uint64_t rng = n - 1;
    if (rng == 0) return 0;
    if (rng <= 0xFFFFFFFFull) {
        if (rng == 0xFFFFFFFFull) return next_uint32(g);
        const uint32_t rng_excl = (uint32_t)rng + 1u;
        uint64_t m = (uint64_t)next_uint32(g) * rng_excl;
        uint32_t leftover = (uint32_t)m;
        if (leftover < rng_excl) {
            const uint32_t threshold = (0xFFFFFFFFu - (uint32_t)rng) % rng_excl;
            while (leftover < threshold) {
                m = (uint64_t)next_uint32(g) * rng_excl;
                leftover = (uint32_t)m;
            }
        }
        return m >> 32;
    }
    if (rng == 0xFFFFFFFFFFFFFFFFull) return next_uint64(g);
    const uint64_t rng_excl = rng + 1;
    u128 m = (u128)next_uint64(g) * rng_excl;
    uint64_t leftover = (uint64_t)m;
    if (leftover < rng_excl) {
        const uint64_t threshold = (0xFFFFFFFFFFFFFFFFull - rng) % rng_excl;
        while (leftover < threshold) {
            m = (u128)next_uint64(g) * rng_excl;
            leftover = (uint64_t)m;
        }
    }
    return (uint64_t)(m >> 64);
}

// `random_interval(max)`: uniform in [0, max] by masked rejection (what shuffle / permutation use)
PGX_NP_HD uint64_t random_interval(Pcg64& g, uint64_t max) {
    if (max == 0) return 0;
    uint64_t mask = max;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    uint64_t v;
    if (max <= 0xFFFFFFFFull) {
        while ((v = (next_uint32(g) & mask)) > max) {}
    } else {
        while ((v = (next_uint64(g) & mask)) > max) {}
    }
    return v;
}
// Generator.shuffle(x) / permutation(n) on a 1-d array or a list: x[i] <-> x[j], i = n-1 .. 1, j = random_interval(i)
template <typename T>
PGX_NP_HD void shuffle(Pcg64& g, T* x, int64_t n) {
    for (int64_t i = n - 1; i > 0; --i) {
        const int64_t j = (int64_t)random_interval(g, (uint64_t)i);
        const T t = x[i];
        x[i] = x[j];
        x[j] = t;
    }
}

// Generator.binomial(1, p): one draw, exactly as `random_binomial` does it for n = 1:
//   P = p <= 0.5 ? p : 1.0 - p;  Q = 1.0 - P;  qn = exp(1 * log(Q));  X = inversion(P, Q, qn);  result = p <= 0.5 ? X : 1 - X
// qn is a per-call constant: binomial1_qn() computes it on the host with libm (numpy calls the same exp/log) and the
// device gets it as an argument.
PGX_NP_HD int64_t binomial1_inversion(Pcg64& g, double p, double q, double qn) {
    const double np_ = 1.0 * p;
    double bound = np_ + 10.0 * __builtin_sqrt(np_ * q + 1.0);
    if (bound > 1.0) bound = 1.0;  // min(n, ...)
    int64_t X = 0;
    double px = qn;
    double U = next_double(g);
    while (U > px) {
        X++;
        if ((double)X > bound) {
            X = 0;
            px = qn;
            U = next_double(g);
        } else {
            U -= px;
            px = ((double)(1 - X + 1) * p * px) / ((double)X * q);
        }
    }
    return X;
}
PGX_NP_HD int64_t binomial1(Pcg64& g, double p, double qn) {
    if (p == 0.0) return 0;
    const double P = p <= 0.5 ? p : 1.0 - p;
    const double Q = 1.0 - P;
    const int64_t X = binomial1_inversion(g, P, Q, qn);
    return p <= 0.5 ? X : 1 - X;
}

}  // namespace pgxnp

// ---- instance generation the way upstream does it (recalled) -----------------------------------------------------------
// Upstream pogema/generator.py, as recalled (conf. medium -- the source is not mounted):
//   generate_obstacles:                    obstacles = default_rng(seed).binomial(1, density, (H, W))
//   generate_positions_and_targets_fast:   order = free cells, row-major;  default_rng(seed).shuffle(order);
//   placing:                               every cell is linked to the NEXT cell of its 4-connected component in `order`;
//                                          walking `order`, a cell with a link becomes a start, the linked cell its target
//                                          (and loses its own link); the first `num_agents` such pairs are the agents.
// One call builds one instance; `scratch` holds 4 * H * W words.  Returns 0, or 1 when fewer than `num_agents` pairs
// exist (upstream: OverflowError).  With `given_map` (H*W bytes, non-zero = obstacle) the obstacles are taken from it and
// only the positions are drawn (upstream with GridConfig.map set).  Host and device run the same code; the numpy arithmetic is bit-exact
// (tests/test_nprng.py), so with the recollection right this reproduces upstream's instances for a given seed.
namespace pgxnp {

PGX_NP_HD int generate_instance(uint64_t seed, int H, int W, int num_agents, double density, double qn,
                                const uint8_t* given_map, uint8_t* map,
                                int32_t* agent_xy, int32_t* target_xy, uint32_t* scratch) {
    const int cells = H * W;
    uint32_t* stack = scratch;               // flood-fill stack, later `colors`
    uint32_t* label = scratch + cells;
    uint32_t* order = scratch + 2 * cells;
    uint32_t* link = scratch + 3 * cells;
    constexpr uint32_t NONE = 0xFFFFFFFFu, OBST = 0xFFFFFFFEu;
    Pcg64 g = default_rng(seed);
    for (int c = 0; c < cells; ++c) {
        map[c] = given_map ? (uint8_t)(given_map[c] != 0) : (uint8_t)binomial1(g, density, qn);
        label[c] = map[c] ? OBST : NONE;
    }
    uint32_t ncomp = 0;
    for (int s = 0; s < cells; ++s) {
        if (label[s] != NONE) continue;
        int sp = 0;
        stack[sp++] = (uint32_t)s;
        label[s] = ncomp;
        while (sp) {
            const int c = (int)stack[--sp];
            const int x = c / W, y = c - x * W;
            if (x > 0 && label[c - W] == NONE) { label[c - W] = ncomp; stack[sp++] = (uint32_t)(c - W); }
            if (x + 1 < H && label[c + W] == NONE) { label[c + W] = ncomp; stack[sp++] = (uint32_t)(c + W); }
            if (y > 0 && label[c - 1] == NONE) { label[c - 1] = ncomp; stack[sp++] = (uint32_t)(c - 1); }
            if (y + 1 < W && label[c + 1] == NONE) { label[c + 1] = ncomp; stack[sp++] = (uint32_t)(c + 1); }
        }
        ++ncomp;
    }
    int n = 0;
    for (int c = 0; c < cells; ++c)
        if (!map[c]) order[n++] = (uint32_t)c;
    Pcg64 g2 = default_rng(seed);
    shuffle(g2, order, (int64_t)n);
    uint32_t* colors = stack;
    for (uint32_t k = 0; k < ncomp; ++k) colors[k] = NONE;
    for (int idx = n - 1; idx >= 0; --idx) {
        const uint32_t color = label[order[idx]];
        link[idx] = colors[color];
        colors[color] = (uint32_t)idx;
    }
    int placed = 0;
    for (int idx = 0; idx < n && placed < num_agents; ++idx) {
        const uint32_t nx = link[idx];
        if (nx == NONE) continue;
        agent_xy[2 * placed] = (int32_t)(order[idx] / (uint32_t)W);
        agent_xy[2 * placed + 1] = (int32_t)(order[idx] % (uint32_t)W);
        target_xy[2 * placed] = (int32_t)(order[nx] / (uint32_t)W);
        target_xy[2 * placed + 1] = (int32_t)(order[nx] % (uint32_t)W);
        link[nx] = NONE;
        ++placed;
    }
    return placed >= num_agents ? 0 : 1;
}

}  // namespace pgxnp
