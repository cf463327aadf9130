// pgx_internal.h -- shared between pgx_kernels.hip (device code + launchers) and pgx_api.cpp (C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pgx {

enum { MODE_STEP = 0, MODE_OBSERVE = 1 };
enum { COLLISION_PRIORITY = 0, COLLISION_BLOCK_BOTH = 1, COLLISION_SOFT = 2 };
enum { ON_TARGET_FINISH = 0, ON_TARGET_RESTART = 1, ON_TARGET_NOTHING = 2 };
// the per-agent `active` byte: bit 0 = `Grid.is_active`; bit 1 = the agent stands on its cell but is MISSING from the
// occupancy array (`Grid.positions`) -- the quirk of the literal `move_without_checks` loop (docs/SPEC.md Q2), see step_body
enum : uint32_t { ACTIVE_BIT = 1u, ACTIVE_GHOST = 2u };

// ---- instance generator "GEN v2" constants shared by the host generator and the device kernels ----
constexpr uint64_t GEN_TAG_OBST = 0x4F42535400000000ull;   // 'OBST'
constexpr uint64_t GEN_TAG_PLACE = 0x504C414300000000ull;  // 'PLAC'
inline uint32_t gen_density_threshold(float density) {      // obstacle <=> 24 hash bits < thr
    double t = (double)density * 16777216.0 + 0.5;
    if (t < 0.0) t = 0.0;
    if (t > 16777216.0) t = 16777216.0;
    return (uint32_t)t;
}
__host__ __device__ inline uint32_t gen_candidate_budget(uint32_t cells) { return 32u * cells + 64u; }

// `empty_outside=False`: obstacles beyond the border ring, a pure function of (seed, global env, generation, cell)
constexpr uint64_t GEN_TAG_OUTSIDE = 0x4F55545300000000ull;  // 'OUTS'
struct OutsideParams {
    int32_t enabled;
    uint32_t thr;            // obstacle <=> 24 hash bits < thr
    uint64_t seed;
    int64_t env_index_base;
    const uint32_t* epoch;   // [B] generation counters
};
__host__ __device__ inline uint64_t gen_sm64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// is padded cell (x, y) beyond the ring? (the ring itself and the map interior are never "outside")
__host__ __device__ inline bool gen_is_outside(int x, int y, int PH, int PW, int r) {
    return x < r - 1 || x > PH - r || y < r - 1 || y > PW - r;
}
__host__ __device__ inline uint64_t gen_outside_hash(uint64_t seed, uint64_t env_global, uint32_t epoch) {
    return gen_sm64(gen_sm64(gen_sm64(seed) ^ env_global) ^ (GEN_TAG_OUTSIDE | epoch));
}
__host__ __device__ inline uint32_t gen_outside_bit(uint64_t h, int x, int y, int PW, uint32_t thr) {
    return (gen_sm64(h ^ (uint64_t)(x * PW + y)) >> 40) < thr ? 1u : 0u;
}

// One numpy Generator (PCG64) per agent, lifelong_rng = NUMPY: {state hi, state lo, inc hi, inc lo, has_uint32 << 32 | uinteger}
struct NpGen {
    uint64_t w[5];
};

// Kernel argument block of the step kernel (passed by value: lands in SGPRs / kernarg segment).
struct StepParams {
    // geometry
    int32_t batch, num_agents, r;
    int32_t wpr;       // 32-bit words per padded bitmap row
    int32_t bm_words;  // words per env bitmap = (H + 2r) * wpr
    int32_t map_w;     // unpadded width
    int32_t map_cells; // unpadded H * W
    uint32_t w_magic;  // ceil(2^32 / (2r+1)): exact division of flat window offsets by the window side
    uint32_t a_magic;  // ceil(2^32 / num_agents): agent slot -> environment of the wave (slots < 4096)
    // behaviour
    int32_t mode, collision, on_target, max_steps, auto_reset, action_dtype;
    int32_t epw;       // environments per wave (single-wave blocks, num_agents <= 64)
    int32_t obs_u8;    // 1: `obs` is uint8 (one byte per cell) instead of float32
    uint32_t obs_one;  // != 0: `obs` is a 16-bit float format and this is its bit pattern of 1.0 (bfloat16 0x3F80, float16 0x3C00)
    int32_t stagger;   // cohort stagger of the single-wave kernel (StepGeometry::stagger)
    int32_t store_policy;  // observation stores: 0 plain, 1 nontemporal, 2 sc1 write-through (StepGeometry::store_policy)
    int32_t state_stores;  // when the small per-step result stores are issued: 0 at once, 1 after the LDS barrier, 2 after the stream
    int32_t soft_rule;    // PGX_SOFT_*  (docs/SPEC.md Q1)
    int32_t soft_occupancy;  // PGX_SOFT_OCCUPANCY_* (Q2; 0 = the literal index-order loop)
    int32_t coop_reward;  // PGX_COOP_REWARD_* (Q4)
    int32_t bad_action;   // PGX_BAD_ACTION_* (Q7)
    int32_t xcd_n[8];     // workgroups (= environment slices) given to each XCD, proportional to its measured store rate
    int32_t xcd_base[8];  // first slice of each XCD (prefix sums of xcd_n): the slices of one XCD stay contiguous
    uint32_t flags;    // diagnostic switches (PGX_FLAGS env var at pgx_create): bit1 generic row path, bit2 time stamps, bit3 identity block mapping
    uint64_t seed;
    int64_t env_index_base;
    // SoA state in HBM
    const uint32_t* obst;  // [B][bm_words]   padded obstacle bitmap, 1 bit per cell
    uint32_t* pos;         // [B][A]          (x << 16) | y, padded coordinates
    uint32_t* tgt;         // [B][A]
    uint8_t* active;       // [B][A]          ACTIVE_BIT | ACTIVE_GHOST
    int32_t* elapsed;      // [B]
    const uint32_t* pos0;  // [B][A]          auto-reset state
    const uint32_t* tgt0;  // [B][A]
    // lifelong (on_target = restart) tables, unpadded cell index = x * map_w + y
    const uint32_t* comp_begin;  // [B][H*W]  offset of the cell's component inside comp_cells
    const uint32_t* comp_len;    // [B][H*W]  size of the cell's component
    const uint32_t* comp_cells;  // [B][H*W]  unpadded packed cells grouped by component, row-major inside
    uint32_t* tcount;            // [B][A]    targets drawn so far
    NpGen* np_state;             // [B][A]    lifelong_rng = NUMPY: the agents' generators (null otherwise)
    const NpGen* np_state0;      // [B][A]    ... as they are right after a reset of the env
    // metric accumulators (pogema/wrappers/metrics.py): {agents solved, sum of solve steps, max solve step, lifelong goals}
    int4* macc;                  // [B]
    float* metrics_out;          // [B][6] ISR, CSR, ep_length, SoC, makespan, avg_throughput (caller-owned, may be null)
    uint8_t* episode_done;       // [B]    1 when the env's episode finished in this step            (may be null)
    uint32_t* bad_count;         // [1]    out-of-range actions of active agents (bad_action = FLAG)
    // I/O (caller-owned device buffers)
    const void* actions;
    float* obs;
    float* rewards;
    uint8_t* terminated;
    uint8_t* truncated;
    uint8_t* act_out;
    const uint8_t* only;      // MODE_OBSERVE: write only workgroups holding a flagged env (may be null)
    unsigned long long* dbg;  // diagnostic (PGX_FLAGS bit2): per-workgroup {start, resolve done, first store, end} clocks
};

// pgx_rollout: what changes from one step of the launch to the next
struct RolloutParams {
    int32_t steps;
    int32_t obs_slots;        // step t writes observation slot t % obs_slots
    int32_t resident_bitmap;  // 1: the obstacle bitmap is staged once and keeps its own LDS region (StepGeometry::resident_bitmap)
    int32_t reserved0;
    int64_t actions_stride;   // bytes between the action tensors of consecutive steps
    int64_t agents_stride;    // batch * num_agents: elements between per-agent outputs of consecutive steps
    int64_t envs_stride;      // batch: elements between per-env outputs of consecutive steps
    int64_t obs_stride;       // bytes between observation slots
    uint64_t policy_seed;     // actions == NULL: uniform random policy keyed by (policy_seed, global env, agent, policy_step0 + t)
    int64_t policy_step0;
    int8_t* actions_out;      // [steps, batch, agents] the policy's actions (may be null)
};

// How one configuration maps onto the step kernel (pgx_kernels.hip: step_geometry()).
struct StepGeometry {
    int G;            // lanes per environment group (power of two, 64 when multi_wave)
    int waves;        // waves per workgroup: 1, ceil(A / 64) when num_agents > 64, or 2..8 helper waves (small launches)
    int epw;          // environments per wave (1 when multi_wave)
    bool multi_wave;  // num_agents > 64: one environment per workgroup
    bool p16;         // window side <= 16: packed 16-bit row masks aliased over the LDS state
    int stagger;      // > 0: odd wave slots sleep this many x 8128 cycles after issuing their loads
    int store_policy; // observation store flavour (pgx_kernels.hip: store_obs16)
    int state_stores; // when the per-step result stores are issued (pgx_kernels.hip: emit_state)
    bool big;         // large-map layout: one env per workgroup, occupancy bitmap only in LDS, obstacles read through the L2
    bool pc;          // rollout launch shape only: resolver / streamer pair of waves per environment group (step_body, PC)
    bool resident_bitmap;  // rollout launch shape only: obstacle bitmap staged once per launch, LDS = bitmap + max(rest, rows)
    size_t lds_bytes;
    // shares of the launch's workgroups per XCD (xcd_partition; equal until pgx_xcd_tune or PGX_XCD_WEIGHTS)
    int grid;             // workgroups to launch: 8 * the largest share
    int32_t xcd_n[8], xcd_base[8];
};
StepGeometry step_geometry(int batch, int A, int bmw, int W, bool allow_p16, int epw_override, int obs_elem_bytes,
                           int waves_override, bool for_rollout = false);
hipError_t prepare_step(const StepGeometry& g, const StepGeometry& roll);
hipError_t launch_step(const StepParams& p, const StepGeometry& g, hipStream_t stream);
// splits `blocks` workgroups over the XCDs by `w`; returns the grid size (8 * the largest share)
int xcd_partition(int blocks, const float w[8], int32_t n[8], int32_t base[8]);
hipError_t launch_rollout(const StepParams& p, const RolloutParams& rp, const StepGeometry& g, hipStream_t stream);

// `only` (device u8 [batch], may be null): pack just the flagged environments
hipError_t launch_pack_obstacles(const uint8_t* obstacles, const uint8_t* only, uint32_t* bm, int batch, int H, int Wd,
                                 int r, int wpr, int bmw, const OutsideParams& outside, hipStream_t stream);

// ---- on-device reset (pgx_reset.hip) ----------------------------------------------------------------
// Kernel argument block of reset_env_kernel: one workgroup per environment of [env_begin, env_begin + env_count),
// scratch slot = workgroup index.
struct ResetParams {
    int32_t env_begin, env_count;
    int32_t H, Wd, A, r, wpr, bmw;
    int32_t lifelong;      // build the component tables (on_target = restart)
    int32_t given_state;   // map already installed (pgx_reset_from_state): components + tables only
    int32_t max_retries;
    uint32_t thr;          // obstacle <=> 24 hash bits < thr
    uint64_t gen_seed;     // env i of the shard draws instance (gen_seed, env_index_base + i)
    int64_t env_index_base;
    const uint8_t* shared_map;  // [H*W] given map for every env, or null
    uint8_t* todo;              // [B] in: envs to build; cleared on success
    const uint32_t* epoch;      // [B] generation counters
    uint8_t* scratch_map;       // [slots][H*W] draft maps
    uint32_t* labels;           // [slots][H*W]
    uint32_t* pending;          // [slots][H*W]
    uint8_t* map_u8;            // [B][H*W] installed maps
    uint32_t* obst_bm;          // [B][bmw] padded bitmaps
    uint32_t *pos, *tgt, *pos0, *tgt0;
    uint8_t* active;
    uint32_t* tcount;           // may be null
    NpGen* np_state;            // may be null (lifelong_rng = NUMPY)
    const NpGen* np_state0;
    int32_t* elapsed;
    int4* macc;
    uint32_t *comp_begin, *comp_len, *comp_cells;  // lifelong only
    uint32_t* fail_count;       // envs that could not be filled
    OutsideParams outside;      // `empty_outside=False`
};
hipError_t launch_reset_begin(const uint8_t* mask, uint8_t* todo, uint8_t* regen, uint32_t* epoch, int batch,
                              hipStream_t s);
hipError_t launch_reset_env(const ResetParams& p, hipStream_t s);
hipError_t launch_pack_agents(const int32_t* agent_xy, const int32_t* target_xy, uint32_t* pos, uint32_t* tgt,
                              uint32_t* pos0, uint32_t* tgt0, uint8_t* active, uint32_t* tcount, size_t n,
                              int r, hipStream_t stream, NpGen* np_state = nullptr, const NpGen* np_state0 = nullptr);
// np_state0[env][a] = generator of agent a of global env (env_index_base + env) right after a reset (lifelong_rng = NUMPY)
hipError_t launch_init_np_lifelong(NpGen* np_state0, uint64_t seed, int64_t env_index_base, int batch, int A, hipStream_t stream);
hipError_t launch_zero_i32(int32_t* v, size_t n, hipStream_t stream);
hipError_t launch_set_targets(const int32_t* target_xy, const uint8_t* mask, uint32_t* tgt, size_t n, int r,
                              hipStream_t stream);
hipError_t launch_unpack_state(const uint32_t* pos, const uint32_t* tgt, const uint8_t* active, int32_t* agent_xy,
                               int32_t* target_xy, uint8_t* act_out, size_t n, int r, hipStream_t stream);
hipError_t launch_occupancy(const uint32_t* pos, const uint8_t* active, uint8_t* occ, size_t n, int A, int PH,
                            int PW, hipStream_t stream);

}  // namespace pgx
