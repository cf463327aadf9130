/*
 * pogema_amd.h -- C-ABI of the MI355X-native vectorized POGEMA step engine.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * is pure Python and exposes no FFI; the entry points below are what a ctypes binding for
 * `pogema/grid.py` + `pogema/envs.py` (reset()/step()) binds instead of the per-agent Python loops.
 * The mounted reference is a stub (/root/reference/README.md:3,5 -- "code is hosted elsewhere"), so
 * upstream files are cited by name only, never by line; see DESIGN.md section 2.
 *
 * Conventions
 *   - every function returns an int status: 0 = ok, negative = error (PGX_E_*); nothing throws
 *     across the ABI; pgx_last_error() returns a thread-local human-readable message.
 *   - all I/O buffers are owned by the caller.  Pointers documented "device" must be device
 *     pointers on the engine's device (e.g. torch tensor .data_ptr()); "host" are host pointers.
 *   - the engine owns its internal SoA state (obstacle bitmaps, agent/target xy, active masks,
 *     step counters) in HBM on its device.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All device work is
 *     enqueued asynchronously on it; no entry point synchronises unless documented.
 *   - one pgx_env per device shard; a handle is not thread-safe, independent handles may be driven
 *     from different host threads.
 *   - coordinates at the ABI are UNPADDED map coordinates, (x = row, y = column), int32 pairs.
 */
#ifndef POGEMA_AMD_H
#define POGEMA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGX_ABI_VERSION 6

/* error codes */
#define PGX_OK 0
#define PGX_E_INVALID (-1)   /* bad argument / config out of the supported range      */
#define PGX_E_HIP (-2)       /* a HIP runtime call failed (message has the HIP error)  */
#define PGX_E_NOMEM (-3)
#define PGX_E_STATE (-4)     /* call order violated (e.g. step before reset)           */
#define PGX_E_PLACEMENT (-5) /* generator could not place the requested agents         */

/* collision systems -- replaces `Pogema.move_agents` branches (upstream pogema/envs.py) */
#define PGX_COLLISION_PRIORITY 0
#define PGX_COLLISION_BLOCK_BOTH 1
#define PGX_COLLISION_SOFT 2

/* on_target modes -- replaces Pogema / PogemaLifeLong / PogemaCoopFinish */
#define PGX_ON_TARGET_FINISH 0
#define PGX_ON_TARGET_RESTART 1
#define PGX_ON_TARGET_NOTHING 2

/* Switches for the semantics the builder recalls with LOW confidence (docs/SPEC.md open questions Q1, Q2, Q4, Q7; the
 * reference source is not mounted, /root/reference/README.md:3,5).  Value 0 is ALWAYS the recalled-literal behaviour of
 * upstream -- quirks included: bit-exactness targets what upstream does, not what would be tidier -- and the default of
 * every host binding; the other value is the plausible alternative, so that pinning against the real package is a config
 * flip, not a kernel edit.  Every variant is implemented in the step kernel and in both oracles and covered by the
 * parity matrix. */
#define PGX_SOFT_LOWEST_INDEX_WINS 0 /* Q1: `soft`, several movers claim one free/vacated cell: the lowest index moves
                                        (literal `used_cells[cell].remove(agent)` + reverse-index loop)               */
#define PGX_SOFT_ALL_STAY 1          /* Q1 alternative: every claimant of a contested cell stays (textbook MAPF)       */
#define PGX_COOP_REWARD_ALL_SOLVED 0 /* Q4: on_target = NOTHING pays 1.0 to every agent iff ALL are on their goals    */
#define PGX_COOP_REWARD_PER_AGENT 1  /* Q4 alternative: 1.0 to each agent standing on its own goal in this step        */
#define PGX_SOFT_OCCUPANCY_INDEX_ORDER 0 /* Q2: the literal `Grid.move_without_checks` loop as recalled -- clear the old cell,
                                           set the new one, agent by agent in index order: an agent entering the cell a
                                           HIGHER-index agent is leaving stands there but is missing from the occupancy
                                           array (`Grid.positions`: the `agents` planes, pgx_get_state's occupancy) until
                                           its next turn in a later step re-sets it.  (ABI <= 4 numbered this 1.)        */
#define PGX_SOFT_OCCUPANCY_EXACT 1       /* Q2 alternative: after a `soft` step the occupancy array is exactly the set of
                                           visible agents' cells (the invariant every other collision system keeps)      */
#define PGX_BAD_ACTION_NOOP 0        /* Q7: an action outside 0..4 is a noop                                           */
#define PGX_BAD_ACTION_FLAG 1        /* Q7 alternative: still a noop on the device, but counted -- pgx_bad_action_count()
                                        lets the host raise the reference's IndexError                               */

/* Stream of the lifelong (on_target = RESTART) target draw.
 *   BUILD  the build's own counter-based stream (docs/SPEC.md S5), keyed by (seed, global env, agent, draw number)
 *   NUMPY  per-agent numpy generators as upstream `PogemaLifeLong._initialize_grid` sets them up (recalled, conf.
 *          medium): main = default_rng(env_seed); seeds = main.integers(2^31 - 1, size=num_agents);
 *          generator[a] = default_rng(seeds[a]); new target = component[generator[a].integers(0, len(component))]
 *          (= `rnd_generator.choice(component, 1)`), re-created at every reset of the env.  env_seed = cfg.seed + global
 *          env index.  The numpy arithmetic is exact (pgx_np_streams); what stays build-defined is the ORDER of a
 *          component's cells (row-major here; upstream: the order its BFS produced). */
#define PGX_LIFELONG_RNG_BUILD 0
#define PGX_LIFELONG_RNG_NUMPY 1

/* dtype of the `actions` buffer handed to pgx_step */
#define PGX_ACTION_I8 0
#define PGX_ACTION_I32 1
#define PGX_ACTION_I64 2

/* dtype of the observation buffer (pgx_config.obs_dtype).  F32 is the reference's dtype (gymnasium Box float32) and
 * the drop-in default; the others write the same 0/1 planes in a lighter format -- non-drop-in modes for callers whose
 * policy network does not want float32 anyway: U8 one byte per cell (4x lighter; the caller casts), BF16 / F16 two bytes
 * per cell (2x lighter, consumed directly by a mixed-precision network: no cast pass at all -- uint8 plus a cast to
 * bfloat16 on the consumer's side moves as many HBM bytes as float32 did).  0.0 and 1.0 are exact in every format. */
#define PGX_OBS_F32 0
#define PGX_OBS_U8 1
#define PGX_OBS_BF16 2
#define PGX_OBS_F16 3

/* hard limits of this build (upstream's GridConfig admits size 2..1024, obs_radius 1..128, num_agents >= 1 -- recalled,
 * pogema/grid_config.py; README.md "Limits"):
 *   PGX_MAX_SIDE        every map GridConfig admits.  Small maps stage both padded bitmaps of an environment in LDS; from 64 KB
 *                       of bitmaps (~500 x 500 cells) the large-map layout keeps the occupancy bitmap only (pgx_geometry.multi_wave = 2)
 *   PGX_MAX_OBS_RADIUS  a window row is ONE 32-bit mask (2r+1 <= 31): obs_radius 16..128 are refused
 *   PGX_MAX_AGENTS      one workgroup (<= 1024 lanes, one lane per agent) per environment: more agents are refused
 *   LDS                 what one environment keeps in a CU's 160 KB: 4 * PH * ceil(PW / 32) bytes per padded bitmap (PH, PW =
 *                       map + 2r) plus 24 bytes per lane of exchange arrays plus the row masks -- 2 bytes per window row
 *                       and (agent, plane) aliased over the bitmaps for windows up to 16 wide, 4 bytes each IN ADDITION for
 *                       wider ones: e.g. 1024 agents with obs_radius >= 8 do not fit; pgx_check_config says so up front */
#define PGX_MAX_OBS_RADIUS 15   /* window side 2r+1 <= 31 (one 32-bit row mask per window row) */
#define PGX_MAX_AGENTS 1024
#define PGX_MAX_SIDE 1024

/* Mirrors the step-relevant fields of the reference's `GridConfig` (upstream
 * pogema/grid_config.py) plus the batch geometry. */
typedef struct pgx_config {
    int32_t batch;             /* envs held by this handle (this device's shard)               */
    int32_t height, width;     /* unpadded map size (rectangular maps allowed)                 */
    int32_t num_agents;
    int32_t obs_radius;
    int32_t collision_system;  /* PGX_COLLISION_*                                              */
    int32_t on_target;         /* PGX_ON_TARGET_*                                              */
    int32_t max_episode_steps; /* MultiTimeLimit; <= 0 disables truncation        */
    int32_t auto_reset;        /* 1: an env whose agents are all terminated or truncated is   */
                               /*    reset to its stored initial state inside the same step    */
    int32_t obs_dtype;         /* PGX_OBS_* (0 = float32, the reference's dtype; 1 u8, 2 bf16, 3 f16) */
    uint64_t seed;             /* lifelong (restart) target stream seed                        */
    int64_t env_index_base;    /* global index of env 0 of this shard (keeps lifelong streams  */
                               /* independent of how the batch is sharded over devices)        */
    int32_t random_outside;    /* 0: cells beyond the border ring are FREE (`empty_outside=True`, the     */
                               /*    reference's default); 1: Bernoulli(outside_density) obstacles there  */
                               /*    (`empty_outside=False`; the stream is this build's own, keyed by      */
                               /*    seed, global env index and the env's generation counter)             */
    float outside_density;
    int32_t soft_vertex_rule;  /* PGX_SOFT_*        (0 = recalled literal algorithm)                    */
    int32_t coop_reward;       /* PGX_COOP_REWARD_* (0 = recalled)                                       */
    int32_t bad_action;        /* PGX_BAD_ACTION_*  (0 = noop)                                           */
    int32_t lifelong_rng;      /* PGX_LIFELONG_RNG_* (0 = the build's counter-based stream)              */
    int32_t soft_occupancy;    /* PGX_SOFT_OCCUPANCY_* (0 = the literal index-order loop, recalled)      */
    int32_t abi_version;       /* must be PGX_ABI_VERSION of the header the caller was COMPILED against: pgx_create   */
                               /* refuses any other value, so a caller built against an older header fails loudly      */
                               /* instead of getting renumbered semantics (ABI 5 swapped the two soft_occupancy        */
                               /* values: 0 became the index-order loop).  ABI <= 5 had `reserved0 = 0` here.          */
} pgx_config;

typedef struct pgx_env pgx_env; /* opaque */

/* ---- lifecycle -------------------------------------------------------------------------------- */
int pgx_abi_version(void);
const char* pgx_last_error(void);

/* Allocates the device-resident SoA state for cfg->batch environments on `device`.
 * Replaces: constructing `batch` reference env objects (upstream pogema/envs.py `_make_pogema`). */
int pgx_create(const pgx_config* cfg, int device, pgx_env** out);
int pgx_destroy(pgx_env* env);
/* Every check of pgx_create that needs no device -- argument ranges, the limits above, the LDS budget of the launch shape --
 * with the same status codes and pgx_last_error() text: lets a host wrapper refuse a configuration (upstream: pydantic's
 * ValidationError of `GridConfig`) before it touches a GPU.  Needs no HIP device. */
int pgx_check_config(const pgx_config* cfg);

/* Sizes of the caller-owned output buffers, in elements. */
int64_t pgx_obs_elems(const pgx_env* env);   /* batch * agents * 3 * (2r+1)^2   (obs_dtype elements) */
int64_t pgx_agent_elems(const pgx_env* env); /* batch * agents                            */

/* ---- reset -------------------------------------------------------------------------------------- */
/* Installs initial states.  Replaces `Grid.__init__` + `add_artificial_border` (upstream
 * pogema/grid.py) for explicitly given maps/positions.
 *   obstacles  device u8  [batch, height, width]   0 = FREE, 1 = OBSTACLE
 *   agent_xy   device i32 [batch, agents, 2]       unpadded (row, col)
 *   target_xy  device i32 [batch, agents, 2]
 * The state is also stored as the auto-reset state.  For on_target = RESTART this call additionally
 * builds the component tables on the device.  Asynchronous on `stream`. */
int pgx_reset_from_state(pgx_env* env, const uint8_t* obstacles, const int32_t* agent_xy,
                         const int32_t* target_xy, void* stream);

/* On-device reset: draws the instances on the GPU.  Replaces `Grid.__init__` with a random map: upstream
 * pogema/generator.py `generate_obstacles` (Bernoulli(density) obstacles), the BFS component labelling and
 * `generate_positions_and_targets_fast` (starts/targets on distinct free cells, each pair inside one
 * 4-connected component) -- SURVEY L1 / section 8f rank 2 -- plus, for on_target = RESTART, the component
 * tables `PogemaLifeLong` draws new targets from.  The random stream is this build's own counter-based
 * generator (numpy's PCG64 is not reproduced -- DESIGN.md); pgx_generate draws the SAME instances on the
 * host.  Env i of the shard draws instance (seed, env_index_base + i): seed and global env index are separate key
 * components (hashed one after the other), so batches reset with different seeds share no instance.
 *   shared_map  device u8 [height, width] or NULL   given map for every env (GridConfig.map): only
 *                                                   starts/targets are drawn, `density` is ignored
 *   env_mask    device u8 [batch] or NULL           NULL: every env, generation 0 (a pure function of
 *                                                   seed and env index).  Otherwise only the flagged envs
 *                                                   get a NEW instance (their generation counter advances);
 *                                                   their step counters and metric accumulators restart.
 * One kernel builds one env per workgroup, retries included; the call then synchronises `stream` once to read the
 * device's failure count: PGX_E_PLACEMENT if an env cannot be filled after `max_retries` re-draws (<= 0: 10). */
int pgx_reset_random(pgx_env* env, float density, uint64_t seed, const uint8_t* shared_map,
                     const uint8_t* env_mask, int32_t max_retries, void* stream);

/* Asynchronous form of the masked reset, for use right after pgx_step: NEW instances for the envs flagged in
 * `env_mask` (device u8 [batch], e.g. the `episode_done` buffer of pgx_set_metrics_buffers), then -- when `obs` is
 * not NULL -- the observations of exactly those envs are rewritten in `obs` (same buffer/dtype as pgx_step's).
 * This is the reference's auto-reset wrapper with `seed=None` (a fresh random instance per episode) for a whole
 * batch, without a host round trip: nothing here synchronises.  An env that cannot be filled within `max_retries`
 * attempts (<= 0: 3) keeps its previous instance and is counted; pgx_regenerate_failures() returns that count
 * (it synchronises `stream`).  Needs one scratch slot per env (9 bytes per cell and env, allocated on first use). */
int pgx_regenerate(pgx_env* env, const uint8_t* env_mask, float density, uint64_t seed, const uint8_t* shared_map,
                   int32_t max_retries, void* obs, void* stream);
int64_t pgx_regenerate_failures(pgx_env* env, void* stream);

/* The unpadded obstacle maps currently installed: device u8 [batch, height, width] (`Grid.get_obstacles`). */
int pgx_get_map(pgx_env* env, uint8_t* obstacles, void* stream);

/* ---- the hot path ------------------------------------------------------------------------------- */
/* One environment step for every env of the shard.  Replaces, per env, `Pogema.step` /
 * `PogemaLifeLong.step` / `PogemaCoopFinish.step` including `move_agents`, `Grid.move`, the
 * `_obs()` gather (`get_obstacles_for_agent`, `get_positions`, `get_square_target`) and
 * `MultiTimeLimit.step` (upstream pogema/envs.py, pogema/grid.py,
 * pogema/wrappers/multi_time_limit.py).
 *   actions      device [batch, agents] of action_dtype, values 0..4 (noop, up, down, left, right)
 *   obs          device f32 (or u8 / bf16 / f16, per pgx_config.obs_dtype) [batch, agents, 3, 2r+1, 2r+1]
 *                (obstacles, agents, target)                                               may be NULL
 *   rewards      device f32 [batch, agents]
 *   terminated   device u8  [batch, agents]
 *   truncated    device u8  [batch, agents]
 *   is_active    device u8  [batch, agents]   infos[i]['is_active'] after the step      may be NULL */
int pgx_step(pgx_env* env, const void* actions, int action_dtype, void* obs, float* rewards,
             uint8_t* terminated, uint8_t* truncated, uint8_t* is_active, void* stream);

/* How pgx_step's workgroups (one per environment, or per group of small environments) are shared out over the 8 XCDs:
 * shares[x] of them run on XCD x (host pointer, 8 entries).  Equal by default.  The XCDs do not get through their
 * observation streams equally fast (the odd ones lag 5-15 %), and with equal shares the fast ones
 * idle at the end of every launch: pgx_xcd_tune() runs the observation pass into `obs` and `obs_alt` in turn (the buffers
 * pgx_step will write; obs_alt may be NULL) a few times, reads when each XCD finished its share, shifts work towards the fast ones and keeps the shares that gave
 * the shortest launch (never worse than equal: equal is the first candidate).  Synchronises `stream`; nothing in the
 * engine's state changes.  us_equal / us_tuned (may be NULL): the pass with equal shares and with the kept ones.
 * PGX_XCD_WEIGHTS=w0,...,w7 sets the shares at pgx_create instead (diagnostic). */
/* Both assume ONE compute partition of 8 XCDs (SPX): on any other device (pgx_geometry.xcd_aware == 0) the mapping is the
 * identity, the shares stay equal and pgx_xcd_tune returns at once with 0 / 0. */
int pgx_xcd_shares(pgx_env* env, int32_t* shares);
int pgx_xcd_tune(pgx_env* env, void* obs, void* obs_alt, int32_t rounds, float* us_equal, float* us_tuned, void* stream);

/* The launch shape pgx_step (for_rollout = 0) or pgx_rollout (1) uses for this handle -- read-only, for tests and
 * benchmarks that must show they ran the SAME kernel variant (tests/test_fullsize_gpu.py asserts that its parity
 * engines and bench.py's engine agree for every BASELINE config) and for reading a profile: which template instance
 * (lanes_per_env, multi_wave, p16), how many waves per workgroup and environments per wave, which store flavour.
 * Chosen once in pgx_create from (batch, num_agents, map, obs_radius, obs_dtype) and the PGX_* tuning overrides. */
typedef struct pgx_geometry {
    int32_t lanes_per_env;  /* G: lanes of a wave one environment occupies (power of two; 64 when multi_wave)       */
    int32_t waves;          /* waves per workgroup                                                                   */
    int32_t envs_per_wave;  /* environments per single-wave workgroup (1 when multi_wave)                            */
    int32_t multi_wave;     /* 1: one environment per workgroup of `waves` waves (num_agents > 64, or helper waves);
                               2: the same in the LARGE-MAP layout (the two padded bitmaps of an environment exceed 64 KB, ~500 x 500
                               cells and up): only the occupancy bitmap lives in LDS, obstacles are read through the L2  */
    int32_t p16;            /* 1: window side <= 16, packed 16-bit row masks                                         */
    int32_t stagger;        /* cohort stagger of the single-wave kernel (0 = off)                                    */
    int32_t store_policy;   /* observation stores AS EXECUTED: 0 plain, 1 nontemporal, 2 sc1 write-through (the lighter */
                            /* formats' generic funnels know 0 and 1 only: a chosen 2 runs -- and is reported -- as 0)  */
    int32_t state_stores;   /* when the small per-step result stores are issued: 0 at once, 1 after the LDS barrier, 2 after the stream */
    int32_t grid;           /* workgroups launched                                                                   */
    int32_t lds_bytes;      /* dynamic LDS per workgroup                                                             */
    int32_t for_rollout;    /* echo of the argument                                                                  */
    int32_t xcd_aware;      /* 1: the device is one 8-XCD compute partition (SPX, 256 CUs): XCD-contiguous workgroup mapping,
                               tunable per-XCD shares, cohort stagger.  0: anything else (CPX/DPX/QPX partitions, other parts):
                               identity mapping, equal shares, no stagger, pgx_xcd_tune is a no-op -- same results        */
} pgx_geometry;
int pgx_get_geometry(const pgx_env* env, int32_t for_rollout, pgx_geometry* out);

/* K steps in ONE launch: the same as `steps` consecutive pgx_step calls with actions[t] -- bit for bit, state and outputs
 * -- for callers that have the actions up front (executing MAPF plans, scripted or random policies, replaying recorded
 * episodes): upstream, the `for t in range(K): env.step(actions[t])` loop around `Pogema.step`.  Every workgroup takes its
 * own environments through all K steps, so there is no launch boundary between steps and one wave's collision resolve
 * runs under the other waves' observation streams.  All pointers are device pointers.
 *   actions       [steps, batch, agents] of action_dtype, or NULL: the engine's uniform random policy -- action
 *                 (h >> 32) * 5 >> 32 of a splitmix64 chain over (policy_seed, cfg.env_index_base + env, agent,
 *                 policy_step0 + t), the same whatever the sharding; written to actions_out ([steps, batch, agents] i8)
 *                 when that is not NULL
 *   obs           [obs_slots, batch, agents, 3, 2r+1, 2r+1] f32 (u8 with PGX_OBS_U8) or NULL; step t writes slot
 *                 t % obs_slots: obs_slots = steps keeps the whole trajectory, 1 only the last observation.
 *                 obs_slot_stride: bytes from one slot to the next; 0 = dense.  (pgx_buffers_stride() for a ring made
 *                 of zone-spread buffers.)
 *   rewards       [steps, batch, agents] f32
 *   terminated, truncated   [steps, batch, agents] u8
 *   is_active     [steps, batch, agents] u8 or NULL
 *   episode_done  [steps, batch] u8 or NULL          1 where the env's episode finished in that step
 *   metrics       [steps, batch, 6] f32 or NULL      written only where episode_done is 1 (as pgx_set_metrics_buffers)
 * The buffers of pgx_set_metrics_buffers are not touched.  auto_reset works as in pgx_step; pgx_regenerate has no
 * place inside the launch (PGX_E_INVALID is never returned for it: the caller simply does not call it). */
typedef struct pgx_rollout_io {
    const void* actions;
    void* obs;
    float* rewards;
    uint8_t* terminated;
    uint8_t* truncated;
    uint8_t* is_active;
    uint8_t* episode_done;
    float* metrics;
    int32_t action_dtype;
    int32_t obs_slots;
    int64_t obs_slot_stride;
    uint64_t policy_seed;
    int64_t policy_step0;
    int8_t* actions_out;
} pgx_rollout_io;
int pgx_rollout(pgx_env* env, int32_t steps, const pgx_rollout_io* io, void* stream);

/* Overwrites the current targets of the agents flagged in `agent_mask` (device u8 [batch, agents]; NULL = all agents).
 *   target_xy  device i32 [batch, agents, 2]  unpadded (row, col); must be free cells inside the map (not checked)
 * `PogemaLifeLong` draws new targets from per-agent numpy generators (upstream pogema/envs.py `_generate_new_target`,
 * pogema/generator.py); the engine's own lifelong stream is a different one (docs/SPEC.md S5).  This call lets a caller
 * supply the targets instead -- a user-defined task generator, or the recorded target sequence of a reference rollout
 * (tests/test_golden_reference.py replays lifelong fixtures this way).  The stored initial targets (auto-reset state)
 * are not touched.  Asynchronous on `stream`. */
int pgx_set_targets(pgx_env* env, const int32_t* target_xy, const uint8_t* agent_mask, void* stream);

/* Number of out-of-range actions (outside 0..4) that ACTIVE agents submitted since the last call (bad_action =
 * PGX_BAD_ACTION_FLAG only; otherwise always 0).  Inactive agents' actions are never looked at, as in the reference's
 * `if self.grid.is_active[agent_idx]` guards.  Synchronises `stream`, then clears the counter.  The host side turns a
 * non-zero count into the reference's IndexError (`MOVES[action]`). */
int64_t pgx_bad_action_count(pgx_env* env, void* stream);

/* Episode metrics, fused into pgx_step.  Replaces the metric wrappers of upstream pogema/wrappers/metrics.py
 * (ISR / CSR / ep_length / SoC / makespan, their non-disappearing forms for on_target = NOTHING, and
 * avg_throughput for lifelong), which put `infos[0]['metrics']` on the step that ends an episode.
 * Registers caller-owned device buffers written by every following pgx_step (NULL disables either):
 *   metrics       f32 [batch, 6]  ISR, CSR, ep_length, SoC, makespan, avg_throughput -- a row is written only
 *                                 on the step in which that env's episode finishes
 *   episode_done  u8  [batch]     1 iff all agents of the env are terminated or truncated in this step */
#define PGX_NUM_METRICS 6
int pgx_set_metrics_buffers(pgx_env* env, float* metrics, uint8_t* episode_done);

/* Observation of the current state without stepping.  Replaces `PogemaBase._obs()` as called by
 * `reset()`. */
int pgx_observe(pgx_env* env, void* obs, void* stream);

/* Placement probe.  On MI355X the speed of the observation stream depends on WHERE the output buffer lives: equal
 * 2 MiB-aligned hipMalloc'd buffers fall into tiers of ~140 / ~144 / ~153 us per configs[2] step on one and the same
 * device (profiles/r1/placement_tiers.txt), presumably by how their physical pages spread over the HBM stacks.  This
 * call writes the current observations into `obs` `reps` times and returns the average duration, so that a caller
 * that owns a pool of candidate buffers can keep the well-placed ones (VecPogema(reuse_buffers=True) does).
 * Synchronises `stream`.  Nothing in the engine's state changes. */
int pgx_time_observe(pgx_env* env, void* obs, int32_t reps, float* microseconds, void* stream);
/* The same into `obs` and `obs_alt` in turn -- how a caller with two alternating output buffers writes.  Two buffers that
 * together exceed the 256 MiB Infinity Cache behave differently from one buffer that fits it. */
int pgx_time_observe_pair(pgx_env* env, void* obs, void* obs_alt, int32_t reps, float* microseconds, void* stream);

/* ---- zone-aware output buffers ------------------------------------------------------------------ */
/* Nothing in the reference corresponds to this: it is where the caller-owned observation buffers SHOULD live on an
 * MI355X.  Physical HBM falls into a few large zones (tens of GiB; three on the devices measured).  A store stream
 * confined to one zone sustains ~5.5 TB/s, the same stream with half of its bytes in another zone ~6.9 TB/s
 * (profiles/r2/placement_*.txt); a plain hipMalloc'd buffer is physically compact and therefore lies in one zone
 * unless it straddles a boundary by luck.  pgx_buffers_create returns `count` buffers of `bytes` bytes, each ONE
 * contiguous virtual range (HIP virtual-memory API) whose second half is backed by physical memory from another zone:
 * the allocator is walked there with temporary spacer allocations (at most `max_spacer_gib` GiB and 90 % of the free
 * memory, released before the call returns) and every candidate is verified by timing a store stream into the buffer.
 * Buffers below 128 MiB are returned without a walk.  The walk assumes that the allocator hands out memory roughly in
 * address order (true for a process that has not fragmented its HBM); when it finds nothing, the buffers are valid but
 * not spread and `spread` says so.  The call blocks the calling thread for ~0.1-2 s (it times kernels on a private
 * non-blocking stream; work the caller has in flight on the device perturbs the timings, so callers synchronise first).
 * WHILE IT RUNS, up to min(max_spacer_gib, 90 % of the free memory) of HBM is held: on a device shared with other
 * processes pass a budget that leaves them room (the Python host defaults to half of the free memory).
 *   max_spacer_gib <= 0  no search: the halves come from wherever the allocator is (still valid buffers)
 * Buffers are usable by any kernel / copy like hipMalloc'd memory and stay valid until pgx_buffers_destroy. */
typedef struct pgx_buffers pgx_buffers; /* opaque */
typedef struct pgx_buffers_info {
    int64_t bytes;        /* usable bytes per buffer                                                         */
    int32_t count;
    int32_t spread;       /* 1: the two halves of every buffer lie in different zones (verified by timing)   */
    int32_t candidates;   /* second-half candidates timed                                                    */
    int32_t reserved0;
    float same_zone_us;   /* probe stream (768 MiB) into two halves allocated back to back (same zone)       */
    float final_us;       /* probe stream with the second half taken where the buffers' second halves are      */
    double spacer_gib;    /* spacer memory held at the end of the search (released before returning)          */
    float buffer_gbs;     /* store-stream rate into the slowest buffer as returned (0 below 128 MiB: no walk --  */
                          /* such a stream is absorbed by the Infinity Cache and says nothing about placement)  */
    float reserved1;
} pgx_buffers_info;
int pgx_buffers_create(int device, size_t bytes, int count, double max_spacer_gib, pgx_buffers** out);
/* The same, with the first `skip_gib` of the walk passed over unprobed: for a caller whose own stream turned out slower
 * into the buffers of a first pool than its probe promised (the fast stretch was narrower than the buffers) and who
 * tries again further on (`info.spacer_gib` of the first pool + 16, say). */
int pgx_buffers_create_at(int device, size_t bytes, int count, double skip_gib, double max_spacer_gib, pgx_buffers** out);
void* pgx_buffers_ptr(pgx_buffers* pool, int index); /* device pointer of buffer `index`, NULL if out of range */
int64_t pgx_buffers_stride(pgx_buffers* pool);       /* all buffers lie in ONE virtual range: ptr(i) = ptr(0) + i * stride,
                                                        stride = bytes rounded up to 2 MiB (pgx_rollout_io.obs_slot_stride) */
int pgx_buffers_drop(pgx_buffers* pool, int index);  /* releases the memory of ONE buffer (its addresses stay reserved and
                                                        must not be touched again): a caller may ask for more buffers
                                                        than it needs, time its own stream into each and keep the best */
int pgx_buffers_get_info(pgx_buffers* pool, pgx_buffers_info* info);
int pgx_buffers_destroy(pgx_buffers* pool);          /* synchronises the device, then unmaps and frees        */
/* Address space (bytes) this process has reserved for pool buffers so far.  These ranges are never handed back to the
 * driver nor re-used (ROCm 7.2 keeps stale translations either way: DESIGN.md section 6, profiles/r3/vmm_va_remap_stale.txt),
 * so the figure grows by count x stride with every pool -- out of 128 TiB; a caller that builds pools in a loop can
 * watch it here.  The probe chunks of the zone walk are plain hipMalloc memory and do not count. */
int64_t pgx_buffers_va_reserved(void);

/* ---- numpy-compatible random primitives ----------------------------------------------------------- */
/* Upstream POGEMA draws everything random from numpy Generators (np.random.default_rng(seed): SeedSequence -> PCG64 ->
 * integers / choice / shuffle / binomial / random; pogema/generator.py and the per-agent generators of PogemaLifeLong).
 * The engine's own generator and lifelong streams are counter-based and different (docs/SPEC.md S5, S6); these entry
 * points provide the numpy PRIMITIVES bit for bit -- pinned against numpy itself (tests/golden/numpy_rng_vectors.npz,
 * tests/test_nprng*.py) -- so that a numpy-stream mode can follow the upstream call sequence once the source is at hand
 * (pgx_config.lifelong_rng = PGX_LIFELONG_RNG_NUMPY already uses them for the lifelong target draw).
 * `streams` independent generators default_rng(seeds[s]) are each advanced `draws` times:
 *   PGX_NP_UINT64       raw 64-bit outputs (bit_generator.random_raw)          out u64 [streams, draws]
 *   PGX_NP_RANDOM       Generator.random()                                     out f64 [streams, draws]
 *   PGX_NP_INTEGERS     Generator.integers(0, n) = Generator.choice(n)         out i64 [streams, draws]
 *   PGX_NP_BINOMIAL1    Generator.binomial(1, p)  (generate_obstacles)         out i64 [streams, draws]
 *   PGX_NP_PERMUTATION  Generator.permutation(draws) = shuffle(arange(draws))  out i64 [streams, draws]
 * pgx_np_streams: device pointers, one GPU thread per stream, asynchronous on `stream`; pgx_np_streams_host: the same
 * arithmetic on the host (host pointers). */
#define PGX_NP_UINT64 0
#define PGX_NP_RANDOM 1
#define PGX_NP_INTEGERS 2
#define PGX_NP_BINOMIAL1 3
#define PGX_NP_PERMUTATION 4
int pgx_np_streams(const uint64_t* seeds, int64_t streams, int32_t op, uint64_t n, double p, int64_t draws, void* out,
                   void* stream);
int pgx_np_streams_host(const uint64_t* seeds, int64_t streams, int32_t op, uint64_t n, double p, int64_t draws, void* out);

/* Instances drawn the way upstream draws them (pogema/generator.py `generate_obstacles` +
 * `generate_positions_and_targets_fast` / `placing`, RECALLED, conf. medium): obstacles =
 * default_rng(seed).binomial(1, density, (H, W)); the free cells, row-major, shuffled with default_rng(seed); every cell
 * linked to the next cell of its 4-connected component in that order; walking the order, a linked cell becomes a start
 * and its link the target; the first `num_agents` pairs are the agents.  Env b uses seeds[b].  With the recollection
 * right these are upstream's instances for that seed; the numpy arithmetic itself is exact.  Outputs are in the format
 * pgx_reset_from_state takes.  `given_map` (u8 [H*W], non-zero = obstacle; may be NULL) replaces the obstacle draw for
 * every env, as GridConfig.map does upstream.  status[b] = 1 when env b has fewer than `num_agents` pairs (upstream: OverflowError).
 *   pgx_np_generate       device pointers; one GPU thread per env; scratch u32 [batch * 4 * H * W]; asynchronous
 *   pgx_np_generate_host  host pointers; scratch u32 [4 * H * W] */
int pgx_np_generate(const uint64_t* seeds, int32_t batch, int32_t height, int32_t width, int32_t num_agents, double density,
                    const uint8_t* given_map, uint8_t* obstacles, int32_t* agent_xy, int32_t* target_xy, uint32_t* scratch, int32_t* status, void* stream);
int pgx_np_generate_host(const uint64_t* seeds, int32_t batch, int32_t height, int32_t width, int32_t num_agents, double density,
                         const uint8_t* given_map, uint8_t* obstacles, int32_t* agent_xy, int32_t* target_xy, uint32_t* scratch, int32_t* status);

/* ---- state export ------------------------------------------------------------------------------- */
/* Replaces `Grid.get_agents_xy` / `get_targets_xy` / `is_active` / the occupancy array (`positions`).
 * Any pointer may be NULL.  All device pointers.
 *   agent_xy, target_xy  i32 [batch, agents, 2] unpadded
 *   is_active            u8  [batch, agents]
 *   elapsed              i32 [batch]                        steps since the env's last reset
 *   occupancy            u8  [batch, height+2r, width+2r]   padded occupancy array */
int pgx_get_state(pgx_env* env, int32_t* agent_xy, int32_t* target_xy, uint8_t* is_active,
                  int32_t* elapsed, uint8_t* occupancy, void* stream);

/* ---- snapshot / restore -------------------------------------------------------------------------- */
/* The complete engine state (maps, bitmaps, agent/target cells, initial state, active flags, step counters, metric
 * accumulators, generation counters, lifelong tables and draw counters) as one opaque device blob of
 * pgx_snapshot_bytes() bytes.  Replaces `PersistentWrapper`'s per-step state history / `step_back` (upstream
 * pogema/wrappers/persistence.py) and gives checkpoint-resume: a loaded snapshot continues bit-identically.
 * The blob is only valid for a handle of the same configuration; the 64-byte header records ABI version, batch,
 * agents, map height/width, obs_radius, on_target, collision_system, max_episode_steps and the total byte count, and
 * pgx_load_snapshot refuses (PGX_E_INVALID) a blob whose header differs from the loading handle's in any of them.
 * Both calls synchronise `stream` once for the header; the payload copies are asynchronous. */
int64_t pgx_snapshot_bytes(pgx_env* env);
int pgx_save_snapshot(pgx_env* env, void* blob, void* stream);
int pgx_load_snapshot(pgx_env* env, const void* blob, void* stream);

/* ---- host-side synthetic map generator ------------------------------------------------------------ */
/* Fills host buffers with `batch` random solvable instances: Bernoulli(density) obstacles, starts and
 * targets on distinct free cells with each start/target pair in one 4-connected component.
 * Plays the role of upstream pogema/generator.py at reset; the random stream is this
 * build's own (numpy's PCG64 sequence is not reproduced -- DESIGN.md).  Env i draws instance
 * (seed0, env_index_base + i) -- the same one pgx_reset_random draws for that seed and global index.
 *   obstacles host u8 [batch, height, width]; agent_xy / target_xy host i32 [batch, agents, 2]
 * nthreads <= 0 picks the number of online cores.  Returns PGX_E_PLACEMENT if an env cannot be
 * filled after `max_retries` re-draws. */
int pgx_generate(int32_t batch, int32_t height, int32_t width, int32_t num_agents, float density,
                 uint64_t seed0, int64_t env_index_base, int32_t max_retries, int32_t nthreads,
                 uint8_t* obstacles, int32_t* agent_xy, int32_t* target_xy);

/* Same placement on GIVEN obstacle maps (custom `GridConfig.map`): only starts/targets are drawn.
 *   obstacles host u8 [batch, height, width], or [height, width] when shared_map != 0. */
int pgx_place_agents(int32_t batch, int32_t height, int32_t width, int32_t num_agents, uint64_t seed0,
                     int64_t env_index_base, int32_t max_retries, int32_t nthreads, const uint8_t* obstacles,
                     int32_t shared_map, int32_t* agent_xy, int32_t* target_xy);

#ifdef __cplusplus
}
#endif
#endif /* POGEMA_AMD_H */
