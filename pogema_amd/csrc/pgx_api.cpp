// pgx_api.cpp -- C-ABI of the engine (include/pogema_amd.h): handle management, reset, step launch,
// state export and the host-side synthetic map generator.  Compiled with hipcc together with
// pgx_kernels.hip into libpogema_amd.so.  There is NO CPU fallback anywhere in this file: every
// compute entry point enqueues HIP kernels and fails with PGX_E_HIP when no device is usable.
#include "../../include/pogema_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "pgx_internal.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define PGX_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(PGX_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            changed = (err == hipSuccess);
        }
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

struct pgx_env {
    pgx_config cfg{};
    int device = 0;
    pgx::StepGeometry geo{};
    int W = 0, PH = 0, PW = 0, wpr = 0, bmw = 0;
    bool has_state = false;
    uint32_t flags = 0;
    // device state
    uint32_t* obst = nullptr;
    uint32_t *pos = nullptr, *tgt = nullptr, *pos0 = nullptr, *tgt0 = nullptr;
    uint8_t* active = nullptr;
    int32_t* elapsed = nullptr;
    int4* macc = nullptr;
    float* metrics_out = nullptr;   // caller-owned (pgx_set_metrics_buffers)
    uint8_t* episode_done = nullptr; // caller-owned
    unsigned long long* dbg = nullptr;
    size_t dbg_elems = 0;
    uint32_t *comp_begin = nullptr, *comp_len = nullptr, *comp_cells = nullptr, *tcount = nullptr;
};

// ================================================================================================
extern "C" {

int pgx_abi_version(void) { return PGX_ABI_VERSION; }

const char* pgx_last_error(void) { return g_err.c_str(); }

int pgx_create(const pgx_config* cfg, int device, pgx_env** out) {
    if (!cfg || !out) return fail(PGX_E_INVALID, "pgx_create: null argument");
    *out = nullptr;
    if (cfg->batch < 1) return fail(PGX_E_INVALID, "batch must be >= 1 (got %d)", cfg->batch);
    if (cfg->height < 1 || cfg->width < 1 || cfg->height > PGX_MAX_SIDE || cfg->width > PGX_MAX_SIDE)
        return fail(PGX_E_INVALID, "map size %dx%d outside [1, %d]", cfg->height, cfg->width, PGX_MAX_SIDE);
    if (cfg->num_agents < 1 || cfg->num_agents > PGX_MAX_AGENTS)
        return fail(PGX_E_INVALID, "num_agents %d outside [1, %d]", cfg->num_agents, PGX_MAX_AGENTS);
    if (cfg->obs_radius < 1 || cfg->obs_radius > PGX_MAX_OBS_RADIUS)
        return fail(PGX_E_INVALID, "obs_radius %d outside [1, %d] supported by this build", cfg->obs_radius,
                    PGX_MAX_OBS_RADIUS);
    if (cfg->collision_system < 0 || cfg->collision_system > 2)
        return fail(PGX_E_INVALID, "unknown collision_system %d", cfg->collision_system);
    if (cfg->on_target < 0 || cfg->on_target > 2) return fail(PGX_E_INVALID, "unknown on_target %d", cfg->on_target);
    if ((int64_t)cfg->num_agents > (int64_t)cfg->height * cfg->width)
        return fail(PGX_E_INVALID, "more agents than cells");

    pgx_env* e = new (std::nothrow) pgx_env();
    if (!e) return fail(PGX_E_NOMEM, "out of host memory");
    e->cfg = *cfg;
    e->device = device;
    const int A = cfg->num_agents, r = cfg->obs_radius;
    e->W = 2 * r + 1;
    e->PH = cfg->height + 2 * r;
    e->PW = cfg->width + 2 * r;
    e->wpr = (e->PW + 31) / 32;
    e->bmw = e->PH * e->wpr;
    if (const char* f = getenv("PGX_FLAGS")) e->flags = (uint32_t)strtoul(f, nullptr, 0);
    int epw_override = 0;  // PGX_EPW: tuning/diagnostic override of the environments-per-wave heuristic
    if (const char* f = getenv("PGX_EPW")) epw_override = atoi(f);
    // PGX_FLAGS bit1: force the generic (32-bit row mask) observation path
    e->geo = pgx::step_geometry(cfg->batch, A, e->bmw, e->W, !(e->flags & 2u), epw_override);
    if (e->geo.lds_bytes > 160 * 1024) {
        const size_t need = e->geo.lds_bytes;
        delete e;
        return fail(PGX_E_INVALID,
                    "configuration needs %zu bytes of LDS per workgroup (> 163840): padded bitmaps of %dx%d "
                    "cells plus %d agents' row masks do not fit one CU",
                    need, cfg->height + 2 * r, cfg->width + 2 * r, A);
    }

    DeviceGuard guard(device);
    if (guard.err != hipSuccess) {
        delete e;
        return fail(PGX_E_HIP, "cannot select HIP device %d: %s", device, hipGetErrorString(guard.err));
    }
    const size_t B = (size_t)cfg->batch, BA = B * A;
    hipError_t err = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) {
        if (err == hipSuccess) err = hipMalloc(p, bytes);
    };
    alloc((void**)&e->obst, B * e->bmw * sizeof(uint32_t));
    alloc((void**)&e->pos, BA * sizeof(uint32_t));
    alloc((void**)&e->tgt, BA * sizeof(uint32_t));
    alloc((void**)&e->pos0, BA * sizeof(uint32_t));
    alloc((void**)&e->tgt0, BA * sizeof(uint32_t));
    alloc((void**)&e->active, BA);
    alloc((void**)&e->elapsed, B * sizeof(int32_t));
    alloc((void**)&e->macc, B * sizeof(int4));
    if (cfg->on_target == PGX_ON_TARGET_RESTART) {
        const size_t cells = B * (size_t)cfg->height * cfg->width;
        alloc((void**)&e->comp_begin, cells * sizeof(uint32_t));
        alloc((void**)&e->comp_len, cells * sizeof(uint32_t));
        alloc((void**)&e->comp_cells, cells * sizeof(uint32_t));
        alloc((void**)&e->tcount, BA * sizeof(uint32_t));
    }
    if (err != hipSuccess) {
        const char* msg = hipGetErrorString(err);
        pgx_destroy(e);
        return fail(err == hipErrorOutOfMemory ? PGX_E_NOMEM : PGX_E_HIP, "hipMalloc failed: %s", msg);
    }
    if ((err = pgx::prepare_step(e->geo)) != hipSuccess) {
        const char* msg = hipGetErrorString(err);
        const size_t need = e->geo.lds_bytes;
        pgx_destroy(e);
        return fail(PGX_E_HIP, "cannot configure the step kernel (%zu bytes of LDS): %s", need, msg);
    }
    if (e->flags & 4u) {  // diagnostic time stamps, one record per workgroup
        e->dbg_elems = (size_t)cfg->batch * 4;
        if (hipMalloc((void**)&e->dbg, e->dbg_elems * sizeof(unsigned long long)) != hipSuccess) e->dbg = nullptr;
        else (void)hipMemset(e->dbg, 0, e->dbg_elems * sizeof(unsigned long long));
    }
    *out = e;
    return PGX_OK;
}

int pgx_destroy(pgx_env* e) {
    if (!e) return PGX_OK;
    DeviceGuard guard(e->device);
    void* ptrs[] = {e->obst,   e->pos,     e->tgt,        e->pos0,     e->tgt0,       e->active,
                    e->elapsed, e->comp_begin, e->comp_len, e->comp_cells, e->tcount, e->dbg, e->macc};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete e;
    return PGX_OK;
}

int64_t pgx_obs_elems(const pgx_env* e) {
    if (!e) return 0;
    return (int64_t)e->cfg.batch * e->cfg.num_agents * 3 * e->W * e->W;
}

int64_t pgx_agent_elems(const pgx_env* e) {
    if (!e) return 0;
    return (int64_t)e->cfg.batch * e->cfg.num_agents;
}

// ---- connected components on the host (lifelong mode only; reset path, not the hot path) ----------
static void label_components_host(const uint8_t* obst, int H, int Wd, uint32_t* comp_begin, uint32_t* comp_len,
                                  uint32_t* comp_cells, std::vector<int32_t>& label, std::vector<int32_t>& stack) {
    const int cells = H * Wd;
    label.assign(cells, -1);
    std::vector<uint32_t> count;
    for (int s = 0; s < cells; ++s) {
        if (obst[s] != 0 || label[s] >= 0) continue;
        const int cid = (int)count.size();
        count.push_back(0);
        label[s] = cid;
        stack.clear();
        stack.push_back(s);
        while (!stack.empty()) {
            const int c = stack.back();
            stack.pop_back();
            count[cid]++;
            const int x = c / Wd, y = c - x * Wd;
            const int nb[4] = {x > 0 ? c - Wd : -1, x + 1 < H ? c + Wd : -1, y > 0 ? c - 1 : -1, y + 1 < Wd ? c + 1 : -1};
            for (int k = 0; k < 4; ++k) {
                const int n = nb[k];
                if (n >= 0 && obst[n] == 0 && label[n] < 0) {
                    label[n] = cid;
                    stack.push_back(n);
                }
            }
        }
    }
    std::vector<uint32_t> begin(count.size() + 1, 0);
    for (size_t i = 0; i < count.size(); ++i) begin[i + 1] = begin[i] + count[i];
    std::vector<uint32_t> fill(begin.begin(), begin.end() - 1);
    for (int c = 0; c < cells; ++c) {
        if (label[c] < 0) {
            comp_begin[c] = 0;
            comp_len[c] = 0;
            continue;
        }
        const int cid = label[c];
        comp_begin[c] = begin[cid];
        comp_len[c] = count[cid];
        const int x = c / Wd, y = c - x * Wd;
        comp_cells[fill[cid]++] = ((uint32_t)x << 16) | (uint32_t)y;
    }
    for (uint32_t i = begin.back(); i < (uint32_t)cells; ++i) comp_cells[i] = 0;
}

int pgx_reset_from_state(pgx_env* e, const uint8_t* obstacles, const int32_t* agent_xy, const int32_t* target_xy,
                         void* stream) {
    if (!e || !obstacles || !agent_xy || !target_xy) return fail(PGX_E_INVALID, "pgx_reset_from_state: null argument");
    DeviceGuard guard(e->device);
    if (guard.err != hipSuccess) return fail(PGX_E_HIP, "cannot select device: %s", hipGetErrorString(guard.err));
    hipStream_t s = (hipStream_t)stream;
    const pgx_config& c = e->cfg;
    const size_t B = (size_t)c.batch, BA = B * c.num_agents;
    PGX_HIP(pgx::launch_pack_obstacles(obstacles, e->obst, c.batch, c.height, c.width, c.obs_radius, e->wpr, e->bmw, s));
    PGX_HIP(pgx::launch_pack_agents(agent_xy, target_xy, e->pos, e->tgt, e->pos0, e->tgt0, e->active, e->tcount, BA,
                                    c.obs_radius, s));
    PGX_HIP(pgx::launch_zero_i32(e->elapsed, B, s));
    PGX_HIP(pgx::launch_zero_i32(reinterpret_cast<int32_t*>(e->macc), B * 4, s));
    if (c.on_target == PGX_ON_TARGET_RESTART) {
        const size_t cells = (size_t)c.height * c.width;
        std::vector<uint8_t> h_obst(B * cells);
        PGX_HIP(hipMemcpyAsync(h_obst.data(), obstacles, B * cells, hipMemcpyDeviceToHost, s));
        PGX_HIP(hipStreamSynchronize(s));
        std::vector<uint32_t> h_begin(B * cells), h_len(B * cells), h_cells(B * cells);
        unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
        nt = (unsigned)std::min<size_t>(nt, B);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t) {
            pool.emplace_back([&, t]() {
                std::vector<int32_t> label, stack;
                for (size_t b = t; b < B; b += nt)
                    label_components_host(h_obst.data() + b * cells, c.height, c.width, h_begin.data() + b * cells,
                                          h_len.data() + b * cells, h_cells.data() + b * cells, label, stack);
            });
        }
        for (auto& th : pool) th.join();
        PGX_HIP(hipMemcpyAsync(e->comp_begin, h_begin.data(), B * cells * 4, hipMemcpyHostToDevice, s));
        PGX_HIP(hipMemcpyAsync(e->comp_len, h_len.data(), B * cells * 4, hipMemcpyHostToDevice, s));
        PGX_HIP(hipMemcpyAsync(e->comp_cells, h_cells.data(), B * cells * 4, hipMemcpyHostToDevice, s));
        PGX_HIP(hipStreamSynchronize(s));
    }
    e->has_state = true;
    return PGX_OK;
}

static void fill_params(const pgx_env* e, pgx::StepParams& p) {
    const pgx_config& c = e->cfg;
    memset(&p, 0, sizeof p);
    p.batch = c.batch;
    p.num_agents = c.num_agents;
    p.r = c.obs_radius;
    p.wpr = e->wpr;
    p.bm_words = e->bmw;
    p.map_w = c.width;
    p.map_cells = c.height * c.width;
    p.w_magic = (uint32_t)((1ull << 32) / (uint64_t)e->W) + 1u;
    p.collision = c.collision_system;
    p.on_target = c.on_target;
    p.max_steps = c.max_episode_steps;
    p.auto_reset = c.auto_reset;
    p.flags = e->flags;
    p.epw = e->geo.epw;
    p.seed = c.seed;
    p.env_index_base = c.env_index_base;
    p.obst = e->obst;
    p.pos = e->pos;
    p.tgt = e->tgt;
    p.active = e->active;
    p.elapsed = e->elapsed;
    p.pos0 = e->pos0;
    p.tgt0 = e->tgt0;
    p.comp_begin = e->comp_begin;
    p.comp_len = e->comp_len;
    p.comp_cells = e->comp_cells;
    p.tcount = e->tcount;
    p.dbg = e->dbg;
    p.macc = e->macc;
    p.metrics_out = e->metrics_out;
    p.episode_done = e->episode_done;
}

int pgx_step(pgx_env* e, const void* actions, int action_dtype, float* obs, float* rewards, uint8_t* terminated,
             uint8_t* truncated, uint8_t* is_active, void* stream) {
    if (!e || !actions || !rewards || !terminated || !truncated) return fail(PGX_E_INVALID, "pgx_step: null argument");
    if (action_dtype < 0 || action_dtype > 2) return fail(PGX_E_INVALID, "pgx_step: bad action_dtype %d", action_dtype);
    if (!e->has_state) return fail(PGX_E_STATE, "pgx_step called before pgx_reset_from_state");
    DeviceGuard guard(e->device);
    if (guard.err != hipSuccess) return fail(PGX_E_HIP, "cannot select device: %s", hipGetErrorString(guard.err));
    pgx::StepParams p;
    fill_params(e, p);
    p.mode = pgx::MODE_STEP;
    p.action_dtype = action_dtype;
    p.actions = actions;
    p.obs = obs;
    p.rewards = rewards;
    p.terminated = terminated;
    p.truncated = truncated;
    p.act_out = is_active;
    PGX_HIP(pgx::launch_step(p, e->geo, (hipStream_t)stream));
    return PGX_OK;
}

int pgx_set_metrics_buffers(pgx_env* e, float* metrics, uint8_t* episode_done) {
    if (!e) return fail(PGX_E_INVALID, "pgx_set_metrics_buffers: null handle");
    e->metrics_out = metrics;
    e->episode_done = episode_done;
    return PGX_OK;
}

int pgx_observe(pgx_env* e, float* obs, void* stream) {
    if (!e || !obs) return fail(PGX_E_INVALID, "pgx_observe: null argument");
    if (!e->has_state) return fail(PGX_E_STATE, "pgx_observe called before pgx_reset_from_state");
    DeviceGuard guard(e->device);
    if (guard.err != hipSuccess) return fail(PGX_E_HIP, "cannot select device: %s", hipGetErrorString(guard.err));
    pgx::StepParams p;
    fill_params(e, p);
    p.mode = pgx::MODE_OBSERVE;
    p.obs = obs;
    PGX_HIP(pgx::launch_step(p, e->geo, (hipStream_t)stream));
    return PGX_OK;
}

int pgx_get_state(pgx_env* e, int32_t* agent_xy, int32_t* target_xy, uint8_t* is_active, int32_t* elapsed,
                  uint8_t* occupancy, void* stream) {
    if (!e) return fail(PGX_E_INVALID, "pgx_get_state: null handle");
    if (!e->has_state) return fail(PGX_E_STATE, "pgx_get_state called before pgx_reset_from_state");
    DeviceGuard guard(e->device);
    if (guard.err != hipSuccess) return fail(PGX_E_HIP, "cannot select device: %s", hipGetErrorString(guard.err));
    hipStream_t s = (hipStream_t)stream;
    const pgx_config& c = e->cfg;
    const size_t B = (size_t)c.batch, BA = B * c.num_agents;
    if (agent_xy || target_xy || is_active)
        PGX_HIP(pgx::launch_unpack_state(e->pos, e->tgt, e->active, agent_xy, target_xy, is_active, BA, c.obs_radius, s));
    if (elapsed) PGX_HIP(hipMemcpyAsync(elapsed, e->elapsed, B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    if (occupancy) {
        PGX_HIP(hipMemsetAsync(occupancy, 0, B * e->PH * e->PW, s));
        PGX_HIP(pgx::launch_occupancy(e->pos, e->active, occupancy, BA, c.num_agents, e->PH, e->PW, s));
    }
    return PGX_OK;
}

// Diagnostic only (not part of include/pogema_amd.h): copies the per-workgroup clock stamps of the last
// launch to the host; needs PGX_FLAGS bit 2 at pgx_create.  Synchronises the device.
int pgx_debug_timestamps(pgx_env* e, unsigned long long* host_out, int64_t max_elems) {
    if (!e || !host_out) return fail(PGX_E_INVALID, "pgx_debug_timestamps: null argument");
    if (!e->dbg) return fail(PGX_E_STATE, "diagnostic stamps not enabled (PGX_FLAGS bit 2)");
    DeviceGuard guard(e->device);
    PGX_HIP(hipDeviceSynchronize());
    const size_t n = std::min<size_t>(e->dbg_elems, (size_t)max_elems);
    PGX_HIP(hipMemcpy(host_out, e->dbg, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return (int)PGX_OK;
}

// ================================================================================================
// host-side synthetic generator
// ================================================================================================
namespace {

struct Xoshiro {
    uint64_t s[4];
    static uint64_t sm(uint64_t& z) {
        z += 0x9E3779B97F4A7C15ull;
        uint64_t x = z;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    }
    explicit Xoshiro(uint64_t seed) {
        uint64_t z = seed;
        for (auto& v : s) v = sm(z);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t result = rotl(s[1] * 5, 7) * 9;
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    float unit() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }
};

// one env; returns true when `A` start/target pairs were placed
// (obst_out == nullptr: place on the given `obst` map instead of drawing one)
bool generate_one(int H, int Wd, int A, float density, uint64_t seed, const uint8_t* obst_in, uint8_t* obst_out,
                  int32_t* axy, int32_t* txy, std::vector<int32_t>& label, std::vector<int32_t>& stack,
                  std::vector<int32_t>& order, std::vector<int32_t>& pending) {
    Xoshiro rng(seed);
    const int cells = H * Wd;
    if (obst_out)
        for (int c = 0; c < cells; ++c) obst_out[c] = rng.unit() < density ? 1 : 0;
    const uint8_t* obst = obst_out ? obst_out : obst_in;
    // components
    label.assign(cells, -1);
    int ncomp = 0;
    order.clear();
    for (int s = 0; s < cells; ++s) {
        if (obst[s]) continue;
        order.push_back(s);
        if (label[s] >= 0) continue;
        const int cid = ncomp++;
        label[s] = cid;
        stack.clear();
        stack.push_back(s);
        while (!stack.empty()) {
            const int c = stack.back();
            stack.pop_back();
            const int x = c / Wd, y = c - x * Wd;
            const int nb[4] = {x > 0 ? c - Wd : -1, x + 1 < H ? c + Wd : -1, y > 0 ? c - 1 : -1, y + 1 < Wd ? c + 1 : -1};
            for (int k = 0; k < 4; ++k) {
                const int n = nb[k];
                if (n >= 0 && !obst[n] && label[n] < 0) {
                    label[n] = cid;
                    stack.push_back(n);
                }
            }
        }
    }
    // Fisher-Yates over the free cells
    for (int i = (int)order.size() - 1; i > 0; --i) {
        const int j = (int)rng.below((uint32_t)(i + 1));
        std::swap(order[i], order[j]);
    }
    // pair cells inside one component: first visit opens a start, second visit closes it with a target
    pending.assign(ncomp, -1);
    int placed = 0;
    for (size_t i = 0; i < order.size() && placed < A; ++i) {
        const int c = order[i];
        const int cid = label[c];
        if (pending[cid] < 0) {
            pending[cid] = c;
        } else {
            const int s = pending[cid];
            pending[cid] = -1;
            axy[2 * placed] = s / Wd;
            axy[2 * placed + 1] = s % Wd;
            txy[2 * placed] = c / Wd;
            txy[2 * placed + 1] = c % Wd;
            ++placed;
        }
    }
    return placed == A;
}

}  // namespace

int pgx_generate(int32_t batch, int32_t height, int32_t width, int32_t num_agents, float density, uint64_t seed0,
                 int32_t max_retries, int32_t nthreads, uint8_t* obstacles, int32_t* agent_xy, int32_t* target_xy) {
    if (batch < 1 || height < 1 || width < 1 || num_agents < 1 || !obstacles || !agent_xy || !target_xy)
        return fail(PGX_E_INVALID, "pgx_generate: bad argument");
    if (!(density >= 0.0f && density <= 1.0f)) return fail(PGX_E_INVALID, "density %.3f outside [0, 1]", (double)density);
    if ((int64_t)2 * num_agents > (int64_t)height * width)
        return fail(PGX_E_PLACEMENT, "%d agents need %d distinct cells, map has %d", num_agents, 2 * num_agents,
                    height * width);
    if (max_retries < 1) max_retries = 10;
    unsigned nt = nthreads > 0 ? (unsigned)nthreads : std::max(1u, std::thread::hardware_concurrency());
    nt = (unsigned)std::min<int64_t>(nt, batch);
    const size_t cells = (size_t)height * width;
    std::vector<int> failed(nt, -1);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t) {
        pool.emplace_back([&, t]() {
            std::vector<int32_t> label, stack, order, pending;
            for (int64_t b = t; b < batch; b += nt) {
                bool ok = false;
                for (int attempt = 0; attempt < max_retries && !ok; ++attempt) {
                    // env b, attempt k draws from stream (seed0 + b) advanced by k * 2^32
                    const uint64_t seed = (seed0 + (uint64_t)b) ^ ((uint64_t)attempt << 40);
                    ok = generate_one(height, width, num_agents, density, seed, nullptr, obstacles + b * cells,
                                      agent_xy + (size_t)b * num_agents * 2, target_xy + (size_t)b * num_agents * 2,
                                      label, stack, order, pending);
                }
                if (!ok && failed[t] < 0) failed[t] = (int)b;
            }
        });
    }
    for (auto& th : pool) th.join();
    for (unsigned t = 0; t < nt; ++t)
        if (failed[t] >= 0)
            return fail(PGX_E_PLACEMENT, "could not place %d agents in env %d after %d attempts (density %.2f, %dx%d)",
                        num_agents, failed[t], max_retries, (double)density, height, width);
    return PGX_OK;
}

int pgx_place_agents(int32_t batch, int32_t height, int32_t width, int32_t num_agents, uint64_t seed0,
                     int32_t max_retries, int32_t nthreads, const uint8_t* obstacles, int32_t shared_map,
                     int32_t* agent_xy, int32_t* target_xy) {
    if (batch < 1 || height < 1 || width < 1 || num_agents < 1 || !obstacles || !agent_xy || !target_xy)
        return fail(PGX_E_INVALID, "pgx_place_agents: bad argument");
    if (max_retries < 1) max_retries = 10;
    unsigned nt = nthreads > 0 ? (unsigned)nthreads : std::max(1u, std::thread::hardware_concurrency());
    nt = (unsigned)std::min<int64_t>(nt, batch);
    const size_t cells = (size_t)height * width;
    std::vector<int> failed(nt, -1);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t) {
        pool.emplace_back([&, t]() {
            std::vector<int32_t> label, stack, order, pending;
            for (int64_t b = t; b < batch; b += nt) {
                bool ok = false;
                const uint8_t* m = obstacles + (shared_map ? 0 : b * cells);
                for (int attempt = 0; attempt < max_retries && !ok; ++attempt) {
                    const uint64_t seed = (seed0 + (uint64_t)b) ^ ((uint64_t)attempt << 40);
                    ok = generate_one(height, width, num_agents, 0.0f, seed, m, nullptr,
                                      agent_xy + (size_t)b * num_agents * 2, target_xy + (size_t)b * num_agents * 2,
                                      label, stack, order, pending);
                }
                if (!ok && failed[t] < 0) failed[t] = (int)b;
            }
        });
    }
    for (auto& th : pool) th.join();
    for (unsigned t = 0; t < nt; ++t)
        if (failed[t] >= 0)
            return fail(PGX_E_PLACEMENT, "could not place %d agents on the given map of env %d after %d attempts",
                        num_agents, failed[t], max_retries);
    return PGX_OK;
}

}  // extern "C"
