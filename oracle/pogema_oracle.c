/*
 * CPU ORACLE (test infrastructure only) -- plain-C batched port of oracle/pogema_oracle.py.
 *
 *     *** PARITY UNPINNED ***  (same status as pogema_oracle.py: /root/reference is a stub README,
 *     README.md:3,5; the algorithm follows SURVEY.md section 8a rows A1..A13 and the builder's
 *     recollection of upstream pogema/grid.py, pogema/envs.py, pogema/wrappers/multi_time_limit.py.)
 *
 * Purpose: (1) parity checker at sizes the pure-Python oracle is too slow for, (2) the
 * `cpu_baseline` leg of bench.py (kind "port").  It is cross-validated against the Python oracle
 * by tests/test_oracle.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
 * load this library; the product never does.
 *
 * Order-dependent parts of the reference (dict-of-lists `used_cells`, `used_edges`, reverse-index
 * revert loop, recursive `_revert_action`) are emulated literally with per-cell insertion-ordered
 * lists, NOT with the closed-form rule the HIP kernel uses -- that is the point of the check.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PO_MAX_CLAIM 8

typedef struct po_config {
    int32_t batch, height, width, num_agents, obs_radius;
    int32_t collision_system; /* 0 priority, 1 block_both, 2 soft */
    int32_t on_target;        /* 0 finish, 1 restart, 2 nothing    */
    int32_t max_episode_steps, auto_reset, reserved0;
    uint64_t seed;
    int64_t env_index_base;
    int32_t random_outside;   /* empty_outside=False: Bernoulli(outside_density) obstacles beyond the ring */
    float outside_density;
    /* switches for the low-confidence recollections (docs/SPEC.md Q1 / Q2 / Q4 / Q7); 0 = recalled literal = default */
    int32_t soft_vertex_rule; /* 0 lowest index wins a contested cell (literal remove + reverse loop), 1 all claimants stay */
    int32_t coop_reward;      /* 0 1.0 to all iff all on goal, 1 1.0 to each agent on its own goal */
    int32_t bad_action;       /* 0 out-of-range action = noop, 1 noop + counted (po_bad_action_count) */
    int32_t soft_occupancy;   /* docs/SPEC.md Q2: 1 occupancy == cells of the visible agents; 0 the literal per-agent
                                 move_without_checks loop (clear old, set new, in index order) as recalled */
} po_config;

typedef struct po_env {
    po_config c;
    int PH, PW;
    /* per env, padded coordinates */
    uint8_t* obst;    /* [B][PH*PW] */
    uint8_t* occ;     /* [B][PH*PW] occupancy array `positions` */
    int32_t* px;      /* [B][A] */
    int32_t* py;
    int32_t* fx;      /* finishes */
    int32_t* fy;
    int32_t* px0;     /* initial (unpadded+r) state for auto reset */
    int32_t* py0;
    int32_t* fx0;
    int32_t* fy0;
    uint8_t* active;  /* [B][A] */
    int32_t* elapsed; /* [B] */
    /* lifelong */
    uint32_t* tcount;      /* [B][A] */
    int32_t* macc;         /* [B][4] metric accumulators: solved, sum of solve steps, max solve step, lifelong goals */
    int32_t* comp_begin;   /* [B][H*W] */
    int32_t* comp_len;     /* [B][H*W] */
    int32_t* comp_cells;   /* [B][H*W] unpadded cell index, grouped by component, row-major inside */
    int64_t bad_actions;   /* out-of-range actions of active agents since the last po_bad_action_count() */
} po_env;

static const int MOVE_DX[5] = {0, -1, 1, 0, 0};
static const int MOVE_DY[5] = {0, 0, 0, -1, 1};
static const int OPPOSITE[5] = {0, 2, 1, 4, 3};

/* ---- lifelong RNG (DESIGN.md "lifelong RNG"; identical in pogema_oracle.py and the HIP kernel) ---- */
static uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint32_t lifelong_draw(uint64_t seed, uint64_t env_index, uint32_t agent, uint32_t counter, uint32_t n) {
    uint64_t h = splitmix64(seed);
    h = splitmix64(h ^ env_index);
    h = splitmix64(h ^ (((uint64_t)agent << 32) | counter));
    return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32);
}

/* ================================================================================================ */
po_env* po_create(const po_config* cfg) {
    po_env* e = (po_env*)calloc(1, sizeof(po_env));
    if (!e) return NULL;
    e->c = *cfg;
    const int r = cfg->obs_radius;
    e->PH = cfg->height + 2 * r;
    e->PW = cfg->width + 2 * r;
    const size_t B = (size_t)cfg->batch, A = (size_t)cfg->num_agents, P = (size_t)e->PH * e->PW;
    const size_t cells = (size_t)cfg->height * cfg->width;
    e->obst = (uint8_t*)calloc(B * P, 1);
    e->occ = (uint8_t*)calloc(B * P, 1);
    int32_t** iv[] = {&e->px, &e->py, &e->fx, &e->fy, &e->px0, &e->py0, &e->fx0, &e->fy0};
    for (unsigned k = 0; k < sizeof iv / sizeof iv[0]; ++k) *iv[k] = (int32_t*)calloc(B * A, sizeof(int32_t));
    e->active = (uint8_t*)calloc(B * A, 1);
    e->elapsed = (int32_t*)calloc(B, sizeof(int32_t));
    e->tcount = (uint32_t*)calloc(B * A, sizeof(uint32_t));
    e->macc = (int32_t*)calloc(B * 4, sizeof(int32_t));
    if (cfg->on_target == 1) {
        e->comp_begin = (int32_t*)calloc(B * cells, sizeof(int32_t));
        e->comp_len = (int32_t*)calloc(B * cells, sizeof(int32_t));
        e->comp_cells = (int32_t*)calloc(B * cells, sizeof(int32_t));
    }
    return e;
}

void po_destroy(po_env* e) {
    if (!e) return;
    free(e->obst); free(e->occ); free(e->px); free(e->py); free(e->fx); free(e->fy);
    free(e->px0); free(e->py0); free(e->fx0); free(e->fy0); free(e->active); free(e->elapsed);
    free(e->tcount); free(e->macc); free(e->comp_begin); free(e->comp_len); free(e->comp_cells);
    free(e);
}

/* components of the UNPADDED map: ids in row-major order of first cell, cells row-major inside */
static void label_components(const uint8_t* obst, int H, int W, int32_t* begin, int32_t* len, int32_t* cells_out) {
    const int n = H * W;
    int32_t* label = (int32_t*)malloc(sizeof(int32_t) * n);
    int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * n);
    int32_t* count = (int32_t*)calloc(n + 1, sizeof(int32_t));
    for (int i = 0; i < n; ++i) label[i] = -1;
    int ncomp = 0;
    for (int s = 0; s < n; ++s) {
        if (obst[s] || label[s] >= 0) continue;
        const int cid = ncomp++;
        int sp = 0;
        label[s] = cid;
        stack[sp++] = s;
        while (sp) {
            const int c = stack[--sp];
            count[cid]++;
            const int x = c / W, y = c % W;
            for (int a = 1; a < 5; ++a) {
                const int nx = x + MOVE_DX[a], ny = y + MOVE_DY[a];
                if (nx < 0 || ny < 0 || nx >= H || ny >= W) continue;
                const int q = nx * W + ny;
                if (!obst[q] && label[q] < 0) { label[q] = cid; stack[sp++] = q; }
            }
        }
    }
    int32_t* start = (int32_t*)malloc(sizeof(int32_t) * (ncomp + 1));
    start[0] = 0;
    for (int k = 0; k < ncomp; ++k) start[k + 1] = start[k] + count[k];
    int32_t* fill = (int32_t*)malloc(sizeof(int32_t) * (ncomp + 1));
    memcpy(fill, start, sizeof(int32_t) * (ncomp + 1));
    for (int c = 0; c < n; ++c) {
        if (label[c] < 0) { begin[c] = 0; len[c] = 0; continue; }
        begin[c] = start[label[c]];
        len[c] = count[label[c]];
        cells_out[fill[label[c]]++] = c;
    }
    free(label); free(stack); free(count); free(start); free(fill);
}

static void install_initial(po_env* e, int b) {
    const int A = e->c.num_agents;
    const size_t P = (size_t)e->PH * e->PW;
    uint8_t* occ = e->occ + (size_t)b * P;
    memset(occ, 0, P);
    for (int i = 0; i < A; ++i) {
        const size_t g = (size_t)b * A + i;
        e->px[g] = e->px0[g]; e->py[g] = e->py0[g];
        e->fx[g] = e->fx0[g]; e->fy[g] = e->fy0[g];
        e->active[g] = 1;
        occ[(size_t)e->px[g] * e->PW + e->py[g]] = 1;
    }
    e->elapsed[b] = 0;
    memset(e->macc + (size_t)b * 4, 0, 4 * sizeof(int32_t));
}

/* Grid.__init__ + add_artificial_border */
int po_reset(po_env* e, const uint8_t* obstacles, const int32_t* agent_xy, const int32_t* target_xy) {
    const int B = e->c.batch, A = e->c.num_agents, H = e->c.height, W = e->c.width, r = e->c.obs_radius;
    const int PH = e->PH, PW = e->PW;
    const size_t P = (size_t)PH * PW;
    for (int b = 0; b < B; ++b) {
        uint8_t* o = e->obst + (size_t)b * P;
        memset(o, 0, P);
        if (e->c.random_outside) { /* docs/SPEC.md S1: generation 0 (explicit states) */
            double thr_d = (double)e->c.outside_density * 16777216.0 + 0.5;
            if (thr_d < 0) thr_d = 0;
            if (thr_d > 16777216.0) thr_d = 16777216.0;
            const uint32_t thr = (uint32_t)thr_d;
            const uint64_t h = splitmix64(splitmix64(splitmix64(e->c.seed) ^ (uint64_t)(e->c.env_index_base + b)) ^
                                          (0x4F55545300000000ull | 0u));
            for (int x = 0; x < PH; ++x)
                for (int y = 0; y < PW; ++y)
                    if (x < r - 1 || x > PH - r || y < r - 1 || y > PW - r)
                        o[(size_t)x * PW + y] = (splitmix64(h ^ (uint64_t)(x * PW + y)) >> 40) < thr ? 1 : 0;
        }
        for (int y = r - 1; y <= PW - r; ++y) { o[(size_t)(r - 1) * PW + y] = 1; o[(size_t)(PH - r) * PW + y] = 1; }
        for (int x = r - 1; x <= PH - r; ++x) { o[(size_t)x * PW + (r - 1)] = 1; o[(size_t)x * PW + (PW - r)] = 1; }
        for (int x = 0; x < H; ++x)
            for (int y = 0; y < W; ++y)
                o[(size_t)(x + r) * PW + (y + r)] = obstacles[((size_t)b * H + x) * W + y] ? 1 : 0;
        for (int i = 0; i < A; ++i) {
            const size_t g = (size_t)b * A + i;
            e->px0[g] = agent_xy[2 * g] + r; e->py0[g] = agent_xy[2 * g + 1] + r;
            e->fx0[g] = target_xy[2 * g] + r; e->fy0[g] = target_xy[2 * g + 1] + r;
            e->tcount[g] = 0;
        }
        install_initial(e, b);
        if (e->c.on_target == 1) {
            const size_t cells = (size_t)H * W;
            label_components(obstacles + (size_t)b * cells, H, W, e->comp_begin + b * cells, e->comp_len + b * cells,
                             e->comp_cells + b * cells);
        }
    }
    return 0;
}

/* ---- Grid.move ---------------------------------------------------------------------- */
static void grid_move(po_env* e, int b, int i, int action) {
    const size_t g = (size_t)b * e->c.num_agents + i;
    const size_t P = (size_t)e->PH * e->PW;
    const uint8_t* o = e->obst + (size_t)b * P;
    uint8_t* occ = e->occ + (size_t)b * P;
    int x = e->px[g], y = e->py[g];
    const int nx = x + MOVE_DX[action], ny = y + MOVE_DY[action];
    if (o[(size_t)nx * e->PW + ny] == 0) {
        if (occ[(size_t)nx * e->PW + ny] == 0) {
            occ[(size_t)x * e->PW + y] = 0;
            x = nx; y = ny;
            occ[(size_t)x * e->PW + y] = 1;
        }
    }
    e->px[g] = x; e->py[g] = y;
}

/* ---- per-thread scratch emulating the reference's dicts ------------------------------------------- */
typedef struct scratch {
    size_t P;
    uint8_t* mark;          /* block_both: 0 absent, 1 'visited', 2 'blocked'                 */
    uint8_t* ccnt;          /* soft: len(used_cells[cell])                                       */
    int32_t* clist;         /* soft: used_cells[cell][k], insertion order, PO_MAX_CLAIM per cell */
    uint8_t* ecnt;          /* soft: len(used_edges[(cell, dir)])  -- key = cell*5 + dir         */
    int32_t* elist;         /* 2 per edge                                                        */
    int32_t* touched;       /* cells to clear afterwards                                         */
    int ntouched;
    int8_t* acts;           /* mutable copy of the actions                                       */
    int32_t* who;           /* all_stay: active agent standing on the cell, -1 = nobody         */
    int32_t* revert;        /* all_stay: agents reverted in the current round                  */
    int64_t* clean;         /* actions with out-of-range values replaced by noop                */
} scratch;

static scratch* scratch_new(size_t P, int A) {
    scratch* s = (scratch*)calloc(1, sizeof(scratch));
    s->P = P;
    s->mark = (uint8_t*)calloc(P, 1);
    s->ccnt = (uint8_t*)calloc(P, 1);
    s->clist = (int32_t*)malloc(sizeof(int32_t) * P * PO_MAX_CLAIM);
    s->ecnt = (uint8_t*)calloc(P * 5, 1);
    s->elist = (int32_t*)malloc(sizeof(int32_t) * P * 5 * 2);
    s->touched = (int32_t*)malloc(sizeof(int32_t) * (size_t)(A + 1) * 8);
    s->acts = (int8_t*)malloc((size_t)A);
    s->who = (int32_t*)malloc(sizeof(int32_t) * P);
    for (size_t k = 0; k < P; ++k) s->who[k] = -1;
    s->revert = (int32_t*)malloc(sizeof(int32_t) * (size_t)(A + 1));
    s->clean = (int64_t*)malloc(sizeof(int64_t) * (size_t)(A + 1));
    return s;
}
static void scratch_free(scratch* s) {
    free(s->mark); free(s->ccnt); free(s->clist); free(s->ecnt); free(s->elist); free(s->touched); free(s->acts); free(s->who); free(s->revert); free(s->clean); free(s);
}
static void touch(scratch* s, int cell) { s->touched[s->ntouched++] = cell; }

static void cells_append(scratch* s, int cell, int agent) {
    if (s->ccnt[cell] >= PO_MAX_CLAIM) abort();
    s->clist[(size_t)cell * PO_MAX_CLAIM + s->ccnt[cell]++] = agent;
}
static void cells_remove(scratch* s, int cell, int agent) { /* list.remove: first occurrence */
    int32_t* l = s->clist + (size_t)cell * PO_MAX_CLAIM;
    const int n = s->ccnt[cell];
    for (int k = 0; k < n; ++k)
        if (l[k] == agent) {
            for (int j = k; j + 1 < n; ++j) l[j] = l[j + 1];
            s->ccnt[cell]--;
            return;
        }
    abort(); /* ValueError in the reference */
}

/* Pogema._revert_action (recursive) */
static void revert_action(po_env* e, int b, scratch* s, int agent, int cell) {
    const size_t g = (size_t)b * e->c.num_agents + agent;
    s->acts[agent] = 0;
    cells_remove(s, cell, agent);
    const int new_cell = e->px[g] * e->PW + e->py[g];
    if (s->ccnt[new_cell] > 0) {
        cells_append(s, new_cell, agent);
        revert_action(e, b, s, s->clist[(size_t)new_cell * PO_MAX_CLAIM], new_cell);
    } else {
        touch(s, new_cell);
        cells_append(s, new_cell, agent);
    }
}

/* Grid.move_without_checks for every active agent once the surviving `soft` moves are known (docs/SPEC.md Q2).
 * soft_occupancy 1: the occupancy array afterwards is exactly the set of active agents' cells; 0: the literal loop as
 * recalled -- clear the old cell, set the new one, agent by agent in index order (an agent entering the cell a HIGHER-index
 * agent is leaving has its new cell cleared again by that agent's turn). */
static void apply_soft_moves(po_env* e, int b, const int8_t* acts) {
    const int A = e->c.num_agents, PW = e->PW;
    const size_t base = (size_t)b * A;
    uint8_t* occ = e->occ + (size_t)b * e->PH * e->PW;
    if (e->c.soft_occupancy != 0) {
        for (int i = 0; i < A; ++i)
            if (e->active[base + i]) occ[(size_t)e->px[base + i] * PW + e->py[base + i]] = 0;
    }
    for (int i = 0; i < A; ++i) {
        if (!e->active[base + i]) continue;
        const int a = acts[i];
        if (e->c.soft_occupancy == 0) occ[(size_t)e->px[base + i] * PW + e->py[base + i]] = 0;
        e->px[base + i] += MOVE_DX[a];
        e->py[base + i] += MOVE_DY[a];
        occ[(size_t)e->px[base + i] * PW + e->py[base + i]] = 1;
    }
}

/* Pogema.move_agents */
static void move_agents(po_env* e, int b, const int64_t* actions, scratch* s) {
    const int A = e->c.num_agents, PW = e->PW;
    const size_t base = (size_t)b * A;
    const size_t P = (size_t)e->PH * e->PW;
    const uint8_t* o = e->obst + (size_t)b * P;
    if (e->c.collision_system == 0) {
        for (int i = 0; i < A; ++i)
            if (e->active[base + i]) grid_move(e, b, i, (int)actions[i]);
    } else if (e->c.collision_system == 1) {
        s->ntouched = 0;
        for (int i = 0; i < A; ++i) {
            if (!e->active[base + i]) continue;
            const int x = e->px[base + i], y = e->py[base + i], a = (int)actions[i];
            const int d = (x + MOVE_DX[a]) * PW + (y + MOVE_DY[a]), c = x * PW + y;
            if (s->mark[d] == 0) touch(s, d);
            s->mark[d] = s->mark[d] ? 2 : 1;
            if (s->mark[c] == 0) touch(s, c);
            s->mark[c] = 2;
        }
        /* agents_xy snapshot: moves below never change an unmoved agent's cell, and each agent reads only its own */
        for (int i = 0; i < A; ++i) {
            if (!e->active[base + i]) continue;
            const int x = e->px[base + i], y = e->py[base + i], a = (int)actions[i];
            const int d = (x + MOVE_DX[a]) * PW + (y + MOVE_DY[a]);
            if (s->mark[d] != 2) grid_move(e, b, i, a);
        }
        for (int k = 0; k < s->ntouched; ++k) s->mark[s->touched[k]] = 0;
    } else if (e->c.soft_vertex_rule == 1) {
        /* docs/SPEC.md Q1 alternative (textbook MAPF): synchronous fixed point -- every round reverts ALL movers whose
         * destination is an obstacle, is claimed by anybody else (a non-mover claims its own cell) or lies across a
         * swapped edge; repeat until nothing changes; apply the surviving moves. */
        for (int i = 0; i < A; ++i) s->acts[i] = (int8_t)actions[i];
        for (int i = 0; i < A; ++i)
            if (e->active[base + i]) s->who[e->px[base + i] * PW + e->py[base + i]] = i;
        for (int changed = 1; changed;) {
            changed = 0;
            s->ntouched = 0;
            for (int i = 0; i < A; ++i) {
                if (!e->active[base + i]) continue;
                const int a = s->acts[i];
                const int d = (e->px[base + i] + MOVE_DX[a]) * PW + (e->py[base + i] + MOVE_DY[a]);
                if (s->ccnt[d] == 0) touch(s, d);
                if (s->ccnt[d] < 255) s->ccnt[d]++;
            }
            int nrev = 0;
            for (int i = 0; i < A; ++i) {
                if (!e->active[base + i] || s->acts[i] == 0) continue;
                const int a = s->acts[i];
                const int c = e->px[base + i] * PW + e->py[base + i];
                const int d = (e->px[base + i] + MOVE_DX[a]) * PW + (e->py[base + i] + MOVE_DY[a]);
                const int ow = s->who[d];
                int swap = 0;
                if (ow >= 0 && s->acts[ow] != 0) {
                    const int ao = s->acts[ow];
                    swap = (e->px[base + ow] + MOVE_DX[ao]) * PW + (e->py[base + ow] + MOVE_DY[ao]) == c;
                }
                if (o[d] || s->ccnt[d] > 1 || swap) s->revert[nrev++] = i;
            }
            for (int k = 0; k < nrev; ++k) { s->acts[s->revert[k]] = 0; changed = 1; }
            for (int k = 0; k < s->ntouched; ++k) s->ccnt[s->touched[k]] = 0;
        }
        for (int i = 0; i < A; ++i)
            if (e->active[base + i]) s->who[e->px[base + i] * PW + e->py[base + i]] = -1;
        apply_soft_moves(e, b, s->acts);
    } else {
        s->ntouched = 0;
        for (int i = 0; i < A; ++i) s->acts[i] = (int8_t)actions[i];
        for (int i = 0; i < A; ++i) {
            if (!e->active[base + i]) continue;
            const int x = e->px[base + i], y = e->py[base + i], a = s->acts[i];
            const int c = x * PW + y, d = (x + MOVE_DX[a]) * PW + (y + MOVE_DY[a]);
            touch(s, c); touch(s, d);
            cells_append(s, d, i);
            s->ecnt[(size_t)c * 5 + a] = 1;                       /* used_edges[x,y,x+dx,y+dy] = [i] */
            s->elist[((size_t)c * 5 + a) * 2] = i;
            if (a != 0) {                                          /* setdefault(reverse edge).append(i) */
                const size_t k = (size_t)d * 5 + OPPOSITE[a];
                if (s->ecnt[k] >= 2) abort();
                s->elist[k * 2 + s->ecnt[k]++] = i;
            }
        }
        for (int i = 0; i < A; ++i) {
            if (!e->active[base + i]) continue;
            const int x = e->px[base + i], y = e->py[base + i], a = s->acts[i];
            const int c = x * PW + y, d = (x + MOVE_DX[a]) * PW + (y + MOVE_DY[a]);
            if (s->ecnt[(size_t)c * 5 + a] > 1) {
                cells_remove(s, d, i);
                cells_append(s, c, i);
                s->acts[i] = 0;
            }
        }
        for (int i = A - 1; i >= 0; --i) {
            if (!e->active[base + i]) continue;
            const int x = e->px[base + i], y = e->py[base + i], a = s->acts[i];
            const int d = (x + MOVE_DX[a]) * PW + (y + MOVE_DY[a]);
            if (s->ccnt[d] > 1 || o[d]) revert_action(e, b, s, i, d);
        }
        apply_soft_moves(e, b, s->acts);
        for (int k = 0; k < s->ntouched; ++k) {
            const int c = s->touched[k];
            s->ccnt[c] = 0;
            memset(s->ecnt + (size_t)c * 5, 0, 5);
        }
    }
}

/* PogemaBase._obs for one env */
static void write_obs(const po_env* e, int b, float* obs) {
    const int A = e->c.num_agents, r = e->c.obs_radius, W = 2 * r + 1, PW = e->PW;
    const size_t P = (size_t)e->PH * e->PW;
    const uint8_t* o = e->obst + (size_t)b * P;
    const uint8_t* occ = e->occ + (size_t)b * P;
    for (int i = 0; i < A; ++i) {
        const size_t g = (size_t)b * A + i;
        float* out = obs + g * 3 * W * W;
        const int x = e->px[g], y = e->py[g];
        for (int wx = 0; wx < W; ++wx)
            for (int wy = 0; wy < W; ++wy) {
                const size_t c = (size_t)(x - r + wx) * PW + (y - r + wy);
                out[wx * W + wy] = (float)o[c];
                out[W * W + wx * W + wy] = (float)occ[c];
                out[2 * W * W + wx * W + wy] = 0.0f;
            }
        int dx = x - e->fx[g], dy = y - e->fy[g];
        dx = dx >= 0 ? (dx < r ? dx : r) : (dx > -r ? dx : -r);
        dy = dy >= 0 ? (dy < r ? dy : r) : (dy > -r ? dy : -r);
        out[2 * W * W + (r - dx) * W + (r - dy)] = 1.0f;
    }
}

/* metric wrappers (upstream pogema/wrappers/metrics.py, recollection -- DESIGN.md open question 9).
 * out[6] = ISR, CSR, ep_length, SoC, makespan, avg_throughput; written only when the episode finished. */
static void compute_metrics(po_env* e, int b, int step, int n_arrived, int n_on_goal, int finished, float* out) {
    int32_t* m = e->macc + (size_t)b * 4;
    const int A = e->c.num_agents;
    if (e->c.on_target == 0) {
        m[0] += n_arrived;
        m[1] += n_arrived * step;
        if (n_arrived) m[2] = step > m[2] ? step : m[2];
    } else if (e->c.on_target == 1) {
        m[3] += n_arrived;
    }
    if (!finished || !out) return;
    if (e->c.on_target == 0) {
        const int unsolved = A - m[0];
        const int total = m[1] + unsolved * step;
        const int mx = unsolved ? step : m[2];
        out[0] = (float)m[0] / (float)A; out[1] = m[0] == A ? 1.0f : 0.0f; out[2] = (float)total / (float)A + 1.0f;
        out[3] = (float)(total + A); out[4] = (float)(mx + 1); out[5] = 0.0f;
    } else if (e->c.on_target == 2) {
        out[0] = (float)n_on_goal / (float)A; out[1] = n_on_goal == A ? 1.0f : 0.0f; out[2] = (float)(step + 1);
        out[3] = (float)(A * (step + 1)); out[4] = (float)(step + 1); out[5] = 0.0f;
    } else {
        const int denom = e->c.max_episode_steps > 0 ? e->c.max_episode_steps : step + 1;
        out[0] = 0.0f; out[1] = 0.0f; out[2] = (float)(step + 1); out[3] = 0.0f; out[4] = 0.0f;
        out[5] = (float)m[3] / (float)denom;
    }
}

static void step_env(po_env* e, int b, const int64_t* actions, float* obs, float* rewards, uint8_t* terminated,
                     uint8_t* truncated, uint8_t* active_out, float* metrics, uint8_t* episode_done, scratch* s) {
    const int A = e->c.num_agents, r = e->c.obs_radius;
    const size_t base = (size_t)b * A;
    const size_t P = (size_t)e->PH * e->PW;
    uint8_t* occ = e->occ + (size_t)b * P;
    /* docs/SPEC.md Q7: out-of-range actions are noops; counted for ACTIVE agents when bad_action = flag */
    int64_t* clean = s->clean;
    int bad = 0;
    for (int i = 0; i < A; ++i) {
        int64_t a = actions[base + i];
        if (a < 0 || a > 4) { bad += e->active[base + i] ? 1 : 0; a = 0; }
        clean[i] = a;
    }
    if (bad && e->c.bad_action == 1) {
#pragma omp atomic
        e->bad_actions += bad;
    }
    move_agents(e, b, clean, s);
    int all_term = 1;
    int n_arrived = 0;  /* was_on_goal: on goal and still active right after the moves */
    for (int i = 0; i < A; ++i) {
        const size_t g = base + i;
        n_arrived += e->active[g] && e->px[g] == e->fx[g] && e->py[g] == e->fy[g];
    }
    if (e->c.on_target == 0) {
        for (int i = 0; i < A; ++i) {
            const size_t g = base + i;
            const int on_goal = e->px[g] == e->fx[g] && e->py[g] == e->fy[g];
            rewards[g] = (on_goal && e->active[g]) ? 1.0f : 0.0f;
            terminated[g] = (uint8_t)on_goal;
        }
        for (int i = 0; i < A; ++i) {
            const size_t g = base + i;
            if (e->px[g] == e->fx[g] && e->py[g] == e->fy[g]) { /* hide_agent */
                if (e->active[g]) occ[(size_t)e->px[g] * e->PW + e->py[g]] = 0;
                e->active[g] = 0;
            }
        }
    } else if (e->c.on_target == 1) {
        const size_t cells = (size_t)e->c.height * e->c.width;
        for (int i = 0; i < A; ++i) {
            const size_t g = base + i;
            const int on_goal = e->px[g] == e->fx[g] && e->py[g] == e->fy[g];
            rewards[g] = (on_goal && e->active[g]) ? 1.0f : 0.0f;
            terminated[g] = 0;
            if (on_goal) {
                const size_t ci = (size_t)b * cells + (size_t)(e->px[g] - r) * e->c.width + (e->py[g] - r);
                const uint32_t k = lifelong_draw(e->c.seed, (uint64_t)(e->c.env_index_base + b), (uint32_t)i,
                                                 e->tcount[g], (uint32_t)e->comp_len[ci]);
                e->tcount[g]++;
                const int cell = e->comp_cells[(size_t)b * cells + e->comp_begin[ci] + k];
                e->fx[g] = cell / e->c.width + r;
                e->fy[g] = cell % e->c.width + r;
            }
        }
    } else {
        int solved = 1;
        for (int i = 0; i < A; ++i) {
            const size_t g = base + i;
            solved = solved && e->active[g] && e->px[g] == e->fx[g] && e->py[g] == e->fy[g];
        }
        for (int i = 0; i < A; ++i) {
            const size_t g = base + i;
            const int mine = e->active[g] && e->px[g] == e->fx[g] && e->py[g] == e->fy[g];
            rewards[g] = (e->c.coop_reward == 1 ? mine : solved) ? 1.0f : 0.0f;
            terminated[g] = (uint8_t)solved;
        }
    }
    for (int i = 0; i < A; ++i) {
        all_term = all_term && terminated[base + i];
        if (active_out) active_out[base + i] = e->active[base + i];
    }
    e->elapsed[b] += 1;
    const int trunc = e->c.max_episode_steps > 0 && e->elapsed[b] >= e->c.max_episode_steps;
    for (int i = 0; i < A; ++i) truncated[base + i] = (uint8_t)trunc;
    const int finished = all_term || trunc;
    compute_metrics(e, b, e->elapsed[b] - 1, n_arrived, n_arrived, finished, metrics ? metrics + (size_t)b * 6 : NULL);
    if (episode_done) episode_done[b] = (uint8_t)finished;
    if (finished) memset(e->macc + (size_t)b * 4, 0, 4 * sizeof(int32_t));
    if (e->c.auto_reset && finished) install_initial(e, b);
    if (obs) write_obs(e, b, obs);
}

/* One step for the whole batch; nthreads > 1 uses OpenMP over environments. */
int po_step(po_env* e, const int64_t* actions, float* obs, float* rewards, uint8_t* terminated, uint8_t* truncated,
            uint8_t* active_out, float* metrics, uint8_t* episode_done, int nthreads) {
    const int B = e->c.batch;
    const size_t P = (size_t)e->PH * e->PW;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        scratch* s = scratch_new(P, e->c.num_agents);
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) step_env(e, b, actions, obs, rewards, terminated, truncated, active_out, metrics, episode_done, s);
        scratch_free(s);
    }
    return 0;
}

int64_t po_bad_action_count(po_env* e) {
    const int64_t n = e->bad_actions;
    e->bad_actions = 0;
    return n;
}

int po_observe(po_env* e, float* obs) {
    for (int b = 0; b < e->c.batch; ++b) write_obs(e, b, obs);
    return 0;
}

/* state export in UNPADDED coordinates */
int po_get_state(const po_env* e, int32_t* agent_xy, int32_t* target_xy, uint8_t* is_active, int32_t* elapsed,
                 uint8_t* occupancy) {
    const size_t BA = (size_t)e->c.batch * e->c.num_agents;
    const int r = e->c.obs_radius;
    for (size_t g = 0; g < BA; ++g) {
        if (agent_xy) { agent_xy[2 * g] = e->px[g] - r; agent_xy[2 * g + 1] = e->py[g] - r; }
        if (target_xy) { target_xy[2 * g] = e->fx[g] - r; target_xy[2 * g + 1] = e->fy[g] - r; }
        if (is_active) is_active[g] = e->active[g];
    }
    if (elapsed) memcpy(elapsed, e->elapsed, sizeof(int32_t) * e->c.batch);
    if (occupancy) memcpy(occupancy, e->occ, (size_t)e->c.batch * e->PH * e->PW);
    return 0;
}

/* ================================================================================================
 * Instance generator "GEN v2" -- plain-C port of oracle/generator_oracle.py (the normative statement).
 * Test infrastructure like everything else here; the product's generators (host pgx_generate, device
 * pgx_reset_random) are compared against it.  Returns 0, or -5 when an env could not be filled.
 *   obstacles  u8  [batch, H, W]   output, or input [H, W] shared by every env when given_map != 0
 *   agent_xy / target_xy   i32 [batch, A, 2] output
 * ================================================================================================ */
#define PO_TAG_OBST 0x4F42535400000000ull
#define PO_TAG_PLACE 0x504C414300000000ull

static uint64_t po_instance_hash(uint64_t seed, uint64_t env, uint32_t epoch, uint32_t attempt) {
    uint64_t h = splitmix64(seed);
    h = splitmix64(h ^ env);
    return splitmix64(h ^ (((uint64_t)epoch << 32) | attempt));
}

static void po_min_index_labels(const uint8_t* obst, int H, int W, int32_t* label, int32_t* stack) {
    const int cells = H * W;
    for (int c = 0; c < cells; ++c) label[c] = -1;
    for (int s = 0; s < cells; ++s) {
        if (obst[s] || label[s] >= 0) continue;
        int sp = 0;
        label[s] = s;
        stack[sp++] = s;
        while (sp) {
            const int c = stack[--sp];
            const int x = c / W, y = c % W;
            const int nb[4] = {x > 0 ? c - W : -1, x + 1 < H ? c + W : -1, y > 0 ? c - 1 : -1, y + 1 < W ? c + 1 : -1};
            for (int k = 0; k < 4; ++k) {
                const int n = nb[k];
                if (n >= 0 && !obst[n] && label[n] < 0) {
                    label[n] = s;
                    stack[sp++] = n;
                }
            }
        }
    }
}

int po_generate(int32_t batch, int32_t H, int32_t W, int32_t A, float density, uint64_t seed, int64_t env_index_base,
                const uint32_t* epochs, int32_t max_retries, int32_t given_map, uint8_t* obstacles, int32_t* agent_xy,
                int32_t* target_xy) {
    const int cells = H * W;
    if (2 * (int64_t)A > cells) return -5;
    double thr_d = (double)density * 16777216.0 + 0.5;
    if (thr_d < 0) thr_d = 0;
    if (thr_d > 16777216.0) thr_d = 16777216.0;
    const uint32_t thr = (uint32_t)thr_d;
    int status = 0;
#pragma omp parallel for schedule(dynamic, 8)
    for (int b = 0; b < batch; ++b) {
        int32_t* label = (int32_t*)malloc(sizeof(int32_t) * cells);
        int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * cells);
        int32_t* pending = (int32_t*)malloc(sizeof(int32_t) * cells);
        uint8_t* taken = (uint8_t*)malloc(cells);
        uint8_t* obst = given_map ? obstacles : obstacles + (size_t)b * cells;
        int32_t* axy = agent_xy + (size_t)b * A * 2;
        int32_t* txy = target_xy + (size_t)b * A * 2;
        int ok = 0;
        for (int attempt = 0; attempt < max_retries && !ok; ++attempt) {
            const uint64_t h = po_instance_hash(seed, (uint64_t)(env_index_base + b), epochs ? epochs[b] : 0u, (uint32_t)attempt);
            if (!given_map)
                for (int c = 0; c < cells; ++c) obst[c] = (splitmix64(h ^ (PO_TAG_OBST | (uint64_t)c)) >> 40) < thr ? 1 : 0;
            if (!given_map || attempt == 0) po_min_index_labels(obst, H, W, label, stack);
            memset(taken, 0, cells);
            for (int c = 0; c < cells; ++c) pending[c] = -1;
            int placed = 0;
            const uint32_t budget = 32u * (uint32_t)cells + 64u;
            for (uint32_t t = 0; t < budget && placed < A; ++t) {
                const uint32_t c = (uint32_t)(((splitmix64(h ^ (PO_TAG_PLACE | (uint64_t)t)) >> 32) * (uint64_t)cells) >> 32);
                if (obst[c] || taken[c]) continue;
                taken[c] = 1;
                const int root = label[c];
                if (pending[root] < 0) {
                    pending[root] = (int32_t)c;
                } else {
                    const int s = pending[root];
                    pending[root] = -1;
                    axy[2 * placed] = s / W; axy[2 * placed + 1] = s % W;
                    txy[2 * placed] = (int32_t)c / W; txy[2 * placed + 1] = (int32_t)c % W;
                    ++placed;
                }
            }
            ok = placed == A;
        }
        if (!ok) {
#pragma omp critical
            status = -5;
        }
        free(label); free(stack); free(pending); free(taken);
    }
    return status;
}
