// pgx_buffers.hip -- zone-aware allocation of the large output (observation) buffers.
//
// Measured on MI355X (profiles/r2/placement_*.txt, tools/placement_zones.hip, tools/placement_layouts.hip): physical HBM
// falls into a few large zones (three on the boxes measured, tens of GiB each -- consistent with the three 4-high rank
// groups of the 12-high HBM3E stacks, 288 GiB / 3).  A pure store stream whose 761 MB all lie in ONE zone sustains
// ~5.5 TB/s (138 us); the same stream with its first half in one zone and its second half in another sustains ~6.9 TB/s
// (110 us) -- reads do not care.  One hipMalloc'd buffer is physically compact, so it lies in one zone unless it
// happens to straddle a zone boundary: that was round 1's "placement lottery" (speed tiers 124 / 137 / 150 us).
//
// Here the placement is REQUESTED instead of searched for: a buffer is one contiguous virtual range assembled with the
// HIP virtual-memory API from two physical halves, and the second half is taken from another zone.  Physical addresses
// are not visible to user space, so "another zone" is found by moving the allocator: spacer allocations are held while
// candidate second halves are created, each candidate is timed against the first half with a plain store stream, and
// the first one that makes the stream >= 10 % faster than a same-zone pair is kept.  Spacers are released afterwards;
// what stays allocated is exactly the buffers.  When no other zone is reachable (spacer budget, little free memory)
// the buffers are still valid, just not spread -- pgx_buffers_info says which.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/pogema_amd.h"

namespace pgx {
// defined in pgx_api.cpp
int fail_msg(int code, const char* fmt, ...);

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Store stream used to classify placements: `nblk` single-wave workgroups, each writes one contiguous chunk, the
// chunks of one XCD contiguous in memory -- the shape of the step kernel's observation stream.
// The first half of the chunks goes to `lo`, the second half to `hi` (two separate allocations: no virtual contiguity
// needed for the probe).
__global__ __launch_bounds__(64) void zone_probe_kernel(f32x4* __restrict__ lo, f32x4* __restrict__ hi, size_t per_block, int nblk) {
    const f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int per_xcd = nblk >> 3;
    int b = blockIdx.x;
    if (b < (per_xcd << 3)) b = (b & 7) * per_xcd + (b >> 3);
    const int half = nblk >> 1;
    f32x4* o = b < half ? lo + (size_t)b * per_block : hi + (size_t)(b - half) * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += 64) o[i] = v;
}
// diagnostic twin of the probe: the same chunks READ (sum kept alive through `sink`)
__global__ __launch_bounds__(64) void zone_read_kernel(const f32x4* __restrict__ lo, const f32x4* __restrict__ hi, size_t per_block,
                                                        int nblk, float* sink) {
    const int per_xcd = nblk >> 3;
    int b = blockIdx.x;
    if (b < (per_xcd << 3)) b = (b & 7) * per_xcd + (b >> 3);
    const int half = nblk >> 1;
    const f32x4* o = b < half ? lo + (size_t)b * per_block : hi + (size_t)(b - half) * per_block;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = threadIdx.x; i < per_block; i += 64) acc += __builtin_nontemporal_load(&o[i]);
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) sink[0] = 1.f;
}
}  // namespace pgx

namespace {

constexpr size_t GRANULE = (size_t)2 << 20;      // physical handles and virtual ranges are multiples of 2 MiB
constexpr size_t GiB = (size_t)1 << 30;
constexpr size_t PROBE_HALF = (size_t)384 << 20; // zone probe: two halves of 384 MiB (well beyond the 256 MiB Infinity Cache)
constexpr size_t SPACER = 8 * GiB;

size_t round_up(size_t v, size_t g) { return (v + g - 1) / g * g; }

// One physical allocation mapped at a fixed virtual address.  hipMemSetAccess is issued by the caller over the WHOLE
// virtual range once every part of it is mapped (per-part calls fail with "invalid argument" for some size pairs on
// ROCm 7.2).
struct Part {
    hipMemGenericAllocationHandle_t handle{};
    void* va = nullptr;
    size_t bytes = 0;
    bool live = false;
};

hipMemAllocationProp device_prop(int device) {
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = device;
    return p;
}

hipError_t map_part(int device, void* va, size_t bytes, Part& part) {
    const hipMemAllocationProp prop = device_prop(device);
    hipError_t e = hipMemCreate(&part.handle, bytes, &prop, 0);
    if (e != hipSuccess) return e;
    e = hipMemMap(va, bytes, 0, part.handle, 0);
    if (e != hipSuccess) {
        (void)hipMemRelease(part.handle);
        return e;
    }
    part.va = va;
    part.bytes = bytes;
    part.live = true;
    return hipSuccess;
}

hipError_t grant_access(int device, void* va, size_t bytes) {
    hipMemAccessDesc acc = {};
    acc.location = device_prop(device).location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    return hipMemSetAccess(va, bytes, &acc, 1);
}

void drop_part(Part& part) {
    if (!part.live) return;
    if (part.va) (void)hipMemUnmap(part.va, part.bytes);
    (void)hipMemRelease(part.handle);
    part.live = false;
}

// Address ranges of the VMM API are never given back to the driver (hipMemAddressFree) NOR re-used for new memory:
//  * ROCm 7.2: a range that is freed, reserved again at the same address and mapped to new memory can keep answering
//    with its OLD translation -- stores that never arrive, then "Memory access fault by GPU"
//    (tools/vmm_ring_repro.py: 2 of 2 runs within 6 pools; 0 of 36 pools with the ranges kept; profiles/r2/vmm_va_reuse_fault.txt);
//  * round 3 tried the gentler form -- keep the reservation, hipMemUnmap, hipMemMap a fresh handle at the same address
//    (a process-wide free list of released ranges): the 12th pool of 2.8 GB buffers read back wrong data
//    (tools/vmm_va_reuse_check.py, profiles/r3/vmm_va_remap_stale.txt) -- some XCD still held the old translation.
// So a pool's own ranges leak address space (count x stride per pool out of 128 TiB; pgx_buffers_va_reserved() reports
// the total), and everything that does NOT need two physical halves behind one address -- the probe chunks of the zone
// walk, ~13 GiB per walk -- is plain hipMalloc/hipFree memory, whose addresses the runtime recycles safely.
// PGX_VA_REUSE=1 switches the free list on, PGX_VA_FREE=1 the hipMemAddressFree (diagnostics for later ROCm versions).
std::mutex g_va_mu;
std::multimap<size_t, void*> g_va_free;  // PGX_VA_REUSE=1 only: reserved, fully unmapped ranges by size
size_t g_va_reserved = 0;               // bytes of address space this process has reserved so far (pgx_buffers_va_reserved)

bool va_reuse() {
    static const bool on = [] { const char* e = getenv("PGX_VA_REUSE"); return e && e[0] == '1'; }();
    return on;
}

// a virtual range that owns its parts
struct Range {
    void* va = nullptr;
    size_t bytes = 0;
    bool plain = false;  // `va` is a hipMalloc allocation (probe chunks), not a VMM reservation
    std::vector<Part> parts;
    hipError_t reserve(size_t n) {
        bytes = n;
        if (va_reuse()) {
            std::lock_guard<std::mutex> lock(g_va_mu);
            auto it = g_va_free.find(n);
            if (it != g_va_free.end()) {
                va = it->second;
                g_va_free.erase(it);
                return hipSuccess;
            }
        }
        const hipError_t e = hipMemAddressReserve(&va, n, GRANULE, nullptr, 0);
        if (e != hipSuccess) va = nullptr;
        else {
            std::lock_guard<std::mutex> lock(g_va_mu);
            g_va_reserved += n;
        }
        return e;
    }
    void release() {
        if (plain) {
            if (va) (void)hipFree(va);
            va = nullptr;
            plain = false;
            return;
        }
        for (Part& p : parts) drop_part(p);
        parts.clear();
        if (va && getenv("PGX_VA_FREE")) (void)hipMemAddressFree(va, bytes);
        else if (va && va_reuse()) {
            std::lock_guard<std::mutex> lock(g_va_mu);
            g_va_free.emplace(bytes, va);
        }
        va = nullptr;
    }
};

// The probes run on a private NON-BLOCKING stream per device: the legacy null stream would synchronise implicitly with
// every blocking stream of the process.  Created once per device, kept for the life of the process.
hipStream_t probe_stream(int device) {
    static std::mutex mu;
    static std::map<int, hipStream_t> streams;
    std::lock_guard<std::mutex> lock(mu);
    auto it = streams.find(device);
    if (it != streams.end()) return it->second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        s = nullptr;  // fall back to the null stream
    }
    streams[device] = s;
    return s;
}

}  // namespace

struct pgx_buffers {
    int device = 0;
    size_t bytes = 0, first = 0, second = 0;  // requested bytes; sizes of the two physical halves
    int count = 0;
    size_t stride = 0;   // bytes between consecutive buffers: round_up(bytes, 2 MiB)
    Range all;           // ONE virtual range: buffer i at all.va + i * stride, parts 2i (first half) and 2i + 1
    void* va_of(int i) const { return (char*)all.va + (size_t)i * stride; }
    pgx_buffers_info info{};
};

namespace {

// average duration of the probe stream: `half_bytes` into `lo` and `half_bytes` into `hi`, concurrently
hipError_t probe_us(hipStream_t st, void* lo, void* hi, size_t half_bytes, float* us) {
    const int nblk = 8192;
    const size_t per_block = half_bytes / 16 / (nblk / 2);
    hipEvent_t a, b;
    hipError_t e = hipEventCreate(&a);
    if (e != hipSuccess) return e;
    e = hipEventCreate(&b);
    if (e != hipSuccess) {
        (void)hipEventDestroy(a);
        return e;
    }
    (void)hipGetLastError();
    const int reps = 6;
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL(pgx::zone_probe_kernel, dim3(nblk), dim3(64), 0, st, (pgx::f32x4*)lo, (pgx::f32x4*)hi, per_block, nblk);
    (void)hipEventRecord(a, st);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(pgx::zone_probe_kernel, dim3(nblk), dim3(64), 0, st, (pgx::f32x4*)lo, (pgx::f32x4*)hi, per_block, nblk);
    (void)hipEventRecord(b, st);
    e = hipEventSynchronize(b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (e == hipSuccess) e = hipGetLastError();
    *us = ms * 1000.f / reps;
    return e;
}

void destroy(pgx_buffers* p) {
    p->all.release();
    delete p;
}

// one allocation of `bytes` at its own virtual address, accessible: a probe chunk of the zone walk.  Plain hipMalloc
// memory (the same VRAM allocator as hipMemCreate, so it marks where the allocator stands just as well) whose address
// the runtime takes back on hipFree -- see the note on address ranges above.  PGX_CHUNK_VMM=1: the round-2 form
// (VMM handle mapped into a reservation that is then leaked), for A/B.
hipError_t make_chunk(int device, size_t bytes, Range& r) {
    static const bool vmm = getenv("PGX_CHUNK_VMM") != nullptr;
    if (!vmm) {
        void* ptr = nullptr;
        const hipError_t e = hipMalloc(&ptr, bytes);
        if (e != hipSuccess) return e;
        r.va = ptr;
        r.bytes = bytes;
        r.plain = true;
        return hipSuccess;
    }
    hipError_t e = r.reserve(bytes);
    if (e != hipSuccess) return e;
    r.parts.resize(1);
    e = map_part(device, r.va, bytes, r.parts[0]);
    if (e == hipSuccess) e = grant_access(device, r.va, bytes);
    if (e != hipSuccess) r.release();
    return e;
}

// physical memory only, mapped nowhere: a spacer is never read or written, it just keeps the allocator from handing the
// same pages out again (hipMemCreate commits the memory at once) -- and takes no address space
hipError_t make_spacer(int device, size_t bytes, Range& r) {
    r.parts.resize(1);
    const hipMemAllocationProp prop = device_prop(device);
    const hipError_t e = hipMemCreate(&r.parts[0].handle, bytes, &prop, 0);
    if (e != hipSuccess) {
        r.parts.clear();
        return e;
    }
    r.parts[0].bytes = bytes;
    r.parts[0].live = true;
    r.bytes = bytes;
    return hipSuccess;
}

// Walks the allocator into another zone.  A reference chunk is allocated first (right behind the buffers' first
// halves: their zone); then, after every spacer, a candidate chunk; the probe stream writes half of its bytes into the
// reference and half into the candidate.  Nothing is freed during the walk (a freed chunk would be handed out again
// and the walk would stand still): `held` keeps spacers, reference and candidates until the caller has allocated the
// second halves.  On return with *found the allocator sits right behind a candidate that lies in another zone.
void find_other_zone(int device, size_t budget, size_t skip, std::vector<Range>& held, pgx_buffers_info& info, bool* found) {
    *found = false;
    hipStream_t st = probe_stream(device);
    held.emplace_back();
    if (make_chunk(device, PROBE_HALF, held.back()) != hipSuccess) { held.pop_back(); return; }
    void* ref = held.back().va;
    held.emplace_back();
    if (make_chunk(device, PROBE_HALF, held.back()) != hipSuccess) { held.pop_back(); return; }
    float t_same = 0.f;
    if (probe_us(st, ref, held.back().va, PROBE_HALF, &t_same) != hipSuccess) return;
    info.same_zone_us = info.final_us = t_same;
    // A one-zone stream sustains 5.4-6.2 TB/s on the devices measured, a 1:1 two-zone stream 6.7-7.0 TB/s: a pair that
    // already reaches 6.6 TB/s straddles a zone boundary as it is -- the allocator stands in the other zone already.
    const float t_spread_abs = (float)(2.0 * (double)PROBE_HALF / 6.6e12 * 1e6);
    if (t_same <= t_spread_abs && skip == 0 && !getenv("PGX_ZONE_SCAN")) {  // (the scan diagnostic maps the whole budget)
        *found = true;
        return;
    }
    auto read_diag = [&]() {
    if (getenv("PGX_ZONE_PAIRS")) {  // diagnostic: how fast does this process READ the same chunks?
            float* sink = nullptr;
            if (hipMalloc((void**)&sink, 4) == hipSuccess && held.size() >= 2) {
                const int nblk = 8192;
                const size_t per_block = PROBE_HALF / 16 / (nblk / 2);
                hipEvent_t a, b;
                (void)hipEventCreate(&a);
                (void)hipEventCreate(&b);
                void* hi = held.back().va ? held.back().va : held[1].va;
                for (int i = 0; i < 2; ++i)
                    hipLaunchKernelGGL(pgx::zone_read_kernel, dim3(nblk), dim3(64), 0, st, (const pgx::f32x4*)ref, (const pgx::f32x4*)hi, per_block, nblk, sink);
                (void)hipEventRecord(a, st);
                for (int i = 0; i < 6; ++i)
                    hipLaunchKernelGGL(pgx::zone_read_kernel, dim3(nblk), dim3(64), 0, st, (const pgx::f32x4*)ref, (const pgx::f32x4*)hi, per_block, nblk, sink);
                (void)hipEventRecord(b, st);
                (void)hipEventSynchronize(b);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, a, b);
                fprintf(stderr, "[pgx_buffers] READ stream over reference + last candidate: %.1f us (%.2f TB/s)\n", ms * 1000.f / 6.f,
                        2.0 * (double)PROBE_HALF / (ms / 6.0 * 1e-3) / 1e12);
                (void)hipEventDestroy(a);
                (void)hipEventDestroy(b);
                (void)hipFree(sink);
            }
        }
    };
    size_t spacer_bytes = 0;
    // `skip`: a previous attempt ended this far into the walk and its buffers did not deliver -- pass over that stretch
    while (spacer_bytes < skip && spacer_bytes + SPACER <= budget) {
        held.emplace_back();
        if (make_spacer(device, SPACER, held.back()) != hipSuccess) { held.pop_back(); (void)hipGetLastError(); return; }
        spacer_bytes += SPACER;
        info.spacer_gib = (double)spacer_bytes / (double)GiB;
    }
    int weak_run = 0;
    bool prev_strong = false;
    while (spacer_bytes + SPACER <= budget) {
        held.emplace_back();
        if (make_spacer(device, SPACER, held.back()) != hipSuccess) { held.pop_back(); (void)hipGetLastError(); break; }
        spacer_bytes += SPACER;
        info.spacer_gib = (double)spacer_bytes / (double)GiB;
        held.emplace_back();
        if (make_chunk(device, PROBE_HALF, held.back()) != hipSuccess) { held.pop_back(); (void)hipGetLastError(); break; }
        float t = 0.f;
        if (probe_us(st, ref, held.back().va, PROBE_HALF, &t) != hipSuccess) break;
        info.candidates += 1;
        if (getenv("PGX_DEBUG")) fprintf(stderr, "[pgx_buffers] %4.0f GiB of spacers: candidate %.1f us (same-zone pair %.1f us)\n", (double)spacer_bytes / (double)GiB, t, t_same);
        // Two kinds of other zone were seen (profiles/r2/placement_walk_scan.txt): one pairs with the reference at
        // ~6.75 TB/s, one at ~6.3-6.4 TB/s (~90 GiB wide, its candidates scatter around 0.88-0.92 of the same-zone time).
        // The first kind is taken at once; the second only after 14 more spacers (112 GiB) have failed to reach the
        // first, when the walk has clearly left it again (a candidate back at the same-zone time), or when the budget
        // ends while still in it.
        constexpr int WEAK_LIMIT = 14;
        const bool strong = t <= t_spread_abs, weak = t < 0.94f * t_same;
        if (weak && !strong) weak_run += 1;
        else if (t >= 0.97f * t_same && weak_run > 0) weak_run = WEAK_LIMIT;  // left it without finding better: take the next one
        const bool last = spacer_bytes + 2 * SPACER > budget;
        // A strong candidate may itself straddle the boundary, and fast stretches narrower than a spacer exist: it
        // takes a second strong candidate, one spacer further, to accept (what is allocated next then lies inside the
        // zone); a lone one is walked past.
        const bool confirmed = strong && (prev_strong || last);
        prev_strong = strong;
        if (!getenv("PGX_ZONE_SCAN") && (confirmed || (weak && !strong && (weak_run > WEAK_LIMIT || last)))) {
            info.final_us = t;
            *found = true;
            read_diag();
            return;
        }
    }
    read_diag();
    if (getenv("PGX_ZONE_PAIRS")) {  // diagnostic: nothing pairs with the reference -- do the candidates pair with each other?
        std::vector<void*> c;
        for (size_t i = 1; i < held.size(); ++i)
            if (held[i].bytes == PROBE_HALF) c.push_back(held[i].va);  // held[1] = baseline candidate, then one per spacer
        fprintf(stderr, "[pgx_buffers] pair scan over %zu candidates (every 3rd), us:\n", c.size());
        for (size_t i = 0; i < c.size(); i += 3) {
            fprintf(stderr, "[pgx_buffers]   %3zu:", i);
            for (size_t j = 0; j < c.size(); j += 3) {
                float t = 0.f;
                if (j <= i || probe_us(st, c[i], c[j], PROBE_HALF, &t) != hipSuccess) fprintf(stderr, "     .");
                else fprintf(stderr, " %5.0f", t);
            }
            fprintf(stderr, "\n");
        }
    }
}

}  // namespace

extern "C" {

int pgx_buffers_create(int device, size_t bytes, int count, double max_spacer_gib, pgx_buffers** out) {
    return pgx_buffers_create_at(device, bytes, count, 0.0, max_spacer_gib, out);
}

int pgx_buffers_create_at(int device, size_t bytes, int count, double skip_gib, double max_spacer_gib, pgx_buffers** out) {
    if (!out) return pgx::fail_msg(PGX_E_INVALID, "pgx_buffers_create: null argument");
    *out = nullptr;
    if (bytes == 0 || count < 1 || count > 64) return pgx::fail_msg(PGX_E_INVALID, "pgx_buffers_create: bad size or count");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess)
        return pgx::fail_msg(PGX_E_HIP, "cannot select HIP device %d", device);
    pgx_buffers* p = new (std::nothrow) pgx_buffers();
    if (!p) {
        (void)hipSetDevice(prev);
        return pgx::fail_msg(PGX_E_NOMEM, "out of host memory");
    }
    p->device = device;
    p->bytes = bytes;
    p->count = count;
    const size_t total = round_up(bytes, GRANULE);
    p->first = total >= 2 * GRANULE ? round_up(total / 2, GRANULE) : total;
    p->second = total - p->first;
    p->stride = total;
    p->info.bytes = (int64_t)bytes;
    p->info.count = count;
    std::vector<Range> held;  // spacers, reference and candidate chunks of the zone walk
    auto bail = [&](int code, const char* what, hipError_t e) {
        const std::string msg = std::string(what) + ": " + hipGetErrorString(e);
        for (Range& s : held) s.release();
        destroy(p);
        (void)hipGetLastError();
        (void)hipSetDevice(prev);
        return pgx::fail_msg(code, "%s", msg.c_str());
    };
    auto code_of = [](hipError_t e) { return e == hipErrorOutOfMemory ? PGX_E_NOMEM : PGX_E_HIP; };
    // 1. one virtual range for all buffers (a rollout's observation ring needs a constant stride) + first halves (the
    //    zone the allocator is in right now)
    {
        const hipError_t e = p->all.reserve((size_t)count * total);
        if (e != hipSuccess) return bail(PGX_E_NOMEM, "hipMemAddressReserve", e);
        p->all.parts.resize((size_t)2 * count);
    }
    for (int i = 0; i < count; ++i) {
        const hipError_t e = map_part(device, p->va_of(i), p->first, p->all.parts[2 * i]);
        if (e != hipSuccess) return bail(code_of(e), "hipMemCreate/hipMemMap (first half)", e);
    }
    // 2. walk the allocator into another zone.  Buffers below 128 MiB are left alone: a repeated stream of that size is
    //    absorbed by the 256 MiB Infinity Cache and too short to be bandwidth-bound (configs[3]'s 190 MB buffers: two
    //    alternating ones exceed the cache together, the spread is worth 2-3 %).
    bool found = false;
    const size_t walk_min = getenv("PGX_ZONE_MIN_MB") ? (size_t)atol(getenv("PGX_ZONE_MIN_MB")) << 20 : (size_t)128 << 20;
    if (p->second && total >= walk_min && max_spacer_gib > 0.0) {
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const size_t need = (size_t)count * p->second + 20 * PROBE_HALF + 4 * GiB;
        size_t budget = (size_t)(max_spacer_gib * (double)GiB);
        budget = std::min(budget, free_b > need ? (free_b - need) / 10 * 9 : 0);
        const size_t skip = skip_gib > 0.0 ? (size_t)(skip_gib * (double)GiB) : 0;
        find_other_zone(device, budget, skip, held, p->info, &found);
        (void)hipGetLastError();
    }
    // 3. second halves, right behind the last candidate of the walk (everything the walk allocated is still held)
    for (int i = 0; i < count && p->second; ++i) {
        const hipError_t e = map_part(device, (char*)p->va_of(i) + p->first, p->second, p->all.parts[2 * i + 1]);
        if (e != hipSuccess) return bail(code_of(e), "hipMemCreate/hipMemMap (second half)", e);
    }
    {
        const hipError_t e = grant_access(device, p->all.va, (size_t)count * total);
        if (e != hipSuccess) return bail(PGX_E_HIP, "hipMemSetAccess", e);
    }
    for (Range& s : held) s.release();
    p->info.spread = found ? 1 : 0;
    // What the buffers themselves sustain (slowest one), for the record only: the probe's 8192 chunks grow with the
    // buffer, and beyond ~1 GB this stream shape saturates near 6.1 TB/s wherever the pages are (configs[4]: the
    // step kernel runs 23 % faster on spread buffers while this figure does not move), so it is not used as a judge.
    if (p->second && total >= walk_min) {
        double worst = 0.0;
        for (int i = 0; i < count; ++i) {
            float t = 0.f;
            if (probe_us(probe_stream(device), p->va_of(i), (char*)p->va_of(i) + p->first, p->second, &t) != hipSuccess || t <= 0.f) continue;
            const double gbs = 2.0 * (double)p->second / ((double)t * 1e-6) / 1e9;
            if (worst == 0.0 || gbs < worst) worst = gbs;
        }
        p->info.buffer_gbs = (float)worst;
    }
    (void)hipGetLastError();
    *out = p;
    (void)hipSetDevice(prev);
    return PGX_OK;
}

void* pgx_buffers_ptr(pgx_buffers* p, int index) {
    if (!p || index < 0 || index >= p->count) return nullptr;
    return p->va_of(index);
}

int64_t pgx_buffers_stride(pgx_buffers* p) { return p ? (int64_t)p->stride : 0; }

int pgx_buffers_drop(pgx_buffers* p, int index) {
    if (!p || index < 0 || index >= p->count) return pgx::fail_msg(PGX_E_INVALID, "pgx_buffers_drop: bad argument");
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();  // nothing may still be writing into the buffer
    drop_part(p->all.parts[(size_t)2 * index]);
    drop_part(p->all.parts[(size_t)2 * index + 1]);
    if (prev >= 0) (void)hipSetDevice(prev);
    return PGX_OK;
}

int pgx_buffers_get_info(pgx_buffers* p, pgx_buffers_info* info) {
    if (!p || !info) return pgx::fail_msg(PGX_E_INVALID, "pgx_buffers_get_info: null argument");
    *info = p->info;
    return PGX_OK;
}

int64_t pgx_buffers_va_reserved(void) {
    std::lock_guard<std::mutex> lock(g_va_mu);
    return (int64_t)g_va_reserved;
}

int pgx_buffers_destroy(pgx_buffers* p) {
    if (!p) return PGX_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();  // nothing may still be writing into the ranges that are about to be unmapped
    destroy(p);
    if (prev >= 0) (void)hipSetDevice(prev);
    return PGX_OK;
}

}  // extern "C"
