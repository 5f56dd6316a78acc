// vd_common.hpp — shared host/device plumbing of libvoidin_hip.so (gfx950 only).
//
// Strict fp32 everywhere: the library is compiled with -ffp-contract=off so that a*b+c is
// never fused (rustc / the reference never contract), with hipcc's default
// correctly-rounded f32 divide/sqrt and f32 denormals kept.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/voidin_abi.h"

struct VdCtx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;      // stream in use (own or caller's)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_mid = nullptr, ev_aux = nullptr;
    bool timed_mid = false;
    bool timed = false;
    bool timing_enabled = false;        // event pairs around kernels cost a few us of GPU idle each: opt-in
    char err[512] = {0};
    int num_cus = 256;
    int cull_variant = 0;               // kernel variant for A/B tuning (VD_OPT_CULL_VARIANT)
    unsigned split_min = 2u << 20;      // inputs of at least this many instances run the split form of the cull
                                        // (VD_OPT_CULL_SPLIT_MIN): below, the fused single launch is faster (tools/ab_split_min.py:
                                        // 1 Mi 39 vs 47 us, 2 Mi 65 vs 65, 3 Mi 97 vs 93, 10 M 312 vs 256)
    long long opt[VD_OPT_COUNT_];       // vd_ctx_set_option: -1 = default (set in vd_ctx_create)
    long long option(int o, long long dflt) const { return opt[o] < 0 ? dflt : opt[o]; }

    // grow-only device scratch arenas
    void* scratch = nullptr;     size_t scratch_bytes = 0;     // general purpose
    void* scan_state = nullptr;  size_t scan_state_bytes = 0;  // look-back granules + ticket word (epoch-tagged)
    void* expand_state = nullptr; size_t expand_state_bytes = 0;  // mask_scan_kernel: done counter + chunk offsets
    void* trace_ovf = nullptr;   size_t trace_ovf_bytes = 0;      // traversal: one bit per ray of the call = "its 128-entry stack overflowed"
    bool trace_ovf_dirty = false;                                 // a call left early (error) and may have left bits set
    void* trace_deep = nullptr;  size_t trace_deep_bytes = 0;     // traversal, second pass over those rays: their list + stack entries beyond 128 (allocated when first needed)
    void* refit_state = nullptr; size_t refit_state_bytes = 0;    // TLAS refit: epoch-tagged {parent, sibling} links + arrival words
    unsigned refit_epoch = 0;                                     // tag of the last refit launch (0 = the arena is freshly zeroed)
    const void* fan_tlas = nullptr; unsigned fan_idle_calls = 0;  // traversal fan-out: top level of the last call that tried it, calls left to run without it
    unsigned fan_nodes = 0, fan_inst = 0;                         // ... and its node / instance counts: the verdict is about THAT scene
    void fan_forget(const void* tlas) { if (tlas == fan_tlas) { fan_tlas = nullptr; fan_idle_calls = 0; } }   // the top level at this address was rebuilt / refitted / released
    unsigned refit_n = 0;                                         // instance count the arena's layout was last used with
    unsigned long long scan_launches = 0;
    void* dbg_ptr = nullptr; unsigned dbg_count = 0; void* dbg_ptr2 = nullptr;   // tuning hooks
    void* stage_in = nullptr;    size_t stage_in_bytes = 0;    // host-pointer API staging
    void* stage_out = nullptr;   size_t stage_out_bytes = 0;
    void* stage_aux = nullptr;   size_t stage_aux_bytes = 0;
    uint32_t* host_pinned = nullptr;                           // 64 u32 of pinned host memory
    unsigned* fault_dev = nullptr;                             // device address of host_pinned[kScanFaultWord] (vd_scan_check_fault)
    void* host_stage = nullptr;  size_t host_stage_bytes = 0;  // grow-only pinned staging (BLAS top tree)
    void* lvl_pinned = nullptr; hipEvent_t ev_lvl[2] = {nullptr, nullptr};   // BLAS level loop: control words of the last two levels (pinned) + their events
    hipStream_t aux_stream = nullptr;                          // copies that overlap a kernel on `stream`
    VdBvhBuildStats bvh_stats = {};                            // vd_bvh_last_build_stats
    bool tlas_chain_lds_opt_in[2] = {false, false};              // tlas_build_kernel<VdTlasNode / VdTlasNodeWide>: dynamic LDS for the slot arrays
    bool mid_lds_opt_in = false;                               // blas_mid_kernel's dynamic-LDS attribute set on this device
    bool tlas_ix_lds_opt_in[8] = {false, false, false, false, false, false, false, false};  // tlas_build_indexed_kernel<VdTlasNode / VdTlasNodeWide> x {plain, with helper waves}, likewise
};

// Every extern "C" entry point runs on the context's device: allocations, event records and launches otherwise go to
// the calling thread's CURRENT device (another thread, or torch.cuda.set_device(j) after the ctx was made for k).
struct VdDeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit VdDeviceGuard(const VdCtx* ctx) {
        if (!ctx) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess;
    }
    ~VdDeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    VdDeviceGuard(const VdDeviceGuard&) = delete;
    VdDeviceGuard& operator=(const VdDeviceGuard&) = delete;
};

#define VD_HIP_CHECK(ctx, call)                                                              \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d %s -> %s", __FILE__, __LINE__,   \
                     #call, hipGetErrorString(e_));                                          \
            (void)hipGetLastError();   /* reported here: a later launch check must not find it again */ \
            return VD_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define VD_FAIL(ctx, code, msg)                                          \
    do {                                                                 \
        snprintf((ctx)->err, sizeof((ctx)->err), "%s", (msg));           \
        return (code);                                                   \
    } while (0)

// Grow-only arena helper. Never called between a kernel's enqueue and its completion on the
// same buffer without a stream sync (callers sync before growing).
int vd_ensure(VdCtx* ctx, void** buf, size_t* cur, size_t need);
int vd_ensure_host(VdCtx* ctx, size_t need);   // ctx->host_stage: grow-only pinned host memory
// tlas.hip: the agglomerative build of tlas.rs:56-105 over ready leaf boxes (six floats {min xyz, max xyz} per leaf; n <= 32 768)
int vd_tlas_build_from_boxes(VdCtx* ctx, const float* d_boxes, uint32_t n, VdTlasNode* d_nodes);
// Look-back scan state for n_tiles tiles: *ticket = 64-bit {epoch | ticket} word, *states = granules.
// Zeroed once when (re)allocated; the epoch tags make per-launch clearing unnecessary.
int vd_scan_scratch(VdCtx* ctx, unsigned n_tiles, unsigned long long** ticket, unsigned long long** states, bool start_timer);
// Has a scan of an earlier launch on this context given up (fault word in pinned host memory)?  Reports it ONCE: VD_ERR_HIP
// with a message, after draining the stream and zeroing the scan state; VD_OK otherwise.  Called at the entry of every vd_*
// that launches a look-back scan, so a renderer that only uses the *_dev forms hears about it on its next frame.
int vd_scan_check_fault(VdCtx* ctx);
static inline unsigned* vd_scan_fault_word(VdCtx* ctx) { return ctx->fault_dev; }

static inline void vd_time_begin(VdCtx* ctx) {
    ctx->timed_mid = false;
    if (ctx->timing_enabled) (void)hipEventRecord(ctx->ev_start, ctx->stream);
}
static inline void vd_time_mid(VdCtx* ctx) {   // boundary between the two passes of a split call
    if (ctx->timing_enabled) { (void)hipEventRecord(ctx->ev_mid, ctx->stream); ctx->timed_mid = true; }
}
static inline void vd_time_end(VdCtx* ctx) {
    if (!ctx->timing_enabled) { ctx->timed = false; return; }
    (void)hipEventRecord(ctx->ev_stop, ctx->stream);
    ctx->timed = true;
}

#ifdef __HIPCC__
// ---- device helpers ------------------------------------------------------------------

// Order-preserving float key (-0 < +0) and Rust's f32::min / f32::max on top of it (glam 0.24 scalar
// Vec3::min/max, crates/bvh/src/blas.rs:190-198, tlas.rs:43,69-70,96-97): a NaN operand is IGNORED - the other
// operand is returned - and -0 < +0 is the tie rule Rust leaves open, which makes bounds
// reductions order-independent and bit-exact against the oracle.  Key reductions (integer atomic min / max) skip
// NaNs through vd_key_lo / vd_key_hi: a NaN maps to the neutral element of the reduction.
__device__ __forceinline__ int vd_key(float f) {
    int i = __float_as_int(f);
    return i ^ ((i >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float vd_unkey(int k) {
    return __int_as_float(k ^ ((k >> 31) & 0x7fffffff));
}
__device__ __forceinline__ int vd_key_lo(float f) { return f != f ? 0x7fffffff : vd_key(f); }        // operand of a min
__device__ __forceinline__ int vd_key_hi(float f) { return f != f ? (int)0x80000000 : vd_key(f); }   // operand of a max
// IEEE minNum / maxNum as gfx950 executes them - one v_min_f32 / v_max_f32: a QUIET NaN operand is ignored, both NaN ->
// NaN, and -0 orders below +0 (the ISA's LT_NEG_ZERO compare).  Exactly the semantics above, without the
// compare-and-select chains (which the compiler turned into six divergent branches per box union: the refit climb went
// from 0.10 to 0.18 ms).  A SIGNALLING NaN is different: in IEEE mode the instruction quiets and RETURNS it, and the
// compiler does not canonicalise the operands of fminf for us (LLVM leaves minnum(sNaN, x) open) - so raw user floats
// pass through vd_quiet once where they are loaded (blas_precompute_kernel); everything arithmetic produces is quiet.
__device__ __forceinline__ float vd_quiet(float x) { return __builtin_canonicalizef(x); }
__device__ __forceinline__ float vd_min_to(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float vd_max_to(float a, float b) { return __builtin_fmaxf(a, b); }

// crates/bvh/src/intersection.rs:16-19
__device__ __forceinline__ float vd_area(float dx, float dy, float dz) {
    return (dx * dy + dx * dz + dy * dz) * 2.0f;
}

__device__ __forceinline__ unsigned vd_lane() { return __lane_id(); }

// count of set bits of `mask` strictly below this lane
__device__ __forceinline__ unsigned vd_mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Compiler-level ordering for same-wave LDS exchange (DS ops of one wave execute in order).
__device__ __forceinline__ void vd_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef unsigned long long vd_u64;
#define VD_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// ---- single-pass ordered scan across workgroups ("decoupled look-back") ----------------
// tile_state[t] is one naturally aligned 8-byte {epoch:30 | status:2 | value:32} granule written
// by ONE agent-scope store and polled by agent-scope loads (write-through / L1-bypassing on
// gfx950), so the data is its own flag and no fence is needed.  A granule whose epoch is not the
// launch's epoch is INVALID, so nothing has to be zeroed between launches: the 64-bit ticket
// word {epoch:32 | next ticket:32} hands out both, and the workgroup that draws the last ticket
// re-arms it for the next launch ({epoch + 1, 0}) - the state advances ON THE DEVICE, so a launch
// recorded in a HIP graph replays correctly (tests/test_gpu_frame_loop.py).  Tickets come from an
// atomic counter, so every predecessor of a running tile is itself running or finished: no
// residency assumption.
// Every wait is BOUNDED, by the wall clock: a tile whose predecessor does not show up within kScanTimeoutTicks of the
// constant 100 MHz counter (two seconds; an ordinary wait is microseconds) - ONE deadline for both of its waiting loops -
// publishes VD_TILE_POISON and returns VD_SCAN_STUCK, and every later tile that runs into the poison does the same at once.
// Giving up is STICKY for the launch: the tile also stores the launch's epoch + 1 into the arena's stuck word, and the last
// tile looks at that word after its own look-back - a tile that had read the quitter's AGGREGATE before it was poisoned and
// then completed normally would otherwise hide the failure behind its INCLUSIVE.  What the last tile then does is
// the kernel's business (cull.hip: count = 0, no list, the fault word in pinned host memory raised - the next vd_* call on the
// context reports VD_ERR_HIP and zeroes the scan state); a launch that lost its LAST tile keeps the count the first tile
// pre-stored, VD_SCAN_STUCK.
// Arena layout (vd_scan_scratch): u64[0] ticket word, u64[1] tuning hook (vd_debug_scan_fault), u64[2] stuck word,
// u64[3] spare, granules from byte 32: the stuck word of a granule array is tile_state[-2].
enum : unsigned { VD_TILE_AGGREGATE = 1u, VD_TILE_INCLUSIVE = 2u, VD_TILE_POISON = 3u };
constexpr unsigned VD_SCAN_STUCK = 0xffffffffu;
constexpr unsigned long long kScanTimeoutTicks = 200000000ull;     // wall_clock64(): 100 MHz on gfx950 -> 2 s
constexpr int kScanFaultWord = 63;                                 // VdCtx::host_pinned[63]: raised by a launch whose scan gave up

__device__ __forceinline__ vd_u64 vd_tile_pack(unsigned epoch, unsigned status, unsigned value) {
    return ((vd_u64)(epoch & 0x3fffffffu) << 34) | ((vd_u64)status << 32) | value;
}

// ONE lane: draw a ticket.  Returns the ticket; *epoch = this launch's epoch.
__device__ __forceinline__ unsigned vd_take_ticket(vd_u64* ticket_word, unsigned n_tiles, unsigned* epoch) {
    const vd_u64 t = __hip_atomic_fetch_add(ticket_word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned ticket = (unsigned)t, ep = (unsigned)(t >> 32);
    *epoch = ep;
    if (ticket == n_tiles - 1u)   // every ticket of this launch is out: re-arm for the next one
        __hip_atomic_store(ticket_word, (vd_u64)(ep + 1u) << 32, VD_RLX_AGENT);
    return ticket;
}

__device__ __forceinline__ bool vd_scan_gave_up(vd_u64* tile_state, unsigned epoch) {      // has any tile of this launch given up?
    return __hip_atomic_load(tile_state - 2, VD_RLX_AGENT) == (vd_u64)(epoch & 0x3fffffffu) + 1ull;
}

// Called by ONE full wave of the workgroup. Returns the exclusive prefix of tile `t`
// (sum of `total` over tiles < t) in every lane, after publishing this tile's state - or VD_SCAN_STUCK.
// `first` marks the first tile of a segment (segmented scan): its prefix restarts at 0.
__device__ __forceinline__ unsigned vd_lookback(vd_u64* tile_state, unsigned epoch, unsigned t, unsigned total, bool first = false) {
    const unsigned lane = vd_lane();
    const unsigned ep = epoch & 0x3fffffffu;
    if (t == 0 || first) {
        if (lane == 0) __hip_atomic_store(&tile_state[t], vd_tile_pack(ep, VD_TILE_INCLUSIVE, total), VD_RLX_AGENT);
        return 0u;
    }
    if (lane == 0) __hip_atomic_store(&tile_state[t], vd_tile_pack(ep, VD_TILE_AGGREGATE, total), VD_RLX_AGENT);
    const unsigned long long deadline = wall_clock64() + kScanTimeoutTicks;      // one deadline for both loops below: they expire together
    unsigned exclusive = 0u;
    int look = (int)t - 1;
    bool stuck = false;
    {   // tiles finish roughly in ticket order: wait for the immediate predecessor with ONE lane
        // (one 8-byte poll per iteration instead of a 64-granule window), then sweep the window
        if (lane == 0) {
            vd_u64 s0 = __hip_atomic_load(&tile_state[look], VD_RLX_AGENT);
            unsigned spins = 0;
            while ((unsigned)(s0 >> 34) != ep || ((unsigned)(s0 >> 32) & 3u) == 0u) {
                if ((++spins & 63u) == 0u && wall_clock64() > deadline) { stuck = true; break; }
                __builtin_amdgcn_s_sleep(8);
                s0 = __hip_atomic_load(&tile_state[look], VD_RLX_AGENT);
            }
        }
        stuck = __builtin_amdgcn_readfirstlane((int)stuck) != 0;
    }
    while (!stuck) {
        const int idx = look - (int)lane;
        vd_u64 s = vd_tile_pack(ep, VD_TILE_INCLUSIVE, 0u);   // virtual tiles before 0
        bool mine_stuck = false;
        if (idx >= 0) {
            s = __hip_atomic_load(&tile_state[idx], VD_RLX_AGENT);
            unsigned spins = 0;
            while ((unsigned)(s >> 34) != ep || ((unsigned)(s >> 32) & 3u) == 0u) {   // not written in this launch yet
                if ((++spins & 63u) == 0u && wall_clock64() > deadline) { mine_stuck = true; break; }
                __builtin_amdgcn_s_sleep(1);
                s = __hip_atomic_load(&tile_state[idx], VD_RLX_AGENT);
            }
        }
        const unsigned status = (unsigned)(s >> 32) & 3u;
        const unsigned value = (unsigned)s;
        const unsigned long long incl = __ballot(!mine_stuck && status == VD_TILE_INCLUSIVE);
        // lanes at or before the first INCLUSIVE one (closest predecessors first) contribute
        const unsigned first_incl = incl ? (unsigned)__builtin_ctzll(incl) : 63u;
        if (__ballot(lane <= first_incl && (mine_stuck || status == VD_TILE_POISON))) { stuck = true; break; }   // a predecessor gave up, or never came
        unsigned v = lane <= first_incl ? value : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        exclusive += v;
        if (incl) break;
        look -= 64;
    }
    if (stuck) {
        if (lane == 0) {
            __hip_atomic_store(tile_state - 2, (vd_u64)ep + 1ull, VD_RLX_AGENT);      // sticky for the launch: the last tile reads it
            __hip_atomic_store(&tile_state[t], vd_tile_pack(ep, VD_TILE_POISON, 0u), VD_RLX_AGENT);
        }
        return VD_SCAN_STUCK;
    }
    if (lane == 0) __hip_atomic_store(&tile_state[t], vd_tile_pack(ep, VD_TILE_INCLUSIVE, exclusive + total), VD_RLX_AGENT);
    return exclusive;
}

// What the LAST tile of a launch stores as the count (one lane): the total - or, when some tile of the launch gave up, 0
// (no list was written: nothing to draw) while the fault word in pinned host memory is raised for the host side.
__device__ __forceinline__ unsigned vd_scan_final_count(vd_u64* tile_state, unsigned epoch, unsigned excl, unsigned tile_total, unsigned* fault_host) {
    if (excl != VD_SCAN_STUCK && !vd_scan_gave_up(tile_state, epoch)) return excl + tile_total;
    __hip_atomic_store(fault_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return 0u;
}
#endif  // __HIPCC__
