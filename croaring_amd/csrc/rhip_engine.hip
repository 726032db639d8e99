// rhip_engine.hip -- host side of libroaring_hip.so: device pools, the batched pairwise
// pipeline (plan -> typed kernels -> directory compaction), portable (de)serialization at the
// boundary, and the C ABI declared in include/roaring_hip.h.
//
// There is deliberately no CPU implementation of any set operation in this file: without a
// usable HIP device every entry point fails (rhip_ctx_create returns NULL).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/roaring_hip.h"
#include "rhip_kernels.h"
#include "rhip_many.h"
#include "rhip_heap.h"
#include "rhip_poolops.h"
#include "rhip_serial.h"
#include "rhip_deser.h"
#include "rhip_frozen.h"
#include "rhip_values.h"
#include "rhip_flip.h"
#include "rhip_prims.h"

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static void set_err(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
// status of the last failed pointer-returning entry point on this thread (they can only return NULL)
static int& last_status() {
    static thread_local int s = 0;
    return s;
}
#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_err("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            throw (int)RHIP_ERR_DEVICE;                                                \
        }                                                                              \
    } while (0)

// ------------------------------------------------------------------ device buffers
static bool release_all_arena_spares();  // (the spare result arenas of every live context: called when an allocation fails)
// tests (rhip_debug_fail_allocs): the next g_fail_allocs device allocations of the library fail as if the device were out
// of memory, after g_fail_skip have been let through
static std::atomic<int> g_fail_skip{0}, g_fail_allocs{0};
static hipError_t dev_malloc(void** p, size_t n) {
    if (g_fail_allocs.load(std::memory_order_relaxed) > 0) {
        if (g_fail_skip.load() > 0) --g_fail_skip;
        else { --g_fail_allocs; *p = nullptr; return hipErrorOutOfMemory; }
    }
    return hipMalloc(p, n);
}
struct DBuf {
    void* p = nullptr;
    size_t cap = 0;
    void* base = nullptr;  // the allocation; p = base + skew
    size_t skew = 0;       // set before the first ensure(): see rhip_ctx_s::arena_skew
    size_t round_to = 0;   // allocations at least this large are rounded up to a multiple of it
    bool pow2_large = false;  // allocations of 1 GiB and more take the next power of two, no slack (see rhip_ctx_s::arena_pow2)
    bool exact = false;       // no slack at all (place_arena: the candidates' sizes decide how the driver composes them)
    uint64_t gen = 0;      // bumped by every (re)allocation: "is this still the memory I initialised?"
    // a result arena the placement search chose or probed (place_arena): against which operand arena (address + generation)
    // and at what probe rate -- so that a parked arena can be taken back for the same operand without probing again
    const void* placed_for = nullptr;
    uint64_t placed_for_gen = 0;
    float placed_gbps = 0.f;
    bool placed_winner = false;  // the arena a search CHOSE (a parked winner is taken back as it is; a loser kept as a spare only if it streamed at the bar)
    // A result arena placed by ADDRESS (place_arena_va): physical memory of its own (hipMemCreate) mapped at `base`, somewhere
    // inside an address range reserved for it -- the memory is released by unmapping, not by hipFree
    bool vmm = false;
    hipMemGenericAllocationHandle_t* vmm_handles = nullptr;  // chunks of vmm_chunk bytes (the last one may be shorter), owned with `base`
    uint32_t vmm_n = 0;
    size_t vmm_chunk = 0;
    void* va_base = nullptr;  // the reserved range
    size_t va_len = 0, map_len = 0;
    void free_mem() {
        if (base && vmm) {
            for (uint32_t k = 0; k < vmm_n; ++k) (void)hipMemUnmap((char*)base + (size_t)k * vmm_chunk, std::min(vmm_chunk, map_len - (size_t)k * vmm_chunk));
            for (uint32_t k = 0; k < vmm_n; ++k) (void)hipMemRelease(vmm_handles[k]);
            delete[] vmm_handles;
            (void)hipMemAddressFree(va_base, va_len);
        } else if (base) {
            (void)hipFree(base);
        }
        vmm = false; vmm_handles = nullptr; vmm_n = 0; vmm_chunk = 0; va_base = nullptr; va_len = map_len = 0;
    }
    void ensure(size_t n) {
        if (n <= cap) return;
        ++gen;
        free_mem();
        p = base = nullptr;
        cap = 0;
        placed_for = nullptr; placed_gbps = 0.f; placed_winner = false;  // (a new allocation: whatever was measured was measured on the old one)
        size_t want = n + n / 8 + 256;
        if (round_to && want + skew >= round_to) want = (want + skew + round_to - 1) / round_to * round_to - skew;
        if (exact) want = (n + 4095) & ~(size_t)4095;
        if (pow2_large && n + skew >= (1ull << 30)) {
            want = 1ull << 30;
            while (want < n + skew) want <<= 1;
            want -= skew;
        }
        hipError_t e = dev_malloc(&base, want + skew);
        if (e != hipSuccess && release_all_arena_spares()) {  // memory the library itself is sitting on (spare arenas)
            (void)hipGetLastError();
            base = nullptr;
            e = dev_malloc(&base, want + skew);
        }
        if (e != hipSuccess && want > n + 256) {
            // the rounded request (slack, round_to, next power of two: up to 2 x n) did not fit: what the caller
            // asked for may still.  hipMalloc's failure is sticky until read.
            (void)hipGetLastError();
            base = nullptr;
            want = (n + 255) & ~(size_t)255;
            e = dev_malloc(&base, want + skew);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            base = nullptr;
            set_err("hipMalloc(%zu) failed: %s", want + skew, hipGetErrorString(e));
            throw (int)RHIP_ERR_ALLOC;
        }
        p = (char*)base + skew;
        cap = want;
    }
    void release() {
        free_mem();
        p = base = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return (T*)p; }
};

// RAII: make `device` current for the calling thread for the duration of an entry point (allocations and
// launches of a context must land on ITS device whatever the caller's current device is), restore on exit
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) {
            if (hipSetDevice(device) != hipSuccess) {
                set_err("hipSetDevice(%d) failed", device);
                throw (int)RHIP_ERR_DEVICE;
            }
            switched = true;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct PartialBuf {  // chunk buffers of rhip_many_partials, recycled through the context
    void* keys = nullptr;
    void* words = nullptr;
    uint64_t cap = 0;  // chunks
};

struct rhip_ctx_s {
    int device = 0;
    std::vector<PartialBuf> partial_cache;
    PartialBuf take_partial_buf(uint64_t chunks);
    hipStream_t stream = nullptr;
    // scratch (grow-only): candidate directory + queues + scan temporaries
    DBuf o_key, o_meta, o_slot, o_off, o_pair, flag, newidx, misc, misc2, prim_tmp, pair_acc;  // (pairwise batches: ss[] below)
    // Batches in flight (rhip_pairwise_begin .. _end): each has a slot = its own pinned staging of the batch description,
    // statistics area and completion word.  Device scratch is shared: the batches' kernels are ordered by the stream.
    static constexpr int N_SLOTS = RHIP_MAX_BATCHES_IN_FLIGHT;
    // slot N_SLOTS belongs to the calls that are not batches (cardinality / predicate): they wait for their own
    // completion, so they never find it busy and never touch the staging, scratch or timing events of a batch in flight
    static constexpr int SYNC_SLOT = N_SLOTS;
    void* h_stage[N_SLOTS + 1] = {};  // grow-only
    void* h_stage_dev[N_SLOTS + 1] = {};  // the same memory as the device addresses it
    bool stage_kernel = true;  // small descriptions are pulled by k_stage_in (RHIP_STAGE_KERNEL=0: always a copy command)
    size_t h_stage_cap[N_SLOTS + 1] = {};
    bool slot_busy[N_SLOTS] = {};
    bool slot_flagjoin[N_SLOTS] = {};  // the slot's batch joins its auxiliary streams by flags (k_join_wait)
    bool bb_timed[N_SLOTS + 1] = {};  // the slot's k_bb events were recorded by its current call (not in a merged launch)
    // device scratch of a pairwise batch (planning arrays, candidate directory, class queues, scan / tail words): one
    // set per slot, so that the planning kernels of a batch can run while the class kernels of the previous one do
    struct SlotScratch {
        DBuf plan_in, match, cand, cand_start, o_key, o_meta, o_off, o_pair, q[N_CLS], misc;
        // X-group histogram of the slot's last grouped batch (inside `cand`): zero between batches -- k_emit's decrements
        // undo k_count's increments -- so it is only cleared when it moves or grows
        void* hist_at = nullptr;
        size_t hist_n = 0;
        uint64_t hist_gen = 0;
        void release() {
            DBuf* all[] = {&plan_in, &match, &cand, &cand_start, &o_key, &o_meta, &o_off, &o_pair, &misc};
            for (auto* b : all) b->release();
            for (auto& b : q) b.release();
        }
    } ss[N_SLOTS + 1];
    hipEvent_t ev_plan[N_SLOTS] = {};
    // A batch over a PREPARED pair list keeps its plan on the list (PlanCache below): the matched container pairs, the class
    // queues and the candidate directory are a pure function of the two pools' directories and the ops, so a repeated batch
    // starts at its class kernels.  RHIP_PLAN_CACHE=0: every batch plans afresh.
    bool plan_cache = true;
    bool last_plan_cached = false;  // the last batch begun on this context took its plan from a pair list's cache (rhip_debug_plan_cached)
    bool plan_overlap = true;  // RHIP_PLAN_OVERLAP=0: the planning kernels of a batch always run on the main stream
    uint64_t plan_overlap_max_bytes = 2ull << 30;
    int in_flight() const { int n = 0; for (bool b : slot_busy) n += b ? 1 : 0; return n; }
    void ensure_stage(int slot, size_t n);
    int acquire_slot();
    DBuf many[20];
    DBuf many_kid, many_dict_buf;  // key dictionary of a chunk table (rhip_many_finalize over 48-bit keys): per call
    bool many_dict = true;         // RHIP_MANY_DICT=0: roaring64 pools group by a radix sort of (key, descriptor) pairs, as until round 5
    DBuf shard[9];  // rhip_many_sharded: send / receive tables and the sparse exchange's staging, kept between calls
    int shard_force_collective = 0;  // RHIP_SHARD_FORCE_COLLECTIVE=1: issue the all-to-all on a one-rank communicator too (tests, timing)
    // many-way path: pinned staging of the selection (ids + member prefix), the event that says the device has read
    // it, totals / completion word / sticky error word inside h_pinned
    void* h_many = nullptr;
    size_t h_many_cap = 0;
    hipEvent_t ev_many_stage = nullptr;
    bool many_stage_pending = false;
    std::vector<rhip_pool_t*> many_free;  // retired results of rhip_or_many / rhip_xor_many: their buffers are recycled (under g_ctx_mu)
    uint64_t gen = 0;                     // generation number of this context (pools remember it: rhip_pool_s::ctx_gen)
    uint64_t dense_pipe = 0, dense_pipe_seq = 0;  // open dense many-way pipeline (rhip_many_partials_dense .. _finalize_dense), 0 = none
    size_t arena_skew = 0;  // result arenas start this many bytes into their allocation (RHIP_ARENA_SKEW)
    size_t arena_round = 0; // RHIP_ARENA_ROUND_MB
    // Result arenas (and the synthetic bitset pool) of 1 GiB and more are allocated as POWERS OF TWO.  The bitset x bitset
    // kernel streams two operands and one result in lockstep, and its time on C2 depends on how the driver's buddy
    // allocator composes the buffers out of physical blocks (round 4, scripts/arena_place.hip, profiles/r04_arena_*):
    // 8 / 16 GiB arenas beside an 8 GiB pool -- one naturally aligned block each -- gave 4.35-4.42 ms in 16 of 16 fresh
    // allocations; 9.4 GB arenas (8 G + 1 G + 256 M + ... blocks) 4.37 or 4.64 ms, 7.8 GiB ones 3.97-4.40, by where the
    // pieces happened to land.  A power of two costs address space (at most 2 x), never traffic.  RHIP_ARENA_POW2=0: off.
    bool arena_pow2 = true;
    // MEASURED placement of large result arenas (place_arena below): when a result arena of at least arena_place_min bytes
    // has to be allocated and the left operand pool is large as well, up to arena_tries candidates are allocated (all kept
    // alive until the choice is made, so that each gets different physical pages), a probe with k_bb's access pattern is
    // timed on each against the operand pool, the fastest is kept -- or the first that streams at arena_good_gbps.  The
    // caller does nothing; a recycled result pool (`reuse`) keeps its placement.  RHIP_ARENA_TRIES (0 / 1: off),
    // RHIP_ARENA_PLACE_MIN_MB.  The search goes on for up to arena_tries more candidates while the best so far is below
    // arena_fair_gbps.
    // Round 6: the mode is a function of the arena's VIRTUAL address (scripts/vmm_place2.hip: the same physical chunks stream
    // at 6.4 or 5.9 TB/s depending on where they are mapped; other chunks at the same address give the same rate), so the
    // search moves ONE physical allocation through an address window instead of allocating candidates: place_arena_va.
    // RHIP_ARENA_VMM=0: the candidate search of round 4; RHIP_ARENA_VA_WINDOW_MB, RHIP_ARENA_VA_STEP_MB: the window walked.
    bool arena_vmm = true;
    int arena_chunk_runs = 2, arena_chunk_stride = 2;  // RHIP_ARENA_CHUNK_RUNS / _STRIDE: passes (the first untimed) and slot stride of a single chunk's probe -- 3 / 1, 3 / 2 and 2 / 2 placed alike on one box (0.781-0.798, twelve processes alternating; gpurun_out/r6w), 2 / 2 in 25-35 ms instead of 45
    uint64_t arena_va_window = 512ull << 30, arena_va_step = 1ull << 30;  // (address space only: 512 GiB = 56 positions of an 8 GiB arena)
    int arena_keep_spares = 1;  // RHIP_ARENA_SPARES=0: the losers of a placement search are released, not kept
    int arena_tries = 10;
    uint64_t arena_place_min = 2ull << 30;
    double arena_good_gbps = 6250.0;
    double arena_fair_gbps = 5850.0;  // (probe scale: slow band 5.2-5.7 TB/s, mid 5.9-6.1, fast windows 6.2-6.4)
    // Candidates that lost a placement search are KEPT (the next search probes them first): memory that is freed has to
    // be scrubbed by the driver before it is handed out again, ~0.3 s per 8 GiB -- a second search right behind the first
    // (bench.py's `or` arena behind its `and` arena) paid 3.2 s for ten allocations out of just-freed memory.  Released by
    // rhip_ctx_trim, rhip_ctx_destroy, and by any allocation of the library that fails (DBuf::ensure retries after it).
    std::vector<DBuf> arena_spares;
    // physical chunks (hipMemCreate) left over by place_arena_chunks, unmapped, with the rate each streamed at beside the
    // operand arena it was probed against: the next result arena for that operand is composed out of them first
    struct ChunkSpare { hipMemGenericAllocationHandle_t h; size_t size; float gbps; const void* placed_for; uint64_t placed_for_gen; };
    std::vector<ChunkSpare> chunk_spares;
    // Address space in which place_arena_chunks probes single chunks, one never-used place per probe, owned by the CONTEXT and
    // freed with it: an address keeps translating to a chunk for as long as that chunk's handle lives (place_arena_va's first
    // limit), and the spare chunks outlive the arena whose search probed them -- in a range that was freed with that arena
    // and handed out again, a later mapping would land on addresses that still translate to the spares
    void* probe_va = nullptr;
    size_t probe_va_len = 0, probe_va_used = 0;
    size_t release_chunk_spares() {
        size_t n = 0;
        for (ChunkSpare& x : chunk_spares) { n += x.size; (void)hipMemRelease(x.h); }
        chunk_spares.clear();
        return n;
    }
    int batches_since_place = 0;  // (trim_spares_when_steady)
    std::vector<float> last_placement;  // probe GB/s of the candidates of the last placement (rhip_debug_last_placement)
    bool debug_plan = false;  // RHIP_DEBUG_PLAN=1: one line per batch on stderr (bounds, fork / merge decision)
    bool merge_classes = true;  // RHIP_MERGE_CLASSES=0: a small batch launches its class kernels one by one
    uint64_t merge_max_items = 256u << 10;  // ... "small" = at most that many matched container pairs (upper bound)
    // X-grouped image queues (rhip_common.h XGroupView): 0 never, 1 when the batch promises enough reuse (at least
    // group_min_reuse matched container pairs -- upper bound -- per container of the operand pools), 2 always (tests)
    int group_x = 1;             // RHIP_GROUP_X
    uint64_t group_min_reuse = 64;  // (C5: 47 partners per container by this bound, almost all interval pairs -- grouping costs it 4 %)
    uint64_t group_min_items = 16u << 10;
    uint32_t group_chunk = 8;    // RHIP_XG_CHUNK: items a wave of the grouped kernels walks in a row, at least
    // Forked class kernels are joined by flags (k_join_signal / k_join_wait, rhip_plan.h) instead of events -- only where
    // kernels of different streams really run side by side: rhip_ctx_create tests that (k_conc_probe) and falls back to
    // events where they do not (a tool that serialises kernels: rocprofv3 --pmc).  RHIP_SPIN_JOIN=0: always events.
    bool spin_join = true;
    uint64_t join_spins = 60000;  // RHIP_JOIN_SPINS: the gate's bound, x ~3.4 us (0.2 s)
    int join_force_fail = 0;      // RHIP_JOIN_FAIL=1 (tests): every gate reports a time-out
    uint64_t join_recovered = 0;  // batches finished through the fallback (rhip_debug_join_recovered)
    u64* join_timeout_word() const { return (u64*)((char*)h_pinned + PINNED_JOIN_TIMEOUT_OFF); }
    bool copy_wide = true;       // RHIP_COPY_WIDE=0: k_copy always takes four items per wave
    int many_pf = 1;  // RHIP_MANY_PF: loads of k_many_l1's ring in flight behind every scatter (1 / 2 / 3 / 4; round 5, after the stream became branch-free: 0.50 / 0.51 / 0.52 / 0.54 ms on C4 -- the kernel is bound by its LDS atomics and instruction issue, five waves per SIMD hide the latency, every further buffer costs spills)
    int many_ch = 0;  // RHIP_MANY_CH: members per piece (tests of the multi-chunk / cut-group paths on small inputs); 0 = by size
    uint64_t many_slots = MANY_RESIDENT;  // RHIP_MANY_SLOTS: pieces of a large call = workgroups of k_many_l1 resident at once (RHIP_MANY_WAVES per CU)
    uint64_t many_t = 0;         // RHIP_MANY_T: members per workgroup of k_many_hist / k_many_scatter; 0 = one workgroup per CU
    int many_reverse = 0;        // RHIP_MANY_REVERSE (tests): k_many_scatter fills its reservations backwards
    int usmall_gp8 = 0;          // RHIP_USMALL_GP8=1: k_usmall's six-per-CU instantiation (measured: no gain on C5, weather or / xor 3-7 % slower; rhip_array.h)
    uint32_t pool_align = 0;     // RHIP_POOL_ALIGN: slot granule of loaded pools, 16 / 128; 0 = by the images' average size (choose_pay_align)
    static constexpr size_t PINNED_MANY_FLAG_OFF = 2304, PINNED_MANY_TOT_OFF = 2560, PINNED_MANY_ERR_OFF = 2816;
    DBuf sel[5];  // pool_select / pool_convert scratch
    void* h_pinned = nullptr;  // small pinned readback area
    // completion word of the pairwise path (inside h_pinned): k_tail's last block writes the call's sequence number
    // after the statistics; the host polls it (the stream's completion signal costs an interrupt / wake-up)
    static constexpr size_t PINNED_STATS_OFF = 1024, PINNED_STATS_STRIDE = 192, PINNED_FLAG_OFF = 2048, PINNED_JOIN_TIMEOUT_OFF = 3072;
    uint64_t seq = 0;
    bool spin_wait = true;
    int explicit_units = 0;  // RHIP_EXPLICIT_UNITS=1: always stage the unit arrays; =2: implicit units, but never four per wave (tests of those paths)
    volatile uint64_t* done_flag(int slot) const { return (volatile uint64_t*)((char*)h_pinned + PINNED_FLAG_OFF + 64 * slot); }
    void* slot_stats(int slot) const { return (char*)h_pinned + PINNED_STATS_OFF + PINNED_STATS_STRIDE * slot; }
    // host-side phase clock (diagnostics, rhip_debug_host_clock): microseconds accumulated per phase of rhip_pairwise
    double hclk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    rhip_stats_t stats{};
    bool timing = false;
    bool class_stats = false;                     // rhip_ctx_set_class_stats: one more kernel + wait per batch
    uint64_t cls_stats[3 * N_CLS] = {};           // items / bytes in / bytes out per work class of the last batch
    hipEvent_t evs[RHIP_MAX_BATCHES_IN_FLIGHT + 1][4]{};  // timing events: [slot][call start, call end, k_bb start, k_bb end]
    hipEvent_t* ev = evs[RHIP_MAX_BATCHES_IN_FLIGHT];     // (calls that are not batches: the synchronous slot's)
    // independent class kernels of one batch run concurrently: fork after planning, join before compaction
    static constexpr int N_AUX = 3;
    hipStream_t aux[N_AUX]{};
    hipEvent_t ev_fork = nullptr, ev_join[N_AUX]{}, ev_runs = nullptr, ev_ba = nullptr;
    bool overlap = true;
    // Class kernels are forked onto the auxiliary streams only when the batch is big enough for their concurrency to
    // pay for the fork / join (two cross-queue dependencies, 40-50 us, and no overlap between consecutive batches):
    // measured break-even on the realdata sets ~150 MB of result-slot bound (RHIP_FORK_MIN_MB overrides; 0 = always)
    uint64_t fork_min_bytes = 144ull << 20;
    // ... unless the batch is LIGHT: at most 128 k matched container pairs and a slot bound below this (RHIP_FORK_MIN_MB
    // sets both: 0 = always fork)
    uint64_t fork_light_bytes = 320ull << 20;
    // A batch of run-dominated pools (C5: 847 000 interval pairs) is forked whatever its bytes once it has this many matched
    // container pairs (upper bound): its interval kernel keeps the MAIN stream and the few filter / probe / image items
    // run beside it instead of behind it (RHIP_FORK_RUNS_MIN; 0 = never)
    uint64_t fork_runs_min_items = 512u << 10;
};

void rhip_ctx_s::ensure_stage(int slot, size_t n) {
    if (n <= h_stage_cap[slot]) return;
    if (h_stage[slot]) (void)hipHostFree(h_stage[slot]);
    h_stage[slot] = nullptr;
    h_stage_cap[slot] = 0;
    const size_t want = n + n / 4 + 4096;
    if (hipHostMalloc(&h_stage[slot], want, hipHostMallocDefault) != hipSuccess) {
        h_stage[slot] = nullptr;
        set_err("hipHostMalloc(%zu) failed", want);
        throw (int)RHIP_ERR_ALLOC;
    }
    h_stage_cap[slot] = want;
    if (hipHostGetDevicePointer(&h_stage_dev[slot], h_stage[slot], 0) != hipSuccess) h_stage_dev[slot] = nullptr;
}
int rhip_ctx_s::acquire_slot() {
    for (int k = 0; k < N_SLOTS; ++k) {
        const int slot = (int)((seq + 1 + k) % N_SLOTS);
        if (!slot_busy[slot]) return slot;
    }
    set_err("%d batches in flight: call rhip_pairwise_end first", N_SLOTS);
    throw (int)RHIP_ERR_ARG;
}

struct rhip_pool_s {
    rhip_ctx_t* ctx = nullptr;
    uint32_t n_bitmaps = 0;
    uint64_t n_cont = 0;
    bool is64 = false;
    DBuf bm_start, key, type, card, nruns, off, arena;
    uint64_t arena_used = 0;
    uint32_t pay_align = 16;    // slot granule the loader / rhip_pool_select used (16, or 128 = whole cache lines); spliced results keep 16
    bool pending = false;       // result of a batch that has begun and not ended: not usable yet
    int in_use = 0;             // batches in flight that read this pool as an operand: not recyclable yet
    bool free_deferred = false; // rhip_pool_free arrived while in_use / pinned: the last batch to end (or list to go) frees it
    int list_pins = 0;          // prepared pair lists that hold this pool as an operand (rhip_pairlist_*): a freed pool must
                                // not be dereferenced by the next rhip_pairwise_list call
    bool from_many = false;     // result of the many-way path: rhip_pool_free hands its buffers back to the context
    uint64_t ctx_gen = 0;       // ... the context it was made by, by GENERATION: an address can be reused by a later context
    uint64_t compact_mark = 0;  // arena_used right after the last compaction (0: never updated in place), see rhip_pairwise_inplace
    // host mirror of the directory (filled lazily for serialization); planning only needs bm_start
    bool host_dir = false;
    bool host_bm = false;
    std::vector<uint64_t> h_bm_start, h_key, h_off;
    std::vector<uint8_t> h_type;
    std::vector<uint32_t> h_card, h_nruns;
    std::vector<uint64_t> h_cards;  // per-bitmap cardinalities cache
    std::vector<uint64_t> h_w;      // per-bitmap result-slot bound (k_bitmap_bounds), see fetch_bounds
    std::vector<uint64_t> h_wm;     // the same bound as the many-way path needs it (runs by rounded cardinality)
    uint64_t wm_total = 0;          // sum of h_wm
    bool has_long_runs = false;     // some run container's payload exceeds a bitset's 8192 bytes (only a hand-made list can)
    uint64_t n_keys_distinct = 0;   // 32-bit pools: distinct container keys in the pool (fetch_bounds)
    uint64_t max_key = 0;           // largest container key in the pool
    std::vector<uint32_t> h_n;      // per-bitmap container count
    uint32_t max_n = 0;             // largest of them
    uint64_t n_run_cont = 0;        // run containers of the pool (fetch_bounds)
    // roaring64 pools: the key dictionary of the many-way path (rhip_many.h: the pool's distinct 48-bit keys, sorted, and
    // every container's dense key id), built once per state of the pool (kd_gen = the bounds_gen it was built for)
    DBuf kd_kid, kd_dict;
    uint64_t kd_K = 0, kd_gen = 0;
    bool host_w = false;
    uint64_t bounds_gen = 0;        // which fetch_bounds() filled the mirrors above (a prepared pair list remembers it)
    int8_t census[3] = {1, 1, 1};   // does the pool hold bitset / array / run containers (1 until known otherwise)
    PoolView view() const {
        PoolView v;
        v.bm_start = bm_start.as<u64>();
        v.key = key.as<u64>();
        v.type = type.as<uint8_t>();
        v.card = card.as<uint32_t>();
        v.nruns = nruns.as<uint32_t>();
        v.off = off.as<u64>();
        v.arena = arena.as<uint8_t>();
        return v;
    }
    void release() {
        bm_start.release(); key.release(); type.release(); card.release(); nruns.release(); off.release();
        arena.release();
        kd_kid.release(); kd_dict.release();
        kd_gen = 0;
    }
};

// sums over a pair list that size a batch (plan()): planning units, matched / candidate containers, result-slot bytes
struct PlanCache;       // (defined behind Plan, further down)
struct PlanCacheEntry;
struct PairSums {
    uint64_t nu_a = 0, nu_b = 0, s_mn = 0, s_na = 0, s_nb = 0, s_wmin = 0, s_wa = 0, s_wb = 0;
};
// A PREPARED pair list (include/roaring_hip.h, rhip_pairlist_*): the indices resident on the device, the sums above
// computed once.  The reference has no pair list at all -- its benchmark walks the bitmaps (benchmarks/benchmark.cpp:
// 2035-2091) -- so validating, summing and copying one is input preparation, not part of the set operation.
struct rhip_pairlist_s {
    rhip_ctx_t* ctx = nullptr;
    rhip_pool_t *A = nullptr, *B = nullptr;
    size_t npairs = 0;
    std::vector<uint32_t> lhs, rhs;  // host copy: explicit-unit batches, the cardinality read-back, re-validation
    DBuf d_idx;                      // lhs | rhs (u32 each) on the device
    PairSums sums;
    uint64_t genA = 0, genB = 0;     // the pools' bounds_gen the sums were taken from (0: never)
    int in_use = 0;                  // batches in flight that read d_idx
    bool free_deferred = false;
    bool pinned = false;             // A / B carry this list's pin (list_pins)
    PlanCache* cache = nullptr;  // the list's cached plans (below): what k_count / k_scan / k_emit leave behind
};

static void pairlist_destroy(rhip_pairlist_t* L);

static void ensure_dir(rhip_pool_t* P, uint32_t n_bitmaps, uint64_t n_cont) {
    P->bm_start.ensure(8 * ((size_t)n_bitmaps + 1));
    P->key.ensure(8 * (size_t)(n_cont + 1));
    P->type.ensure((size_t)n_cont + 16);
    P->card.ensure(4 * (size_t)(n_cont + 1));
    P->nruns.ensure(4 * (size_t)(n_cont + 1));
    P->off.ensure(8 * (size_t)(n_cont + 1));
}

// ------------------------------------------------------------------ context
// contexts that exist: a pool may outlive its context (rhip_pool_free then must not touch it)
static std::mutex g_ctx_mu;
static std::set<rhip_ctx_t*> g_live_ctx;
static uint64_t g_ctx_gen = 0;  // (under g_ctx_mu) every context gets the next generation number
static bool release_all_arena_spares() {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    bool any = false;
    for (rhip_ctx_t* c : g_live_ctx) {
        if (c->arena_spares.empty() && c->chunk_spares.empty()) continue;
        DeviceGuard guard(c->device);  // (the spares of a context live on ITS device)
        for (DBuf& b : c->arena_spares) { any = any || b.base != nullptr; b.release(); }
        c->arena_spares.clear();
        any = c->release_chunk_spares() != 0 || any;
    }
    return any;
}
// Spares are for the searches that follow each other at start-up (a second search right behind the first would otherwise
// be served just-freed memory the driver has to scrub: seconds).  Once a context has run `after` batches without placing
// anything, all but its two best spares go back to the driver: other allocators on the device (a caching allocator,
// another rank) cannot ask this library to let go.
static void trim_spares_when_steady(rhip_ctx_t* c, int after = 16) {
    if ((c->arena_spares.size() <= 2 && c->chunk_spares.empty()) || ++c->batches_since_place != after) return;
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    (void)c->release_chunk_spares();
    std::sort(c->arena_spares.begin(), c->arena_spares.end(), [](const DBuf& a, const DBuf& b) {
        return a.placed_winner != b.placed_winner ? a.placed_winner : a.placed_gbps > b.placed_gbps;  // (parked winners first)
    });
    while (c->arena_spares.size() > 2) {
        c->arena_spares.back().release();
        c->arena_spares.pop_back();
    }
}
static bool ctx_alive(rhip_ctx_t* c) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    return g_live_ctx.count(c) != 0;
}
// The recycle list of many-way results (ctx->many_free) is only touched under g_ctx_mu: rhip_pool_free may come from
// any thread (a finaliser), and the context a pool names may be gone -- or gone and its address taken by a NEW context
// (another device, other buffers), which the generation number tells apart.
static bool many_recycle_push(rhip_pool_t* P);
static rhip_pool_t* many_recycle_pop(rhip_ctx_t* c);
static void arena_park(rhip_pool_t* P);
extern "C" const char* rhip_last_error(void) { return g_err.c_str(); }
extern "C" const char* rhip_version(void) { return "roaring-hip 0.1 (gfx950)"; }

extern "C" rhip_ctx_t* rhip_ctx_create(int device) {
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
            set_err("no HIP device available (libroaring_hip has no CPU fallback)");
            return nullptr;
        }
        if (device < 0) HIPCHK(hipGetDevice(&device));
        if (device >= ndev) { set_err("device %d out of range (%d visible)", device, ndev); return nullptr; }
        DeviceGuard dguard_(device);  // the caller's current device is left as it was
        rhip_ctx_t* c = new rhip_ctx_s();
        c->device = device;
        HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIPCHK(hipHostMalloc(&c->h_pinned, 4096, hipHostMallocDefault));
        for (auto& es : c->evs) for (auto& e : es) HIPCHK(hipEventCreate(&e));
        {
            // One auxiliary stream is created at the device's highest priority: aux2, which under andnot (and in multi-op
            // batches) carries k_union_g beside the main stream's machine-filling k_filter_g -- launched a fork later, its
            // workgroups otherwise only get what the filter leaves.  Measured round 5 (five alternating runs, one box):
            // weather andnot 0.358-0.365 -> 0.333-0.335 ms, census-income or 0.264-0.270 -> 0.247-0.252, everything else
            // within noise; aux1 instead (k_usmall under or / xor): weather or 0.55 -> 0.62 -- the main stream's k_union_g is
            // that batch's critical kernel.  RHIP_AUX_PRIO=<index> picks another stream, -1 none.
            int hi_aux = 2, least = 0, greatest = 0;
            if (const char* e = getenv("RHIP_AUX_PRIO")) hi_aux = atoi(e);
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            for (int k = 0; k < rhip_ctx_s::N_AUX; ++k) {
                if (k == hi_aux) HIPCHK(hipStreamCreateWithPriority(&c->aux[k], hipStreamNonBlocking, greatest));
                else HIPCHK(hipStreamCreateWithFlags(&c->aux[k], hipStreamNonBlocking));
            }
        }
        HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_runs, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_ba, hipEventDisableTiming));
        for (auto& e : c->ev_join) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : c->ev_plan) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_many_stage, hipEventDisableTiming));
        if (const char* e = getenv("RHIP_NO_OVERLAP")) c->overlap = !(e[0] == '1');
        if (const char* e = getenv("RHIP_SPIN_WAIT")) c->spin_wait = !(e[0] == '0');
        if (const char* e = getenv("RHIP_EXPLICIT_UNITS")) c->explicit_units = atoi(e);
        if (const char* e = getenv("RHIP_STAGE_KERNEL")) c->stage_kernel = !(e[0] == '0');
        if (const char* e = getenv("RHIP_FORK_MIN_MB")) c->fork_min_bytes = c->fork_light_bytes = (uint64_t)atoll(e) << 20;
        if (const char* e = getenv("RHIP_PLAN_OVERLAP")) c->plan_overlap = !(e[0] == '0');
        if (const char* e = getenv("RHIP_PLAN_CACHE")) c->plan_cache = atoi(e) != 0;
        if (const char* e = getenv("RHIP_FORK_RUNS_MIN")) c->fork_runs_min_items = (uint64_t)atoll(e);
        if (const char* e = getenv("RHIP_MANY_PF")) c->many_pf = atoi(e);
        if (const char* e = getenv("RHIP_MANY_DICT")) c->many_dict = atoi(e) != 0;
        if (const char* e = getenv("RHIP_MERGE_CLASSES")) c->merge_classes = !(e[0] == '0');
        if (const char* e = getenv("RHIP_DEBUG_PLAN")) c->debug_plan = e[0] == '1';
        if (const char* e = getenv("RHIP_MERGE_MAX_K")) c->merge_max_items = strtoull(e, nullptr, 0) << 10;
        if (const char* e = getenv("RHIP_ARENA_ROUND_MB")) c->arena_round = (size_t)strtoull(e, nullptr, 0) << 20;
        if (const char* e = getenv("RHIP_ARENA_POW2")) c->arena_pow2 = !(e[0] == '0');
        if (const char* e = getenv("RHIP_ARENA_TRIES")) c->arena_tries = atoi(e);
        if (const char* e = getenv("RHIP_ARENA_SPARES")) c->arena_keep_spares = atoi(e);
        if (const char* e = getenv("RHIP_ARENA_VMM")) c->arena_vmm = !(e[0] == '0');
        if (const char* e = getenv("RHIP_ARENA_CHUNK_RUNS")) c->arena_chunk_runs = std::max(2, atoi(e));
        if (const char* e = getenv("RHIP_ARENA_CHUNK_STRIDE")) c->arena_chunk_stride = std::max(1, atoi(e));
        if (const char* e = getenv("RHIP_ARENA_VA_WINDOW_MB")) c->arena_va_window = (uint64_t)strtoull(e, nullptr, 0) << 20;
        if (const char* e = getenv("RHIP_ARENA_VA_STEP_MB")) c->arena_va_step = std::max<uint64_t>(2, strtoull(e, nullptr, 0)) << 20;
        if (const char* e = getenv("RHIP_ARENA_PLACE_MIN_MB")) c->arena_place_min = (uint64_t)strtoull(e, nullptr, 0) << 20;
        if (const char* e = getenv("RHIP_ARENA_SKEW")) c->arena_skew = (size_t)strtoull(e, nullptr, 0) & ~(size_t)255;
        if (const char* e = getenv("RHIP_GROUP_X")) c->group_x = atoi(e);
        if (const char* e = getenv("RHIP_XG_CHUNK")) c->group_chunk = (uint32_t)std::max(1, atoi(e));
        if (const char* e = getenv("RHIP_COPY_WIDE")) c->copy_wide = !(e[0] == '0');
        bool spin_join_forced = false;  // RHIP_SPIN_JOIN=2: on without the self-test (the emulator runs kernels one by one)
        if (const char* e = getenv("RHIP_SPIN_JOIN")) { c->spin_join = !(e[0] == '0'); spin_join_forced = e[0] == '2'; }
        if (const char* e = getenv("RHIP_SHARD_FORCE_COLLECTIVE")) c->shard_force_collective = atoi(e);
        if (const char* e = getenv("RHIP_JOIN_SPINS")) c->join_spins = (uint64_t)std::max(0ll, atoll(e));
        if (const char* e = getenv("RHIP_JOIN_FAIL")) c->join_force_fail = atoi(e);
        if (const char* e = getenv("RHIP_MANY_CH")) c->many_ch = std::max(1, std::min(1 << 20, atoi(e)));
        if (const char* e = getenv("RHIP_MANY_SLOTS")) c->many_slots = (uint64_t)std::max(1, atoi(e));
        if (const char* e = getenv("RHIP_MANY_T")) c->many_t = (uint64_t)std::max(1024, atoi(e));
        if (const char* e = getenv("RHIP_MANY_REVERSE")) c->many_reverse = atoi(e);
        if (const char* e = getenv("RHIP_USMALL_GP8")) c->usmall_gp8 = atoi(e) ? 1 : 0;
        if (const char* e = getenv("RHIP_POOL_ALIGN")) c->pool_align = atoi(e) == 128 ? 128 : (atoi(e) == 16 ? 16 : 0);
        memset(c->h_pinned, 0, 4096);
        if (c->spin_join && !spin_join_forced) {
            // do kernels of two streams overlap here?  (k_conc_probe: bounded wait on the main stream for a flag that a
            // kernel launched 300 us later on an auxiliary stream sets; both words live in the pinned area)
            u64* w = (u64*)((char*)c->h_pinned + 3200);  // [0] flag, [1] verdict
            hipLaunchKernelGGL(k_conc_probe, dim3(1), dim3(64), 0, c->stream, (const u64*)w, w + 1);
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 300.0) {}
            hipLaunchKernelGGL(k_join_signal, dim3(1), dim3(64), 0, c->aux[0], w);
            const bool ok = hipStreamSynchronize(c->stream) == hipSuccess && hipStreamSynchronize(c->aux[0]) == hipSuccess;
            if (!ok || __atomic_load_n(w + 1, __ATOMIC_ACQUIRE) != 1ull) c->spin_join = false;
            (void)hipGetLastError();
            w[0] = w[1] = 0;
        }
        {
            std::lock_guard<std::mutex> lk(g_ctx_mu);
            c->gen = ++g_ctx_gen;
            g_live_ctx.insert(c);
        }
        return c;
    } catch (int) {
        return nullptr;
    }
}
extern "C" void rhip_ctx_destroy(rhip_ctx_t* c) {
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        g_live_ctx.erase(c);
    }
    int prev_dev_ = -1;
    const bool sw_ = hipGetDevice(&prev_dev_) == hipSuccess && prev_dev_ != c->device && hipSetDevice(c->device) == hipSuccess;
    (void)hipStreamSynchronize(c->stream);
    DBuf* all[] = {&c->o_key, &c->o_meta, &c->o_slot, &c->o_off, &c->o_pair,
                   &c->flag, &c->newidx, &c->misc, &c->misc2, &c->prim_tmp, &c->pair_acc};
    for (auto& hs : c->h_stage) if (hs) (void)hipHostFree(hs);
    for (auto* b : all) b->release();
    for (auto& b : c->many) b.release();
    for (auto& b : c->shard) b.release();
    for (auto& b : c->arena_spares) b.release();
    c->arena_spares.clear();
    (void)c->release_chunk_spares();
    if (c->probe_va) { (void)hipMemAddressFree(c->probe_va, c->probe_va_len); c->probe_va = nullptr; }
    for (auto& b : c->sel) b.release();
    for (auto& b : c->partial_cache) { (void)hipFree(b.keys); (void)hipFree(b.words); }
    for (rhip_pool_t* R : c->many_free) { R->release(); delete R; }
    c->many_free.clear();
    for (auto& es : c->evs) for (auto& e : es) if (e) (void)hipEventDestroy(e);
    for (auto& a : c->aux) if (a) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_runs) (void)hipEventDestroy(c->ev_runs);
    if (c->ev_ba) (void)hipEventDestroy(c->ev_ba);
    for (auto& e : c->ev_join) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_plan) if (e) (void)hipEventDestroy(e);
    if (c->ev_many_stage) (void)hipEventDestroy(c->ev_many_stage);
    if (c->h_many) (void)hipHostFree(c->h_many);
    for (auto& sc : c->ss) sc.release();
    (void)hipHostFree(c->h_pinned);
    (void)hipStreamDestroy(c->stream);
    if (sw_) (void)hipSetDevice(prev_dev_);
    delete c;
}
extern "C" void* rhip_ctx_stream(rhip_ctx_t* c) { return (void*)c->stream; }
extern "C" int rhip_ctx_synchronize(rhip_ctx_t* c) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        set_err("hipStreamSynchronize: %s", hipGetErrorString(e));
        return RHIP_ERR_DEVICE;
    }
    if (c->dense_pipe) {  // an open dense many-way pipeline whose stage 1 has failed: say so here as well as in its finalize
        const uint64_t ew = __atomic_load_n((uint64_t*)((char*)c->h_pinned + rhip_ctx_s::PINNED_MANY_ERR_OFF), __ATOMIC_ACQUIRE);
        if ((ew >> 8) == c->dense_pipe && (ew & 0xFF)) {
            set_err("dense many-way stage 1 met a container key >= key_space: use the sparse exchange");
            return RHIP_ERR_ARG;
        }
    }
    return RHIP_OK;
}
extern "C" void rhip_ctx_set_timing(rhip_ctx_t* c, int enabled) { c->timing = enabled != 0; }
extern "C" void rhip_ctx_set_class_stats(rhip_ctx_t* c, int enabled) { c->class_stats = enabled != 0; }
static const char* const k_class_names[N_CLS] = {"k_bb", "k_genw", "k_copy", "(retry)", "k_filter", "k_wave", "k_ivl<32,255>", "k_probe",
                                                 "k_bba", "k_usmall", "k_ivl<8,63>", "k_ivl<16,127>", "k_ba"};
extern "C" int rhip_last_class_stats(rhip_ctx_t* c, rhip_class_stats_t* out, int capacity) {
    if (!c || (!out && capacity > 0)) return RHIP_ERR_ARG;
    int n = 0;
    for (int k = 0; k < N_CLS; ++k) {
        if (k == CLS_RETRY) continue;  // (re-queued items are counted under the class that produced them)
        if (n < capacity) {
            out[n].kernel = k_class_names[k];
            out[n].items = c->cls_stats[3 * k];
            out[n].bytes_in = c->cls_stats[3 * k + 1];
            out[n].bytes_out = c->cls_stats[3 * k + 2];
        }
        ++n;
    }
    return n;
}
extern "C" int rhip_last_stats(rhip_ctx_t* c, rhip_stats_t* out) {
    *out = c->stats;
    return RHIP_OK;
}

// scan helper: out[0..n] = exclusive prefix of in[0..n) (in must have n+1 readable elements)
static void exscan(rhip_ctx_t* c, const uint32_t* in, u64* out, size_t n) {
    size_t tb = 0;
    HIPCHK(prim_exscan_u32_u64(nullptr, tb, in, out, n, c->stream));
    c->prim_tmp.ensure(tb + 16);
    tb = c->prim_tmp.cap;
    HIPCHK(prim_exscan_u32_u64(c->prim_tmp.p, tb, in, out, n, c->stream));
}

// ------------------------------------------------------------------ upload (portable format)
namespace {
struct HostDir {
    std::vector<uint64_t> bm_start, key, off;
    std::vector<uint8_t> type;
    std::vector<uint32_t> card, nruns;
    std::vector<uint8_t> arena;
    void push_payload(const void* src, size_t bytes) {
        size_t o = arena.size();
        size_t padded = (bytes + 15) & ~(size_t)15;
        if (padded < 16) padded = 16;
        arena.resize(o + padded, 0);
        memcpy(arena.data() + o, src, bytes);
    }
};

inline uint16_t rd16(const char* p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline uint32_t rd32(const char* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t rd64(const char* p) { uint64_t v; memcpy(&v, p, 8); return v; }

// Parses one 32-bit portable bitmap (RoaringFormatSpec as implemented by
// ra_portable_deserialize, src/roaring_array.c:633-813) appending its containers with keys
// (key_hi << 16) | key16.  Returns bytes consumed or 0 on a malformed/invalid buffer.
// Containers are validated as roaring_bitmap_internal_validate would (src/roaring.c:454-523).
size_t parse_portable32(const char* buf, size_t len, uint64_t key_hi, HostDir& D) {
    const char* p = buf;
    const char* end = buf + len;
    if (len < 4) return 0;
    uint32_t cookie = rd32(p);
    p += 4;
    uint32_t n;
    bool hasrun = false;
    const uint8_t* runflags = nullptr;
    if ((cookie & 0xFFFF) == 12347u) {
        hasrun = true;
        n = (cookie >> 16) + 1;
        if (p + (n + 7) / 8 > end) return 0;
        runflags = (const uint8_t*)p;
        p += (n + 7) / 8;
    } else if (cookie == 12346u) {
        if (p + 4 > end) return 0;
        n = rd32(p);
        p += 4;
    } else {
        return 0;
    }
    if (n > 65536u) return 0;
    if (p + 4 * (size_t)n > end) return 0;
    const char* desc = p;
    p += 4 * (size_t)n;
    if (!hasrun || n >= 4) {
        if (p + 4 * (size_t)n > end) return 0;
        p += 4 * (size_t)n;
    }
    int32_t prevkey = -1;
    for (uint32_t i = 0; i < n; ++i) {
        uint16_t k16 = rd16(desc + 4 * i);
        uint32_t card = (uint32_t)rd16(desc + 4 * i + 2) + 1;
        if ((int32_t)k16 <= prevkey) return 0;
        prevkey = k16;
        bool isrun = hasrun && ((runflags[i / 8] >> (i % 8)) & 1);
        D.key.push_back((key_hi << 16) | k16);
        D.off.push_back(D.arena.size());
        if (isrun) {
            if (p + 2 > end) return 0;
            uint32_t nr = rd16(p);
            p += 2;
            if (nr == 0 || p + 4 * (size_t)nr > end) return 0;
            uint32_t c = 0;
            int32_t last_end = -2;
            for (uint32_t r = 0; r < nr; ++r) {
                int32_t s = rd16(p + 4 * r), l = rd16(p + 4 * r + 2);
                if (s + l > 65535 || s <= last_end + 1) return 0;  // sorted, non-overlapping, non-adjacent
                last_end = s + l;
                c += (uint32_t)l + 1;
            }
            D.type.push_back(T_RUN);
            D.card.push_back(c);
            D.nruns.push_back(nr);
            D.push_payload(p, 4 * (size_t)nr);
            p += 4 * (size_t)nr;
        } else if (card > 4096) {
            if (p + 8192 > end) return 0;
            uint32_t c = 0;
            for (int w = 0; w < 1024; ++w) c += (uint32_t)__builtin_popcountll(rd64(p + 8 * w));
            if (c != card) return 0;
            D.type.push_back(T_BITSET);
            D.card.push_back(card);
            D.nruns.push_back(0);
            D.push_payload(p, 8192);
            p += 8192;
        } else {
            if (p + 2 * (size_t)card > end) return 0;
            for (uint32_t v = 1; v < card; ++v)
                if (rd16(p + 2 * v) <= rd16(p + 2 * v - 2)) return 0;
            D.type.push_back(T_ARRAY);
            D.card.push_back(card);
            D.nruns.push_back(0);
            D.push_payload(p, 2 * (size_t)card);
            p += 2 * (size_t)card;
        }
    }
    return (size_t)(p - buf);
}

// `into`: a pool that no batch reads any more, to be refilled -- its device buffers are kept (grown when needed), every
// cached host-side fact about the old content is dropped.  A per-call caller (roaring_compat.inc) then pays no
// hipMalloc / hipFree pair per buffer and call.  On failure `into` is destroyed like a fresh pool would be.
rhip_pool_t* upload(rhip_ctx_t* ctx, HostDir& D, uint32_t n_bitmaps, bool is64, rhip_pool_t* into = nullptr) {
    rhip_pool_t* P = into ? into : new rhip_pool_s();
    if (into) {
        rhip_pool_s fresh;
        std::swap(fresh.bm_start, P->bm_start); std::swap(fresh.key, P->key); std::swap(fresh.type, P->type);
        std::swap(fresh.card, P->card); std::swap(fresh.nruns, P->nruns); std::swap(fresh.off, P->off);
        std::swap(fresh.arena, P->arena);
        *P = std::move(fresh);
    }
    try {
        DeviceGuard dguard_(ctx->device);
        P->ctx = ctx;
        P->n_bitmaps = n_bitmaps;
        P->n_cont = D.key.size();
        P->is64 = is64;
        if (P->n_cont >= 0xFFFFFFF0ull) {
            set_err("too many containers for one pool");
            throw (int)RHIP_ERR_ARG;
        }
        ensure_dir(P, n_bitmaps, P->n_cont);
        D.arena.resize(D.arena.size() + 64, 0);  // tail slack for 16-byte over-reads
        P->arena.ensure(D.arena.size());
        P->arena_used = D.arena.size();
        hipStream_t s = ctx->stream;
        const size_t small_total = 8 * D.bm_start.size() + 29 * D.key.size() + D.arena.size() + 7 * 16;
        if (small_total <= (1u << 20)) {
            // A small pool (the per-call drop-ins upload two bitmaps a call): seven copy commands from pageable memory
            // are seven synchronous staging round trips inside the runtime.  Pack the seven arrays into the context's
            // pinned staging buffer (free between synchronous calls) and issue seven asynchronous copies from there.
            ctx->ensure_stage(rhip_ctx_s::SYNC_SLOT, small_total);
            char* hs = (char*)ctx->h_stage[rhip_ctx_s::SYNC_SLOT];
            size_t at = 0;
            auto put = [&](void* dst, const void* src, size_t n) {
                if (!n) return;
                memcpy(hs + at, src, n);
                HIPCHK(hipMemcpyAsync(dst, hs + at, n, hipMemcpyHostToDevice, s));
                at += (n + 15) & ~(size_t)15;
            };
            put(P->bm_start.p, D.bm_start.data(), 8 * D.bm_start.size());
            put(P->key.p, D.key.data(), 8 * D.key.size());
            put(P->type.p, D.type.data(), D.type.size());
            put(P->card.p, D.card.data(), 4 * D.card.size());
            put(P->nruns.p, D.nruns.data(), 4 * D.nruns.size());
            put(P->off.p, D.off.data(), 8 * D.off.size());
            put(P->arena.p, D.arena.data(), D.arena.size());
        } else {
        HIPCHK(hipMemcpyAsync(P->bm_start.p, D.bm_start.data(), 8 * D.bm_start.size(), hipMemcpyHostToDevice, s));
        if (P->n_cont) {
            HIPCHK(hipMemcpyAsync(P->key.p, D.key.data(), 8 * D.key.size(), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(P->type.p, D.type.data(), D.type.size(), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(P->card.p, D.card.data(), 4 * D.card.size(), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(P->nruns.p, D.nruns.data(), 4 * D.nruns.size(), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(P->off.p, D.off.data(), 8 * D.off.size(), hipMemcpyHostToDevice, s));
        }
        HIPCHK(hipMemcpyAsync(P->arena.p, D.arena.data(), D.arena.size(), hipMemcpyHostToDevice, s));
        }
        HIPCHK(hipStreamSynchronize(s));
        // keep the host mirror of the directory: it is already here
        P->h_bm_start.swap(D.bm_start); P->h_key.swap(D.key); P->h_off.swap(D.off);
        P->h_type.swap(D.type); P->h_card.swap(D.card); P->h_nruns.swap(D.nruns);
        P->host_dir = true;
        return P;
    } catch (int) {
        P->release();
        delete P;
        return nullptr;
    }
}
}  // namespace

static bool use_device_parser(size_t n, const size_t* lens, size_t& total);
static rhip_pool_t* portable_via_device(rhip_ctx_t* c, size_t n, const char* const* bufs, const size_t* lens, size_t total,
                                        int is64);

extern "C" rhip_pool_t* rhip_pool_from_portable(rhip_ctx_t* ctx, size_t n, const char* const* bufs,
                                                const size_t* lens) {
    if (!ctx) { set_err("null context"); return nullptr; }
    size_t total = 0;
    if (n && use_device_parser(n, lens, total)) return portable_via_device(ctx, n, bufs, lens, total, 0);
    HostDir D;
    D.bm_start.reserve(n + 1);
    for (size_t i = 0; i < n; ++i) {
        D.bm_start.push_back(D.key.size());
        if (!parse_portable32(bufs[i], lens[i], 0, D)) {
            set_err("bitmap %zu: malformed or invalid portable buffer", i);
            return nullptr;
        }
    }
    D.bm_start.push_back(D.key.size());
    return upload(ctx, D, (uint32_t)n, false);
}

extern "C" rhip_pool_t* rhip_pool_from_portable64(rhip_ctx_t* ctx, size_t n, const char* const* bufs,
                                                  const size_t* lens) {
    if (!ctx) { set_err("null context"); return nullptr; }
    size_t total = 0;
    if (n && use_device_parser(n, lens, total)) return portable_via_device(ctx, n, bufs, lens, total, 1);
    HostDir D;
    for (size_t i = 0; i < n; ++i) {
        D.bm_start.push_back(D.key.size());
        const char* p = bufs[i];
        const char* end = p + lens[i];
        if (lens[i] < 8) { set_err("bitmap %zu: truncated 64-bit buffer", i); return nullptr; }
        uint64_t nb = rd64(p);
        p += 8;
        int64_t prev = -1;
        for (uint64_t b = 0; b < nb; ++b) {
            if (p + 4 > end) { set_err("bitmap %zu: truncated bucket", i); return nullptr; }
            uint32_t high = rd32(p);
            p += 4;
            if ((int64_t)high <= prev) { set_err("bitmap %zu: buckets not ascending", i); return nullptr; }
            prev = high;
            size_t used = parse_portable32(p, (size_t)(end - p), high, D);
            if (!used) { set_err("bitmap %zu bucket %llu: malformed", i, (unsigned long long)b); return nullptr; }
            p += used;
        }
    }
    D.bm_start.push_back(D.key.size());
    return upload(ctx, D, (uint32_t)n, true);
}

extern "C" void rhip_pool_free(rhip_pool_t* P) {
    if (!P) return;
    if (P->in_use > 0 || P->list_pins > 0) {  // an operand of batches in flight / of prepared pair lists: released by the last of them
        P->free_deferred = true;
        return;
    }
    // A many-way result goes back to its context, buffers and all: hipFree synchronises with the device and hipMalloc of
    // the next result costs as much again -- together more than the aggregation of a small bitmap set itself.
    if (P->from_many && many_recycle_push(P)) return;
    arena_park(P);  // (a measured placement is worth keeping: the next result pool for the same operand takes it back)
    P->release();  // hipFree synchronises with the device; the context may already be gone
    delete P;
}
// A result arena that place_arena chose goes to its context's spares instead of back to the driver (at most two parked
// winners per context; rhip_ctx_trim, rhip_ctx_destroy and a failing allocation release them like every spare)
static void arena_park(rhip_pool_t* P) {
    if (!P->arena.base || !P->arena.placed_for || P->pending) return;
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    rhip_ctx_t* c = P->ctx;
    if (!g_live_ctx.count(c) || !c->arena_keep_spares) return;
    int parked = 0;
    for (const DBuf& b : c->arena_spares) parked += b.placed_winner ? 1 : 0;  // (the losers a search left behind do not take a winner's place)
    if (parked >= 2) return;
    c->arena_spares.push_back(P->arena);
    P->arena.base = nullptr; P->arena.p = nullptr; P->arena.cap = 0;  // (ownership moved)
}
static bool many_recycle_push(rhip_pool_t* P) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    rhip_ctx_t* c = P->ctx;
    if (!g_live_ctx.count(c) || c->gen != P->ctx_gen || c->many_free.size() >= 2) return false;
    c->many_free.push_back(P);
    return true;
}
static rhip_pool_t* many_recycle_pop(rhip_ctx_t* c) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (c->many_free.empty()) return nullptr;
    rhip_pool_t* R = c->many_free.back();
    c->many_free.pop_back();
    return R;
}
extern "C" uint32_t rhip_pool_size(const rhip_pool_t* P) { return P->n_bitmaps; }
extern "C" uint64_t rhip_pool_containers(const rhip_pool_t* P) { return P->n_cont; }
extern "C" int rhip_pool_is64(const rhip_pool_t* P) { return P->is64 ? 1 : 0; }
namespace { static void fetch_bounds(rhip_pool_t* P); }
extern "C" int rhip_pool_max_key(rhip_pool_t* P, uint64_t* out) {
    try {
        if (!P || !out) { set_err("null argument"); throw (int)RHIP_ERR_ARG; }
        if (P->pending) { set_err("pool is the result of a batch still in flight"); throw (int)RHIP_ERR_ARG; }
        DeviceGuard dguard_(P->ctx->device);
        fetch_bounds(P);
        *out = P->max_key;
        return RHIP_OK;
    } catch (int e) { return e; }
}

static void pool_payload_stats(rhip_pool_t* P, uint64_t out[4]) {
    rhip_ctx_t* c = P->ctx;
    c->misc.ensure(64);
    HIPCHK(hipMemsetAsync(c->misc.p, 0, 32, c->stream));
    if (P->n_cont)
        hipLaunchKernelGGL(k_payload_stats, dim3((unsigned)((P->n_cont + 255) / 256)), dim3(256), 0, c->stream,
                           P->type.as<uint8_t>(), P->card.as<uint32_t>(), P->nruns.as<uint32_t>(), (u64)P->n_cont,
                           c->misc.as<u64>());
    HIPCHK(hipMemcpyAsync(c->h_pinned, c->misc.p, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(out, c->h_pinned, 32);
}
extern "C" uint64_t rhip_pool_payload_bytes(rhip_pool_t* P) {
    try {
        DeviceGuard dguard_(P->ctx->device);
        uint64_t o[4];
        pool_payload_stats(P, o);
        return o[0];
    } catch (int) { return 0; }
}
extern "C" uint64_t rhip_pool_arena_bytes(const rhip_pool_t* P) { return P ? P->arena_used : 0; }
extern "C" uint32_t rhip_pool_payload_align(const rhip_pool_t* P) { return P ? P->pay_align : 0; }
extern "C" int rhip_pool_type_counts(rhip_pool_t* P, uint64_t out[3]) {
    try {
        DeviceGuard dguard_(P->ctx->device);
        uint64_t o[4];
        pool_payload_stats(P, o);
        out[0] = o[1]; out[1] = o[2]; out[2] = o[3];
        return RHIP_OK;
    } catch (int e) { return e; }
}

extern "C" rhip_pool_t* rhip_pool_synth_bitset(rhip_ctx_t* ctx, uint32_t n_bitmaps, uint32_t n_containers,
                                               uint64_t seed) {
    if (!ctx) { set_err("null context"); return nullptr; }
    rhip_pool_t* P = new rhip_pool_s();
    try {
        DeviceGuard dguard_(ctx->device);
        P->ctx = ctx;
        P->n_bitmaps = n_bitmaps;
        P->n_cont = (uint64_t)n_bitmaps * n_containers;
        if (n_containers > 65536 || P->n_cont >= 0xFFFFFFF0ull) { set_err("bad synth shape"); throw (int)RHIP_ERR_ARG; }
        ensure_dir(P, n_bitmaps, P->n_cont);
        P->arena_used = P->n_cont * 8192ull + 64;
        {   // (a payload that IS a power of two -- 256 x 4096 containers = 8 GiB -- is allocated as exactly that: the 64 bytes
            // of slack would double it, and no kernel reads past the last container of a bitset-only pool)
            const uint64_t pay = P->n_cont * 8192ull;
            P->arena.pow2_large = ctx->arena_pow2;
            P->arena.ensure((ctx->arena_pow2 && pay >= (1ull << 30) && (pay & (pay - 1)) == 0) ? pay : P->arena_used);
            if (P->arena_used > P->arena.cap) P->arena_used = P->arena.cap;  // (the exact allocation: nothing may copy arena_used bytes past it)
        }
        hipLaunchKernelGGL(k_synth_fill, dim3(256 * 16), dim3(256), 0, ctx->stream, P->arena.as<u64>(), n_bitmaps,
                           n_containers, (u64)seed);
        if (P->n_cont)
            hipLaunchKernelGGL(k_synth_dir, dim3((unsigned)((P->n_cont * 64 + 255) / 256)), dim3(256), 0, ctx->stream,
                               P->arena.as<u64>(), n_bitmaps, n_containers, P->bm_start.as<u64>(), P->key.as<u64>(),
                               P->type.as<uint8_t>(), P->card.as<uint32_t>(), P->nruns.as<uint32_t>(),
                               P->off.as<u64>());
        else
            HIPCHK(hipMemsetAsync(P->bm_start.p, 0, 8 * ((size_t)n_bitmaps + 1), ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return P;
    } catch (int) {
        P->release();
        delete P;
        return nullptr;
    }
}

// ------------------------------------------------------------------ download (portable format)
static void fetch_bm_start(rhip_pool_t* P) {
    if (P->host_dir || P->host_bm) return;
    P->h_bm_start.resize((size_t)P->n_bitmaps + 1);
    HIPCHK(hipMemcpyAsync(P->h_bm_start.data(), P->bm_start.p, 8 * P->h_bm_start.size(), hipMemcpyDeviceToHost,
                          P->ctx->stream));
    HIPCHK(hipStreamSynchronize(P->ctx->stream));
    P->host_bm = true;
}
// Device -> host reads of a few small arrays (a per-call result's directory and payload): a copy command into
// pageable memory is a synchronous staging round trip each, so they land in the context's pinned staging buffer
// (free between synchronous calls) with ONE wait, and are copied out from there.
struct SmallReads {
    rhip_ctx_t* c;
    struct Rd { void* dst; size_t at, n; };
    std::vector<Rd> rds;
    size_t at = 0;
    static constexpr size_t LIMIT = 1u << 20;
    explicit SmallReads(rhip_ctx_t* ctx, size_t total) : c(ctx) { c->ensure_stage(rhip_ctx_s::SYNC_SLOT, total + 16 * 8); }
    void get(void* dst, const void* src, size_t n) {
        if (!n) return;
        HIPCHK(hipMemcpyAsync((char*)c->h_stage[rhip_ctx_s::SYNC_SLOT] + at, src, n, hipMemcpyDeviceToHost, c->stream));
        rds.push_back(Rd{dst, at, n});
        at += (n + 15) & ~(size_t)15;
    }
    void finish() {
        HIPCHK(hipStreamSynchronize(c->stream));
        for (const Rd& r : rds) memcpy(r.dst, (char*)c->h_stage[rhip_ctx_s::SYNC_SLOT] + r.at, r.n);
    }
};
static void fetch_dir(rhip_pool_t* P) {
    if (P->host_dir) return;
    hipStream_t s = P->ctx->stream;
    P->h_bm_start.resize((size_t)P->n_bitmaps + 1);
    P->h_key.resize(P->n_cont); P->h_off.resize(P->n_cont); P->h_type.resize(P->n_cont);
    P->h_card.resize(P->n_cont); P->h_nruns.resize(P->n_cont);
    const size_t total = 8 * P->h_bm_start.size() + 29 * (size_t)P->n_cont;
    if (total <= SmallReads::LIMIT) {
        SmallReads R(P->ctx, total);
        R.get(P->h_bm_start.data(), P->bm_start.p, 8 * P->h_bm_start.size());
        R.get(P->h_key.data(), P->key.p, 8 * P->n_cont);
        R.get(P->h_off.data(), P->off.p, 8 * P->n_cont);
        R.get(P->h_type.data(), P->type.p, P->n_cont);
        R.get(P->h_card.data(), P->card.p, 4 * P->n_cont);
        R.get(P->h_nruns.data(), P->nruns.p, 4 * P->n_cont);
        R.finish();
        P->host_dir = true;
        return;
    }
    HIPCHK(hipMemcpyAsync(P->h_bm_start.data(), P->bm_start.p, 8 * P->h_bm_start.size(), hipMemcpyDeviceToHost, s));
    if (P->n_cont) {
        HIPCHK(hipMemcpyAsync(P->h_key.data(), P->key.p, 8 * P->n_cont, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(P->h_off.data(), P->off.p, 8 * P->n_cont, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(P->h_type.data(), P->type.p, P->n_cont, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(P->h_card.data(), P->card.p, 4 * P->n_cont, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(P->h_nruns.data(), P->nruns.p, 4 * P->n_cont, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    P->host_dir = true;
}
static inline size_t h_payload(const rhip_pool_t* P, uint64_t c) {
    uint8_t t = P->h_type[c];
    return t == T_BITSET ? 8192 : (t == T_ARRAY ? 2 * (size_t)P->h_card[c] : 4 * (size_t)P->h_nruns[c]);
}
// size of the 32-bit portable image of containers [c0, c1) (ra_portable_size_in_bytes,
// src/roaring_array.c:445-466)
static size_t portable32_size(const rhip_pool_t* P, uint64_t c0, uint64_t c1) {
    size_t n = (size_t)(c1 - c0);
    bool hasrun = false;
    size_t s = 0;
    for (uint64_t c = c0; c < c1; ++c) {
        if (P->h_type[c] == T_RUN) { hasrun = true; s += 2; }
        s += h_payload(P, c);
    }
    if (hasrun) s += (n < 4) ? 4 + (n + 7) / 8 + 4 * n : 4 + (n + 7) / 8 + 8 * n;
    else s += 8 + 8 * n;
    return s;
}
// ra_portable_serialize, src/roaring_array.c:469-531; payload read from `host_arena` whose
// byte 0 corresponds to device arena offset `base`.
static size_t portable32_write(const rhip_pool_t* P, uint64_t c0, uint64_t c1, const uint8_t* host_arena,
                               uint64_t base, char* buf) {
    char* p = buf;
    uint32_t n = (uint32_t)(c1 - c0);
    bool hasrun = false;
    for (uint64_t c = c0; c < c1; ++c) hasrun |= (P->h_type[c] == T_RUN);
    uint32_t start;
    if (hasrun) {
        uint32_t cookie = 12347u | ((n - 1) << 16);
        memcpy(p, &cookie, 4); p += 4;
        size_t s = (n + 7) / 8;
        memset(p, 0, s);
        for (uint32_t i = 0; i < n; ++i)
            if (P->h_type[c0 + i] == T_RUN) p[i / 8] |= (char)(1 << (i % 8));
        p += s;
        start = (uint32_t)(n < 4 ? 4 + 4 * n + s : 4 + 8 * n + s);
    } else {
        uint32_t cookie = 12346u;
        memcpy(p, &cookie, 4); p += 4;
        memcpy(p, &n, 4); p += 4;
        start = 8 + 8 * n;
    }
    for (uint32_t i = 0; i < n; ++i) {
        uint16_t k = (uint16_t)(P->h_key[c0 + i] & 0xFFFF), cm1 = (uint16_t)(P->h_card[c0 + i] - 1);
        memcpy(p, &k, 2); memcpy(p + 2, &cm1, 2); p += 4;
    }
    if (!hasrun || n >= 4) {
        uint32_t o = start;
        for (uint32_t i = 0; i < n; ++i) {
            memcpy(p, &o, 4); p += 4;
            o += (uint32_t)(h_payload(P, c0 + i) + (P->h_type[c0 + i] == T_RUN ? 2 : 0));
        }
    }
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t c = c0 + i;
        if (P->h_type[c] == T_RUN) { uint16_t nr = (uint16_t)P->h_nruns[c]; memcpy(p, &nr, 2); p += 2; }
        size_t b = h_payload(P, c);
        memcpy(p, host_arena + (P->h_off[c] - base), b);
        p += b;
    }
    return (size_t)(p - buf);
}

extern "C" size_t rhip_pool_portable_size(rhip_pool_t* P, uint32_t i) {
    try {
        if (!P || i >= P->n_bitmaps) { set_err("bitmap index out of range"); return 0; }
        DeviceGuard dguard_(P->ctx->device);
        fetch_dir(P);
        uint64_t c0 = P->h_bm_start[i], c1 = P->h_bm_start[i + 1];
        if (!P->is64) return portable32_size(P, c0, c1);
        size_t s = 8;
        uint64_t c = c0;
        while (c < c1) {
            uint64_t hi = P->h_key[c] >> 16, e = c;
            while (e < c1 && (P->h_key[e] >> 16) == hi) ++e;
            s += 4 + portable32_size(P, c, e);
            c = e;
        }
        return s;
    } catch (int) { return 0; }
}

extern "C" size_t rhip_pool_portable_serialize(rhip_pool_t* P, uint32_t i, char* buf) {
    try {
        if (!P || i >= P->n_bitmaps) { set_err("bitmap index out of range"); return 0; }
        DeviceGuard dguard_(P->ctx->device);
        fetch_dir(P);
        uint64_t c0 = P->h_bm_start[i], c1 = P->h_bm_start[i + 1];
        // payload span of this bitmap: slots are assigned in directory order, so it is one range
        std::vector<uint8_t> host;
        uint64_t base = 0;
        if (c1 > c0) {
            uint64_t lo = ~0ull, hi = 0;
            for (uint64_t c = c0; c < c1; ++c) {
                lo = std::min(lo, P->h_off[c]);
                hi = std::max(hi, P->h_off[c] + h_payload(P, c));
            }
            base = lo;
            host.resize((size_t)(hi - lo));
            HIPCHK(hipMemcpyAsync(host.data(), P->arena.as<uint8_t>() + lo, hi - lo, hipMemcpyDeviceToHost,
                                  P->ctx->stream));
            HIPCHK(hipStreamSynchronize(P->ctx->stream));
        }
        if (!P->is64) return portable32_write(P, c0, c1, host.data(), base, buf);
        char* p = buf;
        uint64_t nb = 0;
        for (uint64_t c = c0; c < c1;) {
            uint64_t hi = P->h_key[c] >> 16, e = c;
            while (e < c1 && (P->h_key[e] >> 16) == hi) ++e;
            ++nb;
            c = e;
        }
        memcpy(p, &nb, 8); p += 8;
        for (uint64_t c = c0; c < c1;) {
            uint64_t hi = P->h_key[c] >> 16, e = c;
            while (e < c1 && (P->h_key[e] >> 16) == hi) ++e;
            uint32_t h32 = (uint32_t)hi;
            memcpy(p, &h32, 4); p += 4;
            p += portable32_write(P, c, e, host.data(), base, p);
            c = e;
        }
        return (size_t)(p - buf);
    } catch (int) { return 0; }
}

extern "C" int rhip_pool_cardinalities(rhip_pool_t* P, uint64_t* out) {
    try {
        DeviceGuard dguard_(P->ctx->device);
        if (P->h_cards.size() != P->n_bitmaps) {
            rhip_ctx_t* c = P->ctx;
            std::vector<uint64_t> tmp(P->n_bitmaps);
            if (P->n_bitmaps) {
                c->misc.ensure(8 * (size_t)P->n_bitmaps);
                hipLaunchKernelGGL(k_bitmap_cards, dim3((unsigned)(((size_t)P->n_bitmaps * 64 + 255) / 256)), dim3(256),
                                   0, c->stream, P->view(), P->n_bitmaps, c->misc.as<u64>());
                HIPCHK(hipMemcpyAsync(tmp.data(), c->misc.p, 8 * (size_t)P->n_bitmaps, hipMemcpyDeviceToHost,
                                      c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
            }
            P->h_cards.swap(tmp);
        }
        if (P->n_bitmaps) memcpy(out, P->h_cards.data(), 8 * (size_t)P->n_bitmaps);
        return RHIP_OK;
    } catch (int e) { return e; }
}

// ------------------------------------------------------------------ pairwise pipeline
namespace {
// Host mirrors a batch plan needs from an operand pool: container counts per bitmap (h_bm_start), the per-bitmap
// result-slot bound W (k_bitmap_bounds) and which container types occur at all.  Fetched once per pool, lazily.
static void fetch_bounds(rhip_pool_t* P) {
    fetch_bm_start(P);
    if (P->host_w) return;
    rhip_ctx_t* c = P->ctx;
    P->h_w.assign((size_t)P->n_bitmaps, 0);
    P->h_wm.assign((size_t)P->n_bitmaps, 0);
    uint32_t census[4] = {0, 0, 0, 0};
    uint64_t nkeys = P->n_cont;
    if (P->n_bitmaps && P->n_cont) {
        const size_t nb = (size_t)P->n_bitmaps;
        // misc2: W[nb] | Wmany[nb] | census (3 x u32, padded to 16 bytes) | distinct-key count, largest key (u64 each) |
        // key bits (2048 x u32)
        c->misc2.ensure(16 * nb + 32 + 8192 + 64);
        u64* dw = c->misc2.as<u64>();
        u64* dwm = dw + nb;
        uint32_t* dcensus = (uint32_t*)(dwm + nb);
        u64* dnk = (u64*)(dcensus + 4);
        uint32_t* dbits = (uint32_t*)(dnk + 2);
        HIPCHK(hipMemsetAsync(dcensus, 0, 32 + 8192, c->stream));
        hipLaunchKernelGGL(k_bitmap_bounds, dim3((unsigned)((nb * 64 + 255) / 256)), dim3(256), 0, c->stream,
                           P->view(), P->n_bitmaps, dw, dwm, dcensus, dnk + 1);
        if (!P->is64) {
            hipLaunchKernelGGL(k_key_mark, dim3((unsigned)std::min<uint64_t>((P->n_cont + 255) / 256, 2048)), dim3(256), 0,
                               c->stream, P->key.as<u64>(), (u64)P->n_cont, dbits);
            hipLaunchKernelGGL(k_key_count, dim3(1), dim3(256), 0, c->stream, dbits, dnk);
            HIPCHK(hipMemcpyAsync(&nkeys, dnk, 8, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(hipMemcpyAsync(P->h_w.data(), dw, 8 * nb, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(P->h_wm.data(), dwm, 8 * nb, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(census, dcensus, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&P->max_key, dnk + 1, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    for (int t = 0; t < 3; ++t) P->census[t] = census[t] ? 1 : 0;
    P->has_long_runs = census[3] != 0;
    P->n_keys_distinct = nkeys;
    P->wm_total = 0;
    P->n_run_cont = 0;
    for (uint64_t& w : P->h_wm) {  // (k_bitmap_bounds packs the bitmap's run-container count above the bound)
        P->n_run_cont += w >> WM_RUNS_SHIFT;
        w &= (1ull << WM_RUNS_SHIFT) - 1ull;
        P->wm_total += w;
    }
    P->h_n.resize((size_t)P->n_bitmaps);
    P->max_n = 0;
    for (uint32_t b = 0; b < P->n_bitmaps; ++b) {
        P->h_n[b] = (uint32_t)(P->h_bm_start[b + 1] - P->h_bm_start[b]);
        P->max_n = std::max(P->max_n, P->h_n[b]);
    }
    static uint64_t next_gen = 0;  // (process-wide: a recycled pool at the same address never repeats a number)
    P->bounds_gen = __atomic_add_fetch(&next_gen, 1, __ATOMIC_RELAXED);
    P->host_w = true;
}

// device scratch of one call (inside ctx->misc, all of it cleared by k_count): u64 words
struct PlanScratch {
    size_t n_scan_tiles, n_tail_tiles;
    size_t w_scan_status, w_tail_status, w_tail_part, w_tickets, w_retry, w_ranges, w_join, n_words;
    void layout(size_t n_scan_elems, size_t ub_cand) {
        n_scan_tiles = (n_scan_elems + SCAN_TILE - 1) / SCAN_TILE + 1;
        n_tail_tiles = (ub_cand + TAIL_TILE - 1) / TAIL_TILE + 1;
        size_t w = 0;
        w_scan_status = w; w += n_scan_tiles;
        w_tail_status = w; w += n_tail_tiles;
        w_tail_part = w; w += 2 * n_tail_tiles;
        // every counter that many blocks hit with atomics sits alone in a 128-byte line: device-scope atomics on one
        // line serialise at the memory side, and plain loads of that line (the section ranges) queue behind them
        w = (w + 15) & ~(size_t)15;
        w_tickets = w; w += 2 * 16;
        w_retry = w; w += 16;
        w_ranges = w; w += (2 * N_SEC + 3 + 15) & ~15;  // (+ 3: the group boundaries of a grouped batch)
        w_join = w; w += 16;  // "auxiliary stream a has finished its class kernels" (k_join_signal -> k_tail)
        n_words = w;
    }
};

struct Plan {
    PlanScratch sc;
    size_t npairs = 0, NU = 0, S = 0;
    uint64_t ub_match = 0, ub_cand = 0, arena_bound = 0, work_bound = 0;
    uint32_t plan_group = 64;  // lanes per planning unit
    int slot = 0;
    hipStream_t plan_stream = nullptr;
    bool may_bb = true, may_filt = true, may_wave = true, may_runs = true, may_copy = true, may_ba = true;
    bool grouped = false;  // the filter / union items are queued by X container (k_filter_g / k_union_g)
    bool runs_dominant = false;  // at least half of the operand pools' containers are run containers (C5, wikileaks): the interval kernel is the batch's longest
    uint32_t copy_per_wave = 4;  // pass-through items a wave of k_copy takes at a time: 16 when the pools hold tiny containers
    const u64* xranges() const { return words + sc.w_ranges + 2 * N_SEC; }
    // where the planning kernels left what the class kernels and the tail only READ -- the class queues (all but the
    // retry queue), the candidate directory: the slot's scratch, or the entry of the pair list's plan cache
    rhip_ctx_s::SlotScratch* Q = nullptr;
    ::PlanCacheEntry* ce = nullptr;  // the cache entry this plan was taken from / saved to (its buffers are Q)
    bool cache_hit = false;
    DBuf& q(rhip_ctx_t* c, int cls) const { return (cls == CLS_RETRY || !Q) ? c->ss[slot].q[cls] : Q->q[cls]; }
    // device pointers
    uint32_t *d_lhs = nullptr, *d_rhs = nullptr, *d_upair = nullptr, *d_utile = nullptr;
    u64* d_pair0 = nullptr;
    u64* words = nullptr;
    CandOut CO{nullptr, nullptr, nullptr};
    u64* ranges() const { return words + sc.w_ranges; }
    uint32_t* retry_count() const { return (uint32_t*)(words + sc.w_retry); }
    LbState scan_lb() const { return LbState{words + sc.w_scan_status, (uint32_t*)(words + sc.w_tickets)}; }
    LbState tail_lb() const { return LbState{words + sc.w_tail_status, (uint32_t*)(words + sc.w_tickets + 16)}; }
    u64* tail_part() const { return words + sc.w_tail_part; }
    u64* join_flags() const { return words + sc.w_join; }
};

}  // namespace
// One cached plan of a prepared pair list: key = (ops, cardinality mode, the operand pools' bounds_gen -- an in-place
// update or a reload of a pool gives it a new one).  `ss` holds the buffers the class kernels only read (queues, candidate
// directory), `ranges` the section ranges k_scan computed (they live in the slot's scratch words, which every call
// clears: k_plan_restore writes them back).  An entry is re-planned only while no batch in flight reads it.
struct PlanCacheEntry {
    bool valid = false;
    uint32_t ops_packed = 0;
    int n_ops = 0, cardmode = 0;
    uint64_t genA = 0, genB = 0, stamp = 0;
    int in_use = 0;
    Plan P;
    rhip_ctx_s::SlotScratch ss;
    DBuf ranges;
};
struct PlanCache {
    static constexpr int N = 6;  // and / or / xor / andnot, a multi-op batch, a cardinality form
    PlanCacheEntry e[N];
    uint64_t clock = 0;
    void release() {
        for (auto& x : e) { x.ss.release(); x.ranges.release(); x.valid = false; }
    }
};
namespace {
constexpr uint32_t N_RANGE_WORDS = 2 * N_SEC + 3;

void check_pair_args(rhip_pool_t* A, rhip_pool_t* B, size_t npairs, const uint32_t* lhs, const uint32_t* rhs) {
    if (!A || !B) { set_err("null pool"); throw (int)RHIP_ERR_ARG; }
    if (A->pending || B->pending) { set_err("operand pool is the result of a batch still in flight"); throw (int)RHIP_ERR_ARG; }
    if (npairs && (!lhs || !rhs)) { set_err("null pair index array"); throw (int)RHIP_ERR_ARG; }
    if (A->is64 != B->is64) { set_err("mixing 32-bit and 64-bit pools"); throw (int)RHIP_ERR_ARG; }
    if (A->ctx->device != B->ctx->device) { set_err("operand pools live on different devices"); throw (int)RHIP_ERR_ARG; }
    if (npairs >= 0x3FFFFFF0ull) { set_err("too many pairs"); throw (int)RHIP_ERR_ARG; }
    // (the bitmap indices themselves are range-checked by the planning pass that reads them anyway)
}

template <int OP>
void launch_bb(rhip_ctx_t* c, hipStream_t st, unsigned grid, const PoolView& A, const PoolView& B, const OutView& O, const Plan& P,
               int cardmode) {
    hipLaunchKernelGGL(k_bb<OP>, dim3(grid), dim3(256), 0, st, A.arena, B.arena, O, P.q(c, CLS_BB).as<BBItem>(),
                       P.ranges() + 2 * SEC_BB, cardmode, c->pair_acc.as<u64>(), P.q(c, CLS_RETRY).as<GenItem>(),
                       P.retry_count());
}

unsigned persistent_grid(uint64_t n_items, unsigned items_per_block, unsigned max_blocks) {
    uint64_t need = (n_items + items_per_block - 1) / items_per_block;
    if (need < 1) need = 1;
    return (unsigned)std::min<uint64_t>(need, max_blocks);
}

// Host half of the plan: units, upper bounds (nothing here waits for the device), ONE host-to-device copy of the
// batch description, then k_count -> k_scan -> k_emit.  On return the class queues are filled at deterministic
// positions and `ranges` (device) holds every section's [begin, end).
struct HostClock {  // phase p accumulates the host time between the previous lap and lap(p)
    rhip_ctx_t* c;
    std::chrono::steady_clock::time_point t;
    explicit HostClock(rhip_ctx_t* c_) : c(c_), t(std::chrono::steady_clock::now()) {}
    void lap(int p) {
        const auto n = std::chrono::steady_clock::now();
        c->hclk[p] += std::chrono::duration<double, std::micro>(n - t).count();
        t = n;
    }
};
// the ops of one batch: one for rhip_pairwise, up to four for rhip_pairwise_multi (result bitmap o * npairs + k =
// op[o] of pair k; on the device such a batch is n x npairs VIRTUAL pairs, see UnitView)
struct OpSet {
    int n = 1;
    int op[4] = {0, 0, 0, 0};
    uint32_t packed() const { uint32_t v = 0; for (int i = 0; i < n; ++i) v |= (uint32_t)op[i] << (2 * i); return v; }
    bool has(int o) const { for (int i = 0; i < n; ++i) if (op[i] == o) return true; return false; }
    int kop() const { return n == 1 ? op[0] : (int)OP_ITEM; }  // kernel argument: the op, or "read it from the item"
};
PairSums pair_sums(rhip_pool_t* A, rhip_pool_t* B, size_t npairs, const uint32_t* lhs, const uint32_t* rhs) {
    PairSums S;
    const uint32_t* nAv = A->h_n.data();
    const uint32_t* nBv = B->h_n.data();
    const uint64_t* wAv = A->h_w.data();
    const uint64_t* wBv = B->h_w.data();
    const uint32_t nbmA = A->n_bitmaps, nbmB = B->n_bitmaps;
    for (size_t i = 0; i < npairs; ++i) {
        const uint32_t l = lhs[i], r = rhs[i];
        if (l >= nbmA || r >= nbmB) { set_err("pair %zu: bitmap index out of range", i); throw (int)RHIP_ERR_ARG; }
        const uint64_t nA = nAv[l], nB = nBv[r], wA = wAv[l], wB = wBv[r];
        S.nu_a += (nA + 255) / 256;
        S.nu_b += (nB + 255) / 256;
        S.s_mn += nA < nB ? nA : nB;
        S.s_na += nA; S.s_nb += nB;
        S.s_wmin += wA < wB ? wA : wB;
        S.s_wa += wA; S.s_wb += wB;
    }
    return S;
}
// L != nullptr: the batch runs over a prepared pair list (lhs / rhs are its host copies): its sums are taken as they
// are while the operand pools' mirrors are the ones they were computed from, and with implicit units nothing is staged
// at all -- the planning kernels read the list's own device copy.
Plan plan(rhip_ctx_t* c, const OpSet& ops, rhip_pool_t* A, rhip_pool_t* B, size_t npairs, const uint32_t* lhs,
          const uint32_t* rhs, int cardmode, int slot, hipStream_t s, HostClock* clk = nullptr, rhip_pairlist_t* L = nullptr) {
    // (s = the stream the planning kernels may use instead of the main one; taken below if the batch is not a huge one)
    rhip_ctx_s::SlotScratch& SS = c->ss[slot];
    Plan P;
    P.npairs = npairs;
    P.slot = slot;
    fetch_bounds(A);
    fetch_bounds(B);
    // ---- the pair list's plan cache: a hit starts the batch at its class kernels
    ::PlanCacheEntry* ce = nullptr;
    if (L && c->plan_cache) {
        if (!L->cache) L->cache = new PlanCache();
        PlanCache& PC = *L->cache;
        const uint32_t opk = ops.packed();
        for (auto& e : PC.e) {
            if (!(e.valid && e.ops_packed == opk && e.n_ops == ops.n && e.cardmode == cardmode && e.genA == A->bounds_gen &&
                  e.genB == B->bounds_gen))
                continue;
            Plan H = e.P;
            H.slot = slot;
            H.cache_hit = true;
            SS.misc.ensure(8 * H.sc.n_words + 64);
            H.words = SS.misc.as<u64>();
            SS.q[CLS_RETRY].ensure(sizeof(GenItem) * (H.ub_match + 1));
            if (cardmode) c->pair_acc.ensure(8 * (npairs + 1));
            else SS.o_meta.ensure(8 * (H.ub_cand + 1));
            if (H.work_bound >= c->plan_overlap_max_bytes) s = c->stream;
            H.plan_stream = s;
            const size_t zt = std::max<size_t>(H.sc.n_words, cardmode ? npairs : 0);
            hipLaunchKernelGGL(k_plan_restore, dim3((unsigned)std::min<size_t>(256, (zt + 255) / 256 + 1)), dim3(256), 0, s, H.words,
                               (uint32_t)H.sc.n_words, (uint32_t)H.sc.w_ranges, (const u64*)e.ranges.as<u64>(), N_RANGE_WORDS,
                               cardmode ? c->pair_acc.as<u64>() : (u64*)nullptr, (uint32_t)npairs);
            e.stamp = ++PC.clock;
            if (clk) { clk->lap(0); clk->lap(2); }
            return H;
        }
        // a miss: the plan made below is left with an entry no batch in flight reads -- a free one, else the least recently used
        for (auto& e : PC.e) {
            if (e.in_use) continue;
            if (!e.valid) { ce = &e; break; }
            if (!ce || e.stamp < ce->stamp) ce = &e;
        }
        if (ce) ce->valid = false;
    }
    rhip_ctx_s::SlotScratch& SQ = ce ? ce->ss : SS;  // where the queues and the candidate directory go
    const size_t nvirt = npairs * (size_t)ops.n;  // virtual pairs = result bitmaps
    if (nvirt >= 0x3FFFFFF0ull) { set_err("too many result bitmaps"); throw (int)RHIP_ERR_ARG; }
    // per op: does the result carry B's unmatched containers (B-side tiles), and which bound formula applies
    bool any_btiles = false;
    int n_b0 = 0, n_b1 = 0, n_b2 = 0;  // ops by bound mode: and-like, andnot, or / xor
    for (int o = 0; o < ops.n; ++o) {
        const int op = ops.op[o];
        if (cardmode || op == OP_AND) ++n_b0;
        else if (op == OP_ANDNOT) ++n_b1;
        else { ++n_b2; any_btiles = true; }
    }
    const bool btiles = any_btiles;  // (a multi-op batch gives EVERY virtual pair its B-side units; dead ones count nothing)
    // ---- pass 1 over the pair list: range check, units and every upper bound (directory mirrors only)
    // No bitmap of either pool above 256 containers (one tile): the units are implicit -- unit = pair (and / andnot /
    // cardinality) or 2 pair + side (or / xor) -- and only the two index lists travel to the device.
    const bool implicit = A->max_n <= 256 && B->max_n <= 256 && c->explicit_units != 1;
    size_t NU = 0;
    uint64_t ub_match = 0, ub = 0, bound = 0;
    {
        PairSums S;
        if (L && L->genA == A->bounds_gen && L->genB == B->bounds_gen) {
            S = L->sums;
        } else {
            S = pair_sums(A, B, npairs, lhs, rhs);
            if (L) { L->sums = S; L->genA = A->bounds_gen; L->genB = B->bounds_gen; }
        }
        NU = (size_t)(ops.n * (S.nu_a + (btiles ? S.nu_b : 0)));
        ub_match = (uint64_t)ops.n * S.s_mn;
        // or / xor: a result holds at most one container per key that occurs in the operand POOLS at all (counted once
        // per pool, fetch_bounds) -- and none above a bitset's 8192 bytes unless a pool holds a longer run list.  Where
        // every bitmap has every key (C2: 4 096 keys) that halves the bound: a 250-pair `or` arena is 8 GiB, not 16.
        uint64_t cand2 = S.s_na + S.s_nb, bytes2 = S.s_wa + S.s_wb;
        if (!A->is64) {
            const uint64_t K = std::min<uint64_t>(65536, A->n_keys_distinct + (A == B ? 0 : B->n_keys_distinct));
            cand2 = std::min<uint64_t>(cand2, (uint64_t)npairs * K);
            if (!A->has_long_runs && !B->has_long_runs) bytes2 = std::min<uint64_t>(bytes2, (uint64_t)npairs * K * 8192ull);
        }
        ub = n_b0 * S.s_mn + n_b1 * S.s_na + n_b2 * cand2;
        bound = n_b0 * S.s_wmin + n_b1 * S.s_wa + n_b2 * bytes2;
    }
    if (implicit) NU = nvirt * (btiles ? 2 : 1);
    if (NU >= 0x7FFFFFF0ull) { set_err("batch too large: %zu planning units", NU); throw (int)RHIP_ERR_ARG; }
    if (ub >= 0xFFFFFFF0ull) { set_err("batch too large: %llu candidate containers", (unsigned long long)ub); throw (int)RHIP_ERR_ARG; }
    // staging layout (host pinned == device): lhs | rhs (u32 each) and, with explicit units, pair0[nvirt+1] u64 |
    // upair | utile (u32 each)
    const size_t o_lhs = 0, o_rhs = o_lhs + 4 * npairs, o_pair0 = (o_rhs + 4 * npairs + 7) & ~(size_t)7,
                 o_upair = o_pair0 + (implicit ? 0 : 8 * (nvirt + 1)), o_utile = o_upair + (implicit ? 0 : 4 * NU),
                 stage_bytes = o_utile + (implicit ? 0 : 4 * NU);
    const bool prestaged = L && implicit;  // the two index lists are all an implicit-unit batch stages
    if (!prestaged) c->ensure_stage(slot, stage_bytes + 16);
    char* hs = prestaged ? nullptr : (char*)c->h_stage[slot];
    if (npairs && !prestaged) {
        memcpy(hs + o_lhs, lhs, 4 * npairs);
        memcpy(hs + o_rhs, rhs, 4 * npairs);
    }
    if (!implicit) {   // ---- pass 2: the units, virtual pair by virtual pair
        uint64_t* pair0 = (uint64_t*)(hs + o_pair0);
        uint32_t* upair = (uint32_t*)(hs + o_upair);
        uint32_t* utile = (uint32_t*)(hs + o_utile);
        const uint32_t* nAv = A->h_n.data();
        const uint32_t* nBv = B->h_n.data();
        size_t u = 0;
        for (int o = 0; o < ops.n; ++o) {
            for (size_t i = 0; i < npairs; ++i) {
                const size_t v = (size_t)o * npairs + i;
                const uint32_t tA = (nAv[lhs[i]] + 255u) >> 8, tB = btiles ? (nBv[rhs[i]] + 255u) >> 8 : 0u;
                pair0[v] = u;
                for (uint32_t t = 0; t < tA; ++t) { upair[u] = (uint32_t)v; utile[u] = t; ++u; }
                for (uint32_t t = 0; t < tB; ++t) { upair[u] = (uint32_t)v; utile[u] = t | UNIT_B; ++u; }
            }
        }
        pair0[nvirt] = NU;
    }
    // beside an HBM-bound multi-gigabyte batch (C2: 24 GB per call) planning kernels cost the bitset kernel more
    // bandwidth than the 3 % of the call they would hide: those batches plan on the main stream
    if (bound >= c->plan_overlap_max_bytes) s = c->stream;
    P.plan_stream = s;
    if (clk) clk->lap(0);
    P.NU = NU;
    P.S = NU + 1;
    P.ub_match = ub_match;
    P.ub_cand = cardmode ? 0 : ub;
    P.arena_bound = cardmode ? 0 : bound;
    P.work_bound = bound;
    // tiny pass-through containers (pool average <= 96 payload bytes: C5, wikileaks): k_copy takes sixteen per wave
    {
        const uint64_t nc = A->n_cont + (A == B ? 0 : B->n_cont), by = A->arena_used + (A == B ? 0 : B->arena_used);
        P.copy_per_wave = (nc && by / nc <= 96 && c->copy_wide) ? 16u : 4u;
    }
    // which classes can occur at all (pool-level type census): a class that cannot is not launched
    auto has = [](const rhip_pool_t* X, int t) { return X->census[t] != 0; };
    const bool aB = has(A, 0), aA = has(A, 1), aR = has(A, 2), bB = has(B, 0), bA = has(B, 1), bR = has(B, 2);
    P.may_bb = aB && bB;
    P.may_runs = aR || bR;  // interval class and the general image class
    P.runs_dominant = 2 * (A->n_run_cont + B->n_run_cont) >= A->n_cont + B->n_cont;
    P.may_filt = P.may_wave = P.may_ba = P.may_copy = false;
    for (int o = 0; o < ops.n; ++o) {
        const int op = ops.op[o];
        if (cardmode || op == OP_AND) {
            P.may_filt |= (aA && (bA || bB)) || (bA && (aA || aB));
        } else if (op == OP_ANDNOT) {
            P.may_filt |= aA && (bA || bB);
            P.may_ba |= aB && bA;       // bitset \ array
            P.may_copy = true;
        } else {
            P.may_wave |= aA && bA;     // two arrays through the image (k_usmall takes the short-operand ones)
            P.may_ba |= (aA && bB) || (aB && bA);
            P.may_copy = true;
        }
    }
    // ---- X-grouped queues: when a container is expected to meet many partners (an all-pairs batch) and the batch has
    // image-class work at all.  ub_match is an upper bound, so the estimate errs towards grouping; a batch that groups
    // without reuse pays one atomic per item in k_count / k_emit and nothing else.
    // Only batches big enough to be forked (run_kernels: the same two bounds): a light batch runs its classes as the one
    // merged launch, which beats grouping it (measured, round 4: census-income andnot 0.24 ms merged, 0.39 grouped;
    // census1881 and 0.135 / 0.160; weather -- forked -- gains 3-5 %).
    const bool same_pool = A == B;
    const uint64_t nxc = same_pool ? A->n_cont : A->n_cont + B->n_cont;
    const bool big = bound >= c->fork_min_bytes && !(ub_match <= (128u << 10) && bound < c->fork_light_bytes);
    P.grouped = c->group_x == 2 ? (P.may_filt || P.may_wave || P.may_ba) && nxc > 0 && nxc < 0x3FFFFFF0ull
                                : c->group_x == 1 && big && (P.may_filt || P.may_wave || P.may_ba) && nxc > 0 && nxc < 0x3FFFFFF0ull &&
                                      ub_match >= c->group_min_items && ub_match >= c->group_min_reuse * nxc;
    const size_t n_hist = P.grouped ? 2 * (size_t)nxc + 1 : 0;
    // ---- device scratch
    const size_t S = P.S;
    P.sc.layout(N_SEC * S + n_hist, P.ub_cand);
    SS.plan_in.ensure(stage_bytes + 16);
    SS.cand.ensure(4 * (N_SEC * S + n_hist + 8));
    SS.cand_start.ensure(8 * (N_SEC * S + n_hist + 8));
    XGroupView XG{nullptr, nullptr, 0, 0, 0};
    if (P.grouped) {
        uint32_t* hist = SS.cand.as<uint32_t>() + N_SEC * S;
        XG = XGroupView{hist, SS.cand_start.as<u64>() + N_SEC * S, (uint32_t)nxc, same_pool ? 0u : (uint32_t)A->n_cont, 1u};
        if (SS.hist_at != (void*)hist || SS.hist_n != n_hist || SS.hist_gen != SS.cand.gen)  // (else: still zero)
            HIPCHK(hipMemsetAsync(hist, 0, 4 * n_hist, s));
    }
    SS.hist_at = nullptr;  // until this batch's k_emit is enqueued the area is not known to come back to zero
    SS.misc.ensure(8 * P.sc.n_words + 64);
    P.words = SS.misc.as<u64>();
    // (a cache entry keeps what it holds for the life of the pair list: only the queues the operand pools can fill at all)
    const bool all_q = ce == nullptr;
    auto qsz = [&](bool may, size_t item, uint64_t n) { return (all_q || may) ? item * (n + 1) : (size_t)64; };
    SQ.q[CLS_BB].ensure(qsz(P.may_bb, sizeof(BBItem), ub_match));
    SQ.q[CLS_GEN].ensure(qsz(P.may_runs, sizeof(GenItem), ub_match));
    SQ.q[CLS_FILT].ensure(qsz(P.may_filt || P.grouped, sizeof(FatItem), ub_match));
    SQ.q[CLS_WAVE].ensure(qsz(P.may_wave, sizeof(FatItem), ub_match));
    SQ.q[CLS_BA].ensure(qsz(P.may_ba, sizeof(FatItem), ub_match));
    SQ.q[CLS_RUNS].ensure(qsz(P.may_runs, sizeof(GenItem), ub_match));
    SS.q[CLS_RETRY].ensure(sizeof(GenItem) * (ub_match + 1));
    SQ.q[CLS_PROBE].ensure(qsz(P.may_filt, sizeof(FatItem), ub_match));
    SQ.q[CLS_BBA].ensure(qsz(P.may_bb, sizeof(BBItem), ub_match));
    SQ.q[CLS_USMALL].ensure(qsz(P.may_wave, sizeof(FatItem), ub_match));
    SQ.q[CLS_RUNS16].ensure(qsz(P.may_runs, sizeof(GenItem), ub_match));
    SQ.q[CLS_RUNS16W].ensure(qsz(P.may_runs, sizeof(GenItem), ub_match));
    SQ.q[CLS_COPY].ensure(qsz(P.may_copy, sizeof(CopyItem), P.ub_cand));
    if (cardmode) c->pair_acc.ensure(8 * (npairs + 1));
    if (!cardmode) {
        SQ.o_key.ensure(8 * (ub + 1)); SS.o_meta.ensure(8 * (ub + 1)); SQ.o_off.ensure(8 * (ub + 2));
        SQ.o_pair.ensure(4 * (ub + 2));
        P.CO = CandOut{SQ.o_key.as<u64>(), SQ.o_off.as<u64>(), SQ.o_pair.as<uint32_t>()};
    }
    P.Q = &SQ;
    // (Letting the planning kernels read two short index lists in place from the pinned staging area was measured:
    // the PCIe round trips inside k_count / k_emit cost 25 us more per batch than the copy command they replace.)
    if (clk) clk->lap(6);  // ([6]: sizing + scratch; [1]: the staging of the batch description alone)
    char* dp = (char*)SS.plan_in.p;
    P.d_pair0 = (u64*)(dp + o_pair0);
    P.d_lhs = (uint32_t*)(dp + o_lhs);
    P.d_rhs = (uint32_t*)(dp + o_rhs);
    P.d_upair = (uint32_t*)(dp + o_upair);
    P.d_utile = (uint32_t*)(dp + o_utile);
    if (prestaged) {
        P.d_lhs = L->d_idx.as<uint32_t>();
        P.d_rhs = P.d_lhs + npairs;
    } else if (c->stage_kernel && c->h_stage_dev[slot] && stage_bytes <= (1u << 20)) {
        const uint32_t n16 = (uint32_t)((stage_bytes + 15) >> 4);  // (staging and plan_in are 16 bytes longer than that)
        if (n16)  // (an empty pair list has nothing to stage, and a zero-block launch is an error)
            hipLaunchKernelGGL(k_stage_in, dim3((n16 + 255) / 256), dim3(256), 0, s, (const uint4*)c->h_stage_dev[slot],
                               (uint4*)dp, n16);
    } else {
        HIPCHK(hipMemcpyAsync(dp, hs, stage_bytes, hipMemcpyHostToDevice, s));
    }
    if (clk) clk->lap(1);
    PoolView VA = A->view(), VB = B->view();
    UnitView UV{P.d_upair, P.d_utile, P.d_pair0, (uint32_t)NU, (uint32_t)nvirt, implicit ? (btiles ? 2u : 1u) : 0u,
                (uint32_t)std::max<size_t>(npairs, 1), ops.packed()};
    PlanZero Z{P.words, (uint32_t)P.sc.n_words, cardmode ? c->pair_acc.as<u64>() : nullptr};
    const size_t zero_threads = std::max<size_t>(P.sc.n_words, cardmode ? npairs : 0);
    // planning waves: one unit per wave, or two / four when every unit is small (no bitmap above 128 / 64 containers)
    const uint32_t max_n = std::max(A->max_n, B->max_n);
    const uint32_t G = (!implicit || c->explicit_units == 2 || max_n > 128) ? 64u : (max_n > 64 ? 32u : 16u);
    P.plan_group = G;
    SS.match.ensure(4 * (size_t)(4 * G) * (NU + 1));  // one word per directory entry of a unit (<= 4 G of them)
    const size_t plan_waves = G == 64 ? NU : (NU + 64 / G - 1) / (64 / G) + 1;
    unsigned gp = (unsigned)std::max<size_t>(1, (std::max<size_t>(plan_waves * 64, std::min<size_t>(zero_threads, 1 << 16)) + 255) / 256);
    const u64 n_scan = (u64)N_SEC * S + n_hist;
    EmitQueues Q{SQ.q[CLS_BB].as<BBItem>(), SQ.q[CLS_GEN].as<GenItem>(), SQ.q[CLS_COPY].as<CopyItem>(),
                 SQ.q[CLS_FILT].as<FatItem>(), SQ.q[CLS_WAVE].as<FatItem>(), SQ.q[CLS_RUNS].as<GenItem>(),
                 SQ.q[CLS_PROBE].as<FatItem>(), SQ.q[CLS_BBA].as<BBItem>(), SQ.q[CLS_USMALL].as<FatItem>(),
                 SQ.q[CLS_RUNS16].as<GenItem>(), SQ.q[CLS_RUNS16W].as<GenItem>(), SQ.q[CLS_BA].as<FatItem>()};
    const unsigned ge = (unsigned)((plan_waves * 64 + 255) / 256);
    auto count = G == 16 ? k_count<16> : G == 32 ? k_count<32> : k_count<64>;
    auto emit = G == 16 ? k_emit<16> : G == 32 ? k_emit<32> : k_emit<64>;
    hipLaunchKernelGGL(count, dim3(gp), dim3(256), 0, s, VA, VB, P.d_lhs, P.d_rhs, UV, cardmode,
                       SS.cand.as<uint32_t>(), SS.match.as<uint32_t>(), Z, XG);
    hipLaunchKernelGGL(k_scan, dim3((unsigned)((n_scan + SCAN_TILE - 1) / SCAN_TILE)), dim3(256), 0, s,
                       SS.cand.as<uint32_t>(), SS.cand_start.as<u64>(), n_scan, P.scan_lb(), P.ranges(), (u64)S, (u64)N_SEC,
                       (u64)XG.nx);
    if (NU)
        hipLaunchKernelGGL(emit, dim3(ge), dim3(256), 0, s, VA, VB, P.d_lhs, P.d_rhs, UV, cardmode,
                           SS.cand_start.as<u64>(), SS.match.as<uint32_t>(), P.CO, Q, XG);
    if (P.grouped) { SS.hist_at = XG.hist; SS.hist_n = n_hist; SS.hist_gen = SS.cand.gen; }
    if (ce) {  // leave the plan with the pair list
        ce->ranges.ensure(8 * N_RANGE_WORDS + 64);
        hipLaunchKernelGGL(k_ranges_save, dim3(1), dim3(64), 0, s, (const u64*)P.ranges(), ce->ranges.as<u64>(), N_RANGE_WORDS);
        P.ce = ce;
        ce->P = P;
        ce->ops_packed = ops.packed(); ce->n_ops = ops.n; ce->cardmode = cardmode;
        ce->genA = A->bounds_gen; ce->genB = B->bounds_gen;
        ce->stamp = ++L->cache->clock;
        ce->valid = true;
    }
    if (clk) clk->lap(2);
    return P;
}

// The class kernels of one batch are independent of each other (disjoint work queues, disjoint result slots;
// pair_acc and the retry counter are only touched with atomics), except that the retry pass of k_genw consumes what
// k_bb and k_runs re-queue.  When more than one class can have work they are forked onto auxiliary streams after
// planning and joined before the tail -- four streams in all, one per hardware queue of the device (a fifth
// stream shares a queue with another and serialises behind it):
//     main : k_bb -> [ev_bb] -> k_usmall | k_probe -> k_bba -> k_copy -> [join] -> k_tail      streaming / light classes
//     aux0 : k_ivl_all (<8,63> | <16,127> | <32,255> in one launch) -> [wait ev_bb] -> k_genw(retry)     interval algebra
//     aux1 : k_filter                                                                and / andnot / cardinality
//     aux2 : k_wave                                                                  or / xor / bitset \ array
//     k_genw(general): on aux1 for or / xor, aux2 for and / cardinality (the stream the op leaves idle), else after k_filter
// Schedules that were measured and dropped (profiles/r02_schedule_notes.md): persistent grids for every class (the
// first kernel's long-lived blocks hold every LDS slot and the others start when it ends); the few-item classes
// alone before the fork (their 15-75 us then add to every batch); stream priority for aux0 (no effect: a 248-VGPR
// k_genw workgroup that does not fit is skipped for a k_filter workgroup that does, whatever the priority).
// k_genw therefore still finishes late next to a machine-filling k_filter / k_wave, but off the critical path.
// Item counts live on the device (`ranges`); grids are sized from the host's upper bounds and a class the operand
// pools cannot produce (type census) is not launched at all.  Except for k_bb the grids are NOT persistent: a wave
// takes at most ITEMS_PER_WAVE items and its block retires, so the workgroup dispatcher can interleave the blocks of
// all four queues as slots free up.  A bitset-only batch (C2) stays on the main stream.
constexpr unsigned ITEMS_PER_WAVE = 4;
unsigned bounded_grid(uint64_t ub_items, unsigned max_blocks = 1u << 16) {
    const uint64_t need = (ub_items + 4 * ITEMS_PER_WAVE - 1) / (4 * ITEMS_PER_WAVE);
    return (unsigned)std::min<uint64_t>(std::max<uint64_t>(need, 1), max_blocks);
}
// join_mask != nullptr: the auxiliary streams are not joined with events -- each ends with k_join_signal and *join_mask says
// which flags the caller's k_join_wait has to wait for (rhip_plan.h)
void run_kernels(rhip_ctx_t* c, const OpSet& ops, const PoolView& VA, const PoolView& VB, const OutView& O, const Plan& P,
                 int cardmode, uint32_t* join_mask = nullptr) {
    if (join_mask) *join_mask = 0;
    hipStream_t s = c->stream;
    const int op = ops.kop();  // the batch's op, or OP_ITEM: a multi-op batch, every work item carries its own
    const bool multi = ops.n > 1;
    const bool any_and_like = ops.has(OP_AND) || ops.has(OP_ANDNOT), any_not_or = ops.n > 1 || ops.op[0] != OP_OR;
    const bool any_union = ops.has(OP_OR) || ops.has(OP_XOR);
    const u64* ranges = P.ranges();
    uint32_t* retry_count = P.retry_count();
    const uint64_t nm = P.ub_match;
    const bool has_bb = P.may_bb && nm, has_runs = P.may_runs && nm, has_filt = P.may_filt && nm;
    const bool has_wave = P.may_wave && nm && !cardmode, has_copy = P.may_copy && P.ub_cand && !cardmode;
    const bool has_bba = has_bb && !cardmode && any_and_like;
    // (a grouped batch types its bitset (op) array results itself: k_union_g re-queues nothing)
    const bool has_retry = !cardmode && ((has_bb && any_not_or) || has_runs || (P.may_ba && nm && any_not_or && !P.grouped));
    const bool has_ba = P.may_ba && nm && !cardmode;
    // (few items AND a moderate slot bound: the class kernels are short whatever the bytes say -- one merged launch beats
    // the fork / join: census-income andnot, 73 k container pairs / 223 MB: 0.30 -> 0.27 ms)
    const bool light = nm <= (128u << 10) && P.work_bound < c->fork_light_bytes;
    // (round 6) interval-dominated batch: forked for its item count, the interval kernel on the main stream
    // (and / andnot / cardinality only: C5 `and` 0.268 -> 0.246 ms, `andnot` 0.424 -> 0.412; under or / xor the pass-through copies
    // and the array unions are as long as the interval kernel and the fork costs more than it hides: C5 `or` 0.515 -> 0.533)
    const bool ivl_main = c->overlap && has_runs && P.runs_dominant && c->fork_runs_min_items && nm >= c->fork_runs_min_items &&
                          !any_union && (has_filt || has_ba || has_bb);
    const bool fork = ivl_main || (c->overlap && (has_runs || has_filt || has_wave || has_ba) && P.work_bound >= c->fork_min_bytes && !light);
    if (c->debug_plan)
        fprintf(stderr, "[rhip plan] ops %d nm %llu work_bound %.1f MB fork %d merge_eligible %d grouped %d copy_per_wave %u\n", ops.n, (unsigned long long)nm,
                P.work_bound / 1048576.0, (int)fork, (int)(nm <= c->merge_max_items), (int)P.grouped, P.copy_per_wave);
    // A small batch (below the fork threshold) runs its class kernels as ONE launch, block ranges per class
    // (rhip_classes.h): their latency chains side by side instead of one after the other.  k_genw follows on its own.
    // (Only batches with few items: the combined kernel has the registers and LDS of its largest body -- 3 waves per SIMD
    // -- which cost the 847 000 interval pairs of a C5 `and` batch 0.35 -> 0.41 ms, while census1881 `and` went 0.152 ->
    // 0.144 ms.)
    if (c->merge_classes && c->overlap && !fork && !P.grouped && (P.work_bound < c->fork_min_bytes || light) && nm <= c->merge_max_items) {
        ClassLaunch L{};
        L.arenaA = VA.arena; L.arenaB = VB.arena; L.O = O; L.ranges = ranges;
        L.q_bb = P.q(c, CLS_BB).as<BBItem>(); L.q_bba = P.q(c, CLS_BBA).as<BBItem>();
        L.q_filt = P.q(c, CLS_FILT).as<FatItem>(); L.q_probe = P.q(c, CLS_PROBE).as<FatItem>();
        L.q_usmall = P.q(c, CLS_USMALL).as<FatItem>(); L.q_wave = P.q(c, CLS_WAVE).as<FatItem>(); L.q_ba = P.q(c, CLS_BA).as<FatItem>();
        L.q_copy = P.q(c, CLS_COPY).as<CopyItem>();
        L.q_r16 = P.q(c, CLS_RUNS16).as<GenItem>(); L.q_r16w = P.q(c, CLS_RUNS16W).as<GenItem>(); L.q_r64 = P.q(c, CLS_RUNS).as<GenItem>();
        L.retry_q = P.q(c, CLS_RETRY).as<GenItem>(); L.retry_count = retry_count; L.pair_acc = c->pair_acc.as<u64>();
        L.kop = op; L.cardmode = cardmode; L.copy_per_wave = P.copy_per_wave;
        const unsigned cap = 512;  // blocks per class: its waves loop over the queue (the real queues are short)
        auto seg = [&](bool present, uint64_t ub) { return present ? ::bounded_grid(ub, cap) : 0u; };
        L.nb[CSEG_IVL16] = seg(has_runs, nm); L.nb[CSEG_IVL16W] = seg(has_runs, nm); L.nb[CSEG_IVL64] = seg(has_runs, nm);
        L.nb[CSEG_FILT] = seg(has_filt, nm); L.nb[CSEG_PROBE] = seg(has_filt, nm);
        L.nb[CSEG_USMALL] = seg(has_wave && any_union, nm); L.nb[CSEG_WAVE] = seg(has_wave, nm);
        L.nb[CSEG_BA] = seg(has_ba, nm); L.nb[CSEG_BBA] = seg(has_bba, nm); L.nb[CSEG_BB] = seg(has_bb, nm);
        L.nb[CSEG_COPY] = seg(has_copy, P.ub_cand);
        unsigned total = 0;
        for (unsigned v : L.nb) total += v;
        if (total) hipLaunchKernelGGL(k_classes, dim3(total), dim3(256), 0, s, L);
        if (has_runs || has_retry) {  // the general image class and / or the re-queued results, after their producers
            const bool gen = has_runs, ret = has_retry;
            if (gen && ret)
                hipLaunchKernelGGL(k_genw<true>, dim3(4 * ::bounded_grid(nm, 512)), dim3(64), 0, s, VA.arena, VB.arena, O,
                                   P.q(c, CLS_GEN).as<GenItem>(), ranges + 2 * SEC_GEN, (const uint32_t*)nullptr, op, cardmode,
                                   c->pair_acc.as<u64>(), (const GenItem*)P.q(c, CLS_RETRY).as<GenItem>(), (const uint32_t*)retry_count);
            else if (gen)
                hipLaunchKernelGGL(k_genw<true>, dim3(4 * ::bounded_grid(nm, 512)), dim3(64), 0, s, VA.arena, VB.arena, O,
                                   P.q(c, CLS_GEN).as<GenItem>(), ranges + 2 * SEC_GEN, (const uint32_t*)nullptr, op, cardmode,
                                   c->pair_acc.as<u64>(), (const GenItem*)nullptr, (const uint32_t*)nullptr);
            else
                hipLaunchKernelGGL(k_genw<false>, dim3(4 * ::bounded_grid(nm, 512)), dim3(64), 0, s, VA.arena, VB.arena, O,
                                   P.q(c, CLS_RETRY).as<GenItem>(), (const u64*)nullptr, retry_count, op, 0,
                                   c->pair_acc.as<u64>(), (const GenItem*)nullptr, (const uint32_t*)nullptr);
        }
        return;
    }
    // Stream roles of a forked batch (round 4).  The batch ends when its LONGEST kernel does -- the image kernel: k_filter(_g)
    // under and / andnot / cardinality, k_union_g / k_wave under or / xor -- so that kernel stays on the MAIN stream:
    // it starts right behind k_emit (a kernel on an auxiliary stream starts ~16 us later: the fork is a cross-queue
    // dependency) and the tail starts right behind it (no signal kernel, no barrier packet on the critical path).  The
    // light chain k_bb -> k_usmall / k_probe -> k_bba -> k_copy, which used to own the main stream, takes the auxiliary
    // stream the image kernel vacated.
    bool used[rhip_ctx_s::N_AUX] = {false, false, false};
    const int crit = !fork ? -1 : ivl_main ? 0 : has_filt ? 1 : (has_wave || has_ba) ? 2 : -1;
    auto on_aux = [&](int a) -> hipStream_t {
        if (!used[a]) {
            HIPCHK(hipStreamWaitEvent(c->aux[a], c->ev_fork, 0));
            used[a] = true;
        }
        return c->aux[a];
    };
    auto on = [&](int a) -> hipStream_t {
        if (!fork || a == crit) return s;
        return on_aux(a);
    };
    if (fork) HIPCHK(hipEventRecord(c->ev_fork, s));
    const hipStream_t lst = (fork && crit >= 0) ? on_aux(crit) : s;  // the stream of the light chain
    if (has_runs) {
        // interval algebra, three size classes in one launch: short lists four pairs per wave (most of a sparse
        // run-compressed batch), long lists one pair per wave.  Few items as a rule, so few blocks (an empty block of an
        // LDS-heavy kernel still queues for a slot); many items simply loop.
        // (forked: at most 2 048 blocks in all -- the queues are short as a rule and every block of this LDS-heavy kernel,
        // empty or not, has to find a slot beside the image kernels: with 8 192 the kernel's last blocks ran 400 us after
        // its first on weather `or`, and the retry pass waits for them; 1 024 + 512 + 512 blocks still fill the machine
        // when the interval pairs ARE the batch -- C5)
        const bool capg = fork && !ivl_main;  // (where the interval pairs ARE the batch the kernel keeps its full grid)
        const unsigned g1 = bounded_grid(nm, capg ? 1024 : 4096), g2 = bounded_grid(nm, capg ? 512 : 2048), g3 = bounded_grid(nm, capg ? 512 : 2048);
        IvlQueues IQ{{P.q(c, CLS_RUNS16).as<GenItem>(), P.q(c, CLS_RUNS16W).as<GenItem>(), P.q(c, CLS_RUNS).as<GenItem>()},
                     {ranges + 2 * SEC_RUNS16, ranges + 2 * SEC_RUNS16W, ranges + 2 * SEC_RUNS}};
        hipLaunchKernelGGL(k_ivl_all, dim3(g1 + g2 + g3), dim3(256), 0, on(0), VA.arena, VB.arena, O, IQ, g1, g2, op,
                           cardmode, c->pair_acc.as<u64>(), P.q(c, CLS_RETRY).as<GenItem>(), retry_count);
    }
    // the general image class: forked, beside the interval chain on the auxiliary stream this op leaves idle (a multi-op
    // batch leaves none idle: ahead of k_filter -- its one-wave blocks slip in wherever two slots are free); on one
    // stream it shares a launch with the retry pass further down
    const bool genw_merged = !fork && has_runs && has_retry;
    auto launch_genw_general = [&](hipStream_t sg) {
        hipLaunchKernelGGL(k_genw<true>, dim3(4 * bounded_grid(nm, 512)), dim3(64), 0, sg, VA.arena, VB.arena, O,
                           P.q(c, CLS_GEN).as<GenItem>(), ranges + 2 * SEC_GEN, (const uint32_t*)nullptr, op,
                           cardmode, c->pair_acc.as<u64>(), (const GenItem*)nullptr, (const uint32_t*)nullptr);
    };
    if (multi && has_runs && !genw_merged) launch_genw_general(on(1));
    // grouped batch: the filter and the union items sit in ONE queue (the filter class's buffer), sorted by X container;
    // a wave walks >= 8 items in a row, so the grids are an eighth of the wave-per-item ones (and at most two rounds of
    // the machine's 5 120 resident image waves)
    const FatItem* xq = P.q(c, CLS_FILT).as<FatItem>();
    const unsigned gx = (unsigned)std::min<uint64_t>(std::max<uint64_t>((nm + 4 * c->group_chunk - 1) / (4 * c->group_chunk), 1), 2560u * 8u / c->group_chunk);
    if (has_filt && P.grouped)
        hipLaunchKernelGGL((op == OP_AND || cardmode) ? k_filter_g<false> : k_filter_g<true>, dim3(gx), dim3(256), 0, on(1), VA.arena,
                           VB.arena, O, xq, P.xranges(), op, cardmode, c->pair_acc.as<u64>(), c->group_chunk);
    else if (has_filt)
        hipLaunchKernelGGL((op == OP_AND || cardmode) ? k_filter<false> : k_filter<true>, dim3(bounded_grid(nm)), dim3(256), 0, on(1), VA.arena, VB.arena, O,
                           P.q(c, CLS_FILT).as<FatItem>(), ranges + 2 * SEC_FILT, op, cardmode, c->pair_acc.as<u64>());
    if ((has_wave || has_ba) && P.grouped) {
        hipStream_t su = on(2);
        if (multi) hipLaunchKernelGGL(k_union_g<OP_ITEM>, dim3(gx), dim3(256), 0, su, VA.arena, VB.arena, O, xq, P.xranges(), c->group_chunk);
        else if (op == OP_OR) hipLaunchKernelGGL(k_union_g<OP_OR>, dim3(gx), dim3(256), 0, su, VA.arena, VB.arena, O, xq, P.xranges(), c->group_chunk);
        else if (op == OP_XOR) hipLaunchKernelGGL(k_union_g<OP_XOR>, dim3(gx), dim3(256), 0, su, VA.arena, VB.arena, O, xq, P.xranges(), c->group_chunk);
        else hipLaunchKernelGGL(k_union_g<OP_ANDNOT>, dim3(gx), dim3(256), 0, su, VA.arena, VB.arena, O, xq, P.xranges(), c->group_chunk);
    } else if (has_wave)
        hipLaunchKernelGGL(k_wave, dim3(bounded_grid(nm)), dim3(256), 0, on(2), VA.arena, VB.arena, O,
                           P.q(c, CLS_WAVE).as<FatItem>(), ranges + 2 * SEC_WAVE, op);
    // The main stream's own kernels are issued right after the big image kernels: the host's launches are what the device
    // waits for in a forked batch (~25 API calls), and k_probe / k_usmall are on its critical path, the few general items are not.
    const bool need_bb_event = fork && has_retry && has_bb;  // the retry pass (another stream) consumes what k_bb re-queues
    if (has_bb) {
        unsigned grid = persistent_grid(nm, 4, 256 * 32);
        if (c->timing) { HIPCHK(hipEventRecord(c->evs[P.slot][2], lst)); c->bb_timed[P.slot] = true; }
        switch (op) {
            case OP_ITEM: launch_bb<OP_ITEM>(c, lst, grid, VA, VB, O, P, cardmode); break;
            case OP_AND: launch_bb<OP_AND>(c, lst, grid, VA, VB, O, P, cardmode); break;
            case OP_OR: launch_bb<OP_OR>(c, lst, grid, VA, VB, O, P, cardmode); break;
            case OP_XOR: launch_bb<OP_XOR>(c, lst, grid, VA, VB, O, P, cardmode); break;
            default: launch_bb<OP_ANDNOT>(c, lst, grid, VA, VB, O, P, cardmode); break;
        }
        if (c->timing) HIPCHK(hipEventRecord(c->evs[P.slot][3], lst));
        if (need_bb_event) HIPCHK(hipEventRecord(c->ev_runs, lst));  // "k_bb done"
    }
    if (has_wave && any_union)  // or / xor of a short array with a long one, by rank: 170 us alone on weather -- a stream of
                                // its own when the filter's is free (or / xor), else the light chain
    {
        auto kus = k_usmall<false, -1>;  // (the op as a template argument where the batch has one: rhip_array.h)
        if (c->usmall_gp8) kus = k_usmall<true, -1>;
#ifndef RHIP_USMALL_NO_OPC  /* (A/B builds) */
        else if (op == OP_OR) kus = k_usmall<false, OP_OR>;
        else if (op == OP_XOR) kus = k_usmall<false, OP_XOR>;
#endif
        hipLaunchKernelGGL(kus, dim3(bounded_grid(nm)), dim3(256), 0, (fork && crit == 2) ? on_aux(1) : lst, VA.arena, VB.arena, O,
                           P.q(c, CLS_USMALL).as<FatItem>(), ranges + 2 * SEC_USMALL, op);
    }
    if (has_filt)  // short streamed arrays: no LDS, 8 waves per SIMD -- co-resides with the LDS-bound kernels
        hipLaunchKernelGGL(k_probe, dim3(bounded_grid(nm)), dim3(256), 0, lst, VA.arena, VB.arena, O,
                           P.q(c, CLS_PROBE).as<FatItem>(), ranges + 2 * SEC_PROBE, op, cardmode, c->pair_acc.as<u64>());
    if (has_bba) {  // bitset pairs expected to give arrays
        if (multi)
            hipLaunchKernelGGL(k_bba<OP_ITEM>, dim3(bounded_grid(nm)), dim3(256), 0, lst, VA.arena, VB.arena, O,
                               P.q(c, CLS_BBA).as<BBItem>(), ranges + 2 * SEC_BBA);
        else if (op == OP_AND)
            hipLaunchKernelGGL(k_bba<OP_AND>, dim3(bounded_grid(nm)), dim3(256), 0, lst, VA.arena, VB.arena, O,
                               P.q(c, CLS_BBA).as<BBItem>(), ranges + 2 * SEC_BBA);
        else
            hipLaunchKernelGGL(k_bba<OP_ANDNOT>, dim3(bounded_grid(nm)), dim3(256), 0, lst, VA.arena, VB.arena, O,
                               P.q(c, CLS_BBA).as<BBItem>(), ranges + 2 * SEC_BBA);
    }
    if (has_copy)
        hipLaunchKernelGGL(k_copy, dim3(bounded_grid(P.copy_per_wave == 16 ? (P.ub_cand + 3) / 4 : P.ub_cand)), dim3(256), 0, lst, VA.arena, VB.arena, O,
                           P.q(c, CLS_COPY).as<CopyItem>(), ranges + 2 * SEC_COPY, P.copy_per_wave);
    // (or / xor: at the end of the light chain, whose stream is idle by then; and / cardinality: the union stream, idle)
    if (!multi && has_runs && !genw_merged) launch_genw_general(!has_filt ? lst : on(!has_wave ? 2 : 1));
    if (has_ba && !P.grouped) {  // bitset (op) array, behind the few general items of its stream: andnot has no k_wave items, so that
                   // stream is free; or / xor: the filter's stream
        hipStream_t sb = multi ? on(0) : op == OP_ANDNOT ? on(2) : lst;  // (multi-op: behind the interval kernel; or / xor: the light chain -- k_usmall has the filter's stream)
        const FatItem* qb = P.q(c, CLS_BA).as<FatItem>();
        const unsigned gb = bounded_grid(nm);
        GenItem* rq = P.q(c, CLS_RETRY).as<GenItem>();
        if (multi) hipLaunchKernelGGL(k_ba<OP_ITEM>, dim3(gb), dim3(256), 0, sb, VA.arena, VB.arena, O, qb, ranges + 2 * SEC_BA, rq, retry_count);
        else if (op == OP_OR) hipLaunchKernelGGL(k_ba<OP_OR>, dim3(gb), dim3(256), 0, sb, VA.arena, VB.arena, O, qb, ranges + 2 * SEC_BA, rq, retry_count);
        else if (op == OP_XOR) hipLaunchKernelGGL(k_ba<OP_XOR>, dim3(gb), dim3(256), 0, sb, VA.arena, VB.arena, O, qb, ranges + 2 * SEC_BA, rq, retry_count);
        else hipLaunchKernelGGL(k_ba<OP_ANDNOT>, dim3(gb), dim3(256), 0, sb, VA.arena, VB.arena, O, qb, ranges + 2 * SEC_BA, rq, retry_count);
        if (fork && op != OP_OR) HIPCHK(hipEventRecord(c->ev_ba, sb));  // "k_ba done" for the retry pass
    }
    if (has_retry) {
        // results that need the LDS image path after all: bitset x bitset results that must become
        // arrays (card <= 4096), interval results that must become bitsets
        // (behind the interval kernel, which re-queues too -- wherever that one runs; never the image kernel's stream)
        hipStream_t sr = !fork || crit == 0 ? s : on_aux(0);
        if (need_bb_event) HIPCHK(hipStreamWaitEvent(sr, c->ev_runs, 0));
        if (fork && has_ba && !P.grouped && op != OP_OR) HIPCHK(hipStreamWaitEvent(sr, c->ev_ba, 0));  // k_ba re-queues its rare array results
        if (genw_merged)  // one stream: the general class and the re-queued results in one launch, after their producers
            hipLaunchKernelGGL(k_genw<true>, dim3(4 * bounded_grid(nm, 512)), dim3(64), 0, sr, VA.arena, VB.arena, O,
                               P.q(c, CLS_GEN).as<GenItem>(), ranges + 2 * SEC_GEN, (const uint32_t*)nullptr, op, 0,
                               c->pair_acc.as<u64>(), (const GenItem*)P.q(c, CLS_RETRY).as<GenItem>(), (const uint32_t*)retry_count);
        else
            hipLaunchKernelGGL(k_genw<false>, dim3(4 * bounded_grid(nm, 512)), dim3(64), 0, sr, VA.arena, VB.arena, O,
                               P.q(c, CLS_RETRY).as<GenItem>(), (const u64*)nullptr, retry_count, op, 0,
                               c->pair_acc.as<u64>(), (const GenItem*)nullptr, (const uint32_t*)nullptr);
    }
    if (fork)
        for (int a = 0; a < rhip_ctx_s::N_AUX; ++a)
            if (used[a]) {
                if (join_mask) {
                    hipLaunchKernelGGL(k_join_signal, dim3(1), dim3(64), 0, c->aux[a], P.join_flags() + a);
                    *join_mask |= 1u << a;
                } else {
                    HIPCHK(hipEventRecord(c->ev_join[a], c->aux[a]));
                    HIPCHK(hipStreamWaitEvent(s, c->ev_join[a], 0));
                }
            }
}

// Wait until the last kernel of a call has published `seq` in its completion word (pinned memory): the host polls the
// word (and the stream's state now and then, so that a device fault still surfaces) instead of blocking on the
// stream, whose completion signal costs an interrupt and a wake-up.  RHIP_SPIN_WAIT=0: block on the stream.
void wait_word(rhip_ctx_t* c, volatile uint64_t* flag, uint64_t seq) {
    hipStream_t s = c->stream;
    if (!c->spin_wait) {
        HIPCHK(hipStreamSynchronize(s));
        return;
    }
    for (uint32_t spins = 0;; ++spins) {
        // (seq | TAIL_EARLY: the tail behind a flag join that gave up -- rhip_pairwise_end takes it from there)
        if ((__atomic_load_n((const uint64_t*)flag, __ATOMIC_ACQUIRE) & ~TAIL_EARLY) == seq) break;
        if ((spins & 0x3FFFu) == 0x3FFFu) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) {  // stream drained: the word is there now, or the kernel never ran
                if ((__atomic_load_n((const uint64_t*)flag, __ATOMIC_ACQUIRE) & ~TAIL_EARLY) == seq) break;
                set_err("the last kernel of the call finished without publishing its completion word");
                throw (int)RHIP_ERR_DEVICE;
            }
            if (q != hipErrorNotReady) { set_err("device error while waiting: %s", hipGetErrorString(q)); throw (int)RHIP_ERR_DEVICE; }
        }
        __builtin_ia32_pause();
    }
}

// many-way / flip paths: their statistics live at a fixed offset of ctx->misc (cleared by the caller)
constexpr size_t MISC_STATS_OFF = 192;
void finish_stats(rhip_ctx_t* c, const Stats* dstats, Stats* out, bool had_bb, uint64_t wait_seq = 0, int slot = 0);
void finish_stats_misc(rhip_ctx_t* c) { finish_stats(c, (const Stats*)((char*)c->misc.p + MISC_STATS_OFF), nullptr, false); }

// the ONE host synchronisation of a call: statistics (and whatever the caller queued before) come back
// wait_seq != 0: the tail kernel publishes that sequence number in pinned memory when everything is written; the host
// polls it (and the stream's state now and then, so that a device fault still surfaces) instead of blocking.
void finish_stats(rhip_ctx_t* c, const Stats* dstats, Stats* out, bool had_bb, uint64_t wait_seq, int slot) {
    hipStream_t s = c->stream;
    // dstats == nullptr: the last kernel of the call already wrote the totals into the pinned area
    if (dstats) HIPCHK(hipMemcpyAsync(c->h_pinned, dstats, sizeof(Stats), hipMemcpyDeviceToHost, s));
    hipEvent_t* ev = c->evs[wait_seq ? slot : rhip_ctx_s::SYNC_SLOT];
    if (c->timing && !wait_seq) HIPCHK(hipEventRecord(ev[1], s));  // (a batch recorded its end when it was enqueued)
    if (wait_seq && !dstats) {
        wait_word(c, c->done_flag(slot), wait_seq);
    } else {
        HIPCHK(hipStreamSynchronize(s));
    }
    Stats st{};
    memcpy(&st, wait_seq ? c->slot_stats(slot) : c->h_pinned, sizeof(Stats));  // a batch's totals are in its slot
    if (out) *out = st;
    c->stats.matched_pairs = st.matched_pairs;
    c->stats.passthrough = st.passthrough;
    c->stats.bytes_in = st.bytes_in;
    c->stats.bytes_out = st.bytes_out;
    c->stats.n_bitset_pairs = st.n_bb;
    c->stats.result_containers = st.result_containers;
    c->stats.ms_bitset_kernel = 0.f;
    c->stats.ms_total = 0.f;
    if (c->timing) {
        if (wait_seq) HIPCHK(hipEventSynchronize(ev[1]));  // the completion word can be seen before the event signals
        (void)hipEventElapsedTime(&c->stats.ms_total, ev[0], ev[1]);
        // (k_bb inside a merged k_classes launch has no events of its own: elapsed time of events never recorded is an
        // error that would surface at the next call's hipGetLastError)
        const int tslot = wait_seq ? slot : rhip_ctx_s::SYNC_SLOT;
        if (had_bb && st.n_bb && c->bb_timed[tslot]) (void)hipEventElapsedTime(&c->stats.ms_bitset_kernel, ev[2], ev[3]);
        c->bb_timed[tslot] = false;
    }
}
}  // namespace

// what it takes to run a batch's k_tail once more (a flag join that gave up: rhip_pairwise_end)
struct TailRedo {
    bool armed = false;     // the batch joined its auxiliary streams by flags
    CandOut CO{nullptr, nullptr, nullptr};
    const u64* meta = nullptr;
    DirOut D{};
    uint32_t n_virt = 0;
    LbState lb{nullptr, nullptr};
    u64* part = nullptr;
    unsigned blocks = 0;
    size_t status_words = 0;  // look-back states + per-tile parts (contiguous from lb.status)
};
struct rhip_batch_s {
    rhip_ctx_t* c;
    rhip_pool_t* R;
    rhip_pool_t *A, *B;
    uint64_t seq;
    int slot;
    bool may_bb;
    const u64* ranges;  // the batch's section ranges (device)
    bool grouped;       // its filter / union items are in the X-grouped queue
    rhip_pairlist_t* L; // the prepared pair list the batch reads (its device copy), or NULL
    TailRedo redo;
    ::PlanCacheEntry* ce = nullptr;  // the cached plan the batch's kernels read (pinned by in_use until the batch ends)
    rhip_ctx_s::SlotScratch* Q = nullptr;  // where its class queues are (class statistics)
};

// Placement of a large result arena by its PHYSICAL CHUNKS (round 6, second half): stage 0 of place_arena.
// scripts/vmm_place5.hip probed every one of sixteen 1 GiB chunks (hipMemCreate) at every GiB position of a C2 arena against
// the pool region it meets there in lockstep: the rate is a property of the CHUNK -- 5.95-6.11 TB/s for chunks 0-5, 14, 15 at
// all eight positions, 5.62-5.84 for chunks 6-13 at all eight -- not of the position or of the pair; chunks created one
// after the other form runs of good and bad ones (the 5-6 GiB wide distance bands of round 4).  And the arena COMPOSED of the
// eight best chunks streamed at 6.33 / 6.34 TB/s where the eight worst gave 5.42 and a plain hipMalloc arena 5.41 (a second
// process: 6.26 / 5.39 / 5.68).  So: three times as many chunks as the arena needs are created, each is probed once ALONE
// (mapped at an address of its own, 1 GiB of k_place_probe against the operand pool: ~0.7 ms), the best are mapped side by
// side as the arena and the usual sampled probe checks the whole; a composition below the bar is tried at the next address
// (the boxes on which the virtual address decides, see place_arena_va).  The chunks that were not taken stay with the
// context, unmapped, WITH their rates: the next result arena for the same operand is composed out of them first
// (trim_spares_when_steady / rhip_ctx_trim release them).  Transient footprint: 3 x the arena.  The driver's limits listed
// at place_arena_va hold here too: every mapping is at an address that was never mapped before.
// Returns 0: nothing done (the calls are missing or failed, no memory); 1: arena placed at or above arena_good_gbps;
// 2: placed below it (arena.placed_gbps says where).
static int place_arena_chunks(rhip_ctx_t* c, DBuf& arena, size_t need, const rhip_pool_t* A, bool fresh_only = false) {
    if (!c->arena_vmm) return 0;
    if (g_fail_allocs.load(std::memory_order_relaxed) > 0) return 0;  // (tests of the allocation-failure paths: through DBuf::ensure)
    const size_t MB2 = 2ull << 20, G1 = 1ull << 30;
    const size_t step = std::max<size_t>(MB2, c->arena_va_step / MB2 * MB2);
    const size_t chunk = std::min<size_t>(G1, step);
    const size_t len = (need + arena.skew + chunk - 1) / chunk * chunk;  // whole chunks: every chunk can take every place
    const uint32_t n_need = (uint32_t)(len / chunk);
    if (n_need > 4096) return 0;
    size_t free_b = 0, tot_b = 0;
    if (hipMemGetInfo(&free_b, &tot_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    struct Ch { hipMemGenericAllocationHandle_t h; float gbps; bool probed; };
    std::vector<Ch> ch;
    // the context's spare chunks that were probed against THIS operand arena come first, rates and all
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        for (size_t k = 0; k < c->chunk_spares.size();) {
            const auto& x = c->chunk_spares[k];
            const bool known_here = x.gbps > 0.f && x.placed_for == A->arena.base && x.placed_for_gen == A->arena.gen;
            if (x.size == chunk && !(fresh_only && known_here)) {  // (one probed beside another operand, or never probed, is probed below)
                const bool known = known_here;
                ch.push_back(Ch{x.h, known ? x.gbps : 0.f, known});
                c->chunk_spares.erase(c->chunk_spares.begin() + (long)k);
            } else {
                ++k;
            }
        }
    }
    auto give_back = [&](size_t from) {  // chunks [from, ..) -> the context's spares (or the driver)
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        for (size_t k = from; k < ch.size(); ++k) {
            if (c->arena_keep_spares) c->chunk_spares.push_back(rhip_ctx_s::ChunkSpare{ch[k].h, chunk, ch[k].probed ? ch[k].gbps : 0.f, A->arena.base, A->arena.gen});
            else (void)hipMemRelease(ch[k].h);
        }
        ch.resize(from);
    };
    const size_t want = 3u * (size_t)n_need;
    const size_t n_spare_in = ch.size();
    while (ch.size() < want) {
        if ((ch.size() + 1 - n_spare_in) * chunk + (256ull << 20) > free_b / 2 && ch.size() >= (size_t)n_need) break;  // (never more than half of what was free)
        hipMemGenericAllocationHandle_t h{};
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        ch.push_back(Ch{h, 0.f, false});
    }
    if (ch.size() < (size_t)n_need) { give_back(0); return 0; }
    // address space: the single chunks are probed in the CONTEXT's probe range, a never-used place each (see probe_va); the
    // composed arena gets a range of its own with up to N_COMP places -- only the arena's own chunks are ever mapped there
    constexpr size_t N_COMP = 6;
    const size_t pitch = len + 2 * step;
    const size_t slot = std::max(chunk, step);
    size_t n_unprobed = 0;
    for (const Ch& x : ch) n_unprobed += x.probed ? 0 : 1;
    if (!c->probe_va) {
        for (size_t want_va : {1ull << 40, 256ull << 30, 32ull << 30}) {
            if (hipMemAddressReserve(&c->probe_va, want_va, 0, nullptr, 0) == hipSuccess) { c->probe_va_len = want_va; break; }
            (void)hipGetLastError();
            c->probe_va = nullptr;
        }
        c->probe_va_used = c->probe_va ? (size_t)((((uintptr_t)c->probe_va + G1 - 1) / G1 * G1 + MB2) - (uintptr_t)c->probe_va) : 0;  // 2 MiB past a GiB boundary
    }
    if (!c->probe_va || c->probe_va_used + n_unprobed * slot > c->probe_va_len) { give_back(0); return 0; }  // (the range is used up: the other stages)
    const size_t va_len = N_COMP * pitch + 2 * G1 + 2 * MB2;
    void* R = nullptr;
    if (hipMemAddressReserve(&R, va_len, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); give_back(0); return 0; }
    uint8_t* base0 = (uint8_t*)c->probe_va + c->probe_va_used;
    const u64 a_items = A->arena.cap / 8192ull;
    hipStream_t s = c->stream;
    hipEvent_t e0 = c->ev[0], e1 = c->ev[1];
    auto timed = [&](uint8_t* out, u64 n_slots, u64 stride, float& gbps, int runs = 3) {
        const u64 n_items = (n_slots + stride - 1) / stride;
        float ms_best = 1e30f;
        for (int r = 0; r < runs; ++r) {  // (the first pass warms the translations)
            if (hipEventRecord(e0, s) != hipSuccess) return false;
            hipLaunchKernelGGL(k_place_probe, dim3(8192), dim3(256), 0, s, A->arena.as<uint8_t>(), a_items, out, n_slots, stride);
            if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return false;
            float ms = 0;
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return false;
            if (r && ms < ms_best) ms_best = ms;
        }
        gbps = (float)((double)n_items * 24576.0 / (double)std::max(ms_best, 1e-6f) / 1e6);
        return true;
    };
    bool ok = true;
    uint8_t* next_place = base0;
    for (size_t k = 0; k < ch.size() && ok; ++k) {
        if (ch[k].probed) continue;
        uint8_t* at = next_place;
        next_place += slot;
        c->probe_va_used += slot;  // (used, whatever happens to the probe)
        ok = hipMemMap(at, chunk, 0, ch[k].h, 0) == hipSuccess;
        if (!ok) break;
        // (every other slot of the chunk, one warm and one timed pass: the two levels are ~5 % apart, and these probes are
        // most of what a placement costs)
        ok = hipMemSetAccess(at, chunk, &acc, 1) == hipSuccess && timed(at, chunk / 8192ull, chunk >= (64ull << 20) ? (u64)c->arena_chunk_stride : 1, ch[k].gbps, c->arena_chunk_runs);
        ok = hipMemUnmap(at, chunk) == hipSuccess && ok;
        ch[k].probed = ok;
        if (ok) c->last_placement.push_back(ch[k].gbps);
        // enough good ones?  (the rates come in two levels, ~5 % apart: n_need chunks within 2.5 % of the best seen, after
        // at least n_need + 4 probes, end the probing -- the chunks not yet probed go back unprobed)
        if (ok && k + 1 >= (size_t)n_need + 4) {
            float top = 0.f;
            for (size_t i = 0; i <= k; ++i) if (ch[i].probed) top = std::max(top, ch[i].gbps);
            size_t n_top = 0;
            for (size_t i = 0; i <= k; ++i) n_top += ch[i].probed && ch[i].gbps >= 0.975f * top ? 1 : 0;
            if (n_top >= (size_t)n_need) break;
        }
    }
    if (!ok) {
        (void)hipGetLastError();
        give_back(0);
        (void)hipMemAddressFree(R, va_len);
        return 0;
    }
    std::stable_sort(ch.begin(), ch.end(), [](const Ch& a, const Ch& b) { return a.probed != b.probed ? a.probed : a.gbps > b.gbps; });
    // the composed arena: the n_need best chunks side by side, at one place after the other until the whole streams at the bar
    uint8_t* comp0 = (uint8_t*)(((uintptr_t)R + G1 - 1) / G1 * G1) + MB2;  // = 2 MiB past a GiB boundary
    if (step % G1) comp0 = (uint8_t*)(((uintptr_t)R + MB2 - 1) / MB2 * MB2);  // (sub-GiB steps: tests on small arenas)
    const u64 n_slots = need / 8192ull;
    const u64 stride = std::max<u64>(1, n_slots / ((1ull << 30) / 8192ull));
    uint8_t* at = nullptr;
    float here = 0.f, best = 0.f, worst = 1e30f;
    for (size_t pos = 0; pos < N_COMP && ok; ++pos) {
        if (at) for (uint32_t k = 0; k < n_need; ++k) ok = hipMemUnmap(at + (size_t)k * chunk, chunk) == hipSuccess && ok;
        at = comp0 + pos * pitch;
        for (uint32_t k = 0; k < n_need && ok; ++k) ok = hipMemMap(at + (size_t)k * chunk, chunk, 0, ch[k].h, 0) == hipSuccess;
        ok = ok && hipMemSetAccess(at, len, &acc, 1) == hipSuccess && timed(at + arena.skew, n_slots, stride, here);
        if (!ok) break;
        c->last_placement.push_back(here);
        if (here >= 0.99f * (float)c->arena_good_gbps) break;  // (a composition of good chunks streams at 6.2-6.35 TB/s by this probe)
        best = std::max(best, here);
        worst = std::min(worst, here);
        if (pos >= 2 && best - worst < 0.012f * best) break;  // the address does not decide here
    }
    if (ok && getenv("RHIP_ARENA_DEBUG")) {  // diagnostics: the chosen chunks' single rates, and what each streams at IN PLACE
        fprintf(stderr, "rhip place_arena_chunks: composition %.0f GB/s at %p; chosen singles:", here, (void*)at);
        for (uint32_t k = 0; k < n_need; ++k) fprintf(stderr, " %.0f", ch[k].gbps);
        fprintf(stderr, "; in place (3 passes, every slot, against the pool region each meets):");
        for (uint32_t k = 0; k < n_need; ++k) {
            const u64 so = (u64)k * (chunk / 8192ull);
            float ms_best = 1e30f;
            for (int r = 0; r < 3 && so + chunk / 8192ull <= a_items; ++r) {
                (void)hipEventRecord(e0, s);
                hipLaunchKernelGGL(k_place_probe, dim3(8192), dim3(256), 0, s, A->arena.as<uint8_t>() + so * 8192ull, a_items - so, at + (size_t)k * chunk, chunk / 8192ull, 1ull);
                (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                if (r && ms < ms_best) ms_best = ms;
            }
            fprintf(stderr, " %.0f", (double)(chunk / 8192ull) * 24576.0 / (double)ms_best / 1e6);
        }
        fprintf(stderr, "; all singles:");
        for (const Ch& x : ch) fprintf(stderr, " %.0f", x.gbps);
        fprintf(stderr, "\n");
    }
    if (!ok) {  // (a mapping may be half made: the handles are released with their mappings)
        (void)hipGetLastError();
        for (Ch& x : ch) (void)hipMemRelease(x.h);
        ch.clear();
        (void)hipMemAddressFree(R, va_len);
        return 0;
    }
    hipMemGenericAllocationHandle_t* hs = new hipMemGenericAllocationHandle_t[n_need];
    for (uint32_t k = 0; k < n_need; ++k) hs[k] = ch[k].h;
    give_back(n_need);
    if (arena.base) arena.release();  // the old, too small arena of a recycled pool
    arena.base = at;
    arena.p = at + arena.skew;
    arena.cap = len - arena.skew;
    arena.vmm = true;
    arena.vmm_handles = hs;
    arena.vmm_n = n_need;
    arena.vmm_chunk = chunk;
    arena.va_base = R;
    arena.va_len = va_len;
    arena.map_len = len;
    arena.exact = false;
    ++arena.gen;
    arena.placed_for = A->arena.base; arena.placed_for_gen = A->arena.gen; arena.placed_gbps = here;
    return here >= 0.99f * (float)c->arena_good_gbps ? 1 : 2;
}

// Placement of a large result arena by ADDRESS (round 6): stage 2 of place_arena, for the boxes on which the arena's
// VIRTUAL address decides the bitset kernel's rate beside a given operand pool.  There the same eight 1 GiB chunks mapped
// at 89 addresses one GiB apart streamed at 6.40 TB/s at three of them and 5.89 at the others, and 17 other chunk sets
// mapped at one address all gave that address's rate to +-0.2 % (scripts/vmm_place2.hip, profiles/r06_vmm2_*.txt) -- the
// translation path, not the HBM channels.  (On other boxes every address of a process streams alike and other physical
// pages do not: stage 1, the candidate allocations.)  The candidates here are ADDRESSES: the arena's memory is created
// once (hipMemCreate, chunks of <= 1 GiB), one range of address space is reserved (hipMemAddressReserve: 512 GiB by
// default -- addresses, not memory), and the memory is mapped at one position of the range after the other -- hipMemMap,
// k_place_probe against the operand pool, hipMemUnmap: ~2 ms each -- until a position streams at arena_good_gbps.  Four
// positions within 1.2 % of one another end the search (the address does not decide here); when the first half of the
// positions holds no fast address, the second half is searched for the first position within 1.5 % of the best rate seen
// (the rates come in a few discrete levels) and the arena stays there.  One allocation, nothing released and created
// again (that stalled 250 ms per 8 GiB: the driver clears released memory).
// What the driver does NOT tolerate, and this function never does (scripts/vmm_place4.hip, gpurun_out/r6j):
//   * mapping at an ADDRESS that was mapped before (while the handle that was mapped there is alive): the address keeps
//     translating to the old memory and the new mapping is silently ignored -- moving eight chunks one GiB at a time left
//     two addresses aliasing one chunk, and the pattern test lost 1 GiB per move; a full-size C2 batch returned zeros in
//     its results.  Every position here is address space of its own (`pitch` apart, longer than the arena) and is visited
//     ONCE: the search never goes back to an earlier position, which is why it continues forward for an equal of the best;
//   * one handle of 8 GiB mapped, unmapped and mapped one position on: memory access fault at the second position, four
//     of four processes.  Chunks of 1 GiB each went through 40 positions in every variant;
//   * a range changing between a 1 GiB page and a table of smaller pages (both addresses and physical chunks GiB-aligned,
//     then not): faulted twice in scripts/vmm_place3.hip.  Positions are 2 MiB past a GiB boundary, never on one.
// Returns false (nothing changed) when the virtual-memory calls are missing or fail; place_arena then chooses among its
// candidate allocations alone.
static bool place_arena_va(rhip_ctx_t* c, DBuf& arena, size_t need, const rhip_pool_t* A) {
    if (!c->arena_vmm) return false;
    const size_t MB2 = 2ull << 20, G1 = 1ull << 30;
    const size_t len = (need + arena.skew + MB2 - 1) / MB2 * MB2;
    size_t free_b = 0, tot_b = 0;
    if (hipMemGetInfo(&free_b, &tot_b) != hipSuccess || free_b < len + (256ull << 20)) { (void)hipGetLastError(); return false; }
    if (g_fail_allocs.load(std::memory_order_relaxed) > 0) return false;  // (tests of the allocation-failure paths: through DBuf::ensure)
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    // positions: `pitch` apart -- the arena's length rounded up to the step, plus one step, so that consecutive positions
    // differ in every address bit from the step upwards (the fast zones are 3-9 GiB wide and have no period we could find)
    const size_t step = std::max<size_t>(MB2, c->arena_va_step / MB2 * MB2);
    const size_t pitch = (len + step - 1) / step * step + 2 * step;
    // ... and in the bits BELOW the step: position `pos` starts sub(pos) past its pitch mark, sub a multiple of 2 MiB below
    // the step (a box of the round's last pass streamed at ONE rate at all 28 positions of a process -- every one of them
    // 2 MiB past a GiB boundary -- and at another rate in the next process, while the hipMalloc'ed candidates of the fallback
    // search, whose addresses differ below the GiB as well, found a fast one at the second try)
    const size_t n_sub = step / MB2 > 1 ? step / MB2 - 1 : 1;  // (never a whole step: never back on a GiB boundary)
    auto sub = [&](size_t pos) { return ((pos * 197u) % n_sub) * MB2; };
    const size_t n_pos = (size_t)std::min<uint64_t>(24, std::max<uint64_t>(2, c->arena_va_window / pitch));
    const size_t chunk = std::min<size_t>(G1, step);
    const uint32_t n_chunks = (uint32_t)((len + chunk - 1) / chunk);
    auto chunk_len = [&](uint32_t k) { return std::min<size_t>(chunk, len - (size_t)k * chunk); };
    hipMemGenericAllocationHandle_t* hs = new hipMemGenericAllocationHandle_t[n_chunks];
    uint32_t n_made = 0;
    for (; n_made < n_chunks; ++n_made)
        if (hipMemCreate(&hs[n_made], chunk_len(n_made), &prop, 0) != hipSuccess) break;
    auto drop_chunks = [&]() { for (uint32_t k = 0; k < n_made; ++k) (void)hipMemRelease(hs[k]); delete[] hs; };
    if (n_made < n_chunks) { (void)hipGetLastError(); drop_chunks(); return false; }
    const size_t va_len = n_pos * pitch + G1 + MB2;
    void* R = nullptr;
    if (hipMemAddressReserve(&R, va_len, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); drop_chunks(); return false; }
    uint8_t* base0 = (uint8_t*)(((uintptr_t)R + G1 - 1) / G1 * G1) + MB2;  // = 2 MiB past a GiB boundary
    if (step % G1) base0 = (uint8_t*)(((uintptr_t)R + MB2 - 1) / MB2 * MB2);  // (sub-GiB steps: tests on small arenas)
    const u64 a_items = A->arena.cap / 8192ull;
    const u64 n_slots = need / 8192ull;
    const u64 stride = std::max<u64>(1, n_slots / ((1ull << 30) / 8192ull));  // ~1 GiB of the arena is written, spread over all of it
    const u64 n_items = (n_slots + stride - 1) / stride;
    hipStream_t s = c->stream;
    hipEvent_t e0 = c->ev[0], e1 = c->ev[1];
    uint8_t* at = nullptr;  // where the memory is mapped now
    uint32_t n_mapped = 0;
    auto unmap = [&]() {
        bool ok = true;
        for (uint32_t k = 0; k < n_mapped; ++k) ok = hipMemUnmap(at + (size_t)k * chunk, chunk_len(k)) == hipSuccess && ok;
        at = nullptr; n_mapped = 0;
        return ok;
    };
    auto bail = [&]() {
        (void)hipGetLastError();
        (void)unmap();
        drop_chunks();
        (void)hipMemAddressFree(R, va_len);
        return false;
    };
    auto map_at = [&](uint8_t* where) {  // (`where`: never mapped before)
        if (!unmap()) return false;
        at = where;
        for (; n_mapped < n_chunks; ++n_mapped)
            if (hipMemMap(where + (size_t)n_mapped * chunk, chunk_len(n_mapped), 0, hs[n_mapped], 0) != hipSuccess) return false;
        return hipMemSetAccess(where, len, &acc, 1) == hipSuccess;
    };
    auto probe = [&](float& gbps) {
        float ms_best = 1e30f;
        for (int r = 0; r < 3; ++r) {  // (the first pass warms the translations)
            if (hipEventRecord(e0, s) != hipSuccess) return false;
            hipLaunchKernelGGL(k_place_probe, dim3(8192), dim3(256), 0, s, A->arena.as<uint8_t>(), a_items, at + arena.skew, n_slots, stride);
            if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return false;
            float ms = 0;
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return false;
            if (r && ms < ms_best) ms_best = ms;
        }
        gbps = (float)((double)n_items * 24576.0 / (double)std::max(ms_best, 1e-6f) / 1e6);
        return true;
    };
    float best = 0.f, worst = 1e30f, here = 0.f;  // (the probe rates are appended to the candidates' in last_placement)
    for (size_t pos = 0; pos < n_pos; ++pos) {
        if (!map_at(base0 + pos * pitch + sub(pos)) || !probe(here)) return bail();
        c->last_placement.push_back(here);
        if (here >= (float)c->arena_good_gbps) break;                     // a fast address
        if (pos >= n_pos / 2 && here >= 0.985f * best) break;             // second half: as good as the best of the first
        best = std::max(best, here);
        worst = std::min(worst, here);
        // a FLAT landscape -- four positions within 1.2 % of one another, below the bar: on such a box (two of the round's
        // six) the address does not decide, every position of a process streams alike and another process, or another
        // allocation, streams at another rate.  The search ends; place_arena goes on with candidate ALLOCATIONS, this arena
        // being the first of them.
        if (pos >= 3 && best - worst < 0.012f * best) break;
    }
    if (arena.base) arena.release();  // the old, too small arena of a recycled pool
    arena.base = at;
    arena.p = at + arena.skew;
    arena.cap = len - arena.skew;
    arena.vmm = true;
    arena.vmm_handles = hs;
    arena.vmm_n = n_chunks;
    arena.vmm_chunk = chunk;
    arena.va_base = R;
    arena.va_len = va_len;
    arena.map_len = len;
    arena.exact = false;
    ++arena.gen;
    arena.placed_for = A->arena.base; arena.placed_for_gen = A->arena.gen; arena.placed_gbps = here;
    return true;
}

// Measured placement of a large result arena (rhip_ctx_s::arena_tries).  The physical address of device memory is not
// visible to a process, and it is what decides: with the operand pool and the result arena in ONE 128 GiB allocation the
// bitset kernel of C2 takes 4.62-4.66 ms while the arena starts within ~24 GiB behind the pool, 3.91-3.98 ms in 5-6 GiB
// wide windows further out (distances 27-31, 66-71, 82-87, 98-103 ... GiB) and 4.32-4.38 ms everywhere else
// (profiles/r04_arena_distance.txt) -- the address hash that spreads a stream over the HBM channels also decides how often
// the lockstep read and write streams of the kernel meet in one channel.  So the candidates are MEASURED: each is
// allocated, probed with the kernel's own access pattern against the operand pool it will be written beside, and the
// fastest is kept; the others are released.  Only when `arena` needs a new allocation of >= arena_place_min bytes, the
// operand arena holds >= 64 MiB and no batch of the context is in flight (the probes wait for the device: this is the one
// place where rhip_pairwise_begin blocks, tens of milliseconds, once per NEW result pool -- a recycled pool keeps its
// placement).  Transient footprint: the candidates stay alive until the choice (a released block comes straight back as
// the next candidate, and releasing mid-search is slow), so the search stops at half of the memory that was free when
// it began (include/roaring_hip.h says so to callers; RHIP_ARENA_TRIES=0 switches the search off).
static void place_arena(rhip_ctx_t* c, DBuf& arena, size_t need, const rhip_pool_t* A) {
    const u64 a_items = A->arena.cap / 8192ull;
    const u64 n_slots = need / 8192ull;
    const u64 stride = std::max<u64>(1, n_slots / ((2ull << 30) / 8192ull));  // ~2 GiB of the candidate is written, spread over all of it
    const u64 n_items = (n_slots + stride - 1) / stride;
    hipStream_t s = c->stream;
    struct Cand { DBuf buf; float gbps = 0; };
    struct Cands : std::vector<Cand> {  // (a HIP error in the middle of the search must not leak the candidates it leaves behind)
        int keep = -1;
        size_t live() const { size_t n = 0; for (const Cand& x : *this) n += x.buf.base ? 1 : 0; return n; }
        ~Cands() { for (int k = 0; k < (int)size(); ++k) if (k != keep) (*this)[k].buf.release(); }
    } cands;
    // A spare that was already probed against THIS operand arena and streamed well -- the arena of a result pool the caller
    // freed (rhip_pool_free parks a placed arena with its context), or a good loser of an earlier search -- is taken as it
    // is: no allocation, no probe.  A caller without `reuse` pays the search once, not per call.
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        int pick = -1;
        for (size_t k = 0; k < c->arena_spares.size(); ++k) {
            const DBuf& sp = c->arena_spares[k];
            if (sp.base && sp.cap >= need && sp.cap <= need + need / 8 && sp.skew == arena.skew && sp.placed_for == A->arena.base &&
                sp.placed_for_gen == A->arena.gen && (sp.placed_winner || sp.placed_gbps >= 0.99f * (float)c->arena_good_gbps) &&  // (a lesser loser is a candidate of the search below)
                (pick < 0 || sp.placed_gbps > c->arena_spares[(size_t)pick].placed_gbps))
                pick = (int)k;
        }
        if (pick >= 0) {
            DBuf take = c->arena_spares[(size_t)pick];
            c->arena_spares.erase(c->arena_spares.begin() + pick);
            if (arena.base) arena.release();
            const bool p2 = arena.pow2_large;
            const uint64_t g = arena.gen;
            arena = take;
            arena.pow2_large = p2;
            arena.exact = false;
            arena.gen = g + 1;
            c->last_placement.clear();
            c->last_placement.push_back(take.placed_gbps);
            return;
        }
    }
    // stage 0 (round 6, second half): the arena composed of the best of three times as many physical chunks as it needs
    c->last_placement.clear();
    c->batches_since_place = 0;
    int by_chunks = place_arena_chunks(c, arena, need, A);
    if (by_chunks == 1) { arena.placed_winner = true; return; }
    if (by_chunks == 2) {
        // One process in ten draws a composition below the bar out of chunks that each stream well ("what is still
        // open"): a SECOND composition, out of chunks the first did not see, before the stages that allocate whole candidates
        DBuf second;
        second.skew = arena.skew; second.round_to = arena.round_to;
        const int again = place_arena_chunks(c, second, need, A, true);
        if (again && second.placed_gbps > arena.placed_gbps) {
            const bool p2 = arena.pow2_large;
            const uint64_t g = arena.gen;
            std::swap(arena, second);  // (the loser is kept as a spare: released memory would have to be cleared for what follows)
            arena.pow2_large = p2; arena.gen = g + 1;
            by_chunks = again;
        }
        if (again) {
            std::lock_guard<std::mutex> lk(g_ctx_mu);
            if (c->arena_keep_spares) c->arena_spares.push_back(second);
            else second.release();
        }
        if (by_chunks == 1) { arena.placed_winner = true; return; }
    }
    size_t free_at_start = 0, tot_mem = 0;
    if (hipMemGetInfo(&free_at_start, &tot_mem) != hipSuccess) { (void)hipGetLastError(); free_at_start = 0; }
    if (free_at_start < 2 * need) { if (by_chunks) arena.placed_winner = true; return; }  // (no room to choose: the caller's ordinary allocation follows)
    hipEvent_t e0 = c->ev[0], e1 = c->ev[1];
    int best = -1;
    if (by_chunks == 2) {  // below the bar: candidate 0 of the stages that follow (its configuration -- skew, rounding -- stays with `arena`)
        Cand cd;
        cd.buf = arena;
        cd.gbps = arena.placed_gbps;
        cands.push_back(cd);
        best = 0;
        arena.base = nullptr; arena.p = nullptr; arena.cap = 0;
        arena.vmm = false; arena.vmm_handles = nullptr; arena.vmm_n = 0; arena.vmm_chunk = 0; arena.va_base = nullptr; arena.va_len = arena.map_len = 0;
    }
    const int n_placed_in = (int)cands.size();
    auto probe = [&](Cand& cur) {
        float ms_best = 1e30f;
        for (int r = 0; r < 3; ++r) {  // (the first pass touches the pages)
            HIPCHK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(k_place_probe, dim3(8192), dim3(256), 0, s, A->arena.as<uint8_t>(), a_items, cur.buf.as<uint8_t>(), n_slots, stride);
            HIPCHK(hipEventRecord(e1, s));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < ms_best) ms_best = ms;
        }
        cur.gbps = (float)((double)n_items * 24576.0 / (double)ms_best / 1e6);
        cur.buf.placed_for = A->arena.base; cur.buf.placed_for_gen = A->arena.gen; cur.buf.placed_gbps = cur.gbps;
        c->last_placement.push_back(cur.gbps);
    };
    // the spares of earlier searches that fit (same exact-size allocation: within 1/8 above the need) are candidates again
    // -- where they lie relative to THIS operand pool is a new question -- and cost no allocation
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        for (size_t k = 0; k < c->arena_spares.size();) {
            DBuf& sp = c->arena_spares[k];
            if (sp.base && sp.cap >= need && sp.cap <= need + need / 8 && sp.skew == arena.skew) {
                Cand cd;
                cd.buf = sp;
                cands.push_back(cd);
                c->arena_spares.erase(c->arena_spares.begin() + (long)k);
            } else {
                ++k;
            }
        }
    }
    for (size_t k = (size_t)n_placed_in; k < cands.size(); ++k) {
        probe(cands[k]);
        if (best < 0 || cands[k].gbps > cands[best].gbps) best = (int)k;
    }
    const int n_spares_in = (int)cands.size();
    // arena_tries candidates -- and up to as many again while even the best of them is in the slow band: consecutive
    // allocations are neighbours (one process of round 4 drew ten slow candidates in a row), and with the losers still
    // alive the driver has to hand out pages further away
    for (int t = n_spares_in; t < 2 * c->arena_tries; ++t) {
        if (best >= 0 && cands[best].gbps >= c->arena_good_gbps) break;
        if (t >= c->arena_tries && best >= 0 && cands[best].gbps >= c->arena_fair_gbps) break;
        if (best >= 0) {  // never run the device out of memory for one more candidate
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < need + need / 4 + (1ull << 30)) { (void)hipGetLastError(); break; }
            // ... and never hold more than half of what was free when the search began (other contexts / ranks sharing the
            // device allocate too).  The search ENDS there: releasing a loser to make room costs ~0.5 s per 16 GiB block
            // once another allocation follows it (a C2 `or` call that had to place its arena took 1.8 s holding four
            // candidates, 5.6 s holding two, 14 ms when nothing was released before the choice) and hands the same pages out again
            if ((cands.size() + 1) * need > free_at_start / 2) break;
        }
        Cand cd;
        cd.buf.skew = arena.skew;
        cd.buf.round_to = arena.round_to;
        // Exactly the bytes needed: the driver composes an allocation out of naturally aligned power-of-two blocks, largest
        // first -- 8 000 MiB = 4 G + 2 G + 1 G + ...; with the usual 12.5 % of slack (9.4 GB = 8 G + ...) or as a power of two
        // the first block is an 8 GiB-aligned 8 GiB one, whose distance to an 8 GiB operand block is a multiple of 8 GiB:
        // never inside a fast window (scripts/arena_place.hip: 0 of 40 such allocations fast, 15 of 28 exact ones).
        cd.buf.pow2_large = false;
        cd.buf.exact = true;
        try {
            cd.buf.ensure(need);
        } catch (int) {
            (void)hipGetLastError();  // (the failed hipMalloc must not surface as the batch's own error later)
            if (best >= 0) break;     // out of memory for another candidate: choose among those there are
            throw;
        }
        cands.push_back(cd);
        Cand& cur = cands.back();
        probe(cur);
        if (best < 0 || cur.gbps > cands[best].gbps) best = (int)cands.size() - 1;
        if (cur.gbps >= c->arena_good_gbps) break;
    }
    // (round 6) no candidate ALLOCATION at the bar: one more allocation, and this one is moved through an address window
    // (place_arena_va above) -- on the boxes where the arena's virtual address decides, a fast position is usually among
    // the first few; on a box where it does not, four positions that stream alike end the attempt.  Candidates first
    // because they were the more dependable of the two across the boxes of round 6 (0.777-0.798 of peak in ten of ten
    // processes; by address alone 0.785-0.801 on four boxes and 0.68-0.78 on two whose addresses all streamed alike), and
    // the address search second because it found 6.3-6.4 TB/s where ten candidates had not (round 5's driver box).
    if (c->arena_vmm && (best < 0 || cands[best].gbps < (float)c->arena_good_gbps)) {
        Cand cd;
        cd.buf.skew = arena.skew;
        cd.buf.round_to = arena.round_to;
        if (place_arena_va(c, cd.buf, need, A)) {
            cd.gbps = cd.buf.placed_gbps;
            cands.push_back(cd);
            if (best < 0 || cd.gbps > cands[best].gbps) best = (int)cands.size() - 1;
        }
    }
    if (best < 0) return;
    // the losers become spares (RHIP_ARENA_SPARES=0: released instead), as long as the spares stay below half of the memory
    // that was free when the search began
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        size_t held = 0;
        for (const DBuf& b : c->arena_spares) held += b.cap;
        for (int k = 0; k < (int)cands.size(); ++k) {
            if (k == best || !cands[k].buf.base) continue;
            if (c->arena_keep_spares && held + cands[k].buf.cap <= free_at_start / 2) {
                held += cands[k].buf.cap;
                c->arena_spares.push_back(cands[k].buf);
                cands[k].buf.base = nullptr; cands[k].buf.p = nullptr; cands[k].buf.cap = 0;  // (ownership moved)
            }
        }
    }
    cands.keep = best;  // (what is left is released when `cands` goes out of scope)
    if (arena.base) {  // the old, too small arena of a recycled pool
        arena.release();
    }
    const bool p2 = arena.pow2_large;
    const uint64_t g = arena.gen;
    arena = cands[best].buf;
    arena.pow2_large = p2;
    arena.exact = false;
    arena.gen = g + 1;
    arena.placed_winner = true;
}

// Everything of a pairwise call up to and including the launch of k_tail: nothing here waits for the device.
static rhip_batch_t* pairwise_begin_ops(rhip_ctx_t* c, size_t n_ops, const rhip_op* ops_, rhip_pool_t* A, rhip_pool_t* B,
                                        size_t npairs, const uint32_t* lhs, const uint32_t* rhs, rhip_pool_t* reuse,
                                        rhip_pairlist_t* L = nullptr) {
    rhip_pool_t* R = nullptr;
    try {
        if (!c) { set_err("null context"); throw (int)RHIP_ERR_ARG; }
        if (n_ops < 1 || n_ops > 4 || !ops_) { set_err("a batch takes one to four ops"); throw (int)RHIP_ERR_ARG; }
        OpSet ops;
        ops.n = (int)n_ops;
        for (size_t o = 0; o < n_ops; ++o) {
            ops.op[o] = (int)ops_[o];
            if (ops.op[o] < 0 || ops.op[o] > 3) { set_err("bad op"); throw (int)RHIP_ERR_ARG; }
        }
        const int op = ops.op[0];
        DeviceGuard dguard_(c->device);
        if (reuse && (reuse->pending || reuse->in_use)) {
            reuse = nullptr;  // still owned / read by a batch
            set_err("`reuse` is the result or an operand of a batch still in flight");
            throw (int)RHIP_ERR_ARG;
        }
        check_pair_args(A, B, npairs, lhs, rhs);
        if (reuse && (reuse == A || reuse == B)) {
            reuse = nullptr;  // not ours to recycle
            set_err("`reuse` must not be one of the operand pools");
            throw (int)RHIP_ERR_ARG;
        }
        const int slot = c->acquire_slot();
        hipStream_t s = c->stream;
        if (c->timing) HIPCHK(hipEventRecord(c->evs[slot][0], s));
        HostClock clk(c);
        // With a batch already in flight, the planning kernels of this one go to an auxiliary stream (the one its op
        // leaves idle when the class kernels are forked): they touch only this slot's scratch and read-only operands,
        // so they run beside the class kernels of the previous batch; the main stream waits for them below.
        hipStream_t ps = s;
        if (c->plan_overlap && c->overlap && c->in_flight() > 0)
            ps = c->aux[(ops.n == 1 && (op == OP_OR || op == OP_XOR)) ? 1 : 2];
        Plan P = plan(c, ops, A, B, npairs, lhs, rhs, 0, slot, ps, &clk, L);
        if (P.plan_stream != s) {
            HIPCHK(hipEventRecord(c->ev_plan[slot], P.plan_stream));
            HIPCHK(hipStreamWaitEvent(s, c->ev_plan[slot], 0));
        }
        const CandOut& CO = P.CO;
        R = reuse ? reuse : new rhip_pool_s();
        reuse = nullptr;
        R->ctx = c;
        R->n_bitmaps = (uint32_t)(npairs * n_ops);
        R->is64 = A->is64;
        R->host_dir = false;
        R->host_bm = false;
        R->host_w = false;
        R->h_cards.clear();
        ensure_dir(R, (uint32_t)(npairs * n_ops), P.ub_cand);
        R->arena.skew = c->arena_skew;
        R->arena.round_to = c->arena_round;
        R->arena.pow2_large = c->arena_pow2;
        // (a batch of this size plans on the main stream, so the probes below are ordered behind everything enqueued so far
        // -- the batches in flight included -- and run alone)
        if (c->arena_tries > 1 && P.arena_bound + 64 > R->arena.cap && P.arena_bound + 64 >= c->arena_place_min &&
            A->arena.cap >= std::min<uint64_t>(64ull << 20, std::max<uint64_t>(c->arena_place_min, 8192)) && P.plan_stream == s &&
            c->in_flight() == 0)
            place_arena(c, R->arena, P.arena_bound + 64, A);
        R->arena.ensure(P.arena_bound + 64);
        OutView O{};
        O.key = CO.key; O.meta = c->ss[slot].o_meta.as<u64>(); O.off = CO.off; O.slot = nullptr;
        O.arena = R->arena.as<uint8_t>();
        PoolView VA = A->view(), VB = B->view();
        const unsigned tail_blocks = (unsigned)std::max<uint64_t>(1, (P.ub_cand + TAIL_TILE - 1) / TAIL_TILE);
        uint32_t join_mask = 0;
        run_kernels(c, ops, VA, VB, O, P, 0, c->spin_join ? &join_mask : nullptr);
        if (join_mask) hipLaunchKernelGGL(k_join_wait, dim3(1), dim3(64), 0, s, (const u64*)P.join_flags(), join_mask, c->join_timeout_word(),
                                          (u64)c->join_spins, c->join_force_fail);
        // drop empty results, build the result directory, totals
        DirOut D{R->bm_start.as<u64>(), R->key.as<u64>(), R->type.as<uint8_t>(), R->card.as<uint32_t>(),
                 R->nruns.as<uint32_t>(), R->off.as<u64>()};
        const uint64_t seq = ++c->seq;
        hipLaunchKernelGGL(k_tail, dim3(tail_blocks), dim3(256), 0,
                           s, P.ranges(), CO, O.meta, D, (uint32_t)(npairs * n_ops), P.tail_lb(), P.tail_part(),
                           (Stats*)c->slot_stats(slot), (u64*)c->done_flag(slot), (u64)seq,
                           join_mask ? (const u64*)c->join_timeout_word() : (const u64*)nullptr);
        if (c->timing) HIPCHK(hipEventRecord(c->evs[slot][1], s));
        HIPCHK(hipGetLastError());  // a refused launch anywhere above must not pass silently
        clk.lap(3);
        rhip_batch_t* b = new rhip_batch_s{c, R, A, B, seq, slot, P.may_bb, P.ranges(), P.grouped, L, TailRedo{}};
        if (join_mask) {
            TailRedo& T = b->redo;
            T.armed = true; T.CO = CO; T.meta = O.meta; T.D = D; T.n_virt = (uint32_t)(npairs * n_ops);
            T.lb = P.tail_lb(); T.part = P.tail_part(); T.blocks = tail_blocks;
            T.status_words = 3 * P.sc.n_tail_tiles;
        }
        c->slot_busy[slot] = true;
        c->slot_flagjoin[slot] = join_mask != 0;
        R->pending = true;
        ++A->in_use;
        ++B->in_use;
        if (L) ++L->in_use;
        b->ce = P.ce;
        b->Q = P.Q;
        if (P.ce) ++P.ce->in_use;
        c->last_plan_cached = P.cache_hit;
        return b;
    } catch (int e) {
        last_status() = e;
        if (R) { R->release(); delete R; }
        if (reuse) { reuse->release(); delete reuse; }
        return nullptr;
    }
}

extern "C" rhip_batch_t* rhip_pairwise_begin(rhip_ctx_t* c, rhip_op op, rhip_pool_t* A, rhip_pool_t* B, size_t npairs,
                                            const uint32_t* lhs, const uint32_t* rhs, rhip_pool_t* reuse) {
    return pairwise_begin_ops(c, 1, &op, A, B, npairs, lhs, rhs, reuse);
}
extern "C" rhip_batch_t* rhip_pairwise_multi_begin(rhip_ctx_t* c, size_t n_ops, const rhip_op* ops, rhip_pool_t* A,
                                                  rhip_pool_t* B, size_t npairs, const uint32_t* lhs, const uint32_t* rhs,
                                                  rhip_pool_t* reuse) {
    return pairwise_begin_ops(c, n_ops, ops, A, B, npairs, lhs, rhs, reuse);
}
extern "C" rhip_pool_t* rhip_pairwise_multi(rhip_ctx_t* c, size_t n_ops, const rhip_op* ops, rhip_pool_t* A, rhip_pool_t* B,
                                            size_t npairs, const uint32_t* lhs, const uint32_t* rhs, rhip_pool_t* reuse) {
    rhip_batch_t* b = pairwise_begin_ops(c, n_ops, ops, A, B, npairs, lhs, rhs, reuse);
    return b ? rhip_pairwise_end(b) : nullptr;
}

// The ONE host wait of the call; the result pool becomes usable.  Batches may be ended in any order.
extern "C" rhip_pool_t* rhip_pairwise_end(rhip_batch_t* b) {
    if (!b) { set_err("null batch"); last_status() = RHIP_ERR_ARG; return nullptr; }
    rhip_ctx_t* c = b->c;
    rhip_pool_t* R = b->R;
    const int slot = b->slot;
    --b->A->in_use;
    --b->B->in_use;
    rhip_pairlist_t* bl = b->L;
    if (bl) --bl->in_use;
    if (b->ce) --b->ce->in_use;
    auto drop_deferred = [](rhip_pool_t* X) {
        if (X->free_deferred && X->in_use == 0 && X->list_pins == 0) { X->release(); delete X; }
    };
    auto drop_list = [](rhip_pairlist_t* X) {  // (after the wait below: nothing reads its device copy any more)
        if (X && X->free_deferred && X->in_use == 0) pairlist_destroy(X);
    };
    try {
        DeviceGuard dguard_(c->device);
        HostClock clk(c);
        Stats st;
        finish_stats(c, nullptr, &st, b->may_bb, b->seq, slot);
        if (b->redo.armed && __atomic_load_n(c->join_timeout_word(), __ATOMIC_ACQUIRE)) {
            // A gate of this context gave up (rhip_plan.h: kernels of different streams did not run side by side -- a
            // tool that serialises them, a busy device): this batch's tail may have run before its class kernels had
            // finished.  Nothing is lost: the class kernels' outputs are complete once their streams are idle, and the
            // tail only reads them.  Wait the ordinary way, clear the tail's scratch, run it again.  The word is per
            // context, so every flag-joined batch that ends while it is set takes this path (its tail may be the early
            // one); it is cleared when no flag-joined batch is left in flight.  Later batches join with events.
            if (c->spin_join) {
                static std::atomic<bool> warned{false};
                if (!warned.exchange(true))
                    fprintf(stderr, "libroaring_hip: kernels of different streams did not run side by side within %.2f s (a tool that "
                                    "serialises kernels, a busy device): batches join their streams with events from here on\n",
                            (double)c->join_spins * 3.4e-6);
            }
            c->spin_join = false;
            for (int a = 0; a < rhip_ctx_s::N_AUX; ++a) HIPCHK(hipStreamSynchronize(c->aux[a]));
            HIPCHK(hipStreamSynchronize(c->stream));
            const TailRedo& T = b->redo;
            HIPCHK(hipMemsetAsync(T.lb.status, 0, 8 * T.status_words, c->stream));
            HIPCHK(hipMemsetAsync(T.lb.ticket, 0, 8, c->stream));
            const uint64_t seq2 = ++c->seq;
            hipLaunchKernelGGL(k_tail, dim3(T.blocks), dim3(256), 0, c->stream, b->ranges, T.CO, T.meta, T.D, T.n_virt, T.lb, T.part,
                               (Stats*)c->slot_stats(slot), (u64*)c->done_flag(slot), (u64)seq2, (const u64*)nullptr);
            HIPCHK(hipGetLastError());
            wait_word(c, c->done_flag(slot), seq2);
            memcpy(&st, c->slot_stats(slot), sizeof(Stats));
            c->stats.matched_pairs = st.matched_pairs; c->stats.passthrough = st.passthrough;
            c->stats.bytes_in = st.bytes_in; c->stats.bytes_out = st.bytes_out;
            c->stats.n_bitset_pairs = st.n_bb; c->stats.result_containers = st.result_containers;
            ++c->join_recovered;
            bool more = false;  // another flag-joined batch still in flight?  (their batch objects are the callers'; count slots)
            for (int k = 0; k < rhip_ctx_s::N_SLOTS; ++k) more = more || (k != slot && c->slot_busy[k] && c->slot_flagjoin[k]);
            if (!more) __atomic_store_n(c->join_timeout_word(), 0ull, __ATOMIC_RELEASE);
        }
        c->slot_flagjoin[slot] = false;
        if (c->class_stats) {  // (diagnostics: the slot's queues and meta words are intact until its next batch)
            rhip_ctx_s::SlotScratch& SS = c->ss[slot];
            rhip_ctx_s::SlotScratch& SQ = b->Q ? *b->Q : SS;  // (a cached plan's queues live with the pair list)
            ClassQueues CQ{SQ.q[CLS_BB].as<BBItem>(), SQ.q[CLS_BBA].as<BBItem>(),
                           {SQ.q[CLS_FILT].as<FatItem>(), SQ.q[CLS_WAVE].as<FatItem>(), SQ.q[CLS_PROBE].as<FatItem>(),
                            SQ.q[CLS_USMALL].as<FatItem>(), SQ.q[CLS_BA].as<FatItem>()},
                           {SQ.q[CLS_GEN].as<GenItem>(), SQ.q[CLS_RUNS].as<GenItem>(), SQ.q[CLS_RUNS16].as<GenItem>(),
                            SQ.q[CLS_RUNS16W].as<GenItem>()},
                           SQ.q[CLS_COPY].as<CopyItem>()};
            c->misc.ensure(8 * 3 * N_CLS + 64);
            hipLaunchKernelGGL(k_class_stats, dim3(N_CLS), dim3(256), 0, c->stream, (const u64*)b->ranges, CQ,
                               (const u64*)SS.o_meta.as<u64>(), c->misc.as<u64>(), b->grouped ? 1 : 0);
            HIPCHK(hipMemcpyAsync(c->cls_stats, c->misc.p, 8 * 3 * N_CLS, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        clk.lap(4);
        R->n_cont = st.result_containers;
        R->arena_used = st.slot_bytes + 64;
        for (int t = 0; t < 3; ++t) R->census[t] = st.n_type[t] ? 1 : 0;
        R->pending = false;
        c->slot_busy[slot] = false;
        if (c->in_flight() == 0) trim_spares_when_steady(c);
        rhip_pool_t *A = b->A, *B = b->B;
        delete b;
        drop_deferred(A);  // (the batch has completed: nothing reads the operands any more)
        if (B != A) drop_deferred(B);
        drop_list(bl);
        clk.lap(5);
        return R;
    } catch (int e) {
        last_status() = e;
        c->slot_busy[slot] = false;
        (void)hipStreamSynchronize(c->stream);
        rhip_pool_t *A = b->A, *B = b->B;
        R->release();
        delete R;
        delete b;
        drop_deferred(A);
        if (B != A) drop_deferred(B);
        drop_list(bl);
        return nullptr;
    }
}

extern "C" rhip_pool_t* rhip_pairwise(rhip_ctx_t* c, rhip_op op_, rhip_pool_t* A, rhip_pool_t* B, size_t npairs,
                                      const uint32_t* lhs, const uint32_t* rhs, rhip_pool_t* reuse) {
    rhip_batch_t* b = rhip_pairwise_begin(c, op_, A, B, npairs, lhs, rhs, reuse);
    return b ? rhip_pairwise_end(b) : nullptr;
}

static int pairwise_cardinality_impl(rhip_ctx_t* c, rhip_op op_, rhip_pool_t* A, rhip_pool_t* B, size_t npairs,
                                     const uint32_t* lhs, const uint32_t* rhs, uint64_t* out, rhip_pairlist_t* L) {
    try {
        if (!c) { set_err("null context"); throw (int)RHIP_ERR_ARG; }
        int op = (int)op_;
        if (op < 0 || op > 3) { set_err("bad op"); throw (int)RHIP_ERR_ARG; }
        DeviceGuard dguard_(c->device);
        check_pair_args(A, B, npairs, lhs, rhs);
        if (npairs && !out) { set_err("null output array"); throw (int)RHIP_ERR_ARG; }
        hipStream_t s = c->stream;
        // per-bitmap cardinalities for inclusion-exclusion (roaring.c:3086-3107)
        std::vector<uint64_t> cA(A->n_bitmaps), cB(B->n_bitmaps);
        if (op != OP_AND) {
            int e;
            if ((e = rhip_pool_cardinalities(A, cA.data())) != 0) throw e;
            if ((e = rhip_pool_cardinalities(B, cB.data())) != 0) throw e;
        }
        if (c->timing) HIPCHK(hipEventRecord(c->ev[0], s));
        Plan P = plan(c, OpSet{}, A, B, npairs, lhs, rhs, 1, rhip_ctx_s::SYNC_SLOT, s, nullptr, L);
        OutView O{};
        PoolView VA = A->view(), VB = B->view();
        run_kernels(c, OpSet{}, VA, VB, O, P, 1);
        hipLaunchKernelGGL(k_card_stats, dim3(1), dim3(64), 0, s, P.ranges(), (Stats*)c->h_pinned);
        if (npairs) HIPCHK(hipMemcpyAsync(out, c->pair_acc.p, 8 * npairs, hipMemcpyDeviceToHost, s));
        finish_stats(c, nullptr, nullptr, P.may_bb);
        for (size_t i = 0; i < npairs; ++i) {
            uint64_t in = out[i];
            switch (op) {
                case OP_OR: out[i] = cA[lhs[i]] + cB[rhs[i]] - in; break;
                case OP_XOR: out[i] = cA[lhs[i]] + cB[rhs[i]] - 2 * in; break;
                case OP_ANDNOT: out[i] = cA[lhs[i]] - in; break;
                default: break;
            }
        }
        return RHIP_OK;
    } catch (int e) { return e; }
}

extern "C" int rhip_pairwise_cardinality(rhip_ctx_t* c, rhip_op op, rhip_pool_t* A, rhip_pool_t* B, size_t npairs,
                                         const uint32_t* lhs, const uint32_t* rhs, uint64_t* out) {
    return pairwise_cardinality_impl(c, op, A, B, npairs, lhs, rhs, out, nullptr);
}

// ------------------------------------------------------------------ prepared pair lists (roaring_hip.h)
// the end of a pair list: its operand pools are unpinned, and released if their rhip_pool_free was waiting for that
static void pairlist_destroy(rhip_pairlist_t* L) {
    if (L->pinned) {
        rhip_pool_t *A = L->A, *B = L->B;
        --A->list_pins;
        --B->list_pins;
        auto drop = [](rhip_pool_t* X) {
            if (X->free_deferred && X->in_use == 0 && X->list_pins == 0) { X->free_deferred = false; rhip_pool_free(X); }
        };
        drop(A);
        if (B != A) drop(B);
    }
    L->d_idx.release();
    if (L->cache) { L->cache->release(); delete L->cache; }
    delete L;
}
static rhip_pairlist_t* pairlist_make(rhip_ctx_t* c, rhip_pool_t* A, rhip_pool_t* B, std::vector<uint32_t>&& lhs,
                                      std::vector<uint32_t>&& rhs) {
    rhip_pairlist_t* L = nullptr;
    try {
        if (!c) { set_err("null context"); throw (int)RHIP_ERR_ARG; }
        check_pair_args(A, B, lhs.size(), lhs.data(), rhs.data());
        DeviceGuard dguard_(c->device);
        L = new rhip_pairlist_s();
        L->ctx = c; L->A = A; L->B = B;
        L->npairs = lhs.size();
        L->lhs = std::move(lhs);
        L->rhs = std::move(rhs);
        fetch_bounds(A);
        fetch_bounds(B);
        L->sums = pair_sums(A, B, L->npairs, L->lhs.data(), L->rhs.data());  // (range-checks every index)
        L->genA = A->bounds_gen; L->genB = B->bounds_gen;
        ++A->list_pins;
        ++B->list_pins;
        L->pinned = true;
        L->d_idx.ensure(8 * L->npairs + 16);
        if (L->npairs) {
            HIPCHK(hipMemcpyAsync(L->d_idx.p, L->lhs.data(), 4 * L->npairs, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync((char*)L->d_idx.p + 4 * L->npairs, L->rhs.data(), 4 * L->npairs, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        return L;
    } catch (int e) {
        last_status() = e;
        if (L) { pairlist_destroy(L); }
        return nullptr;
    }
}
extern "C" rhip_pairlist_t* rhip_pairlist_create(rhip_ctx_t* c, rhip_pool_t* A, rhip_pool_t* B, size_t npairs,
                                                 const uint32_t* lhs, const uint32_t* rhs) {
    if (npairs && (!lhs || !rhs)) { set_err("null pair index array"); last_status() = RHIP_ERR_ARG; return nullptr; }
    return pairlist_make(c, A, B, std::vector<uint32_t>(lhs, lhs + npairs), std::vector<uint32_t>(rhs, rhs + npairs));
}
extern "C" rhip_pairlist_t* rhip_pairlist_all_pairs(rhip_ctx_t* c, rhip_pool_t* A) {
    if (!A) { set_err("null pool"); last_status() = RHIP_ERR_ARG; return nullptr; }
    const uint64_t n = A->n_bitmaps, np = n * (n ? n - 1 : 0) / 2;
    if (np >= 0x3FFFFFF0ull) { set_err("too many pairs"); last_status() = RHIP_ERR_ARG; return nullptr; }
    std::vector<uint32_t> l, r;
    l.reserve((size_t)np); r.reserve((size_t)np);
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = i + 1; j < n; ++j) { l.push_back(i); r.push_back(j); }
    return pairlist_make(c, A, A, std::move(l), std::move(r));
}
extern "C" rhip_pairlist_t* rhip_pairlist_successive(rhip_ctx_t* c, rhip_pool_t* A) {
    if (!A) { set_err("null pool"); last_status() = RHIP_ERR_ARG; return nullptr; }
    std::vector<uint32_t> l, r;
    for (uint32_t i = 0; i + 1 < A->n_bitmaps; ++i) { l.push_back(i); r.push_back(i + 1); }
    return pairlist_make(c, A, A, std::move(l), std::move(r));
}
extern "C" size_t rhip_pairlist_size(const rhip_pairlist_t* L) { return L ? L->npairs : 0; }
extern "C" int rhip_pairlist_pairs(const rhip_pairlist_t* L, uint32_t* lhs, uint32_t* rhs) {
    if (!L) { set_err("null pair list"); return RHIP_ERR_ARG; }
    if (lhs && L->npairs) memcpy(lhs, L->lhs.data(), 4 * L->npairs);
    if (rhs && L->npairs) memcpy(rhs, L->rhs.data(), 4 * L->npairs);
    return RHIP_OK;
}
// forget the list's cached plans (those no batch in flight reads) and release their device memory; the next batch of
// each (ops, form) plans afresh.  Returns how many were dropped.
extern "C" int rhip_pairlist_drop_plans(rhip_pairlist_t* L) {
    if (!L || !L->cache) return 0;
    DeviceGuard dguard_(L->ctx->device);
    int n = 0;
    for (auto& e : L->cache->e) {
        if (e.in_use || !e.valid) continue;
        (void)hipStreamSynchronize(L->ctx->stream);  // (a synchronous call may still be reading it on the stream)
        e.valid = false;
        e.ss.release();
        e.ranges.release();
        ++n;
    }
    return n;
}
extern "C" void rhip_pairlist_free(rhip_pairlist_t* L) {
    if (!L) return;
    if (L->in_use > 0) { L->free_deferred = true; return; }  // released by the last batch that reads it (rhip_pairwise_end)
    pairlist_destroy(L);
}
extern "C" rhip_batch_t* rhip_pairwise_list_begin(rhip_ctx_t* c, size_t n_ops, const rhip_op* ops, rhip_pairlist_t* L,
                                                 rhip_pool_t* reuse) {
    if (!L || L->free_deferred || (c && L->ctx != c)) {
        set_err(!L || L->free_deferred ? "null (or freed) pair list" : "the pair list belongs to another context");
        last_status() = RHIP_ERR_ARG;
        if (reuse && !(reuse->pending || reuse->in_use)) { reuse->release(); delete reuse; }  // consumed, as everywhere
        return nullptr;
    }
    return pairwise_begin_ops(c, n_ops, ops, L->A, L->B, L->npairs, L->lhs.data(), L->rhs.data(), reuse, L);
}
extern "C" rhip_pool_t* rhip_pairwise_list(rhip_ctx_t* c, size_t n_ops, const rhip_op* ops, rhip_pairlist_t* L,
                                           rhip_pool_t* reuse) {
    rhip_batch_t* b = rhip_pairwise_list_begin(c, n_ops, ops, L, reuse);
    return b ? rhip_pairwise_end(b) : nullptr;
}
extern "C" int rhip_pairwise_list_cardinality(rhip_ctx_t* c, rhip_op op, rhip_pairlist_t* L, uint64_t* out) {
    if (!L || L->free_deferred) { set_err("null (or freed) pair list"); return RHIP_ERR_ARG; }
    if (c && L->ctx != c) { set_err("the pair list belongs to another context"); return RHIP_ERR_ARG; }
    return pairwise_cardinality_impl(c, op, L->A, L->B, L->npairs, L->lhs.data(), L->rhs.data(), out, L);
}

// host-side phase clock of rhip_pairwise, microseconds accumulated since the last reset: [0] pair-list passes,
// [1] scratch sizing + host-to-device copy of the batch description, [2] planning launches, [3] class + tail launches,
// [4] wait for completion, [5] result bookkeeping
// (diagnostics for scripts/arena_skew_sweep.py, not part of the boundary)
extern "C" void rhip_debug_set_arena_skew(rhip_ctx_t* c, unsigned long long bytes) { c->arena_skew = (size_t)bytes & ~(size_t)255; }
extern "C" void rhip_debug_set_arena_round(rhip_ctx_t* c, unsigned long long bytes) { c->arena_round = (size_t)bytes; }
extern "C" unsigned long long rhip_debug_pool_arena(rhip_pool_t* P) { return (unsigned long long)(uintptr_t)P->arena.p; }
// probe rates (GB/s) of the candidates of the context's last measured arena placement, in allocation order; returns
// how many there were (0: no placement has happened)
// releases the spare result arenas a context keeps from its placement searches; returns the bytes released
extern "C" unsigned long long rhip_ctx_trim(rhip_ctx_t* c) {
    if (!c) return 0;
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    unsigned long long n = 0;
    for (DBuf& b : c->arena_spares) { n += b.cap; b.release(); }
    c->arena_spares.clear();
    n += c->release_chunk_spares();
    return n;
}
// batches of this context that a flag join gave up on and that were finished through the fallback (rhip_pairwise_end)
// diagnostics (scripts/slab_offsets.py): ONE allocation of slab_bytes, then the placement probe of a `need`-byte result
// arena at offsets 0, step, 2 step ... inside it against pool A: out[k] = GB/s at offset k * step.  Says whether a slab
// holds a fast window at all, how wide it is and whether its position repeats from process to process.
extern "C" int rhip_debug_probe_offsets(rhip_ctx_t* c, rhip_pool_t* A, unsigned long long slab_bytes, unsigned long long need,
                                        unsigned long long step, float* out, int capacity) {
    if (!c || !A || !out || need > slab_bytes || !step) return 0;
    DeviceGuard guard(c->device);
    void* slab = nullptr;
    if (hipMalloc(&slab, slab_bytes) != hipSuccess) { (void)hipGetLastError(); return -1; }
    hipStream_t s = c->stream;
    const u64 a_items = A->arena.cap / 8192ull, n_slots = need / 8192ull;
    const u64 stride = std::max<u64>(1, n_slots / ((2ull << 30) / 8192ull)), n_items = (n_slots + stride - 1) / stride;
    int n = 0;
    for (unsigned long long off = 0; off + need <= slab_bytes && n < capacity; off += step, ++n) {
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(c->ev[0], s);
            hipLaunchKernelGGL(k_place_probe, dim3(8192), dim3(256), 0, s, A->arena.as<uint8_t>(), a_items, (uint8_t*)slab + off, n_slots, stride);
            (void)hipEventRecord(c->ev[1], s);
            (void)hipEventSynchronize(c->ev[1]);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
            if (r && ms < best) best = ms;
        }
        out[n] = (float)((double)n_items * 24576.0 / (double)best / 1e6);
    }
    (void)hipFree(slab);
    return n;
}
// ... and the same probe on memory the caller owns (scripts/seq_candidates.py)
extern "C" float rhip_debug_probe_at(rhip_ctx_t* c, rhip_pool_t* A, void* mem, unsigned long long need) {
    if (!c || !A || !mem) return 0.f;
    DeviceGuard guard(c->device);
    hipStream_t s = c->stream;
    const u64 a_items = A->arena.cap / 8192ull, n_slots = need / 8192ull;
    const u64 stride = std::max<u64>(1, n_slots / ((2ull << 30) / 8192ull)), n_items = (n_slots + stride - 1) / stride;
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(c->ev[0], s);
        hipLaunchKernelGGL(k_place_probe, dim3(8192), dim3(256), 0, s, A->arena.as<uint8_t>(), a_items, (uint8_t*)mem, n_slots, stride);
        (void)hipEventRecord(c->ev[1], s);
        (void)hipEventSynchronize(c->ev[1]);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
        if (r && ms < best) best = ms;
    }
    return (float)((double)n_items * 24576.0 / (double)best / 1e6);
}
// tests: after `skip` more device allocations of the library, the next `count` fail (hipErrorOutOfMemory, no driver call)
extern "C" void rhip_debug_fail_allocs(int skip, int count) { g_fail_skip = skip; g_fail_allocs = count; }
extern "C" int rhip_debug_plan_cached(rhip_ctx_t* c) { return c && c->last_plan_cached ? 1 : 0; }
extern "C" unsigned long long rhip_debug_join_recovered(rhip_ctx_t* c) { return c ? (unsigned long long)c->join_recovered : 0ull; }
extern "C" int rhip_debug_last_placement(rhip_ctx_t* c, float* out, int capacity) {
    if (!c) return 0;
    const int n = (int)c->last_placement.size();
    for (int k = 0; k < n && k < capacity; ++k) out[k] = c->last_placement[k];
    return n;
}
extern "C" int rhip_debug_host_clock(rhip_ctx_t* c, double out[8], int reset) {
    if (!c || !out) return RHIP_ERR_ARG;
    for (int i = 0; i < 8; ++i) out[i] = c->hclk[i];
    if (reset) for (int i = 0; i < 8; ++i) c->hclk[i] = 0;
    return RHIP_OK;
}

#ifdef RHIP_PHASES
// diagnostic builds only: per-phase tick totals of k_filter ([0, 8)) and k_wave ([8, 16)); reset != 0 clears them
extern "C" int rhip_debug_phases(rhip_ctx_t* c, unsigned long long out[32], int reset) {
    if (hipStreamSynchronize(c->stream) != hipSuccess) return RHIP_ERR_DEVICE;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 32) != hipSuccess) return RHIP_ERR_DEVICE;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z) != hipSuccess) return RHIP_ERR_DEVICE;
    }
    return RHIP_OK;
}
#endif

#include "rhip_many_host.inc"
#include "rhip_heap_host.inc"
#include "rhip_sharded.inc"
#include "rhip_synth.inc"
#include "rhip_pool_ops.inc"
#include "roaring_compat.inc"
