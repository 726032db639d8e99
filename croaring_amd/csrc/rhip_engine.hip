// rhip_engine.hip -- host side of libroaring_hip.so: device pools, the batched pairwise
// pipeline (plan -> typed kernels -> directory compaction), portable (de)serialization at the
// boundary, and the C ABI declared in include/roaring_hip.h.
//
// There is deliberately no CPU implementation of any set operation in this file: without a
// usable HIP device every entry point fails (rhip_ctx_create returns NULL).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/roaring_hip.h"
#include "rhip_kernels.h"
#include "rhip_many.h"
#include "rhip_poolops.h"
#include "rhip_serial.h"
#include "rhip_deser.h"
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
struct DBuf {
    void* p = nullptr;
    size_t cap = 0;
    void ensure(size_t n) {
        if (n <= cap) return;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            set_err("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            throw (int)RHIP_ERR_ALLOC;
        }
        cap = want;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
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
    PartialBuf take_partial_buf(uint64_t chunks) {
        for (size_t i = 0; i < partial_cache.size(); ++i)
            if (partial_cache[i].cap >= chunks) {
                PartialBuf b = partial_cache[i];
                partial_cache.erase(partial_cache.begin() + i);
                return b;
            }
        PartialBuf b;
        b.cap = chunks;
        if (hipMalloc(&b.keys, 8 * chunks) != hipSuccess || hipMalloc(&b.words, 8192 * chunks) != hipSuccess) {
            if (b.keys) (void)hipFree(b.keys);
            set_err("hipMalloc of partial chunks failed");
            throw (int)RHIP_ERR_ALLOC;
        }
        return b;
    }
    hipStream_t stream = nullptr;
    // scratch (grow-only): candidate directory + queues + scan temporaries
    DBuf lhs, rhs, u_pair, u_tile, u_pair0, unit_bytes, cand, cand_start, o_key, o_meta, o_slot, o_off, flag, newidx, q[N_CLS], misc, prim_tmp, pair_acc;
    DBuf many[16];
    DBuf sel[5];  // pool_select / pool_convert scratch
    void* h_pinned = nullptr;  // small pinned readback area
    rhip_stats_t stats{};
    bool timing = false;
    hipEvent_t ev[4]{};
    // independent class kernels of one batch run concurrently: fork after planning, join before compaction
    static constexpr int N_AUX = 4;
    hipStream_t aux[N_AUX]{};
    hipEvent_t ev_fork = nullptr, ev_join[N_AUX]{}, ev_runs = nullptr;
    bool overlap = true;
};

struct rhip_pool_s {
    rhip_ctx_t* ctx = nullptr;
    uint32_t n_bitmaps = 0;
    uint64_t n_cont = 0;
    bool is64 = false;
    DBuf bm_start, key, type, card, nruns, off, arena;
    uint64_t arena_used = 0;
    // host mirror of the directory (filled lazily for serialization); planning only needs bm_start
    bool host_dir = false;
    bool host_bm = false;
    std::vector<uint64_t> h_bm_start, h_key, h_off;
    std::vector<uint8_t> h_type;
    std::vector<uint32_t> h_card, h_nruns;
    std::vector<uint64_t> h_cards;  // per-bitmap cardinalities cache
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
    }
};

static void ensure_dir(rhip_pool_t* P, uint32_t n_bitmaps, uint64_t n_cont) {
    P->bm_start.ensure(8 * ((size_t)n_bitmaps + 1));
    P->key.ensure(8 * (size_t)(n_cont + 1));
    P->type.ensure((size_t)n_cont + 16);
    P->card.ensure(4 * (size_t)(n_cont + 1));
    P->nruns.ensure(4 * (size_t)(n_cont + 1));
    P->off.ensure(8 * (size_t)(n_cont + 1));
}

// ------------------------------------------------------------------ context
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
        for (auto& e : c->ev) HIPCHK(hipEventCreate(&e));
        for (auto& a : c->aux) HIPCHK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_runs, hipEventDisableTiming));
        for (auto& e : c->ev_join) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (const char* e = getenv("RHIP_NO_OVERLAP")) c->overlap = !(e[0] == '1');
        return c;
    } catch (int) {
        return nullptr;
    }
}
extern "C" void rhip_ctx_destroy(rhip_ctx_t* c) {
    if (!c) return;
    int prev_dev_ = -1;
    const bool sw_ = hipGetDevice(&prev_dev_) == hipSuccess && prev_dev_ != c->device && hipSetDevice(c->device) == hipSuccess;
    (void)hipStreamSynchronize(c->stream);
    DBuf* all[] = {&c->lhs, &c->rhs, &c->u_pair, &c->u_tile, &c->u_pair0, &c->unit_bytes, &c->cand, &c->cand_start, &c->o_key, &c->o_meta,
                   &c->o_slot, &c->o_off, &c->flag, &c->newidx, &c->misc, &c->prim_tmp, &c->pair_acc};
    for (auto* b : all) b->release();
    for (auto& b : c->q) b.release();
    for (auto& b : c->many) b.release();
    for (auto& b : c->sel) b.release();
    for (auto& b : c->partial_cache) { (void)hipFree(b.keys); (void)hipFree(b.words); }
    for (auto& e : c->ev) (void)hipEventDestroy(e);
    for (auto& a : c->aux) if (a) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_runs) (void)hipEventDestroy(c->ev_runs);
    for (auto& e : c->ev_join) if (e) (void)hipEventDestroy(e);
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
    return RHIP_OK;
}
extern "C" void rhip_ctx_set_timing(rhip_ctx_t* c, int enabled) { c->timing = enabled != 0; }
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

rhip_pool_t* upload(rhip_ctx_t* ctx, HostDir& D, uint32_t n_bitmaps, bool is64) {
    rhip_pool_t* P = new rhip_pool_s();
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
        HIPCHK(hipMemcpyAsync(P->bm_start.p, D.bm_start.data(), 8 * D.bm_start.size(), hipMemcpyHostToDevice, s));
        if (P->n_cont) {
            HIPCHK(hipMemcpyAsync(P->key.p, D.key.data(), 8 * D.key.size(), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(P->type.p, D.type.data(), D.type.size(), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(P->card.p, D.card.data(), 4 * D.card.size(), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(P->nruns.p, D.nruns.data(), 4 * D.nruns.size(), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(P->off.p, D.off.data(), 8 * D.off.size(), hipMemcpyHostToDevice, s));
        }
        HIPCHK(hipMemcpyAsync(P->arena.p, D.arena.data(), D.arena.size(), hipMemcpyHostToDevice, s));
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
    P->release();  // hipFree synchronises with the device; the context may already be gone
    delete P;
}
extern "C" uint32_t rhip_pool_size(const rhip_pool_t* P) { return P->n_bitmaps; }
extern "C" uint64_t rhip_pool_containers(const rhip_pool_t* P) { return P->n_cont; }
extern "C" int rhip_pool_is64(const rhip_pool_t* P) { return P->is64 ? 1 : 0; }

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
        P->arena.ensure(P->arena_used);
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
static void fetch_dir(rhip_pool_t* P) {
    if (P->host_dir) return;
    hipStream_t s = P->ctx->stream;
    P->h_bm_start.resize((size_t)P->n_bitmaps + 1);
    P->h_key.resize(P->n_cont); P->h_off.resize(P->n_cont); P->h_type.resize(P->n_cont);
    P->h_card.resize(P->n_cont); P->h_nruns.resize(P->n_cont);
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
struct PlanResult {
    uint64_t total_cand = 0;
    uint64_t total_bytes = 0;
    uint64_t n_bb = 0, n_gen = 0, n_copy = 0, n_filt = 0, n_wave = 0, n_runs = 0;
};

// misc layout (device): [0, 128) eight {begin,end} u64 section ranges; [128, 132) retry counter;
// [192, ...) Stats
constexpr size_t MISC_RANGES_OFF = 0;
constexpr size_t MISC_RETRY_OFF = 128;
constexpr size_t MISC_STATS_OFF = 192;

__global__ void k_plan_totals(const u64* __restrict__ starts, u64 S, u64* __restrict__ ranges) {
    const uint32_t k = threadIdx.x;
    if (k < N_SEC) {
        ranges[2 * k] = starts[k * S];
        ranges[2 * k + 1] = starts[k * S + (S - 1)];
    }
}

template <int OP>
void launch_bb(rhip_ctx_t* c, unsigned grid, const PoolView& A, const PoolView& B, const OutView& O, int cardmode) {
    hipLaunchKernelGGL(k_bb<OP>, dim3(grid), dim3(256), 0, c->stream, A.arena, B.arena, O, c->q[CLS_BB].as<BBItem>(),
                       (const u64*)((char*)c->misc.p + MISC_RANGES_OFF) + 2 * SEC_BB, cardmode, c->pair_acc.as<u64>(),
                       c->q[CLS_RETRY].as<GenItem>(), (uint32_t*)((char*)c->misc.p + MISC_RETRY_OFF));
}

void check_pair_args(rhip_pool_t* A, rhip_pool_t* B, size_t npairs, const uint32_t* lhs, const uint32_t* rhs) {
    if (!A || !B) { set_err("null pool"); throw (int)RHIP_ERR_ARG; }
    if (npairs && (!lhs || !rhs)) { set_err("null pair index array"); throw (int)RHIP_ERR_ARG; }
    if (A->is64 != B->is64) { set_err("mixing 32-bit and 64-bit pools"); throw (int)RHIP_ERR_ARG; }
    if (A->ctx->device != B->ctx->device) { set_err("operand pools live on different devices"); throw (int)RHIP_ERR_ARG; }
    if (npairs >= 0x3FFFFFF0ull) { set_err("too many pairs"); throw (int)RHIP_ERR_ARG; }
    for (size_t i = 0; i < npairs; ++i)
        if (lhs[i] >= A->n_bitmaps || rhs[i] >= B->n_bitmaps) {
            set_err("pair %zu: bitmap index out of range", i);
            throw (int)RHIP_ERR_ARG;
        }
}

// count -> one scan -> emit -> slot scan.  On return the class queues are filled (at
// deterministic positions) and the host knows the totals (ONE small readback).
PlanResult plan(rhip_ctx_t* c, int op, rhip_pool_t* A, rhip_pool_t* B, size_t npairs, const uint32_t* lhs,
                const uint32_t* rhs, int cardmode, OutView& O) {
    hipStream_t s = c->stream;
    PlanResult R;
    // ---- units (host: directory mirrors) + upper bounds, no sync needed
    fetch_bm_start(A);
    fetch_bm_start(B);
    const bool btiles = !cardmode && (op == OP_OR || op == OP_XOR);
    std::vector<uint32_t> upair, utile;
    std::vector<uint64_t> pair0(npairs + 1);
    uint64_t ub_match = 0, ub = 0;
    for (size_t i = 0; i < npairs; ++i) {
        const uint64_t nA = A->h_bm_start[lhs[i] + 1] - A->h_bm_start[lhs[i]];
        const uint64_t nB = B->h_bm_start[rhs[i] + 1] - B->h_bm_start[rhs[i]];
        pair0[i] = upair.size();
        for (uint64_t t = 0; t < (nA + 255) / 256; ++t) { upair.push_back((uint32_t)i); utile.push_back((uint32_t)t); }
        if (btiles)
            for (uint64_t t = 0; t < (nB + 255) / 256; ++t) { upair.push_back((uint32_t)i); utile.push_back((uint32_t)t | UNIT_B); }
        ub_match += std::min(nA, nB);
        if (cardmode || op == OP_AND) ub += std::min(nA, nB);
        else if (op == OP_ANDNOT) ub += nA;
        else ub += nA + nB;
    }
    const size_t NU = upair.size();
    pair0[npairs] = NU;
    if (ub >= 0xFFFFFFF0ull || NU >= 0x7FFFFFF0ull) { set_err("batch too large: %llu candidate containers", (unsigned long long)ub); throw (int)RHIP_ERR_ARG; }
    const size_t S = NU + 1;
    c->lhs.ensure(4 * (npairs + 1));
    c->rhs.ensure(4 * (npairs + 1));
    c->u_pair.ensure(4 * S); c->u_tile.ensure(4 * S); c->u_pair0.ensure(8 * (npairs + 1));
    c->cand.ensure(4 * (N_SEC * S + 1));
    c->cand_start.ensure(8 * (N_SEC * S + 1));
    c->misc.ensure(512);
    HIPCHK(hipMemcpyAsync(c->lhs.p, lhs, 4 * npairs, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->rhs.p, rhs, 4 * npairs, hipMemcpyHostToDevice, s));
    if (NU) {
        HIPCHK(hipMemcpyAsync(c->u_pair.p, upair.data(), 4 * NU, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(c->u_tile.p, utile.data(), 4 * NU, hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipMemcpyAsync(c->u_pair0.p, pair0.data(), 8 * (npairs + 1), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(c->misc.p, 0, 512, s));
    HIPCHK(hipMemsetAsync(c->cand.p, 0, 4 * (N_SEC * S + 1), s));
    PoolView VA = A->view(), VB = B->view();
    UnitView UV{c->u_pair.as<uint32_t>(), c->u_tile.as<uint32_t>(), c->u_pair0.as<u64>(), (uint32_t)NU};
    unsigned gp = (unsigned)std::max<size_t>(1, (NU * 64 + 255) / 256);
    hipLaunchKernelGGL(k_count, dim3(gp), dim3(256), 0, s, VA, VB, c->lhs.as<uint32_t>(), c->rhs.as<uint32_t>(), UV, op,
                       cardmode, c->cand.as<uint32_t>());
    exscan(c, c->cand.as<uint32_t>(), c->cand_start.as<u64>(), N_SEC * S - 1);
    u64* ranges = (u64*)((char*)c->misc.p + MISC_RANGES_OFF);
    hipLaunchKernelGGL(k_plan_totals, dim3(1), dim3(64), 0, s, c->cand_start.as<u64>(), (u64)S, ranges);
    c->q[CLS_BB].ensure(sizeof(BBItem) * (ub_match + 1));
    c->q[CLS_GEN].ensure(sizeof(GenItem) * (ub_match + 1));
    c->q[CLS_FILT].ensure(sizeof(FatItem) * (ub_match + 1));
    c->q[CLS_WAVE].ensure(sizeof(FatItem) * (ub_match + 1));
    c->q[CLS_RUNS].ensure(sizeof(GenItem) * (ub_match + 1));
    c->unit_bytes.ensure(8 * (NU + 1));
    c->q[CLS_COPY].ensure(sizeof(Item) * (ub + 1));
    if (!cardmode) {
        c->o_key.ensure(8 * (ub + 1)); c->o_meta.ensure(8 * (ub + 1));
        c->o_slot.ensure(4 * (ub + 2)); c->o_off.ensure(8 * (ub + 2));
        HIPCHK(hipMemsetAsync(c->o_slot.p, 0, 4 * (ub + 2), s));
    }
    O.key = c->o_key.as<u64>(); O.meta = c->o_meta.as<u64>();
    O.slot = c->o_slot.as<uint32_t>(); O.off = c->o_off.as<u64>();
    O.arena = nullptr;
    EmitQueues Q{c->q[CLS_BB].as<BBItem>(), c->q[CLS_GEN].as<GenItem>(), c->q[CLS_COPY].as<Item>(), c->q[CLS_FILT].as<FatItem>(), c->q[CLS_WAVE].as<FatItem>(), c->q[CLS_RUNS].as<GenItem>()};
    hipLaunchKernelGGL(k_emit, dim3(gp), dim3(256), 0, s, VA, VB, c->lhs.as<uint32_t>(), c->rhs.as<uint32_t>(), UV, op,
                       cardmode, c->cand_start.as<u64>(), O, Q, c->unit_bytes.as<u64>());
    hipLaunchKernelGGL(k_sum_u64, dim3(1), dim3(1024), 0, s, c->unit_bytes.as<u64>(), (u64)NU,
                       &((Stats*)((char*)c->misc.p + MISC_STATS_OFF))->bytes_in);
    char* hp = (char*)c->h_pinned;
    if (!cardmode) {
        // slots beyond the exact candidate count were zeroed, so scanning the upper bound is exact
        exscan(c, O.slot, c->o_off.as<u64>(), ub);
        HIPCHK(hipMemcpyAsync(hp + 160, c->o_off.as<u64>() + ub, 8, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipMemcpyAsync(hp, ranges, 128, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    uint64_t r[16];
    memcpy(r, hp, 128);
    R.total_cand = r[2 * SEC_CAND + 1] - r[2 * SEC_CAND];
    R.n_bb = r[2 * SEC_BB + 1] - r[2 * SEC_BB];
    R.n_gen = r[2 * SEC_GEN + 1] - r[2 * SEC_GEN];
    R.n_copy = r[2 * SEC_COPY + 1] - r[2 * SEC_COPY];
    R.n_filt = r[2 * SEC_FILT + 1] - r[2 * SEC_FILT];
    R.n_wave = r[2 * SEC_WAVE + 1] - r[2 * SEC_WAVE];
    R.n_runs = r[2 * SEC_RUNS + 1] - r[2 * SEC_RUNS];
    if (!cardmode) memcpy(&R.total_bytes, hp + 160, 8);
    return R;
}

unsigned persistent_grid(uint64_t n_items, unsigned items_per_block, unsigned max_blocks) {
    uint64_t need = (n_items + items_per_block - 1) / items_per_block;
    if (need < 1) need = 1;
    return (unsigned)std::min<uint64_t>(need, max_blocks);
}

// The class kernels of one batch are independent of each other (disjoint work queues, disjoint result slots;
// pair_acc and the retry counter are only touched with atomics), except that the retry pass of k_genw consumes what
// k_bb and k_runs re-queue.  When more than one class has work they are forked onto auxiliary streams after
// planning and joined before compaction, so the latency-bound wave-per-pair kernels overlap each other and the
// bandwidth-bound copies:
//     main : k_bb ----------------------> [wait k_runs] k_genw(retry) -> [join] ...
//     aux0 : k_runs
//     aux1 : k_filter      aux2 : k_wave      aux3 : k_genw(general), k_copy
// A batch with a single class (the bitset-only C2 workload) stays on the main stream with no events at all.
void run_kernels(rhip_ctx_t* c, int op, const PoolView& VA, const PoolView& VB, const OutView& O,
                 const PlanResult& R, int cardmode) {
    hipStream_t s = c->stream;
    const u64* ranges = (const u64*)((char*)c->misc.p + MISC_RANGES_OFF);
    uint32_t* retry_count = (uint32_t*)((char*)c->misc.p + MISC_RETRY_OFF);
    c->q[CLS_RETRY].ensure(sizeof(GenItem) * (R.n_bb + R.n_runs + 1));
    const bool has_wave = R.n_wave && !cardmode, has_copy = R.n_copy && !cardmode;
    const int n_classes = (R.n_bb != 0) + (R.n_runs != 0) + (R.n_filt != 0) + (has_wave ? 1 : 0) + (R.n_gen != 0) +
                          (has_copy ? 1 : 0);
    const bool fork = c->overlap && n_classes > 1;
    bool used[rhip_ctx_s::N_AUX] = {false, false, false, false};
    auto on = [&](int a) -> hipStream_t {
        if (!fork) return s;
        if (!used[a]) {
            HIPCHK(hipStreamWaitEvent(c->aux[a], c->ev_fork, 0));
            used[a] = true;
        }
        return c->aux[a];
    };
    if (fork) HIPCHK(hipEventRecord(c->ev_fork, s));
    // largest, latency-bound kernels first so that they get the machine's first workgroup slots
    if (R.n_filt) {
        unsigned grid = persistent_grid(R.n_filt, 4, 256 * 4);
        hipLaunchKernelGGL(k_filter, dim3(grid), dim3(256), 0, on(1), VA.arena, VB.arena, O, c->q[CLS_FILT].as<FatItem>(),
                           ranges + 2 * SEC_FILT, op, cardmode, c->pair_acc.as<u64>());
    }
    if (has_wave) {
        unsigned grid = persistent_grid(R.n_wave, 4, 256 * 4);
        hipLaunchKernelGGL(k_wave, dim3(grid), dim3(256), 0, on(2), VA.arena, VB.arena, O, c->q[CLS_WAVE].as<FatItem>(),
                           ranges + 2 * SEC_WAVE, op);
    }
    if (R.n_runs) {
        unsigned grid = persistent_grid(R.n_runs, 4, 256 * 4);
        hipStream_t sr = on(0);
        hipLaunchKernelGGL(k_runs, dim3(grid), dim3(256), 0, sr, VA.arena, VB.arena, O, c->q[CLS_RUNS].as<GenItem>(),
                           ranges + 2 * SEC_RUNS, op, cardmode, c->pair_acc.as<u64>(), c->q[CLS_RETRY].as<GenItem>(),
                           retry_count);
        if (fork) HIPCHK(hipEventRecord(c->ev_runs, sr));
    }
    if (R.n_gen) {
        unsigned grid = persistent_grid(R.n_gen, 4, 256 * 2);  // 248 VGPRs: 2 workgroups resident per CU
        hipLaunchKernelGGL(k_genw, dim3(grid), dim3(256), 0, on(3), VA.arena, VB.arena, O, c->q[CLS_GEN].as<GenItem>(),
                           ranges + 2 * SEC_GEN, (const uint32_t*)nullptr, op, cardmode, c->pair_acc.as<u64>());
    }
    if (has_copy) {
        unsigned grid = persistent_grid(R.n_copy, 4, 256 * 8);
        hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, on(3), VA, VB, O, c->q[CLS_COPY].as<Item>(),
                           ranges + 2 * SEC_COPY);
    }
    if (R.n_bb) {
        unsigned grid = persistent_grid(R.n_bb, 4, 256 * 32);
        if (c->timing) HIPCHK(hipEventRecord(c->ev[2], s));
        switch (op) {
            case OP_AND: launch_bb<OP_AND>(c, grid, VA, VB, O, cardmode); break;
            case OP_OR: launch_bb<OP_OR>(c, grid, VA, VB, O, cardmode); break;
            case OP_XOR: launch_bb<OP_XOR>(c, grid, VA, VB, O, cardmode); break;
            default: launch_bb<OP_ANDNOT>(c, grid, VA, VB, O, cardmode); break;
        }
        if (c->timing) HIPCHK(hipEventRecord(c->ev[3], s));
    }
    if (!cardmode && ((R.n_bb && op != OP_OR) || R.n_runs)) {
        // results that need the LDS image path after all: bitset x bitset results that must become
        // arrays (card <= 4096), interval results that must become bitsets
        if (fork && R.n_runs) HIPCHK(hipStreamWaitEvent(s, c->ev_runs, 0));
        unsigned g2 = persistent_grid(R.n_bb + R.n_runs, 4, 256 * 2);  // 248 VGPRs: 2 workgroups resident per CU
        hipLaunchKernelGGL(k_genw, dim3(g2), dim3(256), 0, s, VA.arena, VB.arena, O, c->q[CLS_RETRY].as<GenItem>(),
                           (const u64*)nullptr, retry_count, op, 0, c->pair_acc.as<u64>());
    }
    if (fork)
        for (int a = 0; a < rhip_ctx_s::N_AUX; ++a)
            if (used[a]) {
                HIPCHK(hipEventRecord(c->ev_join[a], c->aux[a]));
                HIPCHK(hipStreamWaitEvent(s, c->ev_join[a], 0));
            }
}

void finish_stats(rhip_ctx_t* c, const PlanResult* R) {
    hipStream_t s = c->stream;
    Stats st;
    HIPCHK(hipMemcpyAsync(c->h_pinned, (char*)c->misc.p + MISC_STATS_OFF, sizeof(Stats), hipMemcpyDeviceToHost, s));
    if (c->timing) HIPCHK(hipEventRecord(c->ev[1], s));
    HIPCHK(hipStreamSynchronize(s));
    memcpy(&st, c->h_pinned, sizeof(Stats));
    c->stats.matched_pairs = R ? R->n_bb + R->n_gen + R->n_filt + R->n_wave + R->n_runs : 0;
    c->stats.passthrough = R ? R->n_copy : 0;
    c->stats.bytes_in = st.bytes_in;
    c->stats.bytes_out = st.bytes_out;
    c->stats.n_bitset_pairs = R ? R->n_bb : 0;
    c->stats.result_containers = st.result_containers;
    c->stats.ms_bitset_kernel = 0.f;
    c->stats.ms_total = 0.f;
    if (c->timing) {
        (void)hipEventElapsedTime(&c->stats.ms_total, c->ev[0], c->ev[1]);
        if (R && R->n_bb) (void)hipEventElapsedTime(&c->stats.ms_bitset_kernel, c->ev[2], c->ev[3]);
    }
}
}  // namespace

extern "C" rhip_pool_t* rhip_pairwise(rhip_ctx_t* c, rhip_op op_, rhip_pool_t* A, rhip_pool_t* B, size_t npairs,
                                      const uint32_t* lhs, const uint32_t* rhs, rhip_pool_t* reuse) {
    rhip_pool_t* R = nullptr;
    try {
        if (!c) { set_err("null context"); throw (int)RHIP_ERR_ARG; }
        int op = (int)op_;
        if (op < 0 || op > 3) { set_err("bad op"); throw (int)RHIP_ERR_ARG; }
        DeviceGuard dguard_(c->device);
        check_pair_args(A, B, npairs, lhs, rhs);
        if (reuse && (reuse == A || reuse == B)) {
            reuse = nullptr;  // not ours to recycle
            set_err("`reuse` must not be one of the operand pools");
            throw (int)RHIP_ERR_ARG;
        }
        hipStream_t s = c->stream;
        if (c->timing) HIPCHK(hipEventRecord(c->ev[0], s));
        OutView O;
        PlanResult P = plan(c, op, A, B, npairs, lhs, rhs, 0, O);
        R = reuse ? reuse : new rhip_pool_s();
        reuse = nullptr;
        R->ctx = c;
        R->n_bitmaps = (uint32_t)npairs;
        R->is64 = A->is64;
        R->host_dir = false;
        R->host_bm = false;
        R->h_cards.clear();
        ensure_dir(R, (uint32_t)npairs, P.total_cand);
        R->arena.ensure(P.total_bytes + 64);
        R->arena_used = P.total_bytes + 64;
        O.arena = R->arena.as<uint8_t>();
        PoolView VA = A->view(), VB = B->view();
        run_kernels(c, op, VA, VB, O, P, 0);
        // drop empty results, build the result directory
        uint64_t n = P.total_cand;
        c->flag.ensure(4 * (n + 2));
        c->newidx.ensure(8 * (n + 2));
        if (n) hipLaunchKernelGGL(k_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, O.meta, (u64)n,
                                  c->flag.as<uint32_t>());
        exscan(c, c->flag.as<uint32_t>(), c->newidx.as<u64>(), n);
        DirOut D{R->bm_start.as<u64>(), R->key.as<u64>(), R->type.as<uint8_t>(), R->card.as<uint32_t>(),
                 R->nruns.as<uint32_t>(), R->off.as<u64>()};
        Stats* st = (Stats*)((char*)c->misc.p + MISC_STATS_OFF);
        if (n) hipLaunchKernelGGL(k_compact, dim3((unsigned)std::min<uint64_t>((n + 4095) / 4096, 1024)), dim3(1024), 0, s,
                                  O, (u64)n, c->newidx.as<u64>(), D, st);
        hipLaunchKernelGGL(k_bm_start, dim3((unsigned)((npairs + 1 + 255) / 256)), dim3(256), 0, s,
                           c->cand_start.as<u64>(), c->u_pair0.as<u64>(), (uint32_t)npairs, c->newidx.as<u64>(),
                           R->bm_start.as<u64>());
        HIPCHK(hipMemcpyAsync((char*)c->h_pinned + 512, c->newidx.as<u64>() + n, 8, hipMemcpyDeviceToHost, s));
        finish_stats(c, &P);
        memcpy(&R->n_cont, (char*)c->h_pinned + 512, 8);
        return R;
    } catch (int e) {
        last_status() = e;
        if (R) { R->release(); delete R; }
        if (reuse) { reuse->release(); delete reuse; }
        return nullptr;
    }
}

extern "C" int rhip_pairwise_cardinality(rhip_ctx_t* c, rhip_op op_, rhip_pool_t* A, rhip_pool_t* B, size_t npairs,
                                         const uint32_t* lhs, const uint32_t* rhs, uint64_t* out) {
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
        c->pair_acc.ensure(8 * (npairs + 1));
        HIPCHK(hipMemsetAsync(c->pair_acc.p, 0, 8 * (npairs + 1), s));
        OutView O;
        PlanResult P = plan(c, OP_AND, A, B, npairs, lhs, rhs, 1, O);
        PoolView VA = A->view(), VB = B->view();
        run_kernels(c, OP_AND, VA, VB, O, P, 1);
        HIPCHK(hipMemcpyAsync(out, c->pair_acc.p, 8 * npairs, hipMemcpyDeviceToHost, s));
        finish_stats(c, &P);
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
#include "rhip_synth.inc"
#include "rhip_pool_ops.inc"
#include "roaring_compat.inc"
