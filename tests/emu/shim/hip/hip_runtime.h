// hipemu -- a single-threaded wave64 SIMT emulator, just large enough to execute the kernels of
// croaring_amd/csrc on a host CPU.  TEST INFRASTRUCTURE ONLY (same status as oracle/): it exists so
// the kernel *logic* (indexing, LDS protocols, wave collectives, result typing) can be exercised by
// `pytest -m "not gpu"` in a container without a GPU.  Nothing in the croaring_amd package loads,
// links or references it; the product library has no CPU path.
//
// Model: one block at a time; every thread of the block is a fiber (hand-rolled x86-64 context
// switch); fibers run until they reach a block barrier or a wave collective.  A wave collective
// (__shfl*, __ballot, wave_barrier, ...) completes when every lane of the wave that has not exited
// has arrived at it, and all arriving lanes must come from the same call site, otherwise the run
// aborts ("divergent collective") -- the kernels promise wave-uniform control flow around
// collectives and the emulator holds them to it.  Lanes of a wave are NOT otherwise in lockstep, so
// cross-lane LDS traffic that is not ordered by a collective or barrier shows up as a wrong result
// (with HIPEMU_SHUFFLE=seed the lane order inside a wave is permuted to shake such races out).
// Device memory from hipMalloc is poisoned (0xA5) so reliance on zeroed memory is caught.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) uint4 {
    unsigned x, y, z, w;
};
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct alignas(8) uint2 {
    unsigned x, y;
};
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

// ---------------------------------------------------------------- runtime API subset
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
typedef struct hipemu_stream* hipStream_t;
struct hipemu_event {
    std::chrono::steady_clock::time_point t;
};
typedef hipemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocMapped = 2 };

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) {
    void* q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 1)) return hipErrorOutOfMemory;
    memset(q, 0xA5, n);
    *p = q;
    return hipSuccess;
}
template <class T>
static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
// virtual-memory API (place_arena_va): an address range is a PROT_NONE mapping, physical memory a memfd, hipMemMap a shared
// MAP_FIXED mapping of it inside the range -- the same bytes at whatever address they are mapped, as on the device
enum hipMemAllocationType { hipMemAllocationTypePinned = 1 };
enum hipMemLocationType { hipMemLocationTypeDevice = 1 };
enum hipMemAccessFlags { hipMemAccessFlagsProtReadWrite = 3 };
struct hipMemLocation { hipMemLocationType type; int id; };
struct hipMemAllocationProp { hipMemAllocationType type; int requestedHandleType; hipMemLocation location; void* win32HandleMetaData; };
struct hipMemAccessDesc { hipMemLocation location; hipMemAccessFlags flags; };
struct hipemu_vmm_handle { int fd; size_t len; };
typedef hipemu_vmm_handle* hipMemGenericAllocationHandle_t;
static inline hipError_t hipMemCreate(hipMemGenericAllocationHandle_t* h, size_t n, const hipMemAllocationProp*, unsigned long long) {
    if (getenv("HIPEMU_NO_VMM")) return hipErrorInvalidValue;
    const int fd = memfd_create("hipemu_vmm", 0);
    if (fd < 0) return hipErrorOutOfMemory;
    if (ftruncate(fd, (off_t)n) != 0) { close(fd); return hipErrorOutOfMemory; }
    *h = new hipemu_vmm_handle{fd, n};
    return hipSuccess;
}
static inline hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) { close(h->fd); delete h; return hipSuccess; }
static inline hipError_t hipMemAddressReserve(void** p, size_t n, size_t, void*, unsigned long long) {
    void* q = mmap(nullptr, n, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (q == MAP_FAILED) return hipErrorOutOfMemory;
    *p = q;
    return hipSuccess;
}
static inline hipError_t hipMemAddressFree(void* p, size_t n) { return munmap(p, n) == 0 ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipMemMap(void* at, size_t n, size_t off, hipMemGenericAllocationHandle_t h, unsigned long long) {
    if (n > h->len) return hipErrorInvalidValue;
    return mmap(at, n, PROT_NONE, MAP_SHARED | MAP_FIXED, h->fd, (off_t)off) == at ? hipSuccess : hipErrorInvalidValue;
}
static inline hipError_t hipMemSetAccess(void* at, size_t n, const hipMemAccessDesc*, size_t) { return mprotect(at, n, PROT_READ | PROT_WRITE) == 0 ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipMemUnmap(void* at, size_t n) {
    return mmap(at, n, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0) == at ? hipSuccess : hipErrorInvalidValue;
}
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = (size_t)1 << 40; *total_b = (size_t)1 << 40; return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T>
static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event(); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemu_event(); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}

// ---------------------------------------------------------------- SIMT core (hipemu_core.cpp)
namespace hipemu {
struct WaveBuf {
    uint64_t vals[64];
    uint64_t present;
    int site;
};
struct Wave {
    WaveBuf buf[2];
    unsigned gen, arrived, live;
};
struct Fiber {
    void* sp;
    dim3 tid;
    unsigned lane;
    Wave* wave;
    int status;  // 0 runnable, 1 waiting on the wave, 2 waiting on the block, 3 done
    unsigned wait_gen;
    char* stack;
};
extern Fiber* cur;
extern dim3 cur_block, cur_bdim, cur_gdim;
// arrive at a wave collective with value v; returns the buffer holding every lane's value
const WaveBuf& wave_sync(int site, uint64_t v);
void block_sync();
void run_grid(dim3 grid, dim3 block, void (*fn)(void*), void* arg);

template <class F>
static void thunk(void* p) { (*(F*)p)(); }
// kernel arguments are converted to the parameter types once, then every thread gets a copy
template <class... P, class... A>
static inline void launch_k(dim3 g, dim3 b, void (*k)(P...), A&&... a) {
    std::tuple<P...> t{P(a)...};
    auto f = [&]() { std::apply(k, t); };
    run_grid(g, b, &thunk<decltype(f)>, &f);
}

template <class T>
static inline uint64_t pack(T v) {
    static_assert(sizeof(T) <= 8, "shuffle operand too wide");
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    return u;
}
template <class T>
static inline T unpack(uint64_t u) {
    T v;
    memcpy(&v, &u, sizeof(T));
    return v;
}
template <class T>
static inline T shfl(int site, T v, int src) {
    const WaveBuf& b = wave_sync(site, pack(v));
    src &= 63;
    return ((b.present >> src) & 1) ? unpack<T>(b.vals[src]) : v;
}
template <class T>
static inline T shfl_up(int site, T v, unsigned d) {
    const unsigned lane = cur->lane;
    const WaveBuf& b = wave_sync(site, pack(v));
    return (lane >= d && ((b.present >> (lane - d)) & 1)) ? unpack<T>(b.vals[lane - d]) : v;
}
template <class T>
static inline T shfl_down(int site, T v, unsigned d) {
    const unsigned lane = cur->lane;
    const WaveBuf& b = wave_sync(site, pack(v));
    return (lane + d < 64 && ((b.present >> (lane + d)) & 1)) ? unpack<T>(b.vals[lane + d]) : v;
}
template <class T>
static inline T shfl_xor(int site, T v, unsigned m) {
    const unsigned lane = cur->lane;
    const WaveBuf& b = wave_sync(site, pack(v));
    const unsigned s = (lane ^ m) & 63;
    return ((b.present >> s) & 1) ? unpack<T>(b.vals[s]) : v;
}
// v_mov_b32 with a DPP control (__builtin_amdgcn_update_dpp): the controls the kernels use -- row_shr:n (0x110 + n),
// row_bcast:15 (0x142), row_bcast:31 (0x143) -- with row_mask, bound_ctrl and `old` as the ISA defines them (a lane whose
// row is masked out, or whose source lane is invalid or absent without bound_ctrl, keeps `old`; with bound_ctrl it reads 0)
static inline int update_dpp(int site, int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const unsigned lane = cur->lane;
    const WaveBuf& b = wave_sync(site, pack(src));
    const unsigned row = lane >> 4, rl = lane & 15u;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (rl >> 2)) & 1)) return old;
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11F) {
        const unsigned n = (unsigned)ctrl - 0x110u;
        if (rl >= n) from = (int)(lane - n);
    } else if (ctrl == 0x142) {
        if (row >= 1) from = (int)(16u * row - 1u);
    } else if (ctrl == 0x143) {
        if (row >= 2) from = 31;
    } else {
        fprintf(stderr, "hipemu: DPP control 0x%x not modelled (site %d)\n", ctrl, site);
        abort();
    }
    if (from < 0 || !((b.present >> from) & 1)) return bound_ctrl ? 0 : old;
    return unpack<int>(b.vals[from]);
}
static inline uint64_t ballot(int site, int pred) {
    const WaveBuf& b = wave_sync(site, pred ? 1 : 0);
    uint64_t m = 0;
    for (int i = 0; i < 64; ++i)
        if (((b.present >> i) & 1) && b.vals[i]) m |= 1ull << i;
    return m;
}
static inline unsigned mbcnt(unsigned mask, unsigned add, int hi) {
    const unsigned lane = cur->lane;
    unsigned below;
    if (!hi)
        below = lane >= 32 ? 0xffffffffu : ((1u << lane) - 1u);
    else
        below = lane <= 32 ? 0u : ((1u << (lane - 32)) - 1u);
    return add + (unsigned)__builtin_popcount(mask & below);
}
}  // namespace hipemu

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::cur_block)
#define blockDim (hipemu::cur_bdim)
#define gridDim (hipemu::cur_gdim)
#define warpSize 64

#define HIPEMU_SITE ((int)(__LINE__ * 131 + sizeof(__FILE__)))
#define __syncthreads() hipemu::block_sync()
#define __threadfence_system() ((void)0)
#define __threadfence() ((void)0)  /* blocks of the emulator run one after another: every write is visible */
#define __shfl(v, src, ...) hipemu::shfl(HIPEMU_SITE, (v), (src))
#define __shfl_up(v, d, ...) hipemu::shfl_up(HIPEMU_SITE, (v), (d))
#define __shfl_down(v, d, ...) hipemu::shfl_down(HIPEMU_SITE, (v), (d))
#define __shfl_xor(v, m, ...) hipemu::shfl_xor(HIPEMU_SITE, (v), (m))
#define __ballot(p) hipemu::ballot(HIPEMU_SITE, (p))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu::update_dpp(HIPEMU_SITE + (ctrl), (old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_readlane(v, l) hipemu::shfl(HIPEMU_SITE, (v), (l))
#define __builtin_amdgcn_wave_barrier() ((void)hipemu::wave_sync(-1, 0))
#define __builtin_amdgcn_mbcnt_lo(m, a) hipemu::mbcnt((m), (a), 0)
#define __builtin_amdgcn_mbcnt_hi(m, a) hipemu::mbcnt((m), (a), 1)
// (only ever applied to values the whole wave shares: the emulated lanes each keep their own copy)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_sleep(n) ((void)0)

static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

// one OS thread executes everything, so the "atomics" are plain read-modify-writes
template <class T, class U>
static inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U>
static inline T atomicSub(T* p, U v) { T o = *p; *p = (T)(o - (T)v); return o; }
template <class T, class U>
static inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U>
static inline T atomicXor(T* p, U v) { T o = *p; *p = (T)(o ^ (T)v); return o; }
template <class T, class U>
static inline T atomicAnd(T* p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class U>
static inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U, class V>
static inline T atomicCAS(T* p, U cmp, V v) { T o = *p; if (o == (T)cmp) *p = (T)v; return o; }
template <class T, class U>
static inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch_k((grid), (block), kernel, ##__VA_ARGS__)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
