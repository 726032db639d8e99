// TEST INFRASTRUCTURE: a stand-in for librccl whose "ranks" are THREADS of one process sharing one GPU, so that
// rhip_many_sharded (croaring_amd/csrc/rhip_sharded.inc) can be run at world sizes 2 .. 8 on the single-GPU test box: the
// real RCCL refuses two ranks on one device, and the library's multi-rank code -- owner partition, packing, send / receive
// offsets -- is otherwise only ever executed by the 8-GPU scaling run.  Loaded through RHIP_RCCL_LIB; only the entry points
// rhip_sharded.inc binds, only 8-byte elements.  Every collective waits for the caller's stream, meets the other ranks at a
// barrier, copies device to device, and meets them again: slow and simple.  Never loaded by the product.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace {
constexpr int MAXW = 16;
struct P2P { const void* p; size_t n; };
struct Group {
    int world = 0;
    pthread_barrier_t bar;
    const void* sendp[MAXW] = {};
    std::vector<P2P> sends[MAXW][MAXW];  // [from][to], in issue order
};
struct Comm { Group* g; int rank; };
struct Pending { int kind; const void* sp; void* rp; size_t n; int peer; Comm* c; hipStream_t s; };  // kind 0 send, 1 recv
thread_local int g_depth = 0;
thread_local std::vector<Pending> g_ops;
int fail(const char* what) { fprintf(stderr, "fake_rccl: %s\n", what); return 1; }
}
extern "C" {
void* fake_group_create(int world) {
    if (world < 1 || world > MAXW) return nullptr;
    Group* g = new Group();
    g->world = world;
    pthread_barrier_init(&g->bar, nullptr, (unsigned)world);
    return g;
}
void* fake_comm_create(void* group, int rank) { return new Comm{(Group*)group, rank}; }
int ncclCommCount(void* c, int* n) { *n = ((Comm*)c)->g->world; return 0; }
int ncclCommUserRank(void* c, int* r) { *r = ((Comm*)c)->rank; return 0; }
const char* ncclGetErrorString(int) { return "fake_rccl error"; }
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* cc, hipStream_t s) {
    Comm* c = (Comm*)cc; Group* g = c->g;
    if (dtype != 5) return fail("only ncclUint64");
    if (hipStreamSynchronize(s) != hipSuccess) return fail("stream sync");
    g->sendp[c->rank] = send;
    pthread_barrier_wait(&g->bar);
    for (int r = 0; r < g->world; ++r) {
        void* dst = (char*)recv + (size_t)r * count * 8;
        if (dst != g->sendp[r] && count && hipMemcpy(dst, g->sendp[r], count * 8, hipMemcpyDeviceToDevice) != hipSuccess) return fail("copy");
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail("sync");
    pthread_barrier_wait(&g->bar);
    return 0;
}
int ncclAllToAll(const void* send, void* recv, size_t count, int dtype, void* cc, hipStream_t s) {
    Comm* c = (Comm*)cc; Group* g = c->g;
    if (dtype != 5) return fail("only ncclUint64");
    if (hipStreamSynchronize(s) != hipSuccess) return fail("stream sync");
    g->sendp[c->rank] = send;
    pthread_barrier_wait(&g->bar);
    for (int r = 0; r < g->world; ++r)
        if (count && hipMemcpy((char*)recv + (size_t)r * count * 8, (const char*)g->sendp[r] + (size_t)c->rank * count * 8, count * 8,
                               hipMemcpyDeviceToDevice) != hipSuccess) return fail("copy");
    if (hipDeviceSynchronize() != hipSuccess) return fail("sync");
    pthread_barrier_wait(&g->bar);
    return 0;
}
int ncclGroupStart() { ++g_depth; return 0; }
int ncclSend(const void* p, size_t n, int dtype, int peer, void* cc, hipStream_t s) {
    if (dtype != 5 || g_depth < 1) return fail("send: ncclUint64 inside a group only");
    g_ops.push_back(Pending{0, p, nullptr, n, peer, (Comm*)cc, s});
    return 0;
}
int ncclRecv(void* p, size_t n, int dtype, int peer, void* cc, hipStream_t s) {
    if (dtype != 5 || g_depth < 1) return fail("recv: ncclUint64 inside a group only");
    g_ops.push_back(Pending{1, nullptr, p, n, peer, (Comm*)cc, s});
    return 0;
}
// (every rank of the group calls GroupEnd once per exchange, with or without operations of its own: the library does)
int ncclGroupEnd() {
    if (--g_depth > 0) return 0;
    // the communicator: from the first pending op, or -- a rank with nothing to send or receive -- there is none, and the
    // barrier cannot be reached.  rhip_sharded.inc's ranks always have a comm; it is passed through a thread-local by the
    // test driver for that case.
    extern thread_local Comm* g_self;
    Comm* c = g_ops.empty() ? g_self : g_ops[0].c;
    if (!c) return fail("GroupEnd without a communicator (fake_set_self)");
    Group* g = c->g;
    for (const Pending& o : g_ops) if (hipStreamSynchronize(o.s) != hipSuccess) return fail("stream sync");
    for (int t = 0; t < g->world; ++t) g->sends[c->rank][t].clear();
    for (const Pending& o : g_ops) if (o.kind == 0) g->sends[c->rank][o.peer].push_back(P2P{o.sp, o.n});
    pthread_barrier_wait(&g->bar);
    int taken[MAXW] = {};
    int rc = 0;
    for (const Pending& o : g_ops) {
        if (o.kind != 1) continue;
        std::vector<P2P>& from = g->sends[o.peer][c->rank];
        if (taken[o.peer] >= (int)from.size() || from[(size_t)taken[o.peer]].n != o.n) { rc = fail("recv without a matching send (count mismatch)"); break; }
        const P2P& m = from[(size_t)taken[o.peer]++];
        if (o.n && hipMemcpy(o.rp, m.p, o.n * 8, hipMemcpyDeviceToDevice) != hipSuccess) { rc = fail("copy"); break; }
    }
    for (int p = 0; p < g->world && !rc; ++p)
        if (taken[p] != (int)g->sends[p][c->rank].size()) rc = fail("a send was never received");
    if (hipDeviceSynchronize() != hipSuccess) rc = fail("sync");
    pthread_barrier_wait(&g->bar);
    g_ops.clear();
    return rc;
}
thread_local Comm* g_self = nullptr;
void fake_set_self(void* c) { g_self = (Comm*)c; }
}
