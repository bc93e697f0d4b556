// TEST INFRASTRUCTURE: a CAPTURABLE in-process stand-in for the RCCL entry points libmg_hip.so uses.  Unlike
// fake_rccl.cpp (host rendezvous + cross-thread events, which cannot be recorded into a hipGraph), point-to-point
// operations here are nothing but kernel launches on the caller's stream: every ordered pair of ranks has a channel in
// device memory and a message is a device-side handshake
//     sender:   k_post (publish the source pointer, ready = ++posted)           ... k_wait_consumed (spin)
//     receiver: k_wait_ready (spin) -> k_copy (many blocks) -> k_done (consumed = ++received)
// so a whole slab V-cycle -- kernels and exchanges on two streams -- can be captured and replayed, which is what the
// library's "graph_comm" path does with the real RCCL.  "Ranks" are threads of one process sharing one GPU.  Spins are
// bounded (15 s, then an error word is set and the kernel gives up: wrong data, never a hang).  Collectives
// that the V-cycle itself never issues (all-reduce of norms) synchronise on the host.  Nothing here is shipped.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace {

struct Channel {
    unsigned long long ready;       // messages published by the source rank
    unsigned long long consumed;    // messages copied by the destination rank
    const void* ptr;                // source buffer of the message in flight
    unsigned long long bytes;
    unsigned long long posted;      // the source's own count (device-resident: replays advance it)
    unsigned long long received;    // the destination's own count
    unsigned long long error;
    unsigned long long pad;
};

constexpr long long kSpinLimit = 1500000000ll;      // wall_clock64 ticks at 100 MHz: 15 s

__global__ void k_post(Channel* ch, const void* ptr, unsigned long long bytes) {
    ch->ptr = ptr;
    ch->bytes = bytes;
    const unsigned long long seq = ++ch->posted;
    __threadfence();
    __hip_atomic_store(&ch->ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// (`gave_up`: one word per world.  After the first time-out every later wait returns at once, so a broken exchange costs
// one spin limit, not one per message.)
__global__ void k_wait_ready(Channel* ch, unsigned long long* gave_up) {
    const unsigned long long seq = ch->received + 1;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(&ch->ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) {
        __builtin_amdgcn_s_sleep(20);
        if (__hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > kSpinLimit) {
            ch->error = 1;
            atomicAdd(gave_up, 1ull);
            break;
        }
    }
}

__global__ void k_copy(const Channel* ch, void* dst, unsigned long long bytes) {
    const double* src = static_cast<const double*>(__hip_atomic_load(&ch->ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (!src || ch->error) return;                  // the wait gave up: leave the buffer alone (the test reads the error count)
    double* d = static_cast<double*>(dst);
    const unsigned long long n = bytes / 8;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        d[i] = __builtin_nontemporal_load(src + i);
}

__global__ void k_done(Channel* ch) {
    const unsigned long long seq = ++ch->received;
    __threadfence();
    __hip_atomic_store(&ch->consumed, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_wait_consumed(Channel* ch, unsigned long long* gave_up) {
    const unsigned long long seq = ch->posted;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(&ch->consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) {
        __builtin_amdgcn_s_sleep(20);
        if (__hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > kSpinLimit) {
            ch->error = 2;
            atomicAdd(gave_up, 1ull);
            break;
        }
    }
}

struct World {
    int nranks = 0, joined = 0;
    std::mutex m;
    std::condition_variable cv;
    Channel* chan = nullptr;        // nranks x nranks, device memory
    unsigned long long* gave_up = nullptr;      // device word: number of waits that timed out
    std::vector<double> scratch;
    int arrived = 0, coll_gen = 0;
};

struct FakeComm {
    World* world;
    int rank, nranks;
};

struct PendingOp {
    bool is_send;
    const void* sbuf;
    void* rbuf;
    size_t bytes;
    int peer;
    FakeComm* comm;
    hipStream_t stream;
};

std::mutex g_m;
std::map<std::string, World*> g_worlds;
thread_local int t_group_depth = 0;
thread_local std::vector<PendingOp> t_ops;

size_t type_size(ncclDataType_t t) { return t == ncclDouble ? 8 : (t == ncclFloat ? 4 : 1); }

void barrier(World* w) {
    std::unique_lock<std::mutex> lk(w->m);
    const int gen = w->coll_gen;
    if (++w->arrived == w->nranks) {
        w->arrived = 0;
        ++w->coll_gen;
        w->cv.notify_all();
    } else {
        w->cv.wait(lk, [&] { return w->coll_gen != gen; });
    }
}

Channel* channel(FakeComm* c, int src, int dst) { return c->world->chan + (size_t)src * c->nranks + dst; }

ncclResult_t flush_group() {
    // every operation is a kernel launch on the caller's stream: publishes first, then the receives, then the waits
    for (auto& op : t_ops)
        if (op.is_send)
            hipLaunchKernelGGL(k_post, dim3(1), dim3(1), 0, op.stream, channel(op.comm, op.comm->rank, op.peer), op.sbuf,
                               (unsigned long long)op.bytes);
    for (auto& op : t_ops) {
        if (op.is_send) continue;
        Channel* ch = channel(op.comm, op.peer, op.comm->rank);
        hipLaunchKernelGGL(k_wait_ready, dim3(1), dim3(1), 0, op.stream, ch, op.comm->world->gave_up);
        const unsigned blocks = (unsigned)std::min<size_t>(512, (op.bytes / 8 + 255) / 256 + 1);
        hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, op.stream, ch, op.rbuf, (unsigned long long)op.bytes);
        hipLaunchKernelGGL(k_done, dim3(1), dim3(1), 0, op.stream, ch);
    }
    for (auto& op : t_ops)
        if (op.is_send)
            hipLaunchKernelGGL(k_wait_consumed, dim3(1), dim3(1), 0, op.stream, channel(op.comm, op.comm->rank, op.peer),
                               op.comm->world->gave_up);
    t_ops.clear();
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::random_device rd;
    for (size_t i = 0; i < sizeof(id->internal); ++i) id->internal[i] = (char)(rd() & 0x7f);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    const std::string key(id.internal, sizeof(id.internal));
    World* w;
    {
        std::lock_guard<std::mutex> lk(g_m);
        auto it = g_worlds.find(key);
        if (it == g_worlds.end()) {
            w = new World();
            w->nranks = nranks;
            if (hipMalloc(reinterpret_cast<void**>(&w->chan), sizeof(Channel) * (size_t)nranks * nranks) != hipSuccess) return ncclUnhandledCudaError;
            if (hipMemset(w->chan, 0, sizeof(Channel) * (size_t)nranks * nranks) != hipSuccess) return ncclUnhandledCudaError;
            if (hipMalloc(reinterpret_cast<void**>(&w->gave_up), 8) != hipSuccess) return ncclUnhandledCudaError;
            if (hipMemset(w->gave_up, 0, 8) != hipSuccess) return ncclUnhandledCudaError;
            g_worlds[key] = w;
        } else {
            w = it->second;
        }
    }
    {
        std::unique_lock<std::mutex> lk(w->m);
        ++w->joined;
        w->cv.notify_all();
        w->cv.wait(lk, [&] { return w->joined >= w->nranks; });
    }
    *comm = reinterpret_cast<ncclComm_t>(new FakeComm{w, rank, nranks});
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete reinterpret_cast<FakeComm*>(comm);
    return ncclSuccess;
}

// test hook: waits that timed out in any world of this process (call after synchronising the streams)
long long fake_rccl_graph_timeouts() {
    std::lock_guard<std::mutex> lk(g_m);
    long long total = 0;
    for (auto& kv : g_worlds) {
        unsigned long long v = 0;
        if (hipMemcpy(&v, kv.second->gave_up, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        total += (long long)v;
    }
    return total;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "fake rccl (graph) error"; }

ncclResult_t ncclGroupStart() {
    ++t_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (--t_group_depth > 0) return ncclSuccess;
    return flush_group();
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    t_ops.push_back(PendingOp{true, buf, nullptr, count * type_size(t), peer, reinterpret_cast<FakeComm*>(comm), s});
    return t_group_depth > 0 ? ncclSuccess : flush_group();
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    t_ops.push_back(PendingOp{false, nullptr, buf, count * type_size(t), peer, reinterpret_cast<FakeComm*>(comm), s});
    return t_group_depth > 0 ? ncclSuccess : flush_group();
}

// host-synchronous (the V-cycle itself never issues it: residual norms only)
ncclResult_t ncclAllReduce(const void* sbuf, void* rbuf, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t s) {
    if (t != ncclDouble || op != ncclSum) return ncclInvalidArgument;
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    World* w = c->world;
    std::vector<double> mine(count);
    if (hipMemcpyAsync(mine.data(), sbuf, count * 8, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::lock_guard<std::mutex> lk(w->m);
        if (w->scratch.size() != count * (size_t)c->nranks) w->scratch.assign(count * (size_t)c->nranks, 0.0);
        for (size_t i = 0; i < count; ++i) w->scratch[(size_t)c->rank * count + i] = mine[i];
    }
    barrier(w);
    std::vector<double> sum(count, 0.0);
    for (int r = 0; r < c->nranks; ++r)
        for (size_t i = 0; i < count; ++i) sum[i] += w->scratch[(size_t)r * count + i];
    barrier(w);
    if (hipMemcpyAsync(rbuf, sum.data(), count * 8, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* sbuf, void* rbuf, size_t count, ncclDataType_t t, int root, ncclComm_t comm,
                           hipStream_t s) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    ncclGroupStart();
    if (c->rank == root) {
        for (int r = 0; r < c->nranks; ++r)
            if (r != root) ncclSend(sbuf, count, t, r, comm, s);
    } else {
        ncclRecv(rbuf, count, t, root, comm, s);
    }
    return ncclGroupEnd();
}

}  // extern "C"
