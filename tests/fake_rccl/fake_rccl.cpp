// TEST INFRASTRUCTURE: an in-process stand-in for the ten RCCL entry points libmg_hip.so uses, so that the
// ASYNCHRONOUS slab transport (ncclSend/ncclRecv groups on a communication stream, event-ordered against the
// sweep kernels) can be exercised with several "ranks" = threads of one process sharing one GPU.  Point-to-point
// operations are matched between threads at ncclGroupEnd and executed as device-to-device copies on the
// RECEIVER's stream, ordered after an event the sender records on ITS stream; the sender's stream then waits
// for the copy.  Collectives synchronise on the host.  Nothing here is shipped.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace {

struct Message {
    const void* buf;
    size_t bytes;
    hipEvent_t ready;       // recorded on the sender's stream when its data is final
    hipEvent_t consumed;    // recorded by the receiver after its copy; the sender's stream waits for it
    bool taken = false;
};

struct World {
    int nranks = 0, joined = 0, generation = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<Message*>> mailbox;     // (src, dst) -> FIFO
    // host-side collectives
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

ncclResult_t flush_group() {
    // sends first: publish (buffer, ready event) to the mailbox
    std::vector<Message*> mine;
    for (auto& op : t_ops) {
        if (!op.is_send) continue;
        Message* msg = new Message();
        msg->buf = op.sbuf;
        msg->bytes = op.bytes;
        if (hipEventCreateWithFlags(&msg->ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventCreateWithFlags(&msg->consumed, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventRecord(msg->ready, op.stream) != hipSuccess) return ncclUnhandledCudaError;
        World* w = op.comm->world;
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->mailbox[{op.comm->rank, op.peer}].push_back(msg);
        }
        w->cv.notify_all();
        mine.push_back(msg);
    }
    // receives: wait for the matching send, copy on the receiver's stream
    for (auto& op : t_ops) {
        if (op.is_send) continue;
        World* w = op.comm->world;
        Message* msg = nullptr;
        {
            std::unique_lock<std::mutex> lk(w->m);
            auto key = std::make_pair(op.peer, op.comm->rank);
            w->cv.wait(lk, [&] { return !w->mailbox[key].empty(); });
            msg = w->mailbox[key].front();
            w->mailbox[key].pop_front();
        }
        if (msg->bytes != op.bytes) return ncclInvalidArgument;
        if (hipStreamWaitEvent(op.stream, msg->ready, 0) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpyAsync(op.rbuf, msg->buf, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess)
            return ncclUnhandledCudaError;
        if (hipEventRecord(msg->consumed, op.stream) != hipSuccess) return ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(w->m);
            msg->taken = true;
        }
        w->cv.notify_all();
    }
    // senders: their stream must not overwrite the buffer before the receiver's copy ran
    size_t si = 0;
    for (auto& op : t_ops) {
        if (!op.is_send) continue;
        Message* msg = mine[si++];
        World* w = op.comm->world;
        {
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { return msg->taken; });
        }
        if (hipStreamWaitEvent(op.stream, msg->consumed, 0) != hipSuccess) return ncclUnhandledCudaError;
        // The receiver is done with the message (`taken` is set after its last use).  Destroying an event that a stream still
        // waits for is legal -- its resources go when it has completed --, and it keeps the number of live events bounded:
        // round 2 leaked two per message, tens of thousands per probe run (see DESIGN.md section 6 on the profiler crash).
        (void)hipEventDestroy(msg->ready);
        (void)hipEventDestroy(msg->consumed);
        delete msg;
    }
    t_ops.clear();
    return ncclSuccess;
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
    FakeComm* c = new FakeComm{w, rank, nranks};
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete reinterpret_cast<FakeComm*>(comm);
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "fake rccl error"; }

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

// Collectives: stream-synchronous host implementations (they are latency-bound scalars / one-off gathers).
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
    for (int r = 0; r < c->nranks; ++r)                      // fixed rank order: deterministic
        for (size_t i = 0; i < count; ++i) sum[i] += w->scratch[(size_t)r * count + i];
    barrier(w);
    if (hipMemcpyAsync(rbuf, sum.data(), count * 8, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* sbuf, void* rbuf, size_t count, ncclDataType_t t, int root, ncclComm_t comm,
                           hipStream_t s) {
    // expressed through the point-to-point machinery: root sends to everybody else
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
