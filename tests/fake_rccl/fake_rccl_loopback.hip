// TEST / MEASUREMENT INFRASTRUCTURE: a one-rank "loopback" stand-in for the RCCL entry points libmg_hip.so uses.  It lets
// ONE rank of an N-slab decomposition run alone on a GPU: communicator set-up never waits for peers, a grouped
// ncclRecv from a neighbour is served by a device copy (ONE kernel per group, as RCCL fuses a group's sends and receives
// into one launch) of what the same group ncclSends to that neighbour
// (same size on a slab: the planes are mirrored, so the values are wrong but the work, the stream ordering and the bytes
// written are those of a real exchange minus the link), broadcasts from other roots and all-reduces leave the buffer as it
// is.  Every operation is a stream operation, so the cycle can also be captured ("graph_comm").  Used by
// tools/slab_rank_probe.py to time what one slab's GPU does per cycle -- the decomposition's compute and launch
// overhead without the artefacts of eight ranks sharing one GPU.  Results are NOT meaningful numerically.  Nothing here
// is shipped.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

namespace {
struct LoopComm { int rank, nranks; };
struct Op { bool is_send; const void* sbuf; void* rbuf; size_t bytes; int peer; hipStream_t stream; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
size_t type_size(ncclDataType_t t) { return t == ncclDouble ? 8 : (t == ncclFloat ? 4 : 1); }

struct Copies { const double* src[4]; double* dst[4]; unsigned long long n[4]; int count; };

// a few workgroups only, like a point-to-point RCCL kernel (blockIdx.y = the copy)
__global__ void k_group_copy(Copies c) {
    const int k = blockIdx.y;
    const double* s = c.src[k];
    double* d = c.dst[k];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < c.n[k]; i += (unsigned long long)gridDim.x * blockDim.x)
        d[i] = s[i];
}

ncclResult_t flush() {
    Copies c{};
    hipStream_t stream = nullptr;
    for (auto& r : t_ops) {
        if (r.is_send) continue;
        for (auto& s : t_ops)
            if (s.is_send && s.peer == r.peer && s.bytes == r.bytes && c.count < 4) {
                c.src[c.count] = static_cast<const double*>(s.sbuf);
                c.dst[c.count] = static_cast<double*>(r.rbuf);
                c.n[c.count] = r.bytes / 8;
                ++c.count;
                stream = r.stream;
                break;
            }
    }
    t_ops.clear();
    if (c.count == 0) return ncclSuccess;
    hipLaunchKernelGGL(k_group_copy, dim3(16, c.count), dim3(512), 0, stream, c);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { std::memset(id->internal, 1, sizeof(id->internal)); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId, int rank) {
    *comm = reinterpret_cast<ncclComm_t>(new LoopComm{rank, nranks});
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete reinterpret_cast<LoopComm*>(comm); return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "loopback rccl error"; }
ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return --t_depth > 0 ? ncclSuccess : flush(); }
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t, hipStream_t s) {
    t_ops.push_back(Op{true, buf, nullptr, count * type_size(t), peer, s});
    return t_depth > 0 ? ncclSuccess : flush();
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t, hipStream_t s) {
    t_ops.push_back(Op{false, nullptr, buf, count * type_size(t), peer, s});
    return t_depth > 0 ? ncclSuccess : flush();
}
ncclResult_t ncclAllReduce(const void* sbuf, void* rbuf, size_t count, ncclDataType_t t, ncclRedOp_t, ncclComm_t, hipStream_t s) {
    if (sbuf != rbuf && hipMemcpyAsync(rbuf, sbuf, count * type_size(t), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return ncclUnhandledCudaError;
    return ncclSuccess;
}
ncclResult_t ncclBroadcast(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) { return ncclSuccess; }
}  // extern "C"
