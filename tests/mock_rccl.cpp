// mock_rccl.cpp — TEST INFRASTRUCTURE ONLY: a stand-in for librccl.so.1 whose transport is files in a shared directory.
//
// RCCL refuses two ranks on one device, and the GPU box of the tests has one.  libptmi.so loads "librccl.so.1" by name on
// the first ptmi_dist_* call (csrc/dist.hip), so a test that puts this library first on LD_LIBRARY_PATH runs the product's
// whole multi-rank exchange - ptmi_dist_init, the ncclSend / N-1 ncclRecv group of ptmi_gather_frame with its exact tile
// sizes and offsets, the placement kernel, barrier and max-reduction - with several processes on ONE GPU.  Only the wire is
// mocked: Send copies the device buffer to a file, Recv waits for the file and copies it to the device buffer; both when the
// group ends, after the stream has drained, in the order they were posted (as NCCL matches them per peer).
//
// Build (tests/test_gpu_parity.py does it): hipcc -shared -fPIC -o <dir>/librccl.so.1 tests/mock_rccl.cpp
// The directory for the files comes from PTMI_MOCK_RCCL_DIR.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

namespace {
struct Comm { std::string dir; int n = 0, rank = 0; std::vector<unsigned long long> sent, received; unsigned long long reductions = 0; };
struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;

size_t typeSize(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}
bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
void writeFile(const std::string& path, const void* data, size_t bytes) {
    const std::string tmp = path + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) std::abort();
    if (bytes && std::fwrite(data, 1, bytes, f) != bytes) std::abort();
    std::fclose(f);
    if (std::rename(tmp.c_str(), path.c_str()) != 0) std::abort();
}
bool readFile(const std::string& path, void* data, size_t bytes, double timeout_s = 120.0) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!exists(path)) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    const size_t got = bytes ? std::fread(data, 1, bytes, f) : 0;
    std::fclose(f);
    return got == bytes;
}
ncclResult_t flush() {
    std::vector<Op> ops; ops.swap(g_ops);
    for (const Op& op : ops) if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<char> host;
    for (const Op& op : ops) {                                   // all sends first: a rank that both sends and receives cannot block itself
        if (!op.send) continue;
        host.resize(op.bytes);
        if (op.bytes && hipMemcpy(host.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        Comm& c = *op.comm;
        writeFile(c.dir + "/msg_" + std::to_string(c.rank) + "_" + std::to_string(op.peer) + "_" + std::to_string(c.sent[op.peer]++), host.data(), op.bytes);
    }
    for (const Op& op : ops) {
        if (op.send) continue;
        host.resize(op.bytes);
        Comm& c = *op.comm;
        const std::string path = c.dir + "/msg_" + std::to_string(op.peer) + "_" + std::to_string(c.rank) + "_" + std::to_string(c.received[op.peer]++);
        if (!readFile(path, host.data(), op.bytes)) return ncclSystemError;     // also: the sender posted another size
        struct stat st; if (stat(path.c_str(), &st) != 0 || (size_t)st.st_size != op.bytes) return ncclInvalidArgument;
        std::remove(path.c_str());
        if (op.bytes && hipMemcpy(op.buf, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "mock-%d-%lld", (int)getpid(), (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    const char* base = std::getenv("PTMI_MOCK_RCCL_DIR");
    if (!base) return ncclInvalidUsage;
    Comm* c = new Comm;
    c->dir = std::string(base) + "/" + std::string(id.internal); c->n = nranks; c->rank = rank;
    c->sent.assign(nranks, 0); c->received.assign(nranks, 0);
    mkdir(c->dir.c_str(), 0700);
    writeFile(c->dir + "/rank_" + std::to_string(rank), "", 0);
    for (int k = 0; k < nranks; k++) { char dummy; if (!readFile(c->dir + "/rank_" + std::to_string(k), &dummy, 0)) return ncclSystemError; }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete reinterpret_cast<Comm*>(comm); return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error"; }
ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { if (--g_depth > 0) return ncclSuccess; g_depth = 0; return flush(); }
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
    g_ops.push_back(Op{true, const_cast<void*>(buf), count * typeSize(t), peer, c, s});
    return g_depth ? ncclSuccess : flush();
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
    g_ops.push_back(Op{false, buf, count * typeSize(t), peer, c, s});
    return g_depth ? ncclSuccess : flush();
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    const size_t bytes = count * typeSize(t);
    std::vector<char> mine(bytes), other(bytes);
    if (hipMemcpy(mine.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const std::string tag = c->dir + "/red_" + std::to_string(c->reductions++) + "_";
    writeFile(tag + std::to_string(c->rank), mine.data(), bytes);
    std::vector<char> acc(mine);
    for (int k = 0; k < c->n; k++) {
        if (k == c->rank) continue;
        if (!readFile(tag + std::to_string(k), other.data(), bytes)) return ncclSystemError;
        for (size_t i = 0; i < count; i++) {
            if (t == ncclFloat64) { double a, b; std::memcpy(&a, &acc[8 * i], 8); std::memcpy(&b, &other[8 * i], 8); a = op == ncclMax ? (a > b ? a : b) : a + b; std::memcpy(&acc[8 * i], &a, 8); }
            else if (t == ncclInt32) { int a, b; std::memcpy(&a, &acc[4 * i], 4); std::memcpy(&b, &other[4 * i], 4); a = op == ncclMax ? (a > b ? a : b) : a + b; std::memcpy(&acc[4 * i], &a, 4); }
            else return ncclInvalidArgument;
        }
    }
    if (hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
}
