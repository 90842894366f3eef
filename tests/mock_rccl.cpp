// mock_rccl.cpp — TEST INFRASTRUCTURE ONLY: a stand-in for librccl.so.1 whose transport is files in a shared directory.
//
// RCCL refuses two ranks on one device, and the GPU box of the tests has one.  libptmi.so loads "librccl.so.1" by name on
// the first ptmi_dist_* call (csrc/dist.hip), so a test that puts this library first on LD_LIBRARY_PATH (or names it in
// PTMI_RCCL_LIB) runs the product's whole multi-rank exchange - ptmi_dist_init, the ncclSend / N-1 ncclRecv group of
// ptmi_gather_frame with its exact tile sizes and offsets, the placement kernel, barrier and max-reduction - with several
// processes on ONE GPU.  Only the wire is mocked.
//
// STREAM-ORDERED AND ASYNCHRONOUS, like the real library: ncclGroupEnd (or a bare ncclSend / ncclRecv) returns at once.  What
// it leaves on the caller's stream is (1) an event marking everything enqueued before the group and (2) a one-wave GATE kernel
// that holds the stream until a flag in host memory is set.  A worker thread - one per process, batches strictly in posting
// order - waits for the event, moves the bytes (device -> file for a send, file -> device for a receive, on a stream of its
// own) and then opens the gate: whatever the caller enqueues behind the group (the placement kernel, the gather_done event
// the next frame's resolve waits for) runs after the data has moved, whatever the caller enqueues on OTHER streams meanwhile
// runs concurrently - so the product's own ordering (resolve_gate, two gathers queued on one stream, staging-buffer reuse) is
// what keeps a frame intact, not a blocking call.  PTMI_MOCK_RCCL_DELAY_MS delays every batch in the worker (makes the window
// in which a missing dependency would corrupt a tile seconds wide); PTMI_MOCK_RCCL_FAIL=recv|send makes that call return
// ncclInvalidArgument (fault injection for the product's error path).  The gate gives up after 120 s.
//
// Build (the tests do it): hipcc -shared -fPIC -o <dir>/librccl.so.1 tests/mock_rccl.cpp
// The directory for the files comes from PTMI_MOCK_RCCL_DIR.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

namespace {
struct Comm { std::string dir; int n = 0, rank = 0; std::vector<unsigned long long> sent, received; unsigned long long reductions = 0; };
struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; unsigned long long seq; };
struct Batch { std::vector<Op> ops; hipEvent_t ready; int* flag; int device; };      // flag: host-mapped, 0 = closed, 1 = open, 2 = failed
thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;

std::mutex g_mu;
std::condition_variable g_cv;
std::deque<Batch> g_queue;
bool g_worker_started = false;
std::atomic<int> g_in_flight{0};
std::atomic<int> g_failed{0};

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

// holds the stream it is launched on until *flag != 0 (or 120 s of the 100 MHz real-time counter have passed)
__global__ void mock_gate(volatile int* flag) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (*flag == 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 120ull * 100000000ull) break;
        __builtin_amdgcn_s_sleep(64);
    }
}

bool moveBatch(const Batch& b, hipStream_t s) {
    std::vector<char> host;
    for (const Op& op : b.ops) {                                   // all sends first: a rank that both sends and receives cannot block itself
        if (!op.send) continue;
        host.resize(op.bytes);
        if (op.bytes && (hipMemcpyAsync(host.data(), op.buf, op.bytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) return false;
        Comm& c = *op.comm;
        writeFile(c.dir + "/msg_" + std::to_string(c.rank) + "_" + std::to_string(op.peer) + "_" + std::to_string(op.seq), host.data(), op.bytes);
    }
    for (const Op& op : b.ops) {
        if (op.send) continue;
        host.resize(op.bytes);
        Comm& c = *op.comm;
        const std::string path = c.dir + "/msg_" + std::to_string(op.peer) + "_" + std::to_string(c.rank) + "_" + std::to_string(op.seq);
        if (!readFile(path, host.data(), op.bytes)) return false;            // also: the sender posted another size
        struct stat st; if (stat(path.c_str(), &st) != 0 || (size_t)st.st_size != op.bytes) return false;
        std::remove(path.c_str());
        if (op.bytes && (hipMemcpyAsync(op.buf, host.data(), op.bytes, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) return false;
    }
    return true;
}

void worker() {
    const char* d = std::getenv("PTMI_MOCK_RCCL_DELAY_MS");
    const int delay_ms = d ? std::atoi(d) : 0;
    hipStream_t s = nullptr; int dev = -1;
    for (;;) {
        Batch b;
        {
            std::unique_lock<std::mutex> lk(g_mu);
            g_cv.wait(lk, [] { return !g_queue.empty(); });
            b = std::move(g_queue.front()); g_queue.pop_front();
        }
        if (b.device != dev) { (void)hipSetDevice(b.device); dev = b.device; if (s) (void)hipStreamDestroy(s); s = nullptr; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
        bool ok = hipEventSynchronize(b.ready) == hipSuccess;      // everything the caller had enqueued before the group
        if (delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
        ok = ok && moveBatch(b, s);
        if (!ok) g_failed.store(1);
        __atomic_store_n(b.flag, ok ? 1 : 2, __ATOMIC_RELEASE);   // opens the gate
        (void)hipEventDestroy(b.ready);
        g_in_flight.fetch_sub(1);
    }
}

// posts the thread's pending ops as one batch: event + gate on every stream involved (the product uses one), then returns
ncclResult_t post() {
    std::vector<Op> ops; ops.swap(g_ops);
    if (ops.empty()) return ncclSuccess;
    if (g_failed.load()) return ncclSystemError;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return ncclUnhandledCudaError;
    const hipStream_t s = ops[0].stream;
    for (const Op& op : ops) if (op.stream != s) return ncclInvalidUsage;       // one stream per group is all this stand-in does
    Batch b; b.ops = std::move(ops); b.device = device;
    if (hipEventCreateWithFlags(&b.ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipHostMalloc((void**)&b.flag, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return ncclUnhandledCudaError;   // leaked on purpose: tiny, and the gate may still be reading it
    *b.flag = 0;
    int* d_flag = nullptr;
    if (hipHostGetDevicePointer((void**)&d_flag, b.flag, 0) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(b.ready, s) != hipSuccess) return ncclUnhandledCudaError;
    hipLaunchKernelGGL(mock_gate, dim3(1), dim3(1), 0, s, (volatile int*)d_flag);
    if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
    g_in_flight.fetch_add(1);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_worker_started) { std::thread(worker).detach(); g_worker_started = true; }
        g_queue.push_back(std::move(b));
    }
    g_cv.notify_one();
    return ncclSuccess;
}
bool failInjected(const char* what) { const char* f = std::getenv("PTMI_MOCK_RCCL_FAIL"); return f && std::strcmp(f, what) == 0; }
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
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { *count = reinterpret_cast<const Comm*>(comm)->n; return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    for (int i = 0; i < 1300000 && g_in_flight.load() > 0; i++) std::this_thread::sleep_for(std::chrono::microseconds(100));   // let the worker finish
    delete reinterpret_cast<Comm*>(comm);
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error"; }
ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { if (--g_depth > 0) return ncclSuccess; g_depth = 0; return post(); }
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (peer < 0 || peer >= c->n || peer == c->rank || failInjected("send")) return ncclInvalidArgument;
    g_ops.push_back(Op{true, const_cast<void*>(buf), count * typeSize(t), peer, c, s, c->sent[peer]++});
    return g_depth ? ncclSuccess : post();
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (peer < 0 || peer >= c->n || peer == c->rank || failInjected("recv")) return ncclInvalidArgument;
    g_ops.push_back(Op{false, buf, count * typeSize(t), peer, c, s, c->received[peer]++});
    return g_depth ? ncclSuccess : post();
}
// the reductions stay host-synchronous (the product follows each with a stream synchronisation anyway): the stream is drained
// first - which waits for every gate in front, i.e. for every transfer posted before - then the values go through files
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    if (g_failed.load()) return ncclSystemError;
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
