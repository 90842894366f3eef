// mock_rccl.cpp — TEST INFRASTRUCTURE ONLY: a stand-in for librccl.so.1 whose transport is files in a shared directory.
//
// RCCL refuses two ranks on one device, and the GPU box of the tests has one.  libptmi.so loads "librccl.so.1" by name on
// the first ptmi_dist_* call (csrc/dist.hip), so a test that puts this library first on LD_LIBRARY_PATH (or names it in
// PTMI_RCCL_LIB) runs the product's whole multi-rank exchange - ptmi_dist_init, the ncclSend / N-1 ncclRecv group of
// ptmi_gather_frame with its exact tile sizes and offsets, the placement kernel, barrier and max-reduction - with several
// processes on ONE GPU.  Only the wire is mocked.
//
// STREAM-ORDERED AND ASYNCHRONOUS, like the real library: ncclGroupEnd (or a bare ncclSend / ncclRecv) returns at once and
// leaves on the caller's stream, in this order:
//   [delay kernel]  PTMI_MOCK_RCCL_DELAY_MS of device-side spinning: makes the window in which a missing dependency would
//                   corrupt a tile that wide (nothing on the host is waited for here)
//   [copy kernels]  every send buffer -> a pinned host staging buffer; an event behind them tells the worker thread when
//   [gate kernel]   (only with receives) one wave that holds the stream until the worker has put the peers' bytes into the
//                   receives' pinned staging buffers and set a flag in host memory; gives up after 120 s
//   [copy kernels]  staging -> every receive buffer
// so whatever the caller enqueues behind the group (the placement kernel, the gather_done event the next frame's resolve
// waits for) runs after the data has moved, and whatever it enqueues on OTHER streams meanwhile runs concurrently: the
// product's own ordering (resolve_gate, two gathers queued on one stream, staging-buffer reuse) is what keeps a frame intact,
// not a blocking call.  The worker thread - one per process, batches strictly in posting order - makes NO call that needs a
// GPU queue (HIP multiplexes a process's streams onto a few hardware queues; a copy of the worker's queued behind the gate it
// is meant to open would never run): it waits for the event, writes the sends' files, reads the receives' files, sets the flag.
// Run the processes with GPU_MAX_HW_QUEUES=8 so that the exchange stream does not share a hardware queue with a render
// stream (sharing only delays, but it would also hide the very race the asynchronous test is there to catch).
// PTMI_MOCK_RCCL_FAIL=recv|send makes that call return ncclInvalidArgument (fault injection for the product's error path).
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
struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; unsigned long long seq; char* staging; };   // staging: pinned host, mapped
struct Batch { std::vector<Op> ops; hipEvent_t copied; int* flag; };      // flag: host-mapped (nullptr without receives), 0 = closed, 1 = open, 2 = failed
thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;

// never destroyed: the worker thread is detached and waits on g_cv for the life of the process; destroying a condition
// variable that has a waiter (a static destructor at exit would) blocks in pthread_cond_destroy for ever
std::mutex& g_mu = *new std::mutex;
std::condition_variable& g_cv = *new std::condition_variable;
std::deque<Batch>& g_queue = *new std::deque<Batch>;
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
__global__ void mock_delay(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
__global__ void mock_copy(const char* __restrict__ src, char* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

void worker() {
    for (;;) {
        Batch b;
        {
            std::unique_lock<std::mutex> lk(g_mu);
            g_cv.wait(lk, [] { return !g_queue.empty(); });
            b = std::move(g_queue.front()); g_queue.pop_front();
        }
        bool ok = hipEventSynchronize(b.copied) == hipSuccess;    // the sends' bytes are in their staging buffers (a host-side wait)
        for (const Op& op : b.ops) {                                // all sends first: a rank that both sends and receives cannot block itself
            if (!op.send || !ok) continue;
            Comm& c = *op.comm;
            writeFile(c.dir + "/msg_" + std::to_string(c.rank) + "_" + std::to_string(op.peer) + "_" + std::to_string(op.seq), op.staging, op.bytes);
        }
        for (const Op& op : b.ops) {
            if (op.send || !ok) continue;
            Comm& c = *op.comm;
            const std::string path = c.dir + "/msg_" + std::to_string(op.peer) + "_" + std::to_string(c.rank) + "_" + std::to_string(op.seq);
            struct stat st;
            ok = readFile(path, op.staging, op.bytes) && stat(path.c_str(), &st) == 0 && (size_t)st.st_size == op.bytes;   // also: the sender posted another size
            std::remove(path.c_str());
        }
        if (!ok) g_failed.store(1);
        if (b.flag) __atomic_store_n(b.flag, ok ? 1 : 2, __ATOMIC_RELEASE);   // opens the gate
        g_in_flight.fetch_sub(1);
    }
}

// posts the thread's pending ops as one batch (see the head of this file), then returns
ncclResult_t post() {
    std::vector<Op> ops; ops.swap(g_ops);
    if (ops.empty()) return ncclSuccess;
    if (g_failed.load()) return ncclSystemError;
    const hipStream_t s = ops[0].stream;
    for (const Op& op : ops) if (op.stream != s) return ncclInvalidUsage;       // one stream per group is all this stand-in does
    const char* d = std::getenv("PTMI_MOCK_RCCL_DELAY_MS");
    const int delay_ms = d ? std::atoi(d) : 0;
    Batch b; b.flag = nullptr;
    bool any_recv = false;
    auto dev = [](void* host) { void* p = nullptr; return hipHostGetDevicePointer(&p, host, 0) == hipSuccess ? (char*)p : (char*)nullptr; };
    // pinned buffers are never freed (hipHostFree may wait for the device; these are test-sized and the process is short-lived)
    for (Op& op : ops) {
        if (hipHostMalloc((void**)&op.staging, op.bytes ? op.bytes : 1, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return ncclUnhandledCudaError;
        any_recv = any_recv || !op.send;
    }
    if (delay_ms > 0) hipLaunchKernelGGL(mock_delay, dim3(1), dim3(1), 0, s, (unsigned long long)delay_ms * 100000ull);
    for (const Op& op : ops)
        if (op.send && op.bytes) hipLaunchKernelGGL(mock_copy, dim3(64), dim3(256), 0, s, (const char*)op.buf, dev(op.staging), op.bytes);
    if (hipEventCreateWithFlags(&b.copied, hipEventDisableTiming) != hipSuccess || hipEventRecord(b.copied, s) != hipSuccess) return ncclUnhandledCudaError;
    if (any_recv) {
        if (hipHostMalloc((void**)&b.flag, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return ncclUnhandledCudaError;
        *b.flag = 0;
        hipLaunchKernelGGL(mock_gate, dim3(1), dim3(1), 0, s, (volatile int*)dev(b.flag));
        for (const Op& op : ops)
            if (!op.send && op.bytes) hipLaunchKernelGGL(mock_copy, dim3(64), dim3(256), 0, s, (const char*)dev(op.staging), (char*)op.buf, op.bytes);
    }
    if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
    b.ops = std::move(ops);
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
    g_ops.push_back(Op{true, const_cast<void*>(buf), count * typeSize(t), peer, c, s, c->sent[peer]++, nullptr});
    return g_depth ? ncclSuccess : post();
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (peer < 0 || peer >= c->n || peer == c->rank || failInjected("recv")) return ncclInvalidArgument;
    g_ops.push_back(Op{false, buf, count * typeSize(t), peer, c, s, c->received[peer]++, nullptr});
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
