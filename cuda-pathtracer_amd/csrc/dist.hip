// dist.hip — the ONE exchange step of a multi-GPU frame, behind the C ABI: every rank's tile of rows goes to the
// destination rank over RCCL (xGMI), where one kernel places the rows into the whole frame.
//
// The reference is single-GPU ("Multi-GPU tiling + RCCL gather is new work defined by BASELINE.json");
// what is kept from it is the pixel -> RNG stream keying by the GLOBAL pixel index (integrator.h:278-279, 378-379),
// which makes the union of any row partition bit-identical to the unsharded frame.
//
// Pattern: xGMI is point-to-point, 7 links per GPU, so the gather is DIRECT: one ncclSend per peer,
// N-1 ncclRecv on the destination inside one ncclGroupStart/End - every peer pushes over its own link concurrently;
// no ring.  Payloads are the exact tile sizes (local_rows * width * 3 elements), 8-bit and/or float, chosen per call.
//
// librccl is loaded on first use (dlopen "librccl.so.1"): single-GPU callers of libptmi.so do not depend on it.
#include "../host/application_state.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>

namespace ptmi {

#define PTMI_HIP(call)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) throw HipError(e_, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)

namespace {
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
};
Rccl load_rccl() {
    Rccl r;                                               // filled completely before anyone can see it
    // PTMI_RCCL_LIB: the file to load instead (a process that has torch in it already holds torch's librccl.so.1, which a bare
    // name would resolve to; the tests' transport stand-in, tests/mock_rccl.cpp, is given by path)
    const char* names[] = {std::getenv("PTMI_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        r.handle = dlopen(n, RTLD_NOW | (n == names[0] ? RTLD_LOCAL : RTLD_GLOBAL));      // a stand-in must not interpose on anyone else's nccl*
        if (r.handle) break;
    }
    if (!r.handle) throw DistError(std::string("cannot load librccl.so.1: ") + dlerror());
    auto sym = [&](const char* name) {
        void* p = dlsym(r.handle, name);
        if (!p) { std::string m = std::string("librccl lacks ") + name; dlclose(r.handle); r.handle = nullptr; throw DistError(m); }
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    return r;
}
// One table per process, published once: a function-local static is initialised under the C++ runtime's guard, so a second
// rank thread (ptmi.h: "one process or host thread per GPU") either waits for the first one's load or sees the finished
// table; a load that throws leaves it uninitialised and the next call tries again.
const Rccl& rccl() {
    static const Rccl r = load_rccl();
    return r;
}
#define PTMI_NCCL(call)                                                                                              \
    do {                                                                                                             \
        ncclResult_t r_ = (call);                                                                                    \
        if (r_ != ncclSuccess) throw DistError(std::string(#call) + ": " + rccl().GetErrorString(r_));              \
    } while (0)
}  // namespace

// ---------------------------------------------------------------------------------------------
// row placement: tiles in rank order (each tile = that rank's local rows, local-row-major) -> whole frame
// ---------------------------------------------------------------------------------------------
// Global row y belongs to rank (y / row_block) % n_ranks and is local row (y / (row_block * n_ranks)) * row_block +
// y % row_block there (ptmi_tiling, include/ptmi.h).  `own` replaces the staging copy of the destination's own tile (it is
// read straight from the render buffers).  Rows are contiguous in the tile and in the frame, and every tile starts on a
// 16-byte boundary of the staging buffer (tileOffsets), so while a row is a whole number of 16-byte vectors (width % 16 == 0
// for the 8-bit image, width % 4 == 0 for float radiance: every BASELINE frame) a row moves as 16-byte vectors: one block row
// of the grid per frame row - the row arithmetic is scalar, once per workgroup - and one thread per vector.
__global__ __launch_bounds__(256) void ptmi_place_rows_v16(const uint4* __restrict__ stage, const uint4* __restrict__ own, int own_rank,
                                                           const long long* __restrict__ tile_offset /* ELEMENTS, per rank */,
                                                           int elems_per_v16, int vec_per_row, int n_ranks, int row_block, uint4* __restrict__ frame) {
    const int y = blockIdx.y;
    const int blk = y / row_block;
    const int rank = blk % n_ranks;
    const int lr = (blk / n_ranks) * row_block + (y - blk * row_block);
    const uint4* src = (rank == own_rank && own ? own : stage + tile_offset[rank] / elems_per_v16) + (long long)lr * vec_per_row;
    uint4* dst = frame + (long long)y * vec_per_row;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < vec_per_row; v += gridDim.x * blockDim.x) dst[v] = src[v];
}
// any other width: element by element
template <typename T>
__global__ __launch_bounds__(256) void ptmi_place_tiles(const T* __restrict__ stage, const T* __restrict__ own, int own_rank,
                                                        const long long* __restrict__ tile_offset /* elements, per rank */,
                                                        int width, int height, int n_ranks, int row_block, T* __restrict__ frame) {
    const long long row_elems = (long long)width * 3;
    const long long total = row_elems * height;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / row_elems);
        const long long in_row = i - (long long)y * row_elems;
        const int blk = y / row_block;
        const int rank = blk % n_ranks;
        const int lr = (blk / n_ranks) * row_block + (y - blk * row_block);
        const long long src = (long long)lr * row_elems + in_row;
        frame[i] = rank == own_rank && own ? own[src] : stage[tile_offset[rank] + src];
    }
}

// first element of every rank's tile in the staging buffers: exact tile sizes, each start rounded up to 16 elements (so that a
// tile of bytes and a tile of floats both start on a 16-byte boundary); returns the elements the staging buffer needs
static long long tileOffsets(int width, int height, int n_ranks, int row_block, std::vector<long long>& off) {
    off.assign((size_t)n_ranks, 0);
    long long acc = 0;
    for (int k = 0; k < n_ranks; k++) {
        acc = (acc + 15) / 16 * 16;
        off[k] = acc;
        acc += (long long)countLocalRows(height, n_ranks, k, row_block) * width * 3;
    }
    return acc;
}

template <typename T>
static void launch_place(const T* stage, const T* own, int own_rank, const long long* d_off, int w, int h, int n_ranks, int row_block,
                         T* frame, hipStream_t s) {
    const long long total = (long long)w * 3 * h;
    if (total <= 0) return;
    const long long row_bytes = (long long)w * 3 * (long long)sizeof(T);
    const bool aligned = ((uintptr_t)stage % 16 == 0) && ((uintptr_t)frame % 16 == 0) && (!own || (uintptr_t)own % 16 == 0);
    if (row_bytes % 16 == 0 && aligned && h <= 65535) {
        const int vec_per_row = (int)(row_bytes / 16);
        hipLaunchKernelGGL(ptmi_place_rows_v16, dim3(std::min((vec_per_row + 255) / 256, 64), h), dim3(256), 0, s, (const uint4*)stage, (const uint4*)own,
                           own_rank, d_off, (int)(16 / sizeof(T)), vec_per_row, n_ranks, row_block, (uint4*)frame);
        return;
    }
    const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL((ptmi_place_tiles<T>), dim3(blocks), dim3(256), 0, s, stage, own, own_rank, d_off, w, h, n_ranks, row_block, frame);
}

// ---------------------------------------------------------------------------------------------
// DistState
// ---------------------------------------------------------------------------------------------
void distUniqueId(void* out128) {
    ncclUniqueId id;
    PTMI_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(out128, &id, sizeof id);
}

void DistState::freeFrame() {
    void* ptrs[] = {d_stage_rgb, d_stage_rad, d_frame_rgb, d_frame_rad, d_tile_offset};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    d_stage_rgb = nullptr; d_stage_rad = nullptr; d_frame_rgb = nullptr; d_frame_rad = nullptr; d_tile_offset = nullptr;
    frame_w = frame_h = frame_row_block = 0; frame_ranks = 0;
    have_rgb = have_rad = false;
}

void DistState::finalize() {
    if (stream) (void)hipStreamSynchronize(stream);
    if (comm) { (void)rccl().CommDestroy((ncclComm_t)comm); comm = nullptr; }
    freeFrame();
    if (d_token) { (void)hipFree(d_token); d_token = nullptr; }
    if (gather_done) { (void)hipEventDestroy(gather_done); gather_done = nullptr; }
    if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
    n_ranks = 1; rank = 0; pending = false; failed = false;
}

void DistState::init(const void* unique_id128, int n, int r) {
    if (n < 1 || r < 0 || r >= n) throw ArgError("ptmi_dist_init: need n_ranks >= 1 and 0 <= rank < n_ranks");
    finalize();
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, sizeof id);
    PTMI_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    PTMI_HIP(hipEventCreateWithFlags(&gather_done, hipEventDisableTiming));
    d_token = (int*)hipMallocSafe(2 * sizeof(int), "dist.token");
    PTMI_HIP(hipMemset(d_token, 0, 2 * sizeof(int)));
    ncclComm_t c = nullptr;
    PTMI_NCCL(rccl().CommInitRank(&c, n, id, r));
    comm = c; n_ranks = n; rank = r;
}

// frame-sized buffers on the destination: allocated on first use for a geometry, reused while it does not change
void DistState::ensureFrame(const TileMap& tm, bool want_rgb, bool want_rad) {
    const bool same = frame_w == tm.width && frame_h == tm.height && frame_row_block == tm.row_block && frame_ranks == tm.n_ranks;
    if (!same) freeFrame();
    const size_t px = (size_t)tm.width * (size_t)tm.height;
    std::vector<long long> off;
    const size_t stage_elems = (size_t)tileOffsets(tm.width, tm.height, tm.n_ranks, tm.row_block, off);
    if (!d_tile_offset) {
        d_tile_offset = (long long*)hipMallocSafe(off.size() * sizeof(long long), "dist.tile_offset");
        PTMI_HIP(hipMemcpy(d_tile_offset, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice));
        h_tile_offset = off;
        frame_w = tm.width; frame_h = tm.height; frame_row_block = tm.row_block; frame_ranks = tm.n_ranks;
    }
    if (want_rgb && !d_frame_rgb) {
        d_stage_rgb = (unsigned char*)hipMallocSafe(stage_elems, "dist.stage_rgb");
        d_frame_rgb = (unsigned char*)hipMallocSafe(px * 3, "dist.frame_rgb");
    }
    if (want_rad && !d_frame_rad) {
        d_stage_rad = (float*)hipMallocSafe(stage_elems * sizeof(float), "dist.stage_rad");
        d_frame_rad = (float*)hipMallocSafe(px * 3 * sizeof(float), "dist.frame_rad");
    }
}

namespace {
// An exception between ncclGroupStart and ncclGroupEnd (a Send / Recv that fails, an argument check) must not leave the
// thread's group open: every later nccl call would be queued into it and never launched - a barrier that returns without
// having synchronised, ranks that silently drift apart.  The guard closes the group on unwind (result ignored: the group is
// being abandoned) and the caller marks the communicator failed.
struct GroupGuard {
    const Rccl& L; bool open = false;
    explicit GroupGuard(const Rccl& l) : L(l) {}
    void start() { PTMI_NCCL(L.GroupStart()); open = true; }
    void end() { open = false; PTMI_NCCL(L.GroupEnd()); }
    ~GroupGuard() { if (open) (void)L.GroupEnd(); }
};
}  // namespace

void DistState::check() const {
    if (!comm) throw ArgError("ptmi_dist_*: call ptmi_dist_init first");
    if (failed) throw DistError("the communicator is in a failed state after an earlier RCCL error: ptmi_dist_finalize and ptmi_dist_init again");
}

int DistState::commCount() const {
    check();
    int n = 0;
    PTMI_NCCL(rccl().CommCount((ncclComm_t)comm, &n));
    return n;
}

void DistState::gatherFrame(const RenderState& r, int dst, int what) {
    check();
    if (dst < 0 || dst >= n_ranks) throw ArgError("ptmi_gather_frame: dst_rank out of range");
    if (!(what & 3) || (what & ~3)) throw ArgError("ptmi_gather_frame: what must be 1 (rgb8), 2 (radiance) or 3 (both)");
    if (!r.d_image) throw ArgError("ptmi_gather_frame: buffers not allocated");
    if (r.tile.n_ranks != n_ranks || r.tile.rank != rank)
        throw ArgError("ptmi_gather_frame: the tiling of ptmi_update_resolution (n_ranks, rank) differs from ptmi_dist_init's");
    const bool want_rgb = what & 1, want_rad = what & 2;
    const Rccl& L = rccl();
    const size_t n_mine = r.n_local * 3;
    if (rank == dst) ensureFrame(r.tile, want_rgb, want_rad);
    // the render stream has been synchronised by renderFrame(); a gather still in flight on this stream simply precedes us
    try {
        GroupGuard group(L);
        group.start();
        if (rank == dst) {
            for (int k = 0; k < n_ranks; k++) {
                if (k == dst) continue;
                const size_t n_k = (size_t)countLocalRows(r.tile.height, n_ranks, k, r.tile.row_block) * (size_t)r.tile.width * 3;
                if (!n_k) continue;
                if (want_rgb) PTMI_NCCL(L.Recv(d_stage_rgb + h_tile_offset[k], n_k, ncclUint8, k, (ncclComm_t)comm, stream));
                if (want_rad) PTMI_NCCL(L.Recv(d_stage_rad + h_tile_offset[k], n_k, ncclFloat32, k, (ncclComm_t)comm, stream));
            }
        } else if (n_mine) {
            if (want_rgb) PTMI_NCCL(L.Send(r.d_image, n_mine, ncclUint8, dst, (ncclComm_t)comm, stream));
            if (want_rad) PTMI_NCCL(L.Send(r.d_radiance, n_mine, ncclFloat32, dst, (ncclComm_t)comm, stream));
        }
        group.end();
    } catch (const DistError&) { failed = true; throw; }
    if (rank == dst) {
        if (want_rgb) launch_place<unsigned char>(d_stage_rgb, r.d_image, rank, d_tile_offset, r.tile.width, r.tile.height, n_ranks, r.tile.row_block, d_frame_rgb, stream);
        if (want_rad) launch_place<float>(d_stage_rad, r.d_radiance, rank, d_tile_offset, r.tile.width, r.tile.height, n_ranks, r.tile.row_block, d_frame_rad, stream);
        PTMI_HIP(hipGetLastError());
        have_rgb = have_rgb || want_rgb; have_rad = have_rad || want_rad;
    }
    PTMI_HIP(hipEventRecord(gather_done, stream));       // the next frame's resolve waits for this before it overwrites the tile
    pending = true;
}

void DistState::wait() {
    if (stream) PTMI_HIP(hipStreamSynchronize(stream));
    pending = false;
}

void DistState::barrier() {
    check();
    try { PTMI_NCCL(rccl().AllReduce(d_token, d_token + 1, 1, ncclInt32, ncclSum, (ncclComm_t)comm, stream)); }
    catch (const DistError&) { failed = true; throw; }
    PTMI_HIP(hipStreamSynchronize(stream));
    pending = false;
}

double DistState::allreduceMax(double v) {
    check();
    double* d = nullptr;
    PTMI_HIP(hipMalloc((void**)&d, 2 * sizeof(double)));
    struct Free { double* p; ~Free() { (void)hipFree(p); } } guard{d};
    const double init[2] = {v, v};                       // the result slot is never read uninitialised
    PTMI_HIP(hipMemcpy(d, init, sizeof init, hipMemcpyHostToDevice));
    try { PTMI_NCCL(rccl().AllReduce(d, d + 1, 1, ncclFloat64, ncclMax, (ncclComm_t)comm, stream)); }
    catch (const DistError&) { failed = true; throw; }
    PTMI_HIP(hipStreamSynchronize(stream));
    PTMI_HIP(hipMemcpy(&v, d + 1, sizeof v, hipMemcpyDeviceToHost));
    return v;
}

// test hook: the destination's placement step alone, tiles supplied by the caller (rank order, exact sizes)
void debugPlaceTiles(int width, int height, int n_ranks, int row_block, const unsigned char* h_tiles_rgb, const float* h_tiles_rad,
                     unsigned char* out_rgb, float* out_rad, hipStream_t s) {
    if (width <= 0 || height <= 0 || n_ranks < 1 || row_block < 1) throw ArgError("bad geometry");
    const size_t n = (size_t)width * height * 3;
    std::vector<long long> off;
    const size_t stage_elems = (size_t)tileOffsets(width, height, n_ranks, row_block, off);      // the staging layout of ptmi_gather_frame
    struct Buf { void* p = nullptr; ~Buf() { if (p) (void)hipFree(p); } };
    Buf d_off;
    d_off.p = hipMallocSafe(off.size() * sizeof(long long), "place.off");
    PTMI_HIP(hipMemcpy(d_off.p, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice));
    auto run = [&](auto* h_in, auto* h_out) {
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(h_out)>>;
        Buf d_in, d_out;
        d_in.p = hipMallocSafe(stage_elems * sizeof(T), "place.in"); d_out.p = hipMallocSafe(n * sizeof(T), "place.out");
        size_t src = 0;
        for (int k = 0; k < n_ranks; k++) {                  // the caller's tiles are packed; the staging buffer's start on 16 elements
            const size_t n_k = (size_t)countLocalRows(height, n_ranks, k, row_block) * (size_t)width * 3;
            if (n_k) PTMI_HIP(hipMemcpy((T*)d_in.p + off[k], h_in + src, n_k * sizeof(T), hipMemcpyHostToDevice));
            src += n_k;
        }
        launch_place<T>((const T*)d_in.p, nullptr, -1, (const long long*)d_off.p, width, height, n_ranks, row_block, (T*)d_out.p, s);
        PTMI_HIP(hipGetLastError());
        PTMI_HIP(hipStreamSynchronize(s));
        PTMI_HIP(hipMemcpy(h_out, d_out.p, n * sizeof(T), hipMemcpyDeviceToHost));
    };
    if (h_tiles_rgb && out_rgb) run(h_tiles_rgb, out_rgb);
    if (h_tiles_rad && out_rad) run(h_tiles_rad, out_rad);
}

}  // namespace ptmi
