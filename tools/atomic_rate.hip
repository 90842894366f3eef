// Throughput of wave-aggregated device-scope atomics the way the refill launches use them (one lane of a wave adds to a cursor, the
// wave waits for the value): 1536 workgroups x 4 waves, every wave does `iters` dependent atomicAdds.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_rate tools/atomic_rate.hip && ./atomic_rate
// stride (ints) between the cursors the waves use: 0 = one cursor for all; the wave picks cursor (wave id % n_cursors).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(int* cur, int n_cursors, int stride, int iters, int* sink, int with_load) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    int* c = cur + (size_t)(wave % n_cursors) * stride;
    int acc = 0;
    for (int i = 0; i < iters; i++) {
        if (with_load) acc += __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int v = 0;
        if ((threadIdx.x & 63) == 0) v = atomicAdd(c, 1);
        acc += __shfl(v, 0);
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}
int main() {
    int *cur, *sink;
    hipMalloc(&cur, 64 << 20); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200, waves = 1536 * 4;
    struct { int n, stride, load; const char* what; } cases[] = {
        {1, 0, 0, "one cursor"}, {1, 0, 1, "one cursor, a load of it before every add"},
        {8, 1, 0, "8 cursors in one 32-byte sector"}, {8, 32, 0, "8 cursors 128 B apart"}, {8, 64, 0, "8 cursors 256 B apart"},
        {8, 1024, 0, "8 cursors 4 KB apart"}, {8, 1 << 18, 0, "8 cursors 1 MB apart"}, {64, 1024, 0, "64 cursors 4 KB apart"},
        {128, 1, 0, "128 cursors, consecutive ints"}, {6144, 16, 0, "one cursor per wave (64 B apart)"},
    };
    for (auto& c : cases) {
        hipMemset(cur, 0, 64 << 20);
        k<<<1536, 256>>>(cur, c.n, c.stride, 10, sink, c.load);
        hipEventRecord(e0);
        k<<<1536, 256>>>(cur, c.n, c.stride, iters, sink, c.load);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-48s %8.2f ms  %7.1f ns per atomic (device-wide)  %6.2f us per wave-iteration\n", c.what, ms, ms * 1e6 / ((double)waves * iters), ms * 1e3 / iters);
    }
    return 0;
}
