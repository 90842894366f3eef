// gather_rate.hip — how many dependent random 32-byte record fetches per second does an MI355X sustain?
// The access pattern of the large-scene walk (ptmi_bounce_phased): every lane follows its own chain, one record
// (2 x float4 at the same 32-byte-aligned address) per step, the next index taken from the record just read.
//   hipcc --offload-arch=gfx950 -O3 -o tools/gather_rate tools/gather_rate.hip && tools/gather_rate
// args: table_MB active_lanes lds_pad_bytes steps valu_pad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>

template <int PAD>
__global__ __launch_bounds__(256) void chase(const float4* __restrict__ tab, unsigned n_rec, int steps, int active, unsigned* out) {
    extern __shared__ float4 pad_lds[];
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    if ((threadIdx.x & 63) >= active) return;
    unsigned idx = (gid * 2654435761u) % n_rec;
    float acc = 0.0f;
    for (int s = 0; s < steps; s++) {
        const float4 a = tab[2 * idx], b = tab[2 * idx + 1];
        float x = a.x + b.y;
#pragma unroll
        for (int q = 0; q < PAD; q++) x = x * 1.0001f + a.z;      // dependent VALU padding
        acc += x;
        idx = __float_as_uint(a.w);                                // next record: stored in the record
    }
    if (acc == 12345.0f) out[0] = idx;
    if (gid == 0) out[1] = idx;
}

int main(int argc, char** argv) {
    const size_t table_mb = argc > 1 ? atoi(argv[1]) : 20;
    const int active = argc > 2 ? atoi(argv[2]) : 64, lds = argc > 3 ? atoi(argv[3]) : 0, steps = argc > 4 ? atoi(argv[4]) : 2000;
    const int pad = argc > 5 ? atoi(argv[5]) : 0;
    const unsigned n_rec = (unsigned)(table_mb * 1024 * 1024 / 32);
    std::vector<float4> h(2 * (size_t)n_rec);
    std::mt19937 rng(1);
    for (unsigned i = 0; i < n_rec; i++) {
        const unsigned nxt = rng() % n_rec;
        float w; memcpy(&w, &nxt, 4);
        h[2 * (size_t)i] = make_float4(1.0f, 2.0f, 0.5f, w); h[2 * (size_t)i + 1] = make_float4(0.1f, 0.2f, 0.3f, 0.4f);
    }
    float4* d; unsigned* out;
    hipMalloc(&d, h.size() * sizeof(float4)); hipMalloc(&out, 8);
    hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
    const int blocks = 256 * 8 * 2;                                 // more than resident
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        if (pad == 0) hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(256), lds, 0, d, n_rec, steps, active, out);
        else if (pad <= 16) hipLaunchKernelGGL(chase<16>, dim3(blocks), dim3(256), lds, 0, d, n_rec, steps, active, out);
        else hipLaunchKernelGGL(chase<48>, dim3(blocks), dim3(256), lds, 0, d, n_rec, steps, active, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double gathers = (double)blocks * 4 * active * steps;
        if (rep == 2) printf("table %zu MB, %d active lanes/wave, lds pad %d B, valu pad %d: %.3f ms, %.3e record fetches/s (%.2f TB/s of 32-B records)\n",
                             table_mb, active, lds, pad, ms, gathers / (ms * 1e-3), gathers * 32 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
