#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "wide_bvh.h"
using namespace ptmi;
__global__ void k(const uint4* nodes, const float* rays, WideStep* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* q = nodes + 8 * (size_t)i;
    const f3 o = mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    const f3 d = mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    const f3 inv = mk3(wide_inv(d.x), wide_inv(d.y), wide_inv(d.z));
    out[i] = wide_node_test(q[0], q[1], q[2], q[3], q[4], q[5], q[6], o, inv, wide_octinv(inv), 1e-4f, 3.0e38f);
}
int main() {
    const int n = 1 << 16;
    std::vector<uint32_t> nodes((size_t)n * 32); std::vector<float> rays((size_t)n * 6);
    srand(1);
    for (int i = 0; i < n; i++) {
        uint32_t* r = &nodes[(size_t)i * 32];
        for (int k = 0; k < 32; k++) r[k] = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
        float p[3] = {(float)(rand() % 100) * 0.1f - 5, (float)(rand() % 100) * 0.1f - 5, (float)(rand() % 100) * 0.1f - 5};
        memcpy(r, p, 12);
        r[3] = (120u + rand() % 4) | ((120u + rand() % 4) << 8) | ((120u + rand() % 4) << 16) | ((uint32_t)(rand() & 0xff) << 24);
        for (int k = 0; k < 6; k++) rays[(size_t)i * 6 + k] = (float)(rand() % 2000) * 0.01f - 10.0f;
        if (i % 7 == 0) rays[(size_t)i * 6 + 3] = 0.0f;
    }
    uint4* dn; float* dr; WideStep* dout;
    hipMalloc(&dn, nodes.size() * 4); hipMalloc(&dr, rays.size() * 4); hipMalloc(&dout, n * sizeof(WideStep));
    hipMemcpy(dn, nodes.data(), nodes.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dr, rays.data(), rays.size() * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dn, dr, dout, n);
    std::vector<WideStep> got(n); hipMemcpy(got.data(), dout, n * sizeof(WideStep), hipMemcpyDeviceToHost);
    int bad[5] = {0, 0, 0, 0, 0}, hits = 0;
    for (int i = 0; i < n; i++) {
        const uint4* q = (const uint4*)&nodes[(size_t)i * 32];
        const f3 o = mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), d = mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        const f3 inv = mk3(wide_inv(d.x), wide_inv(d.y), wide_inv(d.z));
        const WideStep w = wide_node_test(q[0], q[1], q[2], q[3], q[4], q[5], q[6], o, inv, wide_octinv(inv), 1e-4f, 3.0e38f);
        bad[0] += w.child_base != got[i].child_base; bad[1] += w.tri_base != got[i].tri_base; bad[2] += w.imask != got[i].imask;
        bad[3] += w.inner != got[i].inner; bad[4] += w.tris != got[i].tris; hits += w.tris != 0 || w.inner != 0;
        if (i < 2) printf("host %08x %08x %02x %02x %06x | dev %08x %08x %02x %02x %06x\n", w.child_base, w.tri_base, w.imask, w.inner, w.tris, got[i].child_base, got[i].tri_base, got[i].imask, got[i].inner, got[i].tris);
    }
    printf("mismatches: child_base %d tri_base %d imask %d inner %d tris %d of %d (%d with a hit)\n", bad[0], bad[1], bad[2], bad[3], bad[4], n, hits);
    return bad[0] + bad[1] + bad[2] + bad[3] + bad[4] != 0;
}
