// Microbenchmark: throughput of the VALU instructions the 8-wide node test is (or could be) made of, gfx950, 8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O2 tools/valu_rate3.hip -o tools/valu_rate3
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
// eight independent chains of one instruction: r_i = op(r_i, b, c)
#define CHAIN3(OP) asm volatile(OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define CHAIN2(OP) asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#define CHAIN1(OP) asm volatile(OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP16(CHAIN3("v_fma_f32")) }
        if (KIND == 1) { REP16(CHAIN2("v_max_f32")) }
        if (KIND == 2) { REP16(CHAIN3("v_max3_f32")) }
        if (KIND == 3) { REP16(CHAIN3("v_min3_f32")) }
        if (KIND == 4) { REP16(CHAIN2("v_max_i32")) }
        if (KIND == 5) { REP16(CHAIN3("v_max3_i32")) }
        if (KIND == 6) { REP16(CHAIN3("v_min3_i32")) }
        if (KIND == 7) { REP16(CHAIN2("v_max_u32")) }
        if (KIND == 8) { REP16(CHAIN1("v_cvt_f32_ubyte0")) }
        if (KIND == 9) { REP16(CHAIN1("v_cvt_f32_ubyte3")) }
        if (KIND == 10) { REP16(CHAIN2("v_sub_f32")) }
        if (KIND == 11) { REP16(CHAIN2("v_ashrrev_i32")) }
        if (KIND == 12) { REP16(asm volatile("v_bitop3_b32 %0, %0, %8, %9 bitop3:0xf4\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0xf4\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0xf4\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0xf4\n v_bitop3_b32 %4, %4, %8, %9 bitop3:0xf4\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0xf4\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0xf4\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0xf4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (KIND == 13) { REP16(CHAIN3("v_and_or_b32")) }
        if (KIND == 14) { REP16(CHAIN3("v_or3_b32")) }
        if (KIND == 15) { REP16(CHAIN3("v_med3_f32")) }
        if (KIND == 16) { REP16(CHAIN3("v_perm_b32")) }
        if (KIND == 17) { REP16(CHAIN3("v_bfe_u32")) }
        if (KIND == 18) { REP16(CHAIN3("v_mad_u32_u24")) }
        if (KIND == 19) { REP16(CHAIN3("v_lshl_add_u32")) }
        // the accept step of one child as it is now: v_cmp_le_f32 -> vcc, v_cndmask (vcc), or as 8 independent pairs
        if (KIND == 20) { REP16(asm volatile("v_cmp_le_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_le_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_le_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_le_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n v_cmp_le_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_le_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_le_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_le_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %9, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");) }
        // the same through arithmetic: d = b - a; m = d >> 31; a = a | (c & ~m)   (3 instructions, no vcc)
        if (KIND == 21) { REP16(asm volatile("v_sub_f32 %0, %8, %0\n v_ashrrev_i32 %0, 31, %0\n v_bitop3_b32 %0, %0, %9, %8 bitop3:0xf4\n v_sub_f32 %1, %8, %1\n v_ashrrev_i32 %1, 31, %1\n v_bitop3_b32 %1, %1, %9, %8 bitop3:0xf4\n v_sub_f32 %2, %8, %2\n v_ashrrev_i32 %2, 31, %2\n v_bitop3_b32 %2, %2, %9, %8 bitop3:0xf4\n v_sub_f32 %3, %8, %3\n v_ashrrev_i32 %3, 31, %3\n v_bitop3_b32 %3, %3, %9, %8 bitop3:0xf4\n v_sub_f32 %4, %8, %4\n v_ashrrev_i32 %4, 31, %4\n v_bitop3_b32 %4, %4, %9, %8 bitop3:0xf4\n v_sub_f32 %5, %8, %5\n v_ashrrev_i32 %5, 31, %5\n v_bitop3_b32 %5, %5, %9, %8 bitop3:0xf4\n v_sub_f32 %6, %8, %6\n v_ashrrev_i32 %6, 31, %6\n v_bitop3_b32 %6, %6, %9, %8 bitop3:0xf4\n v_sub_f32 %7, %8, %7\n v_ashrrev_i32 %7, 31, %7\n v_bitop3_b32 %7, %7, %9, %8 bitop3:0xf4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (KIND == 22) { REP16(asm volatile("v_cmp_le_i32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_le_i32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_le_i32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_le_i32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n v_cmp_le_i32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_le_i32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_le_i32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_le_i32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %9, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");) }
        if (KIND == 23) { REP16(CHAIN2("v_mul_f32")) }
        if (KIND == 24) { REP16(CHAIN2("v_add_u32")) }
        if (KIND == 25) { REP16(CHAIN3("v_min3_u32")) }
        // r = f32(f16 half of r) * b + c : the f16 plane form of the node test
        if (KIND == 26) { REP16(asm volatile("v_fma_mix_f32 %0, %0, %8, %9 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %1, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %2, %8, %9 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %3, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %4, %8, %9 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %5, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %6, %8, %9 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %7, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (KIND == 27) { REP16(CHAIN2("v_and_b32")) }
        if (KIND == 28) { REP16(CHAIN2("v_lshrrev_b32")) }
        if (KIND == 29) { REP16(CHAIN1("v_cvt_f32_f16")) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void check_mix(unsigned int* bad) {
    const int q = threadIdx.x;
    const _Float16 h = (_Float16)(float)q;
    unsigned short hb; __builtin_memcpy(&hb, &h, 2);
    const unsigned int packed_lo = hb | 0x3c000000u, packed_hi = ((unsigned int)hb << 16) | 0x3c00u;
    unsigned int n = 0;
    for (int i = 0; i < 4096; i++) {
        const float s = __uint_as_float(0x2f800000u + 2654435761u * (unsigned int)i % 0x20000000u), b = -s * 100.5f + (float)i * 1e-3f;
        float r0, r1;
        asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(packed_lo), "v"(s), "v"(b));
        asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(packed_hi), "v"(s), "v"(b));
        const float ref = __builtin_fmaf((float)q, s, b);
        n += (__float_as_uint(r0) != __float_as_uint(ref)) + (__float_as_uint(r1) != __float_as_uint(ref));
    }
    if (n) atomicAdd(bad, n);
}
template <int KIND> void run(const char* name, int per_iter) {
    float* out; hipMalloc(&out, 256 * 8192 * 4);
    const int w = 8, blocks = 256 * w, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10);
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %.2f cyc@2.4GHz per instruction per SIMD\n", name, best, best * 1e-3 * 2.4e9 / ((double)iters * per_iter * w));
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32", 128); run<23>("v_mul_f32", 128); run<10>("v_sub_f32", 128);
    run<1>("v_max_f32", 128); run<2>("v_max3_f32", 128); run<3>("v_min3_f32", 128); run<15>("v_med3_f32", 128);
    run<4>("v_max_i32", 128); run<5>("v_max3_i32", 128); run<6>("v_min3_i32", 128); run<7>("v_max_u32", 128); run<25>("v_min3_u32", 128);
    run<8>("v_cvt_f32_ubyte0", 128); run<9>("v_cvt_f32_ubyte3", 128);
    run<11>("v_ashrrev_i32", 128); run<12>("v_bitop3_b32", 128); run<13>("v_and_or_b32", 128); run<14>("v_or3_b32", 128);
    run<16>("v_perm_b32", 128); run<17>("v_bfe_u32", 128); run<18>("v_mad_u32_u24", 128); run<19>("v_lshl_add_u32", 128); run<24>("v_add_u32", 128);
    run<20>("v_cmp_le_f32 + v_cndmask (vcc) pairs", 256); run<22>("v_cmp_le_i32 + v_cndmask (vcc) pairs", 256);
    run<21>("v_sub_f32 + v_ashrrev + v_bitop3 triples", 384);
    run<26>("v_fma_mix_f32 (f16 lo/hi half x f32 + f32)", 128); run<27>("v_and_b32", 128); run<28>("v_lshrrev_b32", 128); run<29>("v_cvt_f32_f16", 128);
    // semantics of the mixed fma: equal to fmaf((float)half, s, b) for every integer 0..255 in either half
    {
        unsigned int* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(check_mix, dim3(1), dim3(256), 0, 0, bad);
        unsigned int h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("v_fma_mix_f32 vs fmaf((float)half, s, b): %u mismatches\n", h);
    }
    return 0;
}
