// Microbenchmark 2: relative wall-clock cost of the VALU instruction kinds the renderer uses (8 independent chains, 8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define OPS8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i + 1.0f;
    float b = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; it++) {
        REP16(
            if (KIND == 0) { asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c)); }
            if (KIND == 1) { asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "vcc"); }
            if (KIND == 2) { asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "s20", "s21"); }
            if (KIND == 3) { asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "vcc"); }
            if (KIND == 4) { asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b) : "vcc"); }
            if (KIND == 5) { asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b)); }
            if (KIND == 6) { asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b)); }
            if (KIND == 7) { asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])); }
            if (KIND == 8) { asm volatile("v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c)); }
            if (KIND == 9) { asm volatile("v_sub_f32 %0, s20, %0\n v_mul_f32 %1, s21, %1\n v_add_f32 %2, s22, %2\n v_sub_f32 %3, s23, %3\n v_mul_f32 %4, s20, %4\n v_add_f32 %5, s21, %5\n v_sub_f32 %6, s22, %6\n v_mul_f32 %7, s23, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) :: "s20", "s21", "s22", "s23"); }
            if (KIND == 10) { asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b)); }
            if (KIND == 11) { asm volatile("v_xor_b32 %0, %0, %8\n v_lshlrev_b32 %1, 3, %1\n v_add_u32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_lshrrev_b32 %4, 2, %4\n v_add_u32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b)); }
        )
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> double run(const char* name, double base) {
    float* out; hipMalloc(&out, 256 * 2048 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(2048), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(2048), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double per = best * 1e-3 / ((double)iters * 128 * 8) * 2.4e9;   // cycles (at 2.4 GHz) per instruction per SIMD, 8 waves/SIMD
    printf("%-34s %8.3f ms  %.2f cyc@2.4GHz per instr per SIMD  (x%.2f of v_fma)\n", name, best, per, base > 0 ? best / base : 1.0);
    hipFree(out);
    return best;
}
int main() {
    double b = run<0>("v_fma_f32", 0);
    run<6>("v_sub/mul/add_f32 mix", b); run<9>("v_sub/mul/add with SGPR operand", b); run<5>("v_max_f32", b);
    run<1>("v_cndmask_b32 (vcc)", b); run<2>("v_cndmask_b32 (sgpr pair)", b); run<3>("v_cmp_lt_f32 -> vcc", b);
    run<4>("v_cmp + v_cndmask pairs (128 = 64+64)", b); run<7>("v_rcp_f32", b); run<8>("v_div_fixup_f32", b);
    run<10>("v_mov_b32", b); run<11>("int xor/shift/add mix", b);
    return 0;
}
