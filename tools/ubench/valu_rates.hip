// valu_rates.hip — issue cost of the VALU instructions the path kernel uses, relative to v_fma_f32 (MI355X).
// 8 independent dependency chains per lane, 8 waves per SIMD, every CU busy: wave-instructions per cycle per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rates tools/ubench/valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#define CHAINS 8
#define REP 64
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* sink, uint32_t iters, uint32_t seed) {
    uint32_t x[CHAINS];
    float f[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { x[c] = seed + threadIdx.x * 977u + c * 131u; f[c] = 1.0f + (float)((threadIdx.x + c) & 15) * 1e-3f; }
    const uint32_t m = seed | 1u;
    uint32_t tmp = 0, tmp2 = seed;
    asm volatile("s_mov_b64 vcc, 0x5555\n\ts_mov_b64 s[10:11], 0x3333" : : : "vcc", "s10", "s11");
    const float fm = 1.0001f, fa = 1e-7f;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP / CHAINS; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(fm), "v"(fa));
                if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[c]) : "v"(m));
                if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[c]) : "v"(m));
                if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x[c]) : "v"(m));
                if (OP == 4) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[c]));
                if (OP == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[c]));
                if (OP == 6) asm volatile("v_sin_f32 %0, %0" : "+v"(f[c]));
                if (OP == 7) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[c]) : "v"(m));
                if (OP == 8) asm volatile("v_lshrrev_b32 %0, 15, %0" : "+v"(x[c]));
                if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(m));
                if (OP == 10) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[c]) : "v"(m));
                if (OP == 11) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[c]));
                if (OP == 12) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[c]), "v"(fm) : "vcc");
                if (OP == 13) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(uint64_t*)&x[c & ~1]) : "v"(x[c]), "v"(m) : "vcc");
                if (OP == 14) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(m));
                if (OP == 15) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(fm), "v"(fa));
                if (OP == 16) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(uint64_t*)&f[c & ~1]) : "v"(*(uint64_t*)&f[(c & ~1) ^ 2]));
                if (OP == 17) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x[c]) : "v"(m));
                if (OP == 18) asm volatile("v_alignbit_b32 %0, %0, %0, 15" : "+v"(x[c]));
                if (OP == 19) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[c]) : "v"(m));
                if (OP == 20) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x[c]) : "v"(m) : "s10", "s11");
                if (OP == 21) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(x[c]) : "v"(f[c]), "v"(fm), "v"(m) : "vcc", "scc");
                if (OP == 22) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cmp_lt_f32_e64 s[10:11], %1, %4\n\ts_and_b64 vcc, vcc, s[10:11]\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc"
                                           : "+v"(x[c]), "+v"(f[c]) : "v"(fm), "v"(m), "v"(fa) : "vcc", "s10", "s11", "scc");
                if (OP == 23) asm volatile("v_mov_b32 %0, %1" : "+v"(x[c]) : "v"(m));
                if (OP == 24) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[c]) : "v"(fm));
                if (OP == 25) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[c]) : "v"(fa));
                if (OP == 26) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[c]) : "v"(fm));
                if (OP == 27) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[c]) : "v"(fm), "v"(fa));
                if (OP == 28) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x[c]), "v"(m) : "vcc");
                if (OP == 29) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x[c]) : "v"(m));
                if (OP == 36) asm volatile("v_mov_b32 %2, s9\n\tv_cmp_le_f32 vcc, %4, %3\n\tv_cmp_lt_f32_e64 s[10:11], %3, %1\n\ts_and_b64 vcc, vcc, s[10:11]\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc"
                                           : "+v"(x[c]), "+v"(f[c]), "=&v"(tmp) : "v"(f[(c + 1) & 7]), "v"(fa) : "vcc", "s10", "s11", "scc");
                if (OP == 37) asm volatile("v_mov_b32 %2, s9\n\tv_sub_u32 %5, %3, %4\n\tv_cmp_lt_u32 vcc, %5, %1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc"
                                           : "+v"(x[c]), "+v"(f[c]), "=&v"(tmp), "+v"(f[(c + 1) & 7]) : "v"(fa), "v"(tmp2) : "vcc");
                if (OP == 38) asm volatile("v_sub_u32 %2, %3, %4\n\ts_mov_b64 s[10:11], exec\n\tv_cmpx_lt_u32 %2, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %0, s9\n\ts_mov_b64 exec, s[10:11]"
                                           : "+v"(x[c]), "+v"(f[c]), "=&v"(tmp) : "v"(f[(c + 1) & 7]), "v"(fa) : "vcc", "s10", "s11");
                if (OP == 39) asm volatile("v_sub_u32 %2, %3, %4\n\tv_cmp_lt_u32 vcc, %2, %1\n\ts_and_saveexec_b64 s[10:11], vcc\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %0, s9\n\ts_mov_b64 exec, s[10:11]"
                                           : "+v"(x[c]), "+v"(f[c]), "=&v"(tmp) : "v"(f[(c + 1) & 7]), "v"(fa) : "vcc", "s10", "s11", "scc");
                if (OP == 40) asm volatile("v_sub_u32 %2, %3, %4\n\tv_min_u32 %5, %2, %1\n\tv_cmp_ne_u32 vcc, %5, %1\n\tv_mov_b32 %1, %5\n\tv_cndmask_b32 %0, %0, %2, vcc"
                                           : "+v"(x[c]), "+v"(f[c]), "=&v"(tmp), "+v"(f[(c + 1) & 7]) : "v"(fa), "v"(tmp2) : "vcc");
                if (OP == 41) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_add_u32 s12, s12, 1" : "+v"(f[c]) : "v"(fm), "v"(fa) : "s12", "scc");
                if (OP == 42) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2\n\ts_add_u32 s12, s12, 1" : "+v"(f[c]), "+v"(f[(c + 4) & 7]) : "v"(fm), "v"(fa) : "s12", "scc");
                if (OP == 43) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_add_u32 s12, s12, 1\n\ts_and_b32 s13, s12, 7" : "+v"(f[c]) : "v"(fm), "v"(fa) : "s12", "s13", "scc");
                if (OP == 44) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_cmp_eq_u32 s12, 12345\n\ts_cbranch_scc1 1f\n1:" : "+v"(f[c]) : "v"(fm), "v"(fa) : "scc");
                if (OP == 45) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_and_saveexec_b64 s[12:13], exec\n\ts_or_b64 exec, exec, s[12:13]" : "+v"(f[c]) : "v"(fm), "v"(fa) : "s12", "s13", "scc");
                if (OP == 30) asm volatile("s_and_b64 vcc, vcc, s[10:11]\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(m) : "vcc", "scc");
                if (OP == 31) asm volatile("s_and_b64 s[12:13], vcc, s[10:11]\n\tv_cndmask_b32_e64 %0, %0, %1, s[12:13]" : "+v"(x[c]) : "v"(m) : "s12", "s13", "scc");
                if (OP == 32) asm volatile("v_cmp_le_f32 vcc, %2, %1\n\tv_cndmask_b32 %4, %3, %1, vcc\n\tv_cmp_lt_f32 vcc, %4, %5\n\tv_cndmask_b32 %0, %0, %6, vcc\n\tv_cndmask_b32 %5, %5, %4, vcc"
                                           : "+v"(x[c]), "+v"(f[c]) : "v"(fa), "v"(fm), "v"(f[(c + 1) & 7]), "v"(f[(c + 2) & 7]), "v"(m) : "vcc");
                if (OP == 33) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(x[c]), "+v"(f[c]) : "v"(fm), "v"(m) : "vcc");
                if (OP == 34) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_nop 0\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(x[c]) : "v"(f[c]), "v"(fm), "v"(m) : "vcc", "scc");
                if (OP == 35) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_and_b64 vcc, vcc, exec\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(x[c]) : "v"(f[c]), "v"(fm), "v"(m) : "vcc", "scc");
            }
        }
    }
    uint32_t r = 0;
    for (int c = 0; c < CHAINS; ++c) r += x[c] + __float_as_uint(f[c]);
    if (r == 0x12345678u) sink[0] = r;
}

typedef void (*K)(uint32_t*, uint32_t, uint32_t);
int main(int argc, char** argv) {
    const int first = argc > 1 ? atoi(argv[1]) : 0, last = argc > 2 ? atoi(argv[2]) : 45;
    const char* names[] = {"v_fma_f32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_sqrt_f32", "v_rcp_f32", "v_sin_f32", "v_xor_b32",
                           "v_lshrrev_b32", "v_cndmask_b32", "v_mul_hi_u32", "v_cvt_f32_u32", "v_cmp_lt_f32", "v_mad_u64_u32", "v_add_u32",
                           "v_min3_f32", "v_pk_fma_f32", "v_xad_u32", "v_alignbit_b32", "v_lshl_add_u32", "v_cndmask_b32_e64(sgpr pair)",
                           "cmp+cndmask (2 VALU)", "cmp,cmp,s_and,cndmask,cndmask (4 VALU)", "v_mov_b32", "v_min_f32", "v_sub_f32", "v_mul_f32",
                           "v_fmac_f32", "v_cmp_lt_u32", "v_and_or_b32", "s_and vcc + cndmask(vcc) (1 VALU)", "s_and sgpr + cndmask_e64 (1 VALU)",
                           "cmp,cndmask,cmp,cndmask,cndmask (5 VALU, no SALU)", "cmp,cndmask,cndmask (3 VALU)", "cmp,s_nop,cndmask (2 VALU)", "cmp,s_and vcc,cndmask (2 VALU)",
                           "update: mov,cmp,cmp,s_and,cndmask,cndmask (current)", "update: mov,sub_u32,cmp_u32,cndmask,cndmask", "update: sub_u32,cmpx,mov,mov (exec)",
                           "update: sub_u32,cmp_u32,s_and_saveexec,mov,mov", "update: sub,min_u32,cmp_ne,mov,cndmask",
                           "fma + 1 salu (1 VALU)", "2 fma + 1 salu (2 VALU)", "fma + 2 salu (1 VALU)", "fma + s_cmp + s_cbranch (1 VALU)", "fma + saveexec + or exec (1 VALU)"};
    K ks[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>, k<16>, k<17>, k<18>, k<19>, k<20>, k<21>, k<22>, k<23>, k<24>, k<25>, k<26>, k<27>, k<28>, k<29>, k<30>, k<31>, k<32>, k<33>, k<34>, k<35>, k<36>, k<37>, k<38>, k<39>, k<40>, k<41>, k<42>, k<43>, k<44>, k<45>};
    uint32_t* sink;
    if (hipMalloc(&sink, 256) != hipSuccess) return 1;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
    const uint32_t iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double base = 0;
    printf("{\"device\": \"%s\", \"cus\": %d, \"rates\": {", p.name, p.multiProcessorCount);
    for (int o = first; o <= last && o < 46; ++o) {
        hipLaunchKernelGGL(ks[o], dim3(blocks), dim3(256), 0, 0, sink, iters / 10, 12345u);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("\"%s\": \"launch failed\"}}\n", names[o]); return 2; }
        hipEventRecord(e0);
        hipLaunchKernelGGL(ks[o], dim3(blocks), dim3(256), 0, 0, sink, iters, 12345u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double wave_instr = (double)blocks * 4.0 * iters * REP;
        const double per_simd_per_us = wave_instr / (p.multiProcessorCount * 4.0) / (ms * 1e3);
        if (o == 0) base = ms;
        if (base == 0) base = 1.189;
        printf("%s\"%s\": {\"ms\": %.3f, \"wave_instr_per_simd_per_us\": %.1f, \"cost_vs_fma\": %.2f}", o ? ", " : "", names[o], ms, per_simd_per_us, ms / base);
        fflush(stdout);
    }
    printf("}}\n");
    return 0;
}
