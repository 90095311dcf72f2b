// f64_rates.hip — what the FP64 instructions of pt_direct_kernel cost on the MI355X, in issue cycles per wave64
// instruction and per SIMD, with v_fma_f32 beside them; and what hipcc's IEEE division and square root (the sequences the
// `direct` integrator cannot avoid: every operation is one correctly rounded double operation) cost as a whole.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ubench/f64_rates tools/ubench/f64_rates.hip
// usage: f64_rates [waves per SIMD = 8 | 4 | 1]      (CHAINS independent dependency chains per lane)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHAINS 8
#define REP 64
template <int OP>
__global__ __launch_bounds__(256) void k(double* sink, uint32_t iters, double seed) {
    double d[CHAINS];
    float f[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { d[c] = 1.0 + (double)((threadIdx.x + c) & 15) * 1e-3 + seed; f[c] = (float)d[c]; }
    const double dm = 1.0000001 + seed, da = 1e-9 + seed;
    const float fm = 1.0001f, fa = 1e-7f;
    asm volatile("s_mov_b64 vcc, 0x5555" : : : "vcc");
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP / CHAINS; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(fm), "v"(fa));
                if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(dm), "v"(da));
                if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(da));
                if (OP == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[c]) : "v"(dm));
                if (OP == 4) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[c]));
                if (OP == 5) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[c]));
                if (OP == 6) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(d[c]) : "v"(dm) : "vcc");
                if (OP == 7) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(dm), "v"(da) : "vcc");
                if (OP == 8) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(dm), "v"(da));
                if (OP == 9) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(d[c]));
                if (OP == 10) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[c]), "v"(dm) : "vcc");
                if (OP == 11) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(d[c]) : "v"(f[c]));
                if (OP == 12) asm volatile("v_mov_b64 %0, %1" : "+v"(d[c]) : "v"(dm));
                if (OP == 13) d[c] = d[c] / dm;          // IEEE division as hipcc expands it
                if (OP == 14) d[c] = sqrt(d[c]) + 1.0;   // IEEE square root as hipcc expands it, + one add to keep the chain bounded
                if (OP == 15) d[c] = 1.0 / d[c] + 0.5;   // reciprocal form (normalisation), + one add
            }
        }
    }
    double r = 0;
    for (int c = 0; c < CHAINS; ++c) r += d[c] + (double)f[c];
    if (r == 0.12345678) sink[0] = r;
}

typedef void (*K)(double*, uint32_t, double);
int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 8;
    const char* names[] = {"v_fma_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f64", "v_rsq_f64", "v_div_scale_f64", "v_div_fmas_f64",
                           "v_div_fixup_f64", "v_ldexp_f64", "v_cmp_lt_f64", "v_cvt_f64_u32", "v_mov_b64", "a / b (IEEE, hipcc)",
                           "sqrt(a) + 1 (IEEE, hipcc)", "1 / a + 0.5 (IEEE, hipcc)"};
    K ks[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>};
    double* sink;
    if (hipMalloc(&sink, 256) != hipSuccess) return 1;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int threads = 256;                                  // 4 waves per block = one per SIMD
    const int blocks = p.multiProcessorCount * waves;         // `waves` blocks per CU
    const double waves_per_simd = waves;
    const uint32_t iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double clock_ghz = p.clockRate * 1e-6;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f, \"waves_per_simd\": %.2f, \"chains\": %d, \"rates\": {", p.name, p.multiProcessorCount, clock_ghz, waves_per_simd, CHAINS);
    double base = 0;
    for (int o = 0; o < 16; ++o) {
        hipLaunchKernelGGL(ks[o], dim3(blocks), dim3(threads), 0, 0, sink, iters / 10, 0.0);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("\"%s\": \"launch failed\"}}\n", names[o]); return 2; }
        hipEventRecord(e0);
        hipLaunchKernelGGL(ks[o], dim3(blocks), dim3(threads), 0, 0, sink, iters, 0.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // wave-level operations issued per SIMD, and the SIMD's cycles per operation at the reported clock
        const double ops_per_simd = (double)blocks * (threads / 64.0) * iters * REP / (p.multiProcessorCount * 4.0);
        const double cycles_per_op = ms * 1e-3 * clock_ghz * 1e9 / ops_per_simd;
        if (o == 0) base = ms;
        printf("%s\"%s\": {\"ms\": %.3f, \"cycles_per_wave_op_per_simd\": %.2f, \"cost_vs_fma_f32\": %.2f}", o ? ", " : "", names[o], ms, cycles_per_op, ms / base);
        fflush(stdout);
    }
    printf("}}\n");
    return 0;
}
