// valu_rate.hip - issue cost of the VALU instructions the HBM-bound quantizer kernels are made of, on gfx950 (stand-alone:
// hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip).  256 workgroups (one per CU) of NW waves, every wave runs
// 64 independent copies of ONE instruction per loop iteration (8 register chains x 8), so dependent-issue latency is
// hidden from 2 waves per SIMD on; prints SIMD cycles per wave-instruction = ns x clock / (64 x iterations x waves per
// SIMD), clock measured by a v_fma_f32 loop at the documented 2 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define OPS(X)                                                                                                          \
    X(0, "v_fma_f32", "v_fma_f32 %0, %0, %2, %3")                                                                        \
    X(1, "v_pk_fma_f32 (2 elements)", "v_pk_fma_f32 %1, %1, %4, %5")                                                     \
    X(2, "v_pk_mul_f32 (2 elements)", "v_pk_mul_f32 %1, %1, %4")                                                         \
    X(3, "v_pk_add_f32 (2 elements)", "v_pk_add_f32 %1, %1, %4")                                                         \
    X(4, "v_exp_f32", "v_exp_f32 %0, %0")                                                                                \
    X(5, "v_rcp_f32", "v_rcp_f32 %0, %0")                                                                                \
    X(6, "v_rndne_f32", "v_rndne_f32 %0, %0")                                                                            \
    X(7, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %0")                                                                        \
    X(8, "v_cvt_f32_f16 sdwa (high half)", "v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1") \
    X(9, "v_fma_mix_f32 (f16 x f32 + f32)", "v_fma_mix_f32 %0, %0, %2, %3 op_sel_hi:[1,0,0]")                            \
    X(10, "v_fma_mixlo_f16", "v_fma_mixlo_f16 %0, %0, %2, %3 op_sel_hi:[1,0,0]")                                         \
    X(11, "v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 %0, %2, 1, %0")                                                            \
    X(12, "v_max3_f32 |a|,|b|,c", "v_max3_f32 %0, |%0|, |%2|, %3")                                                       \
    X(13, "v_cmp_gt_f32 (to sgpr pair) |a|", "v_cmp_gt_f32 vcc, |%0|, %2")                                                \
    X(14, "v_pk_max_f16 (2 elements)", "v_pk_max_f16 %0, %0, %2")                                                        \
    X(15, "v_cvt_pk_f16_f32 (2 elements)", "v_cvt_pk_f16_f32 %0, %0, %2")                                                \
    X(16, "v_sub_f32", "v_sub_f32 %0, %0, %2")                                                                           \
    X(17, "v_pk_fma_f16 (2 elements)", "v_pk_fma_f16 %0, %0, %2, %3")                                                    \
    X(18, "v_exp_f16", "v_exp_f16 %0, %0")                                                                               \
    X(19, "v_rcp_f16", "v_rcp_f16 %0, %0")                                                                               \
    X(20, "v_sad_u8", "v_sad_u8 %0, %0, %2, %3")                                                                         \
    X(21, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %2, %3")

typedef float float2v __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void probe(float* out, int iters) {
    float v[8];
    float2v p[8];
    for (int i = 0; i < 8; ++i) {
        v[i] = threadIdx.x * 0.001f + i * 0.01f;
        p[i] = float2v{v[i], v[i] + 1.f};
    }
    const float c1 = 0.999f, c2 = 0.001f;
    const float2v q1 = {0.999f, 0.998f}, q2 = {0.001f, 0.002f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
#define X(ID, NAME, ASM) \
    if (OP == ID) asm volatile(ASM : "+v"(v[m & 7]), "+v"(p[m & 7]) : "v"(c1), "v"(c2), "v"(q1), "v"(q2) : "vcc");
            OPS(X)
#undef X
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static double run(int nw) {
    const int iters = 4000, nwg = 256;
    float* out;
    hipMalloc(&out, sizeof(float) * nwg * nw * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<OP><<<nwg, nw * 64>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<OP><<<nwg, nw * 64>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms * 1e6 / (64.0 * iters * (nw / 4));          // ns per wave-instruction per SIMD
}

int main() {
    for (int nw : {4, 8, 16}) {
        const double fma = run<0>(nw);
        const double ghz = 2.0 / fma;                       // v_fma_f32 = 2 cycles (MI355X_MICROARCH.md)
        printf("--- %d wave(s) per SIMD; v_fma_f32 %.3f ns per wave-instruction -> clock %.2f GHz if it is 2 cycles\n", nw / 4, fma, ghz);
#define X(ID, NAME, ASM) printf("%-38s %6.2f cycles\n", NAME, run<ID>(nw) * ghz);
        OPS(X)
#undef X
    }
    return 0;
}
