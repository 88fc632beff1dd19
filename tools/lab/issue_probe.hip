// issue_probe.hip - do MFMA and VALU / transcendental work overlap on ONE SIMD of gfx950?  (stand-alone: hipcc -O3
// --offload-arch=gfx950 -o issue_probe issue_probe.hip; prints cycles per loop iteration for each mix.)
// Every workgroup is NW waves on one CU (launch 256 workgroups); waves of a workgroup go round-robin over the 4 SIMDs,
// so NW = 4 is one wave per SIMD and NW = 16 four per SIMD.  mode (per wave, by role = (wave / 4) & 1 when split):
//   MFMA-only: 8 x v_mfma_f32_32x32x16_f16 on 4 independent accumulators         (8 x 32 = 256 pipe cycles)
//   VALU-only: NV x v_fma_f32 on 8 independent chains
//   EXP-only:  NE x v_exp_f32 on 8 independent chains
//   mixed:     after each MFMA, NV / 8 fmas (+ NE / 8 exps) in the same wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int NV, int NE, bool MF, bool SPLIT, int PRIO = 0, int NROLE = 2>
__global__ void probe(float* out, long long* cyc, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool role_m = !SPLIT || ((wave >> 2) % NROLE) == 0;     // split: SIMD-mates alternate MFMA-only / VALU-only (NROLE 3: one MFMA wave, two VALU waves)
    if (SPLIT && PRIO == 1 && !role_m) __builtin_amdgcn_s_setprio(1);
    if (SPLIT && PRIO == 2 && role_m) __builtin_amdgcn_s_setprio(1);
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f); b[e] = (_Float16)(e * 0.01f); }
    float16v acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    const float c1 = 0.999f, c2 = 0.001f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (SPLIT) {
        // round 6: role-specific loops (the per-step role tests of the generic loop below cost 16 scalar branches per iteration)
        if (role_m) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
#pragma unroll
                    for (int k = 0; k < NV / 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m + k) & 7]) : "v"(c1), "v"(c2));
#pragma unroll
                    for (int k = 0; k < NE / 8; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(m + k) & 7]));
                }
            }
        }
    } else
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MF && role_m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
            if (!SPLIT || !role_m) {
#pragma unroll
                for (int k = 0; k < NV / 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m + k) & 7]) : "v"(c1), "v"(c2));
#pragma unroll
                for (int k = 0; k < NE / 8; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(m + k) & 7]));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int NV, int NE, bool MF, bool SPLIT, int PRIO = 0, int NROLE = 2>
static void run(const char* name, int nw) {
    const int iters = 2000, nwg = 256;
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * nwg * nw * 64);
    hipMalloc(&cyc, sizeof(long long) * nwg * nw);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NV, NE, MF, SPLIT, PRIO, NROLE><<<nwg, nw * 64>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NV, NE, MF, SPLIT, PRIO, NROLE><<<nwg, nw * 64>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nwg * nw);
    hipMemcpy(h.data(), cyc, sizeof(long long) * nwg * nw, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : h) mean += x; mean /= h.size();
    if (SPLIT) {
        double mm = 0, mv = 0; int nm = 0, nv = 0;
        for (size_t i = 0; i < h.size(); ++i) { const int w = (int)(i % nw); if (((w >> 2) % NROLE) == 0) { mm += h[i]; ++nm; } else { mv += h[i]; ++nv; } }
        printf("    (MFMA waves %.1f ticks / iteration, VALU waves %.1f)\n", mm / nm / iters, mv / nv / iters);
    }
    // s_memtime ticks at 100 MHz on gfx950: report wall ns per iteration instead of ticks
    printf("%-44s waves/SIMD %d: %8.1f ns / iteration (event), counter %.1f ticks\n", name, nw / 4, ms * 1e6 / iters, mean / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int nw : {4, 8, 16}) {
        run<0, 0, true, false>("8 MFMA 32x32x16", nw);
        run<64, 0, false, false>("64 v_fma", nw);
        run<0, 32, false, false>("32 v_exp", nw);
        run<64, 0, true, false>("8 MFMA + 64 v_fma interleaved (same wave)", nw);
        run<0, 32, true, false>("8 MFMA + 32 v_exp interleaved (same wave)", nw);
        run<64, 32, true, false>("8 MFMA + 64 v_fma + 32 v_exp (same wave)", nw);
    }
    // round 6: SIMD-mates in OPPOSITE roles (waves 0-3 MFMA-only, 4-7 VALU-only, ...): max of the streams, or their sum?
    for (int nw : {8, 16}) {
        run<64, 0, true, true>("split: 8 MFMA | partner 64 v_fma", nw);
        run<0, 32, true, true>("split: 8 MFMA | partner 32 v_exp", nw);
        run<64, 32, true, true>("split: 8 MFMA | partner 64 v_fma + 32 v_exp", nw);
        run<32, 16, true, true>("split: 8 MFMA | partner 32 v_fma + 16 v_exp", nw);
        run<64, 32, true, true, 1>("split, VALU waves prio 1: 8 MFMA | 64 v_fma + 32 v_exp", nw);
        run<64, 32, true, true, 2>("split, MFMA waves prio 1: 8 MFMA | 64 v_fma + 32 v_exp", nw);
    }
    run<64, 32, true, true, 0, 3>("split 1 MFMA wave + 2 VALU waves (64 fma + 32 exp)", 12);
    run<64, 32, true, true, 1, 3>("split 1 MFMA wave + 2 VALU waves prio 1", 12);
    run<64, 32, true, true, 2, 3>("split 1 MFMA wave prio 1 + 2 VALU waves", 12);
    run<64, 32, false, false>("64 v_fma + 32 v_exp alone", 4);
    run<64, 32, false, false>("64 v_fma + 32 v_exp alone", 8);
    run<64, 32, false, false>("64 v_fma + 32 v_exp alone", 16);
    return 0;
}
