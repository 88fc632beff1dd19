// probe.hip - micro-benchmarks of the MFMA / LDS / barrier mix used by the GEMM main loop.
// Test infrastructure of the library itself (tools/mfma_rate.py); not on the product path.
#include "../../vidit-q_amd/csrc/vq_common.h"

// mode bit0: 13 ds_read_b128 per iteration; bit1: workgroup barrier per iteration;
// bit2: use 32x32x32 MFMAs (9 per iteration) instead of 36 16x16x64
template <int MODE>
__global__ __launch_bounds__(512) void mfma_rate_kernel(int iters, int* out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 512) reinterpret_cast<int4v*>(smem)[i] = int4v{i, i + 1, i + 2, i + 3};
    __syncthreads();
    int4v acc[36];
#pragma unroll
    for (int j = 0; j < 36; ++j) acc[j] = int4v{0, 0, 0, 0};
    int4v xf[4], wf[9];
#pragma unroll
    for (int i = 0; i < 4; ++i) xf[i] = int4v{lane, i, 1, 2};
#pragma unroll
    for (int j = 0; j < 9; ++j) wf[j] = int4v{lane, j, 3, 4};
    const uint8_t* base = smem + (lane & 15) * 64 + (lane >> 4) * 16;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
            const uint8_t* b = base + (it & 1) * 65536;
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const int4v*>(b + i * 1024);
#pragma unroll
            for (int j = 0; j < 9; ++j) wf[j] = *reinterpret_cast<const int4v*>(b + 16384 + j * 1024);
            if (MODE & 8) __builtin_amdgcn_sched_barrier(0);   // loads stay in one burst ahead of the MFMAs
        }
        if (MODE & 4) {
            int16v* a16 = reinterpret_cast<int16v*>(acc);
#pragma unroll
            for (int j = 0; j < 9; ++j)
                a16[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], xf[j & 3], a16[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 9; ++j)
                a16[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], xf[(j + 1) & 3], a16[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 9; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j * 4 + i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], xf[i], acc[j * 4 + i], 0, 0, 0);
        }
        if (MODE & 2) {
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    }
    int x = 0;
#pragma unroll
    for (int j = 0; j < 36; ++j) x ^= acc[j][0] ^ acc[j][1] ^ acc[j][2] ^ acc[j][3];
    if (x == 0x7fffffff) out[tid] = x;
}

// Register-file placement probe: the same 13 ds_read_b128 + 36 MFMA 16x16x64 per iteration, with
//   RF 0: fragments in AGPRs (ds_read ... a[..]; MFMA srcA/srcB = AGPR), accumulators in VGPRs
//   RF 1: fragments in VGPRs, accumulators in AGPRs
//   RF 2: everything in VGPRs, inline asm (control for the asm form itself)
// Question: is the ~14 MFMA-cycles cost of a fragment read a VGPR port conflict between the LDS return
// and the MFMA operand traffic?
template <int RF>
__global__ __launch_bounds__(512) void mfma_rf_kernel(int iters, int* out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 512) reinterpret_cast<int4v*>(smem)[i] = int4v{i, i + 1, i + 2, i + 3};
    __syncthreads();
    int4v acc[36];
#pragma unroll
    for (int j = 0; j < 36; ++j) acc[j] = int4v{0, 0, 0, 0};
    int4v f[13];
    const uint32_t base = (uint32_t)((lane & 15) * 64 + (lane >> 4) * 16);
    for (int it = 0; it < iters; ++it) {
        const uint32_t b = base + (it & 1) * 65536;
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            if (RF == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(f[i]) : "v"(b), "i"(i * 1024));
            else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i]) : "v"(b), "i"(i * 1024));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (RF == 0)
                    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[j * 4 + i]) : "a"(f[4 + j]), "a"(f[i]));
                else if (RF == 1)
                    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[j * 4 + i]) : "v"(f[4 + j]), "v"(f[i]));
                else
                    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[j * 4 + i]) : "v"(f[4 + j]), "v"(f[i]));
            }
    }
    int x = 0;
#pragma unroll
    for (int j = 0; j < 36; ++j) {
        int4v v;
        if (RF == 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[0]) : "a"(acc[j][0]));
        else v = acc[j];
        x ^= v[0];
    }
    if (x == 0x7fffffff) out[tid] = x;
}

// staging-rate probe: every workgroup streams (256+288) rows x 64 B per iteration out of `src`
// (row stride `stride` bytes, k advances 64 B per iteration) either by LDS-DMA (mode 0) or by
// global_load_dwordx4 + ds_write_b128 (mode 1); mode 2 = DMA with two batches in flight.
template <int MODE>
__global__ __launch_bounds__(512) void stage_rate_kernel(const uint8_t* src, int stride, int iters, int* out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rowbase = (blockIdx.x % 64) * 256;         // 64 distinct token panels, 4 blocks share each
    uint32_t soff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int p = wave + i * 8;
        const int r = p * 16 + (lane >> 2);
        const int row = r < 256 ? rowbase + r : 16384 + (r - 256);   // "weights" after the tokens
        soff[i] = (uint32_t)row * (uint32_t)stride + (lane & 3) * 16;
    }
    int x = 0;
    for (int it = 0; it < iters; ++it) {
        const int k = (it % (stride / 64)) * 64;
        uint8_t* stage = smem + (it & 1) * 34816;
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int p = wave + i * 8;
                if (p < 34)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + soff[i] + k),
                                                     (void __attribute__((address_space(3)))*)(stage + p * 1024), 16, 0, 0);
            }
            if (MODE == 0) {
                __syncthreads();
            } else {
                if (wave < 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        } else {
            int4v v[5];
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (wave + i * 8 < 34) v[i] = *reinterpret_cast<const int4v*>(src + soff[i] + k);
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (wave + i * 8 < 34) *reinterpret_cast<int4v*>(stage + (wave + i * 8) * 1024 + lane * 16) = v[i];
            __syncthreads();
        }
        x ^= *reinterpret_cast<const int*>(smem + ((it & 1) ^ 1) * 34816 + tid * 4);
    }
    if (x == 0x7fffffff) out[tid] = x;
}

// combined probe: per k-tile  DMA (0 none / 1 LDS-DMA, 2 batches in flight / 2 global_load_dwordx4 to
// registers, discarded) + RD fragment reads (0/1) + MF MFMA 16x16x64 (0/1): which pairs overlap?
template <int DMA, int RD, int MF>
__global__ __launch_bounds__(512) void combo_rate_kernel(const uint8_t* src, int stride, int iters, int* out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rowbase = (blockIdx.x % 64) * 256;
    uint32_t soff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int p = wave + i * 8;
        const int r = p * 16 + (lane >> 2);
        const int row = r < 256 ? rowbase + r : 16384 + (r - 256);
        soff[i] = (uint32_t)row * (uint32_t)stride + (lane & 3) * 16;
    }
    int4v acc[36];
#pragma unroll
    for (int j = 0; j < 36; ++j) acc[j] = int4v{0, 0, 0, 0};
    int4v xf[4], wfr[9], sink = int4v{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) xf[i] = int4v{lane, i, 1, 2};
#pragma unroll
    for (int j = 0; j < 9; ++j) wfr[j] = int4v{lane, j, 3, 4};
    const int fo = (lane & 15) * 64 + (lane >> 4) * 16;
    auto issue = [&](int it) {
        const int k = (it % (stride / 64)) * 64;
        uint8_t* stage = smem + (it % 4) * 34816;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int p = wave + i * 8;
            if (p < 34) {
                if (DMA == 1)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + soff[i] + k),
                                                     (void __attribute__((address_space(3)))*)(stage + p * 1024), 16, 0, 0);
                else if (DMA == 2) {
                    const int4v v = *reinterpret_cast<const int4v*>(src + soff[i] + k);
                    sink[0] ^= v[0] ^ v[1] ^ v[2] ^ v[3];
                }
            }
        }
    };
    if (DMA == 1) {
        issue(0);
        issue(1);
    }
    for (int it = 0; it < iters; ++it) {
        if (DMA) issue(it + 2);
        if (DMA == 1) {
            if (wave < 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const uint8_t* st = smem + (it % 4) * 34816 + fo;
        if (RD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const int4v*>(st + i * 1024);
#pragma unroll
            for (int j = 0; j < 9; ++j) wfr[j] = *reinterpret_cast<const int4v*>(st + 16384 + j * 1024);
        }
        if (MF) {
#pragma unroll
            for (int j = 0; j < 9; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j * 4 + i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wfr[j], xf[i], acc[j * 4 + i], 0, 0, 0);
        } else if (RD) {
#pragma unroll
            for (int j = 0; j < 9; ++j) sink[1] ^= wfr[j][0] ^ xf[j & 3][1];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    int x = sink[0] ^ sink[1];
#pragma unroll
    for (int j = 0; j < 36; ++j) x ^= acc[j][0] ^ acc[j][1] ^ acc[j][2] ^ acc[j][3];
    if (x == 0x7fffffff) out[tid] = x;
}

// DMA-depth probe: stream 256 "token" rows + 288 "weight" rows, BKB bytes of k per batch, by LDS-DMA
// with DEPTH batches in flight (counted vmcnt), no consumer.  Reports how the L2 -> LDS fill rate of a
// CU depends on bytes in flight and on half-line (64 B) vs full-line (128 B) row chunks.
template <int BKB, int DEPTH>
__global__ __launch_bounds__(512) void dma_depth_kernel(const uint8_t* src, int stride, int iters, int* out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NP = 544 * BKB / 1024, PPW = (NP + 7) / 8, PL = NP - (PPW - 1) * 8;
    constexpr int RPP = 1024 / BKB, CPR = BKB / 16;      // rows per piece, 16-byte chunks per row
    constexpr int STG = 544 * BKB, NST = (163840 / STG) < DEPTH + 1 ? (163840 / STG) : DEPTH + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rowbase = (blockIdx.x % 64) * 256;
    const bool fullw = wave < PL || (NP % 8 == 0);
    uint32_t soff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave + i * 8;
        const int r = p * RPP + lane / CPR;
        const int row = r < 256 ? rowbase + r : 16384 + (r - 256);
        soff[i] = (uint32_t)row * (uint32_t)stride + (lane % CPR) * 16;
    }
    for (int it = 0; it < iters; ++it) {
        const int k = (it % (stride / BKB)) * BKB;
        uint8_t* stage = smem + (it % NST) * STG;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + i * 8;
            if (p < NP)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + soff[i] + k),
                                                 (void __attribute__((address_space(3)))*)(stage + p * 1024), 16, 0, 0);
        }
        if (fullw) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((DEPTH - 1) * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"i"((DEPTH - 1) * (PPW - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int x = *reinterpret_cast<const int*>(smem + tid * 4);
    if (x == 0x7fffffff) out[tid] = x;
}

extern "C" int vq_probe_stage_rate(int mode, const void* src, int stride, int iters, int blocks, int* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = 3 * 34816;
#define GO(M)                                                                                                   \
    {                                                                                                           \
        auto k = stage_rate_kernel<M>;                                                                          \
        static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                             \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        (void)e;                                                                                                \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, st, (const uint8_t*)src, stride, iters, out);       \
    }
    if (mode >= 400 && mode < 500) {
#define DGO(B, D)                                                                                               \
    {                                                                                                           \
        auto k = dma_depth_kernel<B, D>;                                                                        \
        static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                             \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 163840);          \
        (void)e;                                                                                                \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 163840, st, (const uint8_t*)src, stride, iters, out);    \
    }
        switch (mode) {
            case 411: DGO(64, 1) break;
            case 412: DGO(64, 2) break;
            case 413: DGO(64, 3) break;
            case 421: DGO(128, 1) break;
            case 422: DGO(128, 2) break;
            case 441: DGO(32, 1) break;
            case 442: DGO(32, 2) break;
            case 444: DGO(32, 4) break;
            case 446: DGO(32, 6) break;
            default: return VQ_EUNSUP;
        }
#undef DGO
        return vq_check_launch();
    }
    if (mode == 0) GO(0) else if (mode == 1) GO(1) else if (mode == 2) GO(2)
    else if (mode >= 100) {
        const size_t lds4 = 4 * 34816;
#define CGO(D, R, M)                                                                                            \
    {                                                                                                           \
        auto k = combo_rate_kernel<D, R, M>;                                                                    \
        static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                             \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);       \
        (void)e;                                                                                                \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds4, st, (const uint8_t*)src, stride, iters, out);      \
    }
        switch (mode) {
            case 111: CGO(1, 1, 1) break;
            case 101: CGO(1, 0, 1) break;
            case 110: CGO(1, 1, 0) break;
            case 100: CGO(1, 0, 0) break;
            case 211: CGO(2, 1, 1) break;
            case 201: CGO(2, 0, 1) break;
            case 200: CGO(2, 0, 0) break;
            case 311: CGO(0, 1, 1) break;
            case 301: CGO(0, 0, 1) break;
            default: return VQ_EUNSUP;
        }
#undef CGO
    } else return VQ_EUNSUP;
#undef GO
    return vq_check_launch();
}

extern "C" int vq_probe_mfma_rate(int mode, int iters, int blocks, int* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = 131072;
#define GO(M)                                                                                                   \
    {                                                                                                           \
        auto k = mfma_rate_kernel<M>;                                                                           \
        static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                             \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        (void)e;                                                                                                \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, st, iters, out);                                    \
    }
    switch (mode) {
        case 0: GO(0) break;
        case 1: GO(1) break;
        case 2: GO(2) break;
        case 3: GO(3) break;
        case 4: GO(4) break;
        case 5: GO(5) break;
        case 7: GO(7) break;
        case 9: GO(9) break;
        case 11: GO(11) break;
        case 20: { auto k = mfma_rf_kernel<0>; (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, st, iters, out); } break;
        case 21: { auto k = mfma_rf_kernel<1>; (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, st, iters, out); } break;
        case 22: { auto k = mfma_rf_kernel<2>; (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, st, iters, out); } break;
        default: return VQ_EUNSUP;
    }
#undef GO
    return vq_check_launch();
}
