// gemm_half.h - the int8 GEMM as TWO co-resident workgroups per CU (lab variant 22; round-2 experiment, NOT in the product
// library: 10-25 % slower per launch than the full-tile kernel and 22.5 vs 24.2 steps/s in the step; full tiles are
// bit-identical to the product kernel, ragged tails were not debugged).
//
// The full-line kernel (gemm_wide.h) owns its CU: 238 VGPRs x 8 waves and 139 KB of LDS leave room for nothing else,
// so the prologue (first DMA batch, HBM latency), the dequant epilogue (VALU) and the store drain of a tile - 40 % of
// its time - overlap no matrix work at all.  Here a workgroup takes HALF the tile and half the CU:
//   tile 128 tokens x 288 channels, 8 waves (4 x 2), wave tile 32 x 144 -> 72 accumulator registers, <= 128 VGPRs;
//   stages of 64 bytes of k per row (one mfma_i32_16x16x64_i8 k-step), a ring of THREE stages (3 x 26 KB = 78 KB,
//   the same 78 KB the epilogue slabs need), so two workgroups fit the 160 KB of a CU;
// and the hardware interleaves the two: one's prologue / epilogue runs under the other's MFMA loop, and every SIMD
// holds four waves instead of two to cover the fragment-read stalls of the in-order waves.
// Costs accepted: 1.5 x the L2 -> LDS bytes per MAC (intensity 89 instead of 135 MAC/B), 64-byte DMA rows (request-bound
// at ~65 GB/s per CU, tools/dma_depth.py), twice the barriers per k.
// LDS rows are 64 B; the 16-byte chunk of k-group kg of row r sits at chunk position kg ^ g(r), g(r) = (-(r >> 2)) & 3
// on the row's index inside its 16-row MFMA block: with that every lane group of a ds_read_b128 (the hardware serves
// lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, +32) touches 64 distinct banks.
#pragma once
#include "../../vidit-q_amd/csrc/gemm_common.h"

template <int EPI>
__global__ __launch_bounds__(512, 4) void gemm_i8_pair_kernel(GemmArgs a) {
    constexpr int BM = 128, BN = 288, WAVES_M = 4, WAVES_N = 2, NW = 8;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N, TM = WTM / 16, TN = WTN / 16;
    constexpr int XP = BM / 16, WP = BN / 16, PIECES = XP + WP;      // 1 KiB DMA pieces: 16 rows x 64 B
    constexpr int PPW = (PIECES + NW - 1) / NW, PFULL = PIECES - (PPW - 1) * NW;   // waves < PFULL issue PPW pieces
    constexpr int STAGE = (BM + BN) * 64, NSTAGE = 3;
    constexpr int BARJ = TN - 3;       // barrier + next DMA after the last-but-two channel group of a stage
    static_assert(TM == 2 && TN == 9, "fragment rings below");

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int mt_, nt_;
    {
        const int MT_ = (a.M + BM - 1) / BM, NT_ = (a.N + BN - 1) / BN;
        int vb = blockIdx.x;
        if (a.nbatch > 1) {                            // batch-major grid: weight set = blockIdx / tiles
            const int bt = vb / (MT_ * NT_);
            vb -= bt * (MT_ * NT_);
            a.wq += (size_t)bt * a.bs_w;
            a.sw += (size_t)bt * a.bs_ch;
            a.zw += (size_t)bt * a.bs_ch;
            a.cs += (size_t)bt * a.bs_ch;
            if (a.bias) a.bias += (size_t)bt * a.bs_ch;
            a.out += (size_t)bt * a.bs_out;
        } else if (a.ngroups > 1) {                    // group-major grid; uniform selects
            const int g = vb / (MT_ * NT_);
            vb -= g * (MT_ * NT_);
            if (g > 0) {
                const bool g1 = g == 1;
                a.xq = g1 ? a.grp[0].xq : a.grp[1].xq;
                a.sx = g1 ? a.grp[0].sx : a.grp[1].sx;
                a.zx = g1 ? a.grp[0].zx : a.grp[1].zx;
                a.R = g1 ? a.grp[0].R : a.grp[1].R;
                a.wq = g1 ? a.grp[0].wq : a.grp[1].wq;
                a.sw = g1 ? a.grp[0].sw : a.grp[1].sw;
                a.zw = g1 ? a.grp[0].zw : a.grp[1].zw;
                a.cs = g1 ? a.grp[0].cs : a.grp[1].cs;
                a.bias = g1 ? a.grp[0].bias : a.grp[1].bias;
                a.out = g1 ? a.grp[0].out : a.grp[1].out;
            }
        }
        xcd_tile(vb, MT_, NT_, mt_, nt_);
    }
    const int m0 = mt_ * BM, n0 = nt_ * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const bool full_wave = wave < PFULL;               // wave-uniform: issues PPW pieces per stage, the others PPW - 1

    // DMA source offsets: lane -> (row of the piece, chunk position); the chunk fetched is position ^ g(row)
    uint32_t soff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave + i * NW;
        const int rr = lane >> 2, pos = lane & 3;
        const int ch = pos ^ ((-(rr >> 2)) & 3);
        if (p < XP) {
            int gm = m0 + p * 16 + rr;
            gm = gm < a.M ? gm : a.M - 1;
            soff[i] = (uint32_t)gm * (uint32_t)a.Kp + ch * 16;
        } else {
            int gn = n0 + (p - XP) * 16 + rr;
            gn = gn < a.N ? gn : a.N - 1;
            soff[i] = (uint32_t)gn * (uint32_t)a.Kp + ch * 16;
        }
    }
    // LDS-DMA through buffer loads: SGPR resource (x or w base), one VGPR byte offset per piece (constant over k), the
    // k offset in an SGPR - no per-stage 64-bit address arithmetic.  Issued through asm (M0 = LDS destination).
    auto mk_rsrc = [&](const void* base) {
        const unsigned long ba = (unsigned long)base;
        return int4v{(int)__builtin_amdgcn_readfirstlane((unsigned)ba),
                     (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu), (int)0xffffffffu, 0x00020000};
    };
    const int4v rs_x = mk_rsrc(a.xq), rs_w = mk_rsrc(a.wq);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)smem);
    auto issue = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + i * NW;
            if (p < PIECES) {
                const unsigned dst = lds0 + stage * STAGE + p * 1024;
                const int koff = kt * 64;
                if (p < XP)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(soff[i]), "s"(rs_x), "s"(koff)
                                 : "memory", "m0");
                else
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(soff[i]), "s"(rs_w), "s"(koff)
                                 : "memory", "m0");
            }
        }
    };

    int4v acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = int4v{0, 0, 0, 0};

    const int frow = lane & 15, fc = lane >> 4;
    const int fsw = (fc ^ ((-(frow >> 2)) & 3)) * 16;
    const int xf = (wm * WTM + frow) * 64 + fsw;
    const int wf = BM * 64 + (wn * WTN + frow) * 64 + fsw;
    auto ldx = [&](int stage, int i) { return *reinterpret_cast<const int4v*>(smem + stage * STAGE + xf + i * 16 * 64); };
    auto ldw = [&](int stage, int j) { return *reinterpret_cast<const int4v*>(smem + stage * STAGE + wf + j * 16 * 64); };

    const int nkt = a.Kp / 64;
    issue(0, 0);
    if (nkt > 1) {
        issue(1, 1);
        if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPW - 1) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int4v xa[TM], xb[TM] = {}, w[3];
#pragma unroll
    for (int i = 0; i < TM; ++i) xa[i] = ldx(0, i);
    w[0] = ldw(0, 0);
    w[1] = ldw(0, 1);

    // one k-step of 64 per stage; the W ring needs no rotation (TN % 3 == 0: the two fragments prefetched for the next
    // stage land in slots 0 and 1), the next stage's token fragments are read into xb and moved over at the end
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        const int nxt = cur == NSTAGE - 1 ? 0 : cur + 1;
        const int nn2 = nxt == NSTAGE - 1 ? 0 : nxt + 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (j == BARJ && more) {      // stage kt + 1 landed for everyone; stage kt - 1 is free
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (kt + 2 < nkt) issue(nn2, kt + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j + 2 < TN) w[(j + 2) % 3] = ldw(cur, j + 2);
            else if (more) w[(j + 2) % 3] = ldw(nxt, j + 2 - TN);
            if (more && j == TN - 2) xb[0] = ldx(nxt, 0);
            if (more && j == TN - 1) xb[1] = ldx(nxt, 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[j % 3], xa[i], acc[j][i], 0, 0, 0);
            if (j >= TN - 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[i] = xb[i];
        cur = nxt;
    }
    const float* gate_row = EPI == VQ_EPI_GATE_RESID ? ring_tile_gate_row<BM>(a, m0) : nullptr;
    __syncthreads();                                   // every wave is past its last fragment read: the slabs overlay the ring
    int tx = tid;
    asm volatile("" : "+v"(tx));       // opaque: the epilogue's lane-dependent addresses are computed HERE, not before the
                                       // main loop (where they would push its 100 live registers past the 128 of a
                                       // four-waves-per-SIMD kernel)
    ring_stage_params<BM, BN, WAVES_M, WAVES_N, 0>(a, smem, m0, n0, tx, gate_row);
    __syncthreads();
    ring_epilogue<BM, BN, WAVES_M, WAVES_N, EPI, 0>(a, smem, acc, m0, n0, nullptr, tx, gate_row != nullptr);
}

template <int EPI>
static int launch_gemm_pair_e(const GemmArgs& a, hipStream_t st) {
    constexpr int BM = 128, BN = 288;
    constexpr size_t RING = 3 * (size_t)(BM + BN) * 64;
    constexpr size_t EPIL = (size_t)8 * 32 * (144 * 2) + 4 * BN * 4 + 12 * BM;
    constexpr size_t LDS = RING > EPIL ? RING : EPIL;
    static_assert(2 * LDS <= 163840, "two workgroups per CU");
    const int MT = (a.M + BM - 1) / BM, NTl = (a.N + BN - 1) / BN;
    const int tiles = MT * NTl * (a.nbatch > 1 ? a.nbatch : a.ngroups > 1 ? a.ngroups : 1);
    auto k = gemm_i8_pair_kernel<EPI>;
    static hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);  // once
    if (e != hipSuccess) {
        g_vq_last_hip_error = (int)e;
        return VQ_ELAUNCH;
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(512), LDS, st, a);
    return vq_check_launch();
}

static int launch_gemm_pair(const GemmArgs& a, hipStream_t st) {
    switch (a.epilogue) {
        case VQ_EPI_NONE: return launch_gemm_pair_e<VQ_EPI_NONE>(a, st);
        case VQ_EPI_GELU: return launch_gemm_pair_e<VQ_EPI_GELU>(a, st);
        case VQ_EPI_GATE_RESID: return launch_gemm_pair_e<VQ_EPI_GATE_RESID>(a, st);
        default: return launch_gemm_pair_e<VQ_EPI_RESID>(a, st);
    }
}
