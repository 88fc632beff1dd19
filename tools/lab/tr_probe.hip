// Semantics probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): prints which LDS elements each lane receives.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/tr_probe tools/lab/tr_probe.hip && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __fp16 h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(const _Float16* in, _Float16* out, int rs) {
    __shared__ __attribute__((aligned(16))) _Float16 sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, grp = l >> 4;
    // lane i of a 16-lane group points at row grp*4 + i/4, columns 4*(i%4)..+3 of a row-major image (row stride rs)
    const _Float16* p = sm + (grp * 4 + i / 4) * rs + 4 * (i % 4);
    h4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (_Float16)v[j];
}
int main() {
    std::vector<_Float16> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (_Float16)(float)(i % 2048);
    _Float16 *d, *o;
    hipMalloc(&d, 8192); hipMalloc(&o, 512);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    for (int rs : {64, 72}) {
        k<<<1, 64>>>(d, o, rs);
        std::vector<_Float16> r(256);
        hipMemcpy(r.data(), o, 512, hipMemcpyDeviceToHost);
        printf("row stride %d: lane -> (row, col) of its 4 elements\n", rs);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) { int e = (int)(float)r[l * 4 + j]; printf(" (%d,%d)", e / rs, e % rs); }
            printf("\n");
        }
    }
    return 0;
}
