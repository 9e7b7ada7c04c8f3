// Why does ONE wave per SIMD walk the bf16 GEMM at ~60 cycles per MFMA inside the staggered kernel when the same walk alone reaches
// 33?  The kernel's own gemm_p on waves 0-3 (one per SIMD) while waves 4-7 (their SIMD partners) are: A gone, B parked at the
// workgroup barrier, C in a vector-instruction loop, D storing to LDS.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/gemm_p_lone.hip -o /tmp/gemm_p_lone.bin && /tmp/gemm_p_lone.bin
#include "../../dl4vc_amd/csrc/dan_kernels_bf16p.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace dan;
namespace dan {
void launch_fc(const float*, long long, const float*, long long, const float*, float*, long long, int, int, int, int, hipStream_t, float*, long long) {}
}
template <int MODE, int MT, int VALU_N = 300, int PRIO = 0>
__global__ __launch_bounds__(512, 2) void k(const char* wblk, float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[P_LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < P_LDS_BYTES / 16; i += blockDim.x) {
        bf8 v;
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)(float)((i * 8 + j) % 13 - 6);
        *(bf8*)(lds + (size_t)i * 16) = v;
    }
    __syncthreads();
    const int n = lane & 31, hh = lane >> 5, q = wave & 3;
    const int row0 = P_HALO + n;
    v16f acc[MT];
    for (int m = 0; m < MT; ++m) acc[m] = (v16f)(0.f);
    float junk = (float)tid;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (wave < 4) {
            int r = row0;
            asm volatile("" : "+v"(r));
            const unsigned xb0 = cell_addr(r - 2, hh), xb1 = cell_addr(r, hh), xb2 = cell_addr(r + 2, hh);
            gbf8p w = (gbf8p)(wblk + (size_t)(it % 7) * WP_LAYER_BYTES + WP_CONV_OFF) + q * 64 + lane;
            bf8 first[4];
            load_first(first, w);
            PFENCE();
            if (PRIO) __builtin_amdgcn_s_setprio(3);
            gemm_p<MT, 3, false>(acc, lds, xb0, xb1, xb2, w, first);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            PFENCE();
        } else if (MODE == 2) {
            float j0 = junk, j1 = junk + 1.f, j2 = junk + 2.f, j3 = junk + 3.f;       // four chains: ~VALU issue-bound, like an epilogue
            for (int j = 0; j < VALU_N / 4; ++j) { j0 = j0 * 1.0001f + 0.5f; j1 = j1 * 1.0002f + 0.5f; j2 = j2 * 1.0003f + 0.5f; j3 = j3 * 1.0004f + 0.5f; }
            junk = j0 + j1 + j2 + j3;
        } else if (MODE == 3) {
            bf8 v;
            for (int j = 0; j < 8; ++j) v[j] = (__bf16)junk;
            for (int j = 0; j < 40; ++j) *(bf8*)(lds + P_IMG_BYTES + ((size_t)(j * 256 + tid) * 16) % (P_IMG_BYTES - 16)) = v;
        }
        if (MODE >= 1) __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = junk;
    for (int m = 0; m < MT; ++m) for (int j = 0; j < 16; ++j) sum += acc[m][j];
    out[(size_t)blockIdx.x * blockDim.x + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE, int MT, int VALU_N = 300, int PRIO = 0>
static void run(const char* name, const char* dw, float* dout, unsigned long long* dcyc) {
    const int iters = 210, wgs = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, MT, VALU_N, PRIO>), dim3(wgs), dim3(512), 0, 0, dw, dout, iters, dcyc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<unsigned long long> c(wgs * 8);
    (void)hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> per;
    for (int b = 0; b < wgs; ++b) for (int w = 0; w < 4; ++w) per.push_back((double)c[b * 8 + w] / (iters * 24 * MT));
    std::sort(per.begin(), per.end());
    printf("%-52s MT=%d: %.2f cycles per MFMA (one wave per SIMD walking)\n", name, MT, per[per.size() / 2]);
}
int main() {
    char* dw; float* dout; unsigned long long* dcyc;
    std::vector<__bf16> hw((size_t)7 * WP_LAYER_BYTES / 2);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (__bf16)(float)((int)(i % 7) - 3);
    (void)hipMalloc(&dw, hw.size() * 2); (void)hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&dout, 256 * 512 * 4); (void)hipMalloc(&dcyc, 256 * 8 * 8);
    run<0, 5>("partners gone", dw, dout, dcyc);
    run<1, 5>("partners parked at the barrier", dw, dout, dcyc);
    run<2, 5, 300>("partners: 300 vector instructions, then barrier", dw, dout, dcyc);
    run<2, 5, 600>("partners: 600 vector instructions, then barrier", dw, dout, dcyc);
    run<2, 5, 1000>("partners: 1000 vector instructions, then barrier", dw, dout, dcyc);
    run<2, 5, 600, 1>("partners: 600 vector instructions; walker at s_setprio 3", dw, dout, dcyc);
    run<2, 5, 1000, 1>("partners: 1000 vector instructions; walker at s_setprio 3", dw, dout, dcyc);
    run<3, 5>("partners storing 40 x 16 B to LDS, then barrier", dw, dout, dcyc);
    run<0, 6>("partners gone", dw, dout, dcyc);
    run<1, 6>("partners parked at the barrier", dw, dout, dcyc);
    return 0;
}
