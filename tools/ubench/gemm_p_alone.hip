// The bf16 segment kernel's own GEMM walk (dan::gemm_p of dan_kernels_bf16p.hip) timed ALONE on every CU: 8 waves, the kernel's
// image layout and weight block, nothing else in flight.  Compare with mfma_bf16_feed.hip (the same walk written out by hand)
// and with the cadence the stamped kernel shows (tools/segp_probe.hip).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/gemm_p_alone.hip -o tools/ubench/gemm_p_alone.bin && tools/ubench/gemm_p_alone.bin
#include "../../dl4vc_amd/csrc/dan_kernels_bf16p.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace dan;
namespace dan {
void launch_fc(const float*, long long, const float*, long long, const float*, float*, long long, int, int, int, int, hipStream_t, float*, long long) {}
}
template <int TAPS>
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
    const int n = lane & 31, hh = lane >> 5, q = wave & 3, half = wave >> 2;
    const int row0 = P_HALO + half * 160 + n;
    v16f acc[5];
    for (int m = 0; m < 5; ++m) acc[m] = (v16f)(0.f);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        int r = row0;
        asm volatile("" : "+v"(r));
        const unsigned xb0 = cell_addr(r - 2, hh), xb1 = cell_addr(r, hh), xb2 = cell_addr(r + 2, hh);
        gbf8p w = (gbf8p)(wblk + (size_t)(it % 7) * WP_LAYER_BYTES + WP_CONV_OFF) + q * 64 + lane;
        bf8 first[4];
        load_first(first, w);
        PFENCE();
        gemm_p<5, TAPS, false>(acc, lds + (it & 1) * P_IMG_BYTES, xb0, xb1, xb2, w, first);
        PFENCE();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int m = 0; m < 5; ++m) for (int j = 0; j < 16; ++j) sum += acc[m][j];
    out[(size_t)blockIdx.x * blockDim.x + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int TAPS>
static void run(const char* name, const char* dw, float* dout, unsigned long long* dcyc) {
    const int iters = 210, wgs = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<TAPS>), dim3(wgs), dim3(512), 0, 0, dw, dout, iters, dcyc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<unsigned long long> c(wgs * 8);
    (void)hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> per;
    for (size_t i = 0; i < c.size(); ++i) per.push_back((double)c[i] / (iters * TAPS * 8 * 5));
    std::sort(per.begin(), per.end());
    printf("%-28s %.2f cycles per MFMA per wave = %.2f at the SIMD (incl. the first-fragment wait of every call)\n", name, per[per.size() / 2], per[per.size() / 2] / 2);
}
int main() {
    char* dw; float* dout; unsigned long long* dcyc;
    std::vector<__bf16> hw((size_t)7 * WP_LAYER_BYTES / 2);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (__bf16)(float)((int)(i % 7) - 3);
    (void)hipMalloc(&dw, hw.size() * 2); (void)hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&dout, 256 * 512 * 4); (void)hipMalloc(&dcyc, 256 * 8 * 8);
    run<3>("gemm_p<5,3> (conv)", dw, dout, dcyc);
    run<1>("gemm_p<5,1> (residual 1x1)", dw, dout, dcyc);
    return 0;
}
