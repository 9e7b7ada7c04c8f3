// Microbenchmark: what MFMA cadence does the bf16 segment kernel's k-major walk reach on one CU (8 waves, two per SIMD)?
//   v_mfma_f32_32x32x16_bf16, 5 accumulator tiles per wave, per MFMA one ds_read_b128 (B fragment, 5 MFMAs ahead), per 5 MFMAs one
//   16-byte global load per lane (A fragment from an L2-resident block, 4 steps ahead).
// Variants switch the feeds off one at a time:  LDS 0/1, W 0/1 (weights from global), WAVES 4/8.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_bf16_feed.hip -o tools/ubench/mfma_bf16_feed.bin && tools/ubench/mfma_bf16_feed.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) bf8* gbf8p;
constexpr int MT = 5, S = 24, NA = 4;

template <int LDS, int W, int NV = 0>
__global__ __launch_bounds__(512, 2) void k(const bf8* __restrict__ wts, float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[159744];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 159744 / 16; i += blockDim.x) {
        bf8 v;
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)(float)((i * 8 + j) % 13 - 6);
        *(bf8*)(lds + (size_t)i * 16) = v;
    }
    __syncthreads();
    const int n = lane & 31, hh = lane >> 5, q = wave & 3, half = wave >> 2;
    const int row0 = 4 + half * 160 + n;
    auto cell = [&](int row, int chunk) { return (unsigned)row * 256u + (unsigned)((chunk ^ row) & 15) * 16u; };
    const unsigned xb[3] = {cell(row0 - 2, hh), cell(row0, hh), cell(row0 + 2, hh)};
    gbf8p w = (gbf8p)wts + q * 64 + lane;
    v16f acc[MT];
    for (int m = 0; m < MT; ++m) acc[m] = (v16f)(0.f);
    bf8 a[NA], b[2][MT];
    float f[8];
    for (int j = 0; j < 8; ++j) f[j] = (float)(tid + j);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        for (int j = 0; j < NA; ++j) a[j] = w[(size_t)j * 4 * 64];
        for (int m = 0; m < MT; ++m) b[0][m] = *(const bf8*)(lds + xb[0] + m * 8192);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int sn = s + 1;
            const unsigned xa = xb[(sn / 8) % 3] ^ (unsigned)((sn % 8) << 5);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s % NA], b[LDS ? (s & 1) : 0][m], acc[m], 0, 0, 0);
                if (LDS && sn < S) b[sn & 1][m] = *(const bf8*)(lds + xa + m * 8192);
#pragma unroll
                for (int v = 0; v < NV; ++v) f[v % 8] = f[v % 8] * 1.0001f + 0.5f;        // the SAME wave's vector work between its MFMAs
            }
            if (W && s + NA < S) a[s % NA] = w[(size_t)(s + NA) * 4 * 64];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (LDS && sn < S) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            }
            if (W && s + NA < S) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int j = 0; j < 8; ++j) sum += f[j];
    for (int m = 0; m < MT; ++m) for (int j = 0; j < 16; ++j) sum += acc[m][j];
    out[(size_t)blockIdx.x * blockDim.x + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int LDS, int W, int NV = 0>
static void run(const char* name, int threads, const bf8* dw, float* dout, unsigned long long* dcyc) {
    const int iters = 200, wgs = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<LDS, W, NV>), dim3(wgs), dim3(threads), 0, 0, dw, dout, iters, dcyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(wgs * 8);
    hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> per;
    const int waves = threads / 64;
    for (int b = 0; b < wgs; ++b) for (int w = 0; w < waves; ++w) per.push_back((double)c[b * 8 + w] / (iters * S * MT));
    std::sort(per.begin(), per.end());
    const double med = per[per.size() / 2];
    // s_memtime runs at 100 MHz on this part?  report raw ticks per MFMA per wave and per SIMD (waves / 4 share a SIMD)
    printf("%-34s %d waves: %.2f ticks per MFMA per wave = %.2f per MFMA at the SIMD\n", name, waves, med, med / (waves / 4));
}

int main() {
    bf8* dw; float* dout; unsigned long long* dcyc;
    std::vector<__bf16> hw((size_t)(S + NA) * 4 * 64 * 8);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (__bf16)(float)((int)(i % 7) - 3);
    hipMalloc(&dw, hw.size() * 2); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&dout, 256 * 512 * 4); hipMalloc(&dcyc, 256 * 8 * 8);
    run<0, 0>("no feeds", 512, dw, dout, dcyc);
    run<1, 0>("LDS reads", 512, dw, dout, dcyc);
    run<0, 1>("weight loads", 512, dw, dout, dcyc);
    run<1, 1>("LDS reads + weight loads", 512, dw, dout, dcyc);
    run<1, 1, 2>("feeds + 2 own vector instr per MFMA", 512, dw, dout, dcyc);
    run<1, 1, 4>("feeds + 4 own vector instr per MFMA", 512, dw, dout, dcyc);
    run<1, 1, 6>("feeds + 6 own vector instr per MFMA", 512, dw, dout, dcyc);
    run<1, 1, 2>("feeds + 2 own vector instr per MFMA", 256, dw, dout, dcyc);
    run<1, 1, 4>("feeds + 4 own vector instr per MFMA", 256, dw, dout, dcyc);
    run<1, 1, 6>("feeds + 6 own vector instr per MFMA", 256, dw, dout, dcyc);
    run<0, 0>("no feeds", 256, dw, dout, dcyc);
    run<1, 1>("LDS reads + weight loads", 256, dw, dout, dcyc);
    return 0;
}
