// Microbenchmark: how do VALU operand-forming ops and accumulator-chain order affect a v_mfma_f32_16x16x4_f32 stream?
// Mimics the Winograd conv loop of dan_kernels.hip (7 tiles x 4 GEMMs x 4 k-steps, B operands formed by VALU adds).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

// ORDER 0: k-major (4 dependent MFMAs in a row)   1: two chains interleaved   2: four chains interleaved
// VALU  0: none (B operands constant)  1: 16 VALU before the tile's MFMAs  2: one VALU after every MFMA (forms the next tile's V)
template <int ORDER, int VALU>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, unsigned long long* cyc, float seed) {
    const int tid = threadIdx.x;
    v4f acc[7][4], a[4], v[4], r[4], vn[4];
#pragma unroll
    for (int m = 0; m < 7; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][q] = (v4f){0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a[q] = (v4f){1.f, 0.5f, 0.25f, 2.f} * (seed + q);
        r[q] = (v4f){seed, seed * 2, seed * 3, seed * 4} + (float)(tid + q);
        v[q] = r[q];
        vn[q] = r[q];
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            if (VALU == 1) {
                v[0] = r[0] - r[2]; v[1] = r[1] + r[2]; v[2] = r[2] - r[1]; v[3] = r[1] - r[3];
                r[0] = r[2] * 0.5f; r[1] = r[3] * 0.5f;       // stand-in for the row rotation (keeps values changing)
                const v4f t = r[0]; r[0] = r[2]; r[2] = t; const v4f u = r[1]; r[1] = r[3]; r[3] = u;
            }
            if (ORDER == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[m][q] = MFMA(a[q][s], v[q][s], acc[m][q]);
            } else if (ORDER == 1) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[m][2 * h] = MFMA(a[2 * h][s], v[2 * h][s], acc[m][2 * h]);
                        acc[m][2 * h + 1] = MFMA(a[2 * h + 1][s], v[2 * h + 1][s], acc[m][2 * h + 1]);
                    }
                    if (VALU == 2) {
                        // 8 VALU forming two of the next tile's operands while this half's MFMAs run
                        vn[2 * h] = r[2 * h] - r[(2 * h + 2) & 3];
                        vn[2 * h + 1] = r[2 * h + 1] + r[(2 * h + 2) & 3];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                        }
                    }
                }
                if (VALU == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] = vn[q]; r[q] = vn[(q + 1) & 3]; }
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[m][q] = MFMA(a[q][s], v[q][s], acc[m][q]);
            }
            if (VALU == 1) {
                __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    v4f s = (v4f){0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < 7; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) s += acc[m][q];
    out[blockIdx.x * blockDim.x + tid] = s[0] + s[1] + s[2] + s[3];
    if ((tid & 63) == 0) { cyc[(blockIdx.x * 8 + (tid >> 6)) * 2] = t0; cyc[(blockIdx.x * 8 + (tid >> 6)) * 2 + 1] = t1; }
}

template <int O, int V>
void run(const char* name, int threads) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 2 * 8);
    const int iters = 1000;
    hipLaunchKernelGGL((k<O, V>), dim3(256), dim3(threads), 0, 0, out, iters, cyc, 1.0f);
    hipLaunchKernelGGL((k<O, V>), dim3(256), dim3(threads), 0, 0, out, iters, cyc, 1.0f);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256 * 8 * 2]; (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const int wps = threads / 256, nw = threads / 64;
    // SIMD-level: a block's span (first wave start .. last wave end) over the MFMAs one SIMD issued; and the fastest wave alone
    double span = 0, fast = 0;
    for (int b = 0; b < 256; ++b) {
        unsigned long long lo = ~0ull, hi = 0, best = ~0ull;
        for (int w = 0; w < nw; ++w) {
            lo = h[(b * 8 + w) * 2] < lo ? h[(b * 8 + w) * 2] : lo;
            hi = h[(b * 8 + w) * 2 + 1] > hi ? h[(b * 8 + w) * 2 + 1] : hi;
            const unsigned long long d = h[(b * 8 + w) * 2 + 1] - h[(b * 8 + w) * 2];
            best = d < best ? d : best;
        }
        span += (double)(hi - lo); fast += (double)best;
    }
    printf("%-58s %d waves/SIMD: %.2f cycles per MFMA at the SIMD (fastest wave alone: %.2f per own MFMA)\n", name, wps,
           span / 256 / iters / 112.0 / wps, fast / 256 / iters / 112.0);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int threads : {256, 512}) {
        run<0, 0>("k-major chains (4 dependent in a row), no VALU", threads);
        run<1, 0>("two chains interleaved, no VALU", threads);
        run<2, 0>("four chains interleaved, no VALU", threads);
        run<0, 1>("k-major chains, 24 VALU before each tile", threads);
        run<2, 1>("four chains interleaved, 24 VALU before each tile", threads);
        run<1, 2>("two chains interleaved, one VALU after every MFMA", threads);
    }
    return 0;
}
