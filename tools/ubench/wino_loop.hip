// Microbenchmark of the two conv GEMM cores of dan_kernels.hip in isolation (LDS-resident read, weights from L2):
// SIMD-level cycles per MFMA of conv_gemm (direct, 3 taps) and conv_gemm_wino, at 1 and 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/wino_loop.hip -o tools/ubench/wino_loop.bin
#include "../../dl4vc_amd/csrc/dan_kernels.hip"
#include <cstdio>
#include <vector>
using namespace dan;

// Experimental (not in the product): Winograd GEMM with TWO channel tiles per wave and half the position tiles, so one
// input transform / LDS row feeds 32 MFMAs instead of 16.  CNT = tiles of this wave (4 or 3).
template <int CNT>
__device__ __forceinline__ void conv_gemm_wino2(v4f (&acc)[4][4][2], const float* xrow, gv4f_ptr wl, int m0) {
    const float* x0 = xrow + 4 * m0 * LDS_S;
    v4f xa = *(const v4f*)(x0), xb = *(const v4f*)(x0 + 2 * LDS_S), xc = *(const v4f*)(x0 + 4 * LDS_S), xd = *(const v4f*)(x0 + 6 * LDS_S);
    float neg1 = -1.f;
    asm volatile("" : "+v"(neg1));
    const v2f m1 = {neg1, neg1};
    for (int g = 0; g < KGC; ++g) {
        v4f a[4][2];                                        // single-buffered: the SIMD's other wave covers the L2 latency
        const int gn = (g + 1 < KGC) ? g + 1 : g;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int n = 0; n < 2; ++n) a[k][n] = wl[(size_t)(k * KGC + g) * (KGC * 64) + n * 64];
        const float* xg = x0 + g * 16;
        const float* xn = x0 + gn * 16;
#pragma unroll
        for (int m = 0; m < CNT; ++m) {
            v4f v[4];
            v[0] = pk_sub(xa, xc, m1); v[1] = xb + xc; v[2] = pk_sub(xc, xb, m1); v[3] = pk_sub(xb, xd, m1);
            if (m + 1 < CNT) {
                xa = xc; xb = xd;
                xc = *(const v4f*)(xg + (4 * m + 8) * LDS_S);
                xd = *(const v4f*)(xg + (4 * m + 10) * LDS_S);
            } else {
                xa = *(const v4f*)(xn); xb = *(const v4f*)(xn + 2 * LDS_S);
                xc = *(const v4f*)(xn + 4 * LDS_S); xd = *(const v4f*)(xn + 6 * LDS_S);
            }
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][k][n] = mfma16(a[k][n][s], v[k][s], acc[m][k][n]);
            __builtin_amdgcn_s_setprio(0);
        }
    }
}

template <int WHICH>
__global__ __launch_bounds__(SEG_THREADS, 2) void k(const float* wl, float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float xs[LDS_ROWS * LDS_S];
    __shared__ __attribute__((aligned(16))) float cst[MAX_LAYERS * CST_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kk = lane >> 4;
    for (int i = tid; i < LDS_ROWS * LDS_S; i += blockDim.x) xs[i] = (float)((i * 7) & 15) * 0.0625f;
    for (int i = tid; i < MAX_LAYERS * CST_FLOATS; i += blockDim.x) cst[i] = 0.5f;
    __syncthreads();
    float s = 0.f;
    unsigned long long t0 = 0, t1 = 0;
    if (WHICH == 0) {
        const int cq = wave & 3, ph = (wave >> 2) & 1;
        const int m_base = ph * MTW, cnt = ph ? MT - MTW : MTW;
        gv4f_ptr w_conv = (gv4f_ptr)(wl + W_OFF) + (cq * NT) * 64 + lane;
        v4f acc[MTW][NT], pre[NT];
        for (int m = 0; m < MTW; ++m) for (int n = 0; n < NT; ++n) acc[m][n] = splat(0.f);
        for (int n = 0; n < NT; ++n) pre[n] = w_conv[n * 64];
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) conv_gemm(acc, xs, w_conv, pre, KGC, 3, 2, lane, m_base, cnt);
        t1 = __builtin_amdgcn_s_memtime();
        for (int m = 0; m < MTW; ++m) for (int n = 0; n < NT; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    } else if (WHICH == 2) {
        const int cq = wave & 3, ph = (wave >> 2) & 1;
        const int wP0 = wino_base(lane);
        gv4f_ptr w_w = (gv4f_ptr)(wl + WW_OFF) + (cq * 2) * 64 + lane;
        v4f acc[4][4][2];
        for (int m = 0; m < 4; ++m) for (int q = 0; q < 4; ++q) for (int n = 0; n < 2; ++n) acc[m][q][n] = splat(0.f);
        const float* xrow = xs + (HALO + wP0 - 2) * LDS_S + kk * 4;
        t0 = __builtin_amdgcn_s_memtime();
        if (ph == 0) for (int it = 0; it < iters; ++it) conv_gemm_wino2<4>(acc, xrow, w_w, 0);
        else for (int it = 0; it < iters; ++it) conv_gemm_wino2<3>(acc, xrow, w_w, 4);
        t1 = __builtin_amdgcn_s_memtime();
        for (int m = 0; m < 4; ++m) for (int q = 0; q < 4; ++q) for (int n = 0; n < 2; ++n) s += acc[m][q][n][0] + acc[m][q][n][1] + acc[m][q][n][2] + acc[m][q][n][3];
    } else {
        const int wP0 = wino_base(lane);
        gv4f_ptr w_w = (gv4f_ptr)(wl + WW_OFF) + wave * 64 + lane;
        v4f acc[MW][4], pre[4];
        for (int m = 0; m < MW; ++m) for (int q = 0; q < 4; ++q) acc[m][q] = splat(0.f);
        for (int q = 0; q < 4; ++q) pre[q] = w_w[(size_t)q * KGC * (KGC * 64)];
        const float* xrow = xs + (HALO + wP0 - 2) * LDS_S + kk * 4;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) conv_gemm_wino(acc, xrow, w_w, pre);
        t1 = __builtin_amdgcn_s_memtime();
        for (int m = 0; m < MW; ++m) for (int q = 0; q < 4; ++q) s += acc[m][q][0] + acc[m][q][1] + acc[m][q][2] + acc[m][q][3];
    }
    out[blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = t0; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = t1; }
}

template <int W>
void run(const char* name, int threads, const float* d_wl, double mfma_per_simd_per_iter_2w) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 2 * 8);
    const int iters = 50;
    hipLaunchKernelGGL(k<W>, dim3(256), dim3(threads), 0, 0, d_wl, out, iters, cyc);
    hipLaunchKernelGGL(k<W>, dim3(256), dim3(threads), 0, 0, d_wl, out, iters, cyc);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256 * 8 * 2]; (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double span = 0, fast = 0, slow = 0;
    for (int b = 0; b < 256; ++b) {
        unsigned long long lo = ~0ull, hi = 0, best = ~0ull, worst = 0;
        for (int w = 0; w < nw; ++w) {
            const unsigned long long a = h[(b * 8 + w) * 2], e = h[(b * 8 + w) * 2 + 1];
            lo = a < lo ? a : lo; hi = e > hi ? e : hi;
            best = (e - a) < best ? (e - a) : best; worst = (e - a) > worst ? (e - a) : worst;
        }
        span += (double)(hi - lo); fast += (double)best; slow += (double)worst;
        if (b == 0) { printf("    block 0 per-wave cycles/call:"); for (int w = 0; w < nw; ++w) printf(" %llu", (h[(b * 8 + w) * 2 + 1] - h[(b * 8 + w) * 2]) / iters); printf("\n"); }
    }
    const double per_simd = mfma_per_simd_per_iter_2w * (nw == 8 ? 1.0 : 0.5) * iters;
    printf("%-28s %d waves/SIMD: %.2f cycles per MFMA at the SIMD   (block span %.0f, fastest wave %.0f, slowest %.0f cycles per call)\n",
           name, nw / 4, span / 256 / per_simd, span / 256 / iters, fast / 256 / iters, slow / 256 / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    std::vector<float> wl(LAYER_STRIDE);
    for (size_t i = 0; i < wl.size(); ++i) wl[i] = (float)((i * 13) & 31) * 0.01f - 0.15f;
    float* d_wl; (void)hipMalloc(&d_wl, wl.size() * 4); (void)hipMemcpy(d_wl, wl.data(), wl.size() * 4, hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        // per SIMD with 2 waves: direct (7 + 6) tiles x 2 x 3 taps x 8 kg x 4 = 2496; wino 2 x 7 x 16 x 8 = 1792
        run<0>("direct 3-tap conv_gemm", threads, d_wl, threads == 512 ? 2496.0 : 2.0 * 1344.0);
        run<1>("conv_gemm_wino", threads, d_wl, 1792.0);
        if (threads == 512) run<2>("wino, 2 channel tiles per wave", threads, d_wl, 1792.0);
    }
    return 0;
}
