// What bounds the highway compression kernels (3.9-4.5 TB/s of h where the read-mean kernels reach 5.5-6.0)?  Variants of
// highway16_kernel (bf16 h, dan_kernels_bf16p.hip) on one chunk of synthetic h at BASELINE config 5's shape, each timed alone:
//   D   positions in flight per wave (4 in the product), RT row tiles of 16 reads per workgroup (4),
//   NOW no weight loads (one fragment set re-used: wrong results, shows what the L2 weight stream costs),
//   NOM no MFMAs (loads only, summed with integer ops so that they are not eliminated),
//   SEQ lanes of a 16-lane group read CONSECUTIVE 16-byte pieces of one row (coalesced 256-B runs; wrong operand layout: shows what
//       the 16-rows-per-instruction pattern costs).
// GPU box:  hipcc -O3 --offload-arch=gfx950 tools/highway_probe.hip -o /tmp/highway_probe && /tmp/highway_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 hbf8 __attribute__((ext_vector_type(8)));
constexpr int HPAD = 32;
__device__ __forceinline__ v4f mfma16b(hbf8 a, hbf8 b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int D, int RT, bool NOW, bool NOM, bool SEQ>
__global__ __launch_bounds__(512) void hw16(const uint16_t* __restrict__ h, long long hls, const hbf8* __restrict__ wc, long long wcls,
                                            float* __restrict__ out, int n_rows, int L) {
    __shared__ float part[8][RT][2][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int layer = blockIdx.y, row0 = blockIdx.x * (16 * RT);
    const size_t K = (size_t)L * HPAD;
    const int g_lo = (int)((long long)L * wave / 8), g_hi = (int)((long long)L * (wave + 1) / 8);
    const uint16_t* arow[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int row = min(row0 + 16 * i + (SEQ ? (lane >> 2) : r16), n_rows - 1);
        arow[i] = h + (size_t)layer * hls + (size_t)row * K + (SEQ ? (lane & 3) : kk) * 8;
    }
    const hbf8* wl = wc + (size_t)layer * wcls + lane;
    v4f acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i) { acc[i][0] = (v4f){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (v4f){0.f, 0.f, 0.f, 0.f}; }
    hbf8 ar[D][RT], bw[D][4];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int g = min(g_lo + d, g_hi - 1);
#pragma unroll
        for (int i = 0; i < RT; ++i) ar[d][i] = *(const hbf8*)(arow[i] + (size_t)g * HPAD);
#pragma unroll
        for (int j = 0; j < 4; ++j) bw[d][j] = wl[((size_t)(NOW ? 0 : g) * 4 + j) * 64];
    }
    for (int g = g_lo; g < g_hi; g += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            hbf8 av[RT], w4[4];
#pragma unroll
            for (int i = 0; i < RT; ++i) av[i] = ar[d][i];
#pragma unroll
            for (int j = 0; j < 4; ++j) w4[j] = bw[d][j];
            const int gn = min(g + d + D, g_hi - 1);
#pragma unroll
            for (int i = 0; i < RT; ++i) ar[d][i] = *(const hbf8*)(arow[i] + (size_t)gn * HPAD);
            if (!NOW) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bw[d][j] = wl[((size_t)gn * 4 + j) * 64];
            }
            if (g + d < g_hi) {
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    if (NOM) {
                        v4f t = __builtin_bit_cast(v4f, av[i]);
                        acc[i][0] += t; acc[i][1] += __builtin_bit_cast(v4f, w4[i & 3]);
                    } else {
                        acc[i][0] = mfma16b(av[i], w4[0], acc[i][0]);
                        acc[i][0] = mfma16b(av[i], w4[1], acc[i][0]);
                        acc[i][1] = mfma16b(av[i], w4[2], acc[i][1]);
                        acc[i][1] = mfma16b(av[i], w4[3], acc[i][1]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) part[wave][i][j][lane * 4 + jj] = acc[i][j][jj];
    __syncthreads();
    for (int idx = tid; idx < RT * 2 * 256; idx += 512) {
        const int i = idx >> 9, j = (idx >> 8) & 1, e = idx & 255;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sum += part[w][i][j][e];
        const int row = row0 + 16 * i + 4 * ((e >> 2) >> 4) + (e & 3), o = 16 * j + ((e >> 2) & 15);
        if (row < n_rows) out[((size_t)layer * n_rows + row) * 32 + o] = sum;
    }
}


// v2: the eight waves of a workgroup walk the SAME positions on different row tiles (wave = 16 reads, all positions: no cross-wave sum),
// so a position's four weight fragments (4 KiB) are fetched ONCE per workgroup -- wave w loads those of position 8 ph + w -- into
// a two-phase LDS ring and read from there by all eight; h: DH positions in flight per wave.
template <int DH>
__global__ __launch_bounds__(512) void hw16v2(const uint16_t* __restrict__ h, long long hls, const hbf8* __restrict__ wc, long long wcls,
                                              float* __restrict__ out, int n_rows, int L) {
    __shared__ __attribute__((aligned(16))) char ring[2][8][4][1024];       // [phase parity][position in phase][fragment][lane * 16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int layer = blockIdx.y, row0 = blockIdx.x * 128 + wave * 16;
    const size_t K = (size_t)L * HPAD;
    const int row = min(row0 + r16, n_rows - 1);
    const uint16_t* arow = h + (size_t)layer * hls + (size_t)row * K + kk * 8;
    const hbf8* wl = wc + (size_t)layer * wcls + lane;
    v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const int n_ph = (L + 7) / 8;
    hbf8 ar[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) ar[d] = *(const hbf8*)(arow + (size_t)min(d, L - 1) * HPAD);
    hbf8 wq[4];
    {   // phase 0's weights
        const int p = min(wave, L - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[j] = wl[((size_t)p * 4 + j) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) *(hbf8*)(&ring[0][wave][j][lane * 16]) = wq[j];
    }
    __syncthreads();
    for (int ph = 0; ph < n_ph; ++ph) {
        const int pn = min(8 * (ph + 1) + wave, L - 1);          // next phase: this wave's position
        if (ph + 1 < n_ph) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wq[j] = wl[((size_t)pn * 4 + j) * 64];
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const int p = 8 * ph + d;
            const hbf8 av = ar[(d) % DH];
            // refill the slot with position p + DH (DH is a multiple of 8: the slot index is compile-time)
            ar[d % DH] = *(const hbf8*)(arow + (size_t)min(p + DH, L - 1) * HPAD);
            if (p < L) {
                const hbf8 w0 = *(const hbf8*)(&ring[ph & 1][d][0][lane * 16]), w1 = *(const hbf8*)(&ring[ph & 1][d][1][lane * 16]);
                const hbf8 w2 = *(const hbf8*)(&ring[ph & 1][d][2][lane * 16]), w3 = *(const hbf8*)(&ring[ph & 1][d][3][lane * 16]);
                acc0 = mfma16b(av, w0, acc0); acc0 = mfma16b(av, w1, acc0);
                acc1 = mfma16b(av, w2, acc1); acc1 = mfma16b(av, w3, acc1);
            }
        }
        if (ph + 1 < n_ph) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *(hbf8*)(&ring[(ph + 1) & 1][wave][j][lane * 16]) = wq[j];
        }
        __syncthreads();
    }
    // C layout: row = 4 kk + jj (of the 16 reads), column o = r16
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int rr = row0 + 4 * kk + jj;
        if (rr < n_rows) { out[((size_t)layer * n_rows + rr) * 32 + r16] = acc0[jj]; out[((size_t)layer * n_rows + rr) * 32 + 16 + r16] = acc1[jj]; }
    }
}

int main() {
    const int R = 128, L = 301, ns = 1024, NL = 7, n_rows = ns * R;
    const size_t n_h = (size_t)NL * n_rows * L * HPAD;
    uint16_t* hb; hbf8* wc; float* out;
    CK(hipMalloc(&hb, n_h * 2)); CK(hipMalloc(&wc, (size_t)NL * L * 4 * 64 * 16)); CK(hipMalloc(&out, (size_t)NL * n_rows * 32 * 4));
    CK(hipMemset(hb, 0x3c, n_h * 2)); CK(hipMemset(wc, 0x3c, (size_t)NL * L * 4 * 64 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long hls = (long long)n_rows * L * HPAD, wcls = (long long)L * 4 * 64;
    auto run = [&](const char* name, auto kern, int rt) -> int {
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3((n_rows + 16 * rt - 1) / (16 * rt), NL), dim3(512), 0, 0, hb, hls, wc, wcls, out, n_rows, L);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) best = std::min(best, ms);
        }
        printf("%-44s %7.3f ms  -> %.2f TB/s of h\n", name, best, n_h * 2 / 1e9 / best);
        return 0;
    };
    run("product form (D 4, RT 4)", hw16<4, 4, false, false, false>, 4);
    run("D 8", hw16<8, 4, false, false, false>, 4);
    run("D 2", hw16<2, 4, false, false, false>, 4);
    run("RT 2 (32 reads per workgroup), D 8", hw16<8, 2, false, false, false>, 2);
    run("no weight loads", hw16<4, 4, true, false, false>, 4);
    run("no MFMAs", hw16<4, 4, false, true, false>, 4);
    run("no weight loads, no MFMAs", hw16<4, 4, true, true, false>, 4);
    run("consecutive pieces per row (coalesced), no W/M", hw16<4, 4, true, true, true>, 4);
    run("consecutive pieces, D 8, no W/M", hw16<8, 4, true, true, true>, 4);
    auto run2 = [&](const char* name, auto kern) -> int {
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3((n_rows + 127) / 128, NL), dim3(512), 0, 0, hb, hls, wc, wcls, out, n_rows, L);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) best = std::min(best, ms);
        }
        printf("%-44s %7.3f ms  -> %.2f TB/s of h\n", name, best, n_h * 2 / 1e9 / best);
        return 0;
    };
    run2("v2: weights once per workgroup via LDS, DH 8", hw16v2<8>);
    run2("v2, DH 16", hw16v2<16>);
    run2("v2, DH 24", hw16v2<24>);
    run2("v2, DH 32", hw16v2<32>);
    return 0;
}
