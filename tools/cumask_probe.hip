// tools/cumask_probe.hip -- can the HBM-bound tail kernels of chunk k run UNDER the MFMA-bound segment kernel of chunk k+1?
//
// Three questions, answered on the GPU box (hipcc --offload-arch=gfx950 -O3 -o gpurun_out/cumask_probe tools/cumask_probe.hip):
//   1. how do the bits of hipExtStreamCreateWithCUMask map to (XCC, SE, CU)?   (where_kernel under one-bit masks)
//   2. what HBM read rate do k CUs reach, k = 8 .. 64, spread evenly over the XCDs?   (stream_kernel under a k-CU mask)
//   3. does a matrix-core-bound kernel on the other 256 - k CUs keep its rate while the k CUs stream?   (mfma_kernel beside it)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void where_kernel(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc;
    }
}

// grid-stride read of n float4, eight 16-byte loads in flight per lane
__global__ __launch_bounds__(256) void stream_kernel(const f4* __restrict__ src, long long n, float* out) {
    f4 acc = {0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(src + i + j * stride);
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    for (; i < n; i += stride) { f4 v = src[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}

// persistent: one workgroup per CU (120 KB of LDS claimed), 8 waves, `iters` x 32 independent fp32 MFMAs each
__global__ __launch_bounds__(512) void mfma_kernel(int iters, float* out) {
    extern __shared__ float lds[];
    f4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = {0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.678f) { lds[threadIdx.x] = s; out[blockIdx.x] = lds[0]; }
}

static hipStream_t masked_stream(const std::vector<int>& cus) {
    uint32_t mask[8] = {0};
    for (int c : cus) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t s;
    CHK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    return s;
}

int main() {
    int n_cus = 0;
    CHK(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs: %d\n", n_cus);
    unsigned* d_where; CHK(hipMalloc(&d_where, 4096 * 8));
    float* d_out; CHK(hipMalloc(&d_out, 1 << 20));
    // ---- 1. the mapping of mask bits
    printf("-- mask bit -> where 256 one-wave blocks ran (distinct (xcc, se, cu) triples)\n");
    for (int bit : {0, 9}) {
        hipStream_t s = masked_stream({bit});
        where_kernel<<<256, 64, 0, s>>>(d_where);
        CHK(hipStreamSynchronize(s));
        std::vector<unsigned> w(512);
        CHK(hipMemcpy(w.data(), d_where, 512 * 4, hipMemcpyDeviceToHost));
        std::vector<unsigned> keys;
        for (int b = 0; b < 256; ++b) {
            const unsigned hw = w[2 * b], xcc = w[2 * b + 1] & 15;
            keys.push_back((xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15));
        }
        std::sort(keys.begin(), keys.end()); keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        printf("bit %3d:", bit);
        for (unsigned k : keys) printf(" (xcc %u se %u cu %u)", k >> 16, (k >> 8) & 255, k & 255);
        printf("\n");
        CHK(hipStreamDestroy(s));
    }
    // ---- 2. HBM read rate of k CUs
    const long long bytes = 8ll << 30, n4 = bytes / 16;
    f4* d_src; CHK(hipMalloc(&d_src, bytes)); CHK(hipMemset(d_src, 0, bytes));
    hipEvent_t e0, e1, f0, f1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreate(&f0)); CHK(hipEventCreate(&f1));
    auto pick = [&](int k, bool low) {                       // k CUs: the lowest k bits, or every (256 / k)-th bit
        std::vector<int> v;
        for (int i = 0; i < k; ++i) v.push_back(low ? i : i * (n_cus / k));
        return v;
    };
    printf("-- stream_kernel (8 GiB read) on k CUs, alone\n");
    for (bool low : {true})
        for (int k : {8, 16, 24, 32}) {
            hipStream_t s = masked_stream(pick(k, low));
            for (int wg_per_cu : {4, 8}) {
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    CHK(hipEventRecord(e0, s));
                    stream_kernel<<<k * wg_per_cu, 256, 0, s>>>(d_src, n4, d_out);
                    CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
                    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
                }
                printf("k %3d (%s) %d wg/cu: %8.2f ms  %7.1f GB/s  (%5.1f GB/s per CU)\n", k, low ? "low bits" : "spread  ", wg_per_cu, best,
                       bytes / best / 1e6, bytes / best / 1e6 / k);
            }
            CHK(hipStreamDestroy(s));
        }
    // ---- 3. the matrix-core kernel on the other CUs, alone and beside the stream
    printf("-- mfma_kernel on 256 - k CUs (one workgroup each), alone | beside stream_kernel on the k CUs\n");
    CHK(hipFuncSetAttribute((const void*)mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    const int iters = 40000;
    for (int k : {0, 8, 16, 32}) {
        std::vector<int> tails = pick(std::max(k, 1), true), conv;      // (symmetric: k / 8 CUs of every XCC)
        if (k == 0) tails.clear();
        for (int i = 0; i < n_cus; ++i) if (std::find(tails.begin(), tails.end(), i) == tails.end()) conv.push_back(i);
        hipStream_t sc = masked_stream(conv), st = k ? masked_stream(tails) : nullptr;
        for (int mode = 0; mode < (k ? 2 : 1); ++mode) {
            float best_c = 1e9, best_t = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipDeviceSynchronize());
                CHK(hipEventRecord(e0, sc));
                mfma_kernel<<<(int)conv.size(), 512, 120 * 1024, sc>>>(iters, d_out);
                CHK(hipEventRecord(e1, sc));
                if (mode == 1) {
                    CHK(hipEventRecord(f0, st));
                    stream_kernel<<<k * 8, 256, 0, st>>>(d_src, n4, d_out);
                    CHK(hipEventRecord(f1, st));
                }
                CHK(hipDeviceSynchronize());
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best_c) { best_c = ms; if (mode == 1) CHK(hipEventElapsedTime(&best_t, f0, f1)); }
            }
            const double flop = (double)conv.size() * 8 * iters * 32 * 2048.0;
            printf("k %2d %s: mfma %8.2f ms = %6.1f TF on %3zu CUs", k, mode ? "beside" : "alone ", best_c, flop / best_c / 1e9, conv.size());
            if (mode) printf("   stream %8.2f ms = %7.1f GB/s", best_t, bytes / best_t / 1e6);
            printf("\n");
        }
        CHK(hipStreamDestroy(sc)); if (st) CHK(hipStreamDestroy(st));
    }
    return 0;
}
