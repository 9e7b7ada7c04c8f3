// tools/coresident_probe.hip -- which workgroups get onto a CU BESIDE a persistent, matrix-core-bound workgroup that holds
// 120 KiB of LDS and 2 x ~216 registers per SIMD lane (the bf16x3 segment kernel's footprint)?  A streaming kernel of 256
// threads with 0 .. 36 KiB of LDS is launched on a second stream while the hog runs; its duration beside the hog against its
// duration alone says whether it ran (tools/cumask_probe.hip is the sibling experiment with CU masks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, int mode>
__global__ __launch_bounds__(512) void hog_kernel(int iters, float* out) {
    extern __shared__ float lds[];
    f4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = {0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if (mode == 2) { acc[j][0] = __builtin_fmaf(a, b, acc[j][0]); acc[j][1] = __builtin_fmaf(a, b, acc[j][1]); acc[j][2] = __builtin_fmaf(a, b, acc[j][2]); acc[j][3] = __builtin_fmaf(a, b, acc[j][3]); }
            else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
        if (mode == 1) __builtin_amdgcn_s_sleep(2);
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.678f) { lds[threadIdx.x] = s; out[blockIdx.x] = lds[0]; }
}

__global__ __launch_bounds__(256) void stream_kernel(const f4* __restrict__ src, long long n, float* out, int touch, int prio) {
    extern __shared__ float lds[];
    if (prio) __builtin_amdgcn_s_setprio(3);
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
    if (touch) lds[threadIdx.x] = acc.x;
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = touch ? lds[0] : 1.f;
}

int main() {
    float* d_out; CHK(hipMalloc(&d_out, 1 << 20));
    const long long bytes = 4ll << 30, n4 = bytes / 16;
    f4* d_src; CHK(hipMalloc(&d_src, bytes)); CHK(hipMemset(d_src, 0, bytes));
    hipStream_t sa, sb; CHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1, f0, f1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreate(&f0)); CHK(hipEventCreate(&f1));
    CHK(hipFuncSetAttribute((const void*)hog_kernel<50, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHK(hipFuncSetAttribute((const void*)hog_kernel<50, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHK(hipFuncSetAttribute((const void*)hog_kernel<50, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHK(hipFuncSetAttribute((const void*)hog_kernel<24, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHK(hipFuncSetAttribute((const void*)stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    auto run = [&](int hog_acc, int hog_lds, int co_lds, int wgs_per_cu, int mode, int prio) {
        float alone = 1e9, beside = 1e9, hog_ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(f0, sb));
            stream_kernel<<<256 * wgs_per_cu, 256, co_lds, sb>>>(d_src, n4, d_out, co_lds > 0, prio);
            CHK(hipEventRecord(f1, sb)); CHK(hipDeviceSynchronize());
            float ms; CHK(hipEventElapsedTime(&ms, f0, f1)); alone = std::min(alone, ms);
        }
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, sa));
            if (hog_acc == 24) hog_kernel<24, 0><<<256, 512, hog_lds, sa>>>(12000, d_out);
            else if (mode == 0) hog_kernel<50, 0><<<256, 512, hog_lds, sa>>>(6000, d_out);
            else if (mode == 1) hog_kernel<50, 1><<<256, 512, hog_lds, sa>>>(6000, d_out);
            else hog_kernel<50, 2><<<256, 512, hog_lds, sa>>>(6000, d_out);
            CHK(hipEventRecord(e1, sa));
            CHK(hipEventRecord(f0, sb));
            stream_kernel<<<256 * wgs_per_cu, 256, co_lds, sb>>>(d_src, n4, d_out, co_lds > 0, prio);
            CHK(hipEventRecord(f1, sb)); CHK(hipDeviceSynchronize());
            float ms; CHK(hipEventElapsedTime(&ms, f0, f1));
            if (ms < beside) { beside = ms; CHK(hipEventElapsedTime(&hog_ms, e0, e1)); }
        }
        printf("hog mode %d (0 mfma, 1 mfma + sleep, 2 valu), co-runner prio %d | hog: %2d accumulators, %6d B LDS | co-runner: %5d B LDS, %d wg/cu: alone %7.2f ms  beside %7.2f ms (hog %7.2f ms)  -> %s\n", mode, prio, hog_acc, hog_lds, co_lds, wgs_per_cu,
               alone, beside, hog_ms, beside < 0.6f * hog_ms ? "RAN beside the hog" : "waited for the hog");
    };
    for (int mode : {0, 1, 2})
        for (int prio : {0, 1})
            for (int co_lds : {0, 32768}) run(50, 122880, co_lds, 4, mode, prio);
    run(24, 65536, 0, 4, 0, 0);
    run(24, 0, 0, 4, 0, 0);
    run(24, 0, 0, 1, 0, 0);
    return 0;
}
