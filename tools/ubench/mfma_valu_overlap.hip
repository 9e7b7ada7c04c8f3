// tools/ubench/mfma_valu_overlap.hip -- do the MFMAs of one wave and the vector instructions of ANOTHER wave of the same SIMD overlap?
// One workgroup of eight waves per CU: waves 0-3 (one per SIMD) issue back-to-back independent 16x16x32 bf16 MFMAs, waves 4-7 issue
// independent packed fp32 FMAs.  Times of: MFMA waves alone, VALU waves alone, both -- at s_setprio (mfma, valu) in {0, 3}^2.
// And the same mix inside ONE wave (3 FMAs behind every MFMA), which is how the fp32 segment kernel hides its input transform.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int PA, int PB>
__global__ __launch_bounds__(512) void two_roles(int iters, int do_a, int do_b, float* out) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (!do_a) return;
        __builtin_amdgcn_s_setprio(PA);
        v4f acc[8];
        bf8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f); b[j] = (__bf16)1.f; }
        for (int j = 0; j < 8; ++j) acc[j] = (v4f){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 8; ++j) r += acc[j][0];
    } else {
        if (!do_b) return;
        __builtin_amdgcn_s_setprio(PB);
        v2f x[16];
        const v2f m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
        for (int j = 0; j < 16; ++j) x[j] = (v2f){threadIdx.x * 1e-3f + j, 1.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 6; ++rep)                      // 96 packed FMAs per iteration beside 32 MFMAs: 3 per MFMA
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = __builtin_elementwise_fma(x[j], m, c);
        }
        for (int j = 0; j < 16; ++j) r += x[j][0] + x[j][1];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

// the same mix in one instruction stream (all eight waves): 3 FMAs behind every MFMA
__global__ __launch_bounds__(512) void one_stream(int iters, int with_valu, float* out) {
    v4f acc[8];
    bf8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f); b[j] = (__bf16)1.f; }
    for (int j = 0; j < 8; ++j) acc[j] = (v4f){0, 0, 0, 0};
    v2f x[12];
    const v2f m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
    for (int j = 0; j < 12; ++j) x[j] = (v2f){threadIdx.x * 1e-3f + j, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)                          // (two waves per SIMD share the pipe: 16 MFMAs + 48 FMAs per wave and iteration)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
                if (with_valu) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) x[(j * 3 + k) % 12] = __builtin_elementwise_fma(x[(j * 3 + k) % 12], m, c);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
            }
    }
    float r = 0.f;
    for (int j = 0; j < 8; ++j) r += acc[j][0];
    for (int j = 0; j < 12; ++j) r += x[j][0] + x[j][1];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int PA, int PB>
static float run(int iters, int a, int b, float* d_out) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        two_roles<PA, PB><<<256, 512>>>(iters, a, b, d_out);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float* d_out; CHK(hipMalloc(&d_out, 1 << 16));
    const int iters = 20000;                                       // 640 k MFMAs x 16 cycles = 10.2 M cycles = 4.3 ms at 2.4 GHz
    printf("MFMA waves alone (one per SIMD):          %7.3f ms\n", run<0, 0>(iters, 1, 0, d_out));
    printf("VALU waves alone (one per SIMD):          %7.3f ms   (3 packed FMAs per MFMA of the other role)\n", run<0, 0>(iters, 0, 1, d_out));
    printf("both, s_setprio mfma 0 / valu 0:          %7.3f ms\n", run<0, 0>(iters, 1, 1, d_out));
    printf("both, s_setprio mfma 3 / valu 0:          %7.3f ms\n", run<3, 0>(iters, 1, 1, d_out));
    printf("both, s_setprio mfma 0 / valu 3:          %7.3f ms\n", run<0, 3>(iters, 1, 1, d_out));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int wv = 0; wv < 2; ++wv) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0));
            one_stream<<<256, 512>>>(iters, wv, d_out);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("one stream, two waves per SIMD, %s: %7.3f ms\n", wv ? "MFMA + 3 FMAs each" : "MFMAs only        ", best);
    }
    return 0;
}
