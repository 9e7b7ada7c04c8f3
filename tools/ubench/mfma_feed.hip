// Microbenchmark: does interleaving LDS fragment reads (and which kind) slow a v_mfma_f32_16x16x4_f32 stream?
// One workgroup per CU, 4 or 8 waves, 104 MFMAs per iteration with 13 x {8 MFMA, variant read}.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_feed.hip -o /tmp/mfma_feed && /tmp/mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[216 * 136];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 216 * 136; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    v4f acc[13];
    v4f b[13];
    const float* xrow = lds + (4 + (lane & 15)) * 136 + (lane >> 4) * 4;
#pragma unroll
    for (int m = 0; m < 13; ++m) { acc[m] = (v4f){0, 0, 0, 0}; b[m] = *(const v4f*)(xrow + m * 16 * 136); }
    v4f a = (v4f){1.f, 0.5f, 0.25f, 2.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const float* xn = xrow + (it & 7) * 16;
#pragma unroll
        for (int m = 0; m < 13; ++m) {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[m][s], acc[m], 0, 0, 0);
            if (VARIANT == 1) b[m] = *(const v4f*)(xn + m * 16 * 136);
            if (VARIANT == 2) { v2f lo = *(const v2f*)(xn + m * 16 * 136), hi = *(const v2f*)(xn + m * 16 * 136 + 2); b[m] = (v4f){lo[0], lo[1], hi[0], hi[1]}; }
            if (VARIANT == 3 && (m & 1)) b[m] = *(const v4f*)(xn + m * 16 * 136);      // half the reads
        }
#pragma unroll
        for (int m = 0; m < 13; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (VARIANT == 1 || (VARIANT == 3 && (m & 1))) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (VARIANT == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    v4f s = (v4f){0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < 13; ++m) s += acc[m];
    out[blockIdx.x * blockDim.x + tid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, int threads) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
    const int waves_per_simd = threads / 256;
    printf("%-34s %d waves/SIMD: %.2f cycles per MFMA (SIMD-level)\n", name, waves_per_simd, avg / iters / 52.0 / waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int threads : {256, 512}) {
        run<0>("no LDS reads", threads);
        run<1>("1 ds_read_b128 per 4 MFMA", threads);
        run<2>("2 ds_read_b64 per 4 MFMA", threads);
        run<3>("1 ds_read_b128 per 8 MFMA", threads);
    }
    return 0;
}
