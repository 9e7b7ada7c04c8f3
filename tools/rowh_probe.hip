// Times the half-read 3-tap launch of the training step (train_rowh_kernel) alone, on synthetic rows, for several start offsets
// of a CU's second workgroup -- and the whole-read kernel (train_row_kernel) beside it.  (The variants without loads / stores
// quoted in dan_train.hip were a temporary flag in the kernel.)  GPU box: bash tools/rowh_cycle.sh, or
//   hipcc -O3 --offload-arch=gfx950 -Idl4vc_amd/csrc tools/rowh_probe.hip dl4vc_amd/csrc/dan_train.o dl4vc_amd/csrc/dan_kernels.o -o /tmp/rowh_probe && /tmp/rowh_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "dan_train.h"
namespace dan {
template <bool SRC2, bool POOL, bool F2> __global__ void train_rowh_kernel(RowArgs a, int n_rows, int stagger);
extern template __global__ void train_rowh_kernel<false, false, false>(RowArgs, int, int);
extern template __global__ void train_rowh_kernel<true, false, false>(RowArgs, int, int);
__global__ void train_row_kernel(RowArgs a, int n_rows);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    using namespace dan;
    const int n_rows = argc > 1 ? atoi(argv[1]) : 6400, L = 201, R = 100;
    const size_t n = (size_t)n_rows * L * CPAD;
    float *src1, *src2, *out1, *w, *coef, *stats;
    CK(hipMalloc(&src1, n * 4)); CK(hipMalloc(&src2, n * 4)); CK(hipMalloc(&out1, n * 4));
    CK(hipMalloc(&w, (size_t)3 * KGC * KGC * 256 * 4)); CK(hipMalloc(&coef, 3 * CPAD * 4)); CK(hipMalloc(&stats, (size_t)2 * n_rows * 2 * CPAD * 4));
    std::vector<float> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20 & 1023) / 1024.f - 0.5f;
    for (size_t o = 0; o < n; o += h.size()) {
        const size_t m = std::min(h.size(), n - o);
        CK(hipMemcpy(src1 + o, h.data(), m * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(src2 + o, h.data(), m * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(w, h.data(), (size_t)3 * KGC * KGC * 256 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(coef, h.data(), 3 * CPAD * 4, hipMemcpyHostToDevice));
    RowArgs a{};
    a.R = R; a.L = L; a.mode = 1; a.src1 = src1; a.s1_stride = CPAD; a.w1 = w; a.taps = 3; a.kg = KGC; a.dil = 2; a.out1 = out1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int kind, int grid, int stagger, bool dgrad) -> int {
        RowArgs b = a;
        if (dgrad) { b.src2 = src2; b.coef = coef; b.mask_src2 = 1; } else { b.stats = stats; b.relu_out = 1; }
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            if (kind == 0) hipLaunchKernelGGL(train_row_kernel, dim3(grid), dim3(512), 0, 0, b, n_rows);
            else if (dgrad) hipLaunchKernelGGL((train_rowh_kernel<true, false, false>), dim3(grid), dim3(256), 0, 0, b, n_rows, stagger);
            else hipLaunchKernelGGL((train_rowh_kernel<false, false, false>), dim3(grid), dim3(256), 0, 0, b, n_rows, stagger);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) best = std::min(best, ms);
        }
        printf("%-44s %s  %.3f ms\n", name, dgrad ? "data gradient" : "forward      ", best);
        return 0;
    };
    for (int dg = 0; dg < 2; ++dg) {
        run("whole read, 256 workgroups", 0, 256, 0, dg);
        run("half read, 512 wgs, no offset", 1, 512, 0, dg);
        run("half read, 512 wgs, upper half +1 sleep", 1, 512, 1, dg);
        run("half read, 512 wgs, upper half +3", 1, 512, 3, dg);
        run("half read, 512 wgs, upper half +5", 1, 512, 5, dg);
        run("half read, 512 wgs, odd +3", 1, 512, 256 + 3, dg);
        run("half read, 512 wgs, odd +5", 1, 512, 256 + 5, dg);
        run("half read, 256 wgs (one per CU)", 1, 256, 0, dg);
    }
    return 0;
}
