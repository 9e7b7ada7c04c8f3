// Times the pointwise launches of the training step (train_point_kernel through launch_train_row) alone, on synthetic rows,
// with parts of their work switched off by argument: which tensor or stage a launch's time follows.  GPU box:
// bash tools/rowh_cycle.sh point
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "dan_train.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    using namespace dan;
    const int n_rows = argc > 1 ? atoi(argv[1]) : 6400, L = 201, R = 100;
    const size_t n = (size_t)n_rows * L * CPAD, nh = (size_t)n_rows * L * HPAD;
    float *t[5], *hb[3], *w, *coef, *stats;
    for (auto& p : t) CK(hipMalloc(&p, n * 4));
    for (auto& p : hb) CK(hipMalloc(&p, nh * 4));
    CK(hipMalloc(&w, (size_t)KGC * KGC * 256 * 4)); CK(hipMalloc(&coef, 3 * CPAD * 4)); CK(hipMalloc(&stats, ((size_t)n_rows * L / 64 + 2) * 2 * CPAD * 4));
    std::vector<float> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20 & 1023) / 1024.f - 0.5f;
    for (auto p : t) for (size_t o = 0; o < n; o += h.size()) CK(hipMemcpy(p + o, h.data(), std::min(h.size(), n - o) * 4, hipMemcpyHostToDevice));
    for (auto p : hb) for (size_t o = 0; o < nh; o += h.size()) CK(hipMemcpy(p + o, h.data(), std::min(h.size(), nh - o) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h.data(), (size_t)KGC * KGC * 256 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(coef, h.data(), 3 * CPAD * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, const RowArgs& a, double gb) -> int {
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            launch_train_row(a, n_rows, nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) best = std::min(best, ms);
        }
        printf("%-72s %.3f ms  (%.2f GB -> %.2f TB/s)\n", name, best, gb, gb / best);
        return 0;
    };
    const double T = n * 4 / 1e9, Hh = nh * 4 / 1e9;
    RowArgs b{};
    b.R = R; b.L = L; b.mode = 1; b.s1_stride = CPAD;
    {   // forward, residual layer: x = W_r bn(a) + b_r + x_prev; h = relu(W_b x + b_b)
        RowArgs a = b; a.src1 = t[0]; a.coef = coef; a.w1 = w; a.taps = 1; a.kg = KGC; a.bias1 = coef; a.add1 = t[1]; a.out1 = t[2];
        a.w2 = w; a.bias2 = coef; a.out2 = hb[0];
        run("fwd residual: bn(a) -> 1x1 GEMM + x_prev -> x, bottleneck -> h", a, 3 * T + Hh);
        RowArgs c = a; c.w2 = nullptr; run("   without the bottleneck stage", c, 3 * T);
        c = a; c.add1 = nullptr; run("   without the addend x_prev", c, 2 * T + Hh);
        c = a; c.out1 = nullptr; run("   without the store of x", c, 2 * T + Hh);
        c = a; c.w1 = nullptr; run("   without the 1x1 GEMM (x = bn(a) + x_prev)", c, 3 * T + Hh);
        c = a; c.w1 = nullptr; c.w2 = nullptr; c.add1 = nullptr; run("   copy only: x = bn(a)", c, 2 * T);
        c = a; c.w1 = nullptr; c.add1 = nullptr; c.out1 = nullptr; run("   lazy form: h = bottleneck(bn(a)) only", c, T + Hh);
    }
    {   // backward: dn = W_r^T g, stats (sum dn, sum dn a)
        RowArgs a = b; a.src1 = t[0]; a.w1 = w; a.taps = 1; a.kg = KGC; a.out1 = t[2]; a.stats = stats; a.stat_aux = t[1];
        run("bwd dn = W_r^T g, stats with a", a, 3 * T);
        RowArgs c = a; c.stats = nullptr; c.stat_aux = nullptr; run("   without the statistics", c, 2 * T);
        c = a; c.out1 = nullptr; run("   without the store of dn", c, 2 * T);
    }
    {   // backward: g = du + W_b^T (dh * (h > 0)) [+ g_next], stats (sum g, sum g a)
        RowArgs a = b; a.src1 = hb[0]; a.s1_stride = HPAD; a.src2 = hb[1]; a.mask_src2 = 1; a.w1 = w; a.taps = 1; a.kg = 2; a.add1 = t[0]; a.out1 = t[2];
        a.stats = stats; a.stat_aux = t[1];
        run("bwd g = du + W_b^T (dh * (h > 0)), stats with a", a, 3 * T + 2 * Hh);
        RowArgs c = a; c.add2 = t[3]; run("   with the residual skip g_next as well", c, 4 * T + 2 * Hh);
        c = a; c.stats = nullptr; c.stat_aux = nullptr; run("   without the statistics", c, 2 * T + 2 * Hh);
    }
    return 0;
}
