// Times the HBM-bound tail kernels of the forward (read mean, final pool, highway compression; fp32 and bf16-input forms) alone,
// on synthetic buffers of one chunk, and prints the rate at which each reads its input.  GPU box: bash tools/rowh_cycle.sh tails
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "dan_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    using namespace dan;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto fill = [&](void* p, size_t bytes) -> int {
        std::vector<uint16_t> h(1 << 20);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(0x3c00 + ((i * 2654435761u) >> 22 & 0x3ff));      // bf16 / fp32-safe bit patterns
        for (size_t o = 0; o < bytes; o += h.size() * 2) CK(hipMemcpy((char*)p + o, h.data(), std::min(h.size() * 2, bytes - o), hipMemcpyHostToDevice));
        return 0;
    };
    for (int cfgi = 0; cfgi < 2; ++cfgi) {
        const int R = cfgi ? 128 : 64, L = cfgi ? 301 : 201, ns = cfgi ? 1024 : 2048, H = 32, NL = 7, C = 128;
        const size_t rows = (size_t)ns * R, n_y = rows * L * CPAD, n_h = (size_t)NL * rows * L * HPAD;
        const long long fs = ((2LL * C * L + (long long)NL * H * R) + 15) / 16 * 16;
        void *y, *hb; float *pool, *feat, *wc, *bc;
        CK(hipMalloc(&y, n_y * 4)); CK(hipMalloc(&hb, n_h * 4)); CK(hipMalloc(&pool, (size_t)ns * L * CPAD * 4)); CK(hipMalloc(&feat, (size_t)ns * fs * 4));
        CK(hipMalloc(&wc, (size_t)NL * L * 2 * 2 * 64 * 4 * 4)); CK(hipMalloc(&bc, NL * HPAD * 4));
        if (fill(y, n_y * 4) || fill(hb, n_h * 4) || fill(wc, (size_t)NL * L * 2 * 2 * 64 * 4 * 4) || fill(bc, NL * HPAD * 4)) return 1;
        auto run = [&](const char* name, double gb, auto&& f) -> int {
            float best = 1e9f;
            for (int it = 0; it < 4; ++it) {
                CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it) best = std::min(best, ms);
            }
            printf("%d x %d x %d sites  %-28s %8.3f ms  %6.1f GB -> %.2f TB/s\n", R, L, ns, name, best, gb, gb / best);
            return 0;
        };
        const long long hls = (long long)rows * L * HPAD, wcls = (long long)L * 2 * 2 * 64 * 4;
        run("read_mean (fp32 y)", n_y * 4 / 1e9, [&] { launch_read_mean((const float*)y, pool, ns, R, L, nullptr, nullptr); });
        run("final_pool (fp32 y)", n_y * 4 / 1e9, [&] { launch_final_pool((const float*)y, feat, fs, ns, R, L, C, nullptr, nullptr); });
        run("highway (fp32 h)", n_h * 4 / 1e9, [&] { launch_highway((const float*)hb, hls, wc, wcls, bc, feat, fs, 2 * C * L, ns, R, L, H, NL, nullptr, nullptr); });
        run("read_mean16 (bf16 y)", n_y * 2 / 1e9, [&] { launch_read_mean16((const uint16_t*)y, pool, ns, R, L, nullptr, nullptr); });
        run("final_pool16 (bf16 y)", n_y * 2 / 1e9, [&] { launch_final_pool16((const uint16_t*)y, feat, fs, ns, R, L, C, nullptr, nullptr); });
        run("highway16 (bf16 h)", n_h * 2 / 1e9, [&] { launch_highway16((const uint16_t*)hb, hls, wc, wcls, bc, feat, fs, 2 * C * L, ns, R, L, H, NL, nullptr, nullptr); });
        run("read_meanx (two bf16 planes)", n_y * 4 / 1e9, [&] { launch_read_meanx((const uint16_t*)y, pool, ns, R, L, nullptr, nullptr); });
        run("final_poolx (two bf16 planes)", n_y * 4 / 1e9, [&] { launch_final_poolx((const uint16_t*)y, feat, fs, ns, R, L, C, nullptr, nullptr); });
        CK(hipFree(y)); CK(hipFree(hb)); CK(hipFree(pool)); CK(hipFree(feat)); CK(hipFree(wc)); CK(hipFree(bc));
    }
    return 0;
}
