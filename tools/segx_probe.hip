// Diagnostic build of the bf16x3 split segment kernel (dan_kernels_bf16x.hip) with per-phase s_memtime stamps (never shipped, never timed for
// throughput: stamps serialise the schedule -- read the SHARES, not the length).
//   hipcc -O3 --offload-arch=gfx950 -DDAN_STAMPS tools/segx_probe.hip -o tools/segx_probe.bin && tools/segx_probe.bin [l_begin l_end [L [R]]]
#include "../dl4vc_amd/csrc/dan_kernels_bf16x.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace dan;
namespace dan {   // (launch_fcx's combine step lives in dan_kernels.hip; the probe never calls it)
void launch_fc_combine(const float*, int, const float*, float*, long long, int, int, int, hipStream_t) {}
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int l_begin = argc > 2 ? atoi(argv[1]) : 2, l_end = argc > 2 ? atoi(argv[2]) : 7;
    const int L = argc > 3 ? atoi(argv[3]) : 201;
    const int R = argc > 4 ? atoi(argv[4]) : 64, sites = 128, layers = 7, nrows = sites * R, nwg = 256;
    std::vector<uint16_t> wl((size_t)layers * WX_LAYER_BYTES / 2);
    srand(1);
    for (auto& v : wl) v = (uint16_t)(0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15));     // small bf16 values of either sign
    char* d_wl;
    CK(hipMalloc(&d_wl, wl.size() * 2));
    CK(hipMemcpy(d_wl, wl.data(), wl.size() * 2, hipMemcpyHostToDevice));
    for (int l = 0; l < layers; ++l) CK(hipMemset(d_wl + (size_t)l * WX_LAYER_BYTES + WX_CST_OFF, 0, (CST_FLOATS + 32) * 4));
    const size_t ny = (size_t)nrows * 2 * L * CPAD;
    std::vector<uint16_t> y(ny);
    for (auto& v : y) v = (uint16_t)(0x3e00 + (rand() & 0xff));
    uint16_t* d_y;
    float *d_h, *d_pool, *d_emb, *d_pe;
    uint8_t* d_u8;
    CK(hipMalloc(&d_y, ny * 2));
    CK(hipMemcpy(d_y, y.data(), ny * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pool, (size_t)sites * L * CPAD * 4)); CK(hipMemset(d_pool, 0, (size_t)sites * L * CPAD * 4));
    const size_t hls = (size_t)nrows * L * HPAD;
    CK(hipMalloc(&d_h, hls * layers * 4));
    CK(hipMalloc(&d_emb, VOCAB * EMBED * 4)); CK(hipMemset(d_emb, 0, VOCAB * EMBED * 4));
    CK(hipMalloc(&d_pe, L * EMBED * 4)); CK(hipMemset(d_pe, 0, L * EMBED * 4));
    CK(hipMalloc(&d_u8, (size_t)nrows * L)); CK(hipMemset(d_u8, 1, (size_t)nrows * L));
    unsigned long long* d_st;
    const size_t nst = (size_t)nwg * NWAVE * 64;
    CK(hipMalloc(&d_st, nst * 8)); CK(hipMemset(d_st, 0, nst * 8));
#ifdef DAN_STAMPS
    CK(hipMemcpyToSymbol(HIP_SYMBOL(x3::g_xstamps), &d_st, sizeof d_st));
#endif
    SegmentXArgs a{};
    a.wl = d_wl; a.l_begin = l_begin; a.l_end = l_end; a.n_layers = layers; a.dil_mid = 2; a.dil_final = 2;
    a.res_mask = 0x70; a.has_hw = 1; a.R = R; a.L = L;
    a.reads = a.qual = a.strand = a.ref = a.ref_mask = a.var_mask = d_u8;
    a.emb = d_emb; a.pe = d_pe; a.y = d_y; a.pool = l_begin ? d_pool : nullptr; a.h = d_h; a.h_layer_stride = (long long)hls;
    a.tap = nullptr; a.tap_layer = -1; a.stagger = getenv("DAN_X_STAGGER") ? atoi(getenv("DAN_X_STAGGER")) : 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        launch_segmentx(a, sites, nwg, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("launch %d: %.3f ms for %d rows = %.1f us per row per CU\n", rep, ms, nrows, ms * 1e3 * 256 / nrows);
    }
    std::vector<unsigned long long> st(nst);
    CK(hipMemcpy(st.data(), d_st, nst * 8, hipMemcpyDeviceToHost));
    auto med = [&](int k0, int k1, int wave_sel) {
        std::vector<long long> d;
        for (int wg = 0; wg < nwg; ++wg)
            for (int w = 0; w < NWAVE; ++w) {
                if (wave_sel >= 0 && w != wave_sel) continue;
                const unsigned long long* s = &st[((size_t)wg * NWAVE + w) * 64];
                if (s[k0] && s[k1]) d.push_back((long long)(s[k1] - s[k0]));
            }
        if (d.empty()) return -1LL;
        std::sort(d.begin(), d.end());
        return d[d.size() / 2];
    };
    printf("segment [%d,%d) L=%d  median cycles of the third row of every workgroup\n", l_begin, l_end, L);
    printf("prologue (encode | DMA wait, barrier)  %8lld\n", med(0, 1, -1));
    for (int l = l_begin; l < l_end; ++l) {
        const int sb = 2 + (l - l_begin) * 8;
        const bool res = (a.res_mask >> l) & 1;
        printf("L%d  pre %6lld | conv %7lld w0 %7lld w7 %7lld | epi %6lld | deferred bottleneck w0 %6lld w7 %6lld | barrier wait w0 %6lld w7 %6lld | "
               "res: seed+store+bar+gemm %7lld | (pack+bar+)store+bar %6lld | own bottleneck %6lld\n",
               l + 1, med(l == l_begin ? 1 : sb - 2, sb, -1), med(sb, sb + 1, -1), med(sb, sb + 1, 0), med(sb, sb + 1, 7),
               med(sb + 7, sb + 2, -1), med(sb + 1, sb + 7, 0), med(sb + 1, sb + 7, 7), med(sb + 2, sb + 3, 0), med(sb + 2, sb + 3, 7),
               res ? med(sb + 3, sb + 4, -1) : 0LL, med(res ? sb + 4 : sb + 3, sb + 5, -1), med(sb + 5, sb + 6, -1));
    }
    printf("copy_out + next DMA issue   %8lld\n", med(62, 63, -1));
    printf("total                       %8lld\n", med(0, 63, -1));
    return 0;
}
