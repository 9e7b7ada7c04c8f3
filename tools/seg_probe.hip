// Diagnostic build of the conv-stack segment kernel with per-phase s_memtime stamps (never shipped,
// never timed for throughput: stamps serialise the schedule -- read the SHARES, not the length).
//   hipcc -O3 --offload-arch=gfx950 -DDAN_STAMPS tools/seg_probe.hip -o /tmp/seg_probe && /tmp/seg_probe [l_begin l_end [0 [wino [L [table 0|1]]]]]
#include "../dl4vc_amd/csrc/dan_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace dan;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int l_begin = argc > 2 ? atoi(argv[1]) : 2, l_end = argc > 2 ? atoi(argv[2]) : 7;
    const int precision = argc > 3 ? atoi(argv[3]) : 0;
    const int wino = argc > 4 ? atoi(argv[4]) : 0;
    const int L = argc > 5 ? atoi(argv[5]) : 201;
    const int R = 64, sites = 64, layers = 7, nwg = sites * R;
    std::vector<float> wl((size_t)layers * LAYER_STRIDE);
    srand(1);
    for (auto& v : wl) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    float *d_wl, *d_y, *d_pool, *d_h, *d_emb, *d_pe;
    uint8_t* d_u8;
    CK(hipMalloc(&d_wl, wl.size() * 4));
    CK(hipMemcpy(d_wl, wl.data(), wl.size() * 4, hipMemcpyHostToDevice));
    const size_t ny = (size_t)nwg * L * CPAD;
    std::vector<float> y(ny);
    for (auto& v : y) v = rand() / (float)RAND_MAX - 0.5f;
    CK(hipMalloc(&d_y, ny * 4));
    CK(hipMemcpy(d_y, y.data(), ny * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pool, (size_t)sites * L * CPAD * 4));
    CK(hipMemcpy(d_pool, y.data(), (size_t)sites * L * CPAD * 4, hipMemcpyHostToDevice));
    const size_t hls = (size_t)nwg * L * HPAD;
    CK(hipMalloc(&d_h, hls * layers * 4));
    CK(hipMalloc(&d_emb, VOCAB * EMBED * 4)); CK(hipMemset(d_emb, 0, VOCAB * EMBED * 4));
    CK(hipMalloc(&d_pe, L * EMBED * 4)); CK(hipMemset(d_pe, 0, L * EMBED * 4));
    CK(hipMalloc(&d_u8, (size_t)nwg * L)); CK(hipMemset(d_u8, 1, (size_t)nwg * L));
    unsigned long long* d_st;
    const size_t nst = (size_t)nwg * NWAVE * NSTAMP;
    CK(hipMalloc(&d_st, nst * 8)); CK(hipMemset(d_st, 0, nst * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof d_st));
    SegmentArgs a{};
    a.wl = d_wl; a.l_begin = l_begin; a.l_end = l_end; a.n_layers = layers; a.dil_mid = 2; a.dil_final = 2;
    a.res_mask = 0x70; a.has_hw = 1; a.R = R; a.L = L;
    a.reads = a.qual = a.strand = a.ref = a.ref_mask = a.var_mask = d_u8;
    a.emb = d_emb; a.pe = d_pe; a.y = d_y; a.pool = l_begin ? d_pool : nullptr; a.h = d_h; a.h_layer_stride = (long long)hls;
    a.tap = nullptr; a.tap_layer = -1; a.wino = wino;
    // (precision != 0 selected the round-3 bf16 kernel, deleted in round 4: the bf16 families have probes of their own, segx / segp)
    if (precision != 0) { printf("seg_probe covers the fp32 kernel only: use tools/segx_probe.hip / tools/segp_probe.hip\n"); return 1; }
    // layer 1 by table (dan_kernels.h L0_*): an all-zero table exercises the walk's instruction stream
    float* d_l0;
    CK(hipMalloc(&d_l0, l0_tab_floats(L) * 4)); CK(hipMemset(d_l0, 0, l0_tab_floats(L) * 4));
    a.l0_tab = argc > 6 && atoi(argv[6]) == 0 ? nullptr : d_l0;
    for (int rep = 0; rep < 3; ++rep) {
        launch_segment(a, sites, 0, 0);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> st(nst);
    CK(hipMemcpy(st.data(), d_st, nst * 8, hipMemcpyDeviceToHost));
    // median over workgroups/waves of each phase delta
    auto med = [&](int k0, int k1, int wave_sel) {
        std::vector<long long> d;
        for (int wg = 0; wg < nwg; ++wg)
            for (int w = 0; w < NWAVE; ++w) {
                if (wave_sel >= 0 && w != wave_sel) continue;
                const unsigned long long* s = &st[((size_t)wg * NWAVE + w) * NSTAMP];
                if (s[k0] && s[k1]) d.push_back((long long)(s[k1] - s[k0]));
            }
        if (d.empty()) return -1LL;
        std::sort(d.begin(), d.end());
        return d[d.size() / 2];
    };
    printf("segment [%d,%d)  median cycles (all waves | wave0 | wave3)\n", l_begin, l_end);
    printf("prologue (zero+load)        %8lld\n", med(0, 1, -1));
    for (int l = l_begin; l < l_end; ++l) {
        const int sb = 2 + (l - l_begin) * 8;
        printf("L%d  pre(const/init) %7lld | conv %7lld %7lld %7lld | epi %6lld | barrier %6lld | swap %6lld | res %7lld | write+bar %6lld | bottleneck %6lld %6lld %6lld\n",
               l + 1, med(l == l_begin ? 1 : sb - 1, sb, -1), med(sb, sb + 1, -1), med(sb, sb + 1, 0), med(sb, sb + 1, 3),
               med(sb + 1, sb + 2, -1), med(sb + 2, sb + 3, -1), med(sb + 3, sb + 4, -1), med(sb + 4, sb + 5, -1),
               med(((a.res_mask >> l) & 1) ? sb + 5 : sb + 3, sb + 6, -1), med(sb + 6, sb + 7, -1), med(sb + 6, sb + 7, 0), med(sb + 6, sb + 7, 3));
    }
    {   // per-wave view of the second layer of the segment: conv start relative to wave 0's, conv length, barrier wait
        const int l = l_begin + 1 < l_end ? l_begin + 1 : l_begin, sb = 2 + (l - l_begin) * 8;
        printf("L%d per wave (start vs wave 0 | conv | epilogue+barrier wait):", l + 1);
        for (int w = 0; w < NWAVE; ++w) {
            std::vector<long long> off, len, wait;
            for (int wg = 0; wg < nwg; ++wg) {
                const unsigned long long* s0 = &st[((size_t)wg * NWAVE + 0) * NSTAMP];
                const unsigned long long* sw = &st[((size_t)wg * NWAVE + w) * NSTAMP];
                if (s0[sb] && sw[sb] && sw[sb + 1] && sw[sb + 3]) {
                    off.push_back((long long)(sw[sb] - s0[sb])); len.push_back((long long)(sw[sb + 1] - sw[sb]));
                    wait.push_back((long long)(sw[sb + 3] - sw[sb + 1]));
                }
            }
            std::sort(off.begin(), off.end()); std::sort(len.begin(), len.end()); std::sort(wait.begin(), wait.end());
            if (!off.empty()) printf("  w%d %lld|%lld|%lld", w, off[off.size() / 2], len[len.size() / 2], wait[wait.size() / 2]);
        }
        printf("\n");
    }
    printf("copy_out                    %8lld\n", med(62, 63, -1));
    printf("total                       %8lld\n", med(0, 63, -1));
    return 0;
}
