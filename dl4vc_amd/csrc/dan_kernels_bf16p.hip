// Plain-bf16 conv-stack segment kernel for gfx950, "ping-pong" form (dan_config.precision = 2; BASELINE config 5:
// 128 reads x 301 bp).  dl4vc/model.py:728-778 for one read resident in LDS, as in the other families, but built around
// what bounded those at 8x the fp32 matrix rate:
//
//   * v_mfma_f32_32x32x16_bf16, wave = (channel quarter q = wave & 3) x (position half = wave >> 2): a 1-KiB ds_read_b128 of
//     activations feeds 16 k MACs, a quarter of the LDS bytes per MAC of the 16-channel-per-wave mapping (whose conv stage sat
//     on the LDS bandwidth limit and on the matrix pipe at once);
//   * TWO images of the read in LDS (2 x 312 rows x 256 B = 156 KiB): a layer reads one and writes the other, so there is one
//     barrier per layer instead of two, no in-place hazard, and a residual layer's 1x1 GEMM reads BN(ReLU(conv)) from the
//     second image while the layer input, still in the first, is each lane's own accumulator seed (x + b) and is then
//     overwritten by that lane alone;
//   * unpadded 256-byte rows, 16-byte chunk c of row r stored at chunk c ^ (r & 15): the 16 lanes of a ds_read_b128 group
//     touch 16 distinct rows -> all 64 banks once; the 8 lanes of a ds_write_b128 group likewise;
//   * the rows of every weight fragment are permuted on the host so that a lane's 16 accumulator registers are 16 CONSECUTIVE
//     output channels of one position: the epilogue is two 16-byte LDS stores per tile and lane;
//   * activations cross HBM as bf16 (y at segment ends, the bottleneck h of every layer): half the bytes of the fp32 spill
//     format for the kernel's prologue / copy-out and for the three HBM-bound reductions behind it;
//   * persistent workgroups (one per CU) over XCD-contiguous slices of whole sites: no per-row dispatch cost, the site's
//     pool image and the weights stay in one L2, LDS is zeroed once per workgroup.
//
// Numerics (the oracle's bf16 = "storage" mode, oracle/dan_oracle.py::conv_layer): GEMM operands bf16, sums fp32, bias / ReLU /
// BatchNorm / residual add in fp32, every stored activation rounded to bf16 (ties to even, v_cvt_pk_bf16_f32).
#include "dan_kernels.h"

namespace dan {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) bf8* gbf8p;

__device__ __forceinline__ v16f mfma32(bf8 a, bf8 b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float relu1(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()); }

__device__ __forceinline__ bf8 lds_read(const char* lds, unsigned addr) { return *(const bf8*)(lds + addr); }
__device__ __forceinline__ void lds_write(char* lds, unsigned addr, bf8 v) { *(bf8*)(lds + addr) = v; }
__device__ __forceinline__ unsigned cell_addr(int row, int chunk) { return (unsigned)row * P_ROW_BYTES + (unsigned)((chunk ^ row) & 15) * 16; }

// acc[m] += W(this wave's 32 channels) x X(32 positions of tile m) over TAPS x KS k-steps of 16 channels.
//   xb0..2: this lane's byte address of chunk (lane >> 5) of its row for tap t in the source image (tile 0); chunk
//           2 ks + (lane >> 5) lies at xb ^ (ks << 5), tile m a further m * 32 rows on.
//   w:      this wave's first fragment (+ lane); step (t, ks) is (t * P_KSC + ks) * 4 fragments on.
// A run-time loop over chunks of NA = 4 k-steps: the four weight fragments of the NEXT chunk are requested slot by slot as
// the current chunk's are consumed (an L2 round trip = 4 x MT MFMAs ahead), the activations of the next k-step while the
// current one's MFMAs issue (double-buffered registers, one ds_read_b128 behind each MFMA).
template <int MT, int TAPS, int KS>
__device__ __forceinline__ void gemm_p(v16f (&acc)[MT], const char* lds, unsigned xb0, unsigned xb1, unsigned xb2, gbf8p w) {
    constexpr int NA = 4;
    static_assert(KS % NA == 0, "k-steps per tap come in chunks of four");
    constexpr int CH = KS / NA, NCH = TAPS * CH;
    bf8 a[NA], b[2][MT];
#pragma unroll
    for (int j = 0; j < NA; ++j) a[j] = w[j * 4 * 64];
#pragma unroll
    for (int m = 0; m < MT; ++m) b[0][m] = lds_read(lds, xb0 + m * (32 * P_ROW_BYTES));
    unsigned xcur = xb0;
    int ks0 = 0;                                                 // first k-step of the current chunk within its tap
    gbf8p wp = w;
    for (int c = 0; c < NCH; ++c) {
        const int cn = c + 1;
        const bool last = cn == NCH;
        unsigned xnext = xcur;
        int ksn = ks0 + NA;
        gbf8p wn = wp + NA * 4 * 64;
        if (ksn == KS) {                                         // the next chunk opens the next tap
            ksn = 0;
            wn += (P_KSC - KS) * 4 * 64;
            if (TAPS == 3) xnext = (cn == CH) ? xb1 : xb2;
        }
        if (last) { wn = wp; xnext = xcur; ksn = ks0; }          // (uniform) nothing follows: re-request what is at hand
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const bf8 as = a[j];
            a[j] = wn[j * 4 * 64];
            const unsigned xa = (j + 1 < NA) ? (xcur ^ (unsigned)((ks0 + j + 1) << 5)) : (xnext ^ (unsigned)(ksn << 5));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = mfma32(as, b[j & 1][m], acc[m]);
                b[(j + 1) & 1][m] = lds_read(lds, xa + m * (32 * P_ROW_BYTES));
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        xcur = xnext; ks0 = ksn; wp = wn;
    }
}

// 16 fp32 values of one lane (16 consecutive channels of one position) -> two 16-byte chunks of bf16
__device__ __forceinline__ void pack16(const v16f& v, bf8& lo, bf8& hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo[j] = (__bf16)v[j]; hi[j] = (__bf16)v[8 + j]; }
}

template <int MT>
__global__ __launch_bounds__(SEG_THREADS, NWAVE / 4) void segmentp_kernel(SegmentPArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[P_LDS_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wave & 3, half = wave >> 2;
    const int n = lane & 31, hh = lane >> 5;
    const int L = a.L;
    const int pbase = half * (MT * 32);

    // zero both images and the tail once: the halo rows, the rows past the window and the bytes a last tile reads beyond its
    // image are never written with anything but zeros afterwards (see the epilogue's mask)
    for (int i = tid; i < P_LDS_BYTES / 16; i += SEG_THREADS) *(v4f*)(lds + (size_t)i * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // persistent walk: workgroups b and b + 8 share an XCD (round-robin dealing), so slice x = b & 7 of the rows -- whole sites --
    // belongs to the workgroups b = x, x + 8, ...: the reads of a site and the weights stay in one L2
    const int n_work = a.work_count ? *a.work_count : a.n_rows;
    const int slice = a.work_count ? (n_work + 7) / 8 : a.slice_rows;
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, nj = gridDim.x >> 3;
    for (int k = jw; k < slice; k += nj) {
        const int wk = xcd * slice + k;
        if (wk >= n_work) break;
        const int row_index = __builtin_amdgcn_readfirstlane(a.work_count ? a.work[wk] : wk);
        const int site = row_index / a.R;
        const size_t read_idx = (size_t)row_index;
        int cur = 0;                                             // image holding the current layer input
        {
            char* img = lds;                                     // image 0
            if (a.l_begin == 0) {
                // ---- encode (dl4vc/model.py:450-627), canonical 48-channel order, rounded to bf16
                const size_t rbase = read_idx * (size_t)L, sbase = (size_t)site * L;
                int ok_ref = 1, ok_var = 1;
                for (int p = tid; p < L; p += SEG_THREADS) {
                    const int tok = a.reads[rbase + p], rm = a.ref_mask[sbase + p], vm = a.var_mask[sbase + p];
                    ok_ref &= (rm == 0) || (tok == rm);
                    ok_var &= (vm == 0) || (tok == vm);
                }
                // workgroup-wide AND through eight flag words at the very end of the array (__syncthreads_and would add its own
                // LDS on top of the 160 KiB this kernel declares)
                int* flags = (int*)(lds + P_LDS_BYTES - 64);
                {
                    const int w_ref = __all(ok_ref), w_var = __all(ok_var);
                    if (lane == 0) { flags[wave] = w_ref; flags[8 + wave] = w_var; }
                }
                __syncthreads();
                int agree_ref = 1, agree_var = 1;
#pragma unroll
                for (int w8 = 0; w8 < NWAVE; ++w8) { agree_ref &= flags[w8]; agree_var &= flags[8 + w8]; }
                for (int p = tid; p < L; p += SEG_THREADS) {
                    const int tok = a.reads[rbase + p], qv = a.qual[rbase + p], st = a.strand[rbase + p];
                    const int rf = a.ref[sbase + p], rm = a.ref_mask[sbase + p], vm = a.var_mask[sbase + p];
                    const float* er = a.emb + min(tok, VOCAB - 1) * EMBED;
                    const float* ef = a.emb + min(rf, VOCAB - 1) * EMBED;
                    const float* pp = a.pe + p * EMBED;
                    float row[CIN0];
#pragma unroll
                    for (int e = 0; e < EMBED; ++e) { const float pv = pp[e]; row[e] = er[e] + pv; row[EMBED + e] = ef[e] + pv; }
                    row[40] = (float)qv * 0.01f;
                    row[41] = (float)st * 0.5f;
                    row[42] = (rm != 0 && agree_ref) ? 1.f : 0.f;
                    row[43] = (vm != 0 && agree_var) ? 1.f : 0.f;
                    row[44] = (rm != 0) ? 1.f : 0.f;
                    row[45] = row[46] = row[47] = 0.f;
                    const int r = P_HALO + p;
#pragma unroll
                    for (int c = 0; c < CIN0 / 8; ++c) {
                        bf8 v;
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = (__bf16)row[c * 8 + j];
                        lds_write(img, cell_addr(r, c), v);
                    }
                }
            } else {
                // ---- resume: x = bf16(y + pool) (model.py:742); the whole read and its site's pool image in flight at once
                const bf8* ysrc = (const bf8*)(a.y + read_idx * (size_t)L * CPAD);
                const v4f* psrc = a.pool ? (const v4f*)(a.pool + (size_t)site * L * CPAD) : nullptr;
                const int n16 = L * (CPAD / 8);
                constexpr int NPF = (P_LMAX * (CPAD / 8) + SEG_THREADS - 1) / SEG_THREADS;
                bf8 vy[NPF];
                v4f vp[NPF][2];
#pragma unroll
                for (int u = 0; u < NPF; ++u) {
                    const int i = tid + u * SEG_THREADS;
                    if (i < n16) vy[u] = ysrc[i];
                }
#pragma unroll
                for (int u = 0; u < NPF; ++u) {
                    const int i = tid + u * SEG_THREADS;
                    vp[u][0] = vp[u][1] = (v4f){0.f, 0.f, 0.f, 0.f};
                    if (psrc && i < n16) { vp[u][0] = psrc[2 * i]; vp[u][1] = psrc[2 * i + 1]; }
                }
#pragma unroll
                for (int u = 0; u < NPF; ++u) {
                    const int i = tid + u * SEG_THREADS;
                    if (i < n16) {
                        bf8 v;
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = (__bf16)((float)vy[u][j] + vp[u][j >> 2][j & 3]);
                        lds_write(img, cell_addr(P_HALO + (i >> 4), i & 15), v);
                    }
                }
            }
        }
        __syncthreads();

        auto copy_tap = [&](int img_i, int nch) {
            // image -> fp32 [L][CPAD] (debug tap)
            float* dst = a.tap + read_idx * (size_t)L * CPAD;
            const char* img = lds + img_i * P_IMG_BYTES;
            for (int i = tid; i < L * (CPAD / 8); i += SEG_THREADS) {
                const int p = i >> 4, c = i & 15;
                const bf8 v = lds_read(img, cell_addr(P_HALO + p, c));
                v4f o0, o1;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o0[j] = (c * 8 + j < nch) ? (float)v[j] : 0.f; o1[j] = (c * 8 + 4 + j < nch) ? (float)v[4 + j] : 0.f; }
                *(v4f*)(dst + (size_t)i * 8) = o0;
                *(v4f*)(dst + (size_t)i * 8 + 4) = o1;
            }
        };
        if (a.tap && a.tap_layer == 0 && a.l_begin == 0) copy_tap(0, CIN0);

        for (int l = a.l_begin; l < a.l_end; ++l) {
            const char* blk = a.wl + (size_t)l * WP_LAYER_BYTES;
            const float* cst = (const float*)(blk + WP_CST_OFF);
            const bool residual = (a.res_mask >> l) & 1u;
            const int dil = (l == 0) ? 1 : (l + 1 < a.n_layers ? a.dil_mid : a.dil_final);
            const char* src = lds + cur * P_IMG_BYTES;
            char* dst = lds + (cur ^ 1) * P_IMG_BYTES;
            const int c0 = 32 * q + 16 * hh;                      // this lane's 16 output channels
            const int row0 = P_HALO + pbase + n;                  // its row in tile 0

            v16f acc[MT];
            {
                v16f bias;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4f bv = *(const v4f*)(cst + CST_BIAS + c0 + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) bias[4 * g + j] = bv[j];
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = bias;
            }
            const unsigned xb0 = cell_addr(row0 - dil, hh), xb1 = cell_addr(row0, hh), xb2 = cell_addr(row0 + dil, hh);
            gbf8p wconv = (gbf8p)(blk + WP_CONV_OFF) + q * 64 + lane;
            // layer 1: 48 encoded channels as four k-steps (the fourth meets zero weights; whatever finite values an earlier
            // layer left in chunks 6, 7 of the image contribute exactly 0)
            if (l == 0) gemm_p<MT, 3, 4>(acc, src, xb0, xb1, xb2, wconv);
            else        gemm_p<MT, 3, P_KSC>(acc, src, xb0, xb1, xb2, wconv);

            // ---- epilogue: ReLU, BatchNorm (folded), rows past the window forced to zero, bf16, two 16-byte stores per tile
            const unsigned wa = cell_addr(row0, 4 * q + 2 * hh);
            {
                const float* cst2 = cst;
                asm volatile("" : "+s"(cst2));                   // (requested here, not ahead of the GEMM: 32 registers)
                v16f sc, sh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4f s4 = *(const v4f*)(cst2 + CST_SCALE + c0 + 4 * g), h4 = *(const v4f*)(cst2 + CST_SHIFT + c0 + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { sc[4 * g + j] = s4[j]; sh[4 * g + j] = h4[j]; }
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int p = pbase + 32 * m + n;
                    v16f v = acc[m];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = (p < L) ? relu1(v[i]) * sc[i] + sh[i] : 0.f;
                    bf8 lo, hi;
                    pack16(v, lo, hi);
                    if (p < P_LMAX + P_HALO) {
                        lds_write(dst, wa + m * (32 * P_ROW_BYTES), lo);
                        lds_write(dst, (wa ^ 16u) + m * (32 * P_ROW_BYTES), hi);
                    }
                }
            }
            __syncthreads();
            if (residual) {
                // y = Wr * t + bres + x   (model.py:753-761): t = the image just written, x = this lane's own cells of the input image
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bf8 xl = lds_read(src, wa + m * (32 * P_ROW_BYTES)), xh = lds_read(src, (wa ^ 16u) + m * (32 * P_ROW_BYTES));
#pragma unroll
                    for (int j = 0; j < 8; ++j) { acc[m][j] = (float)xl[j]; acc[m][8 + j] = (float)xh[j]; }
                }
                {
                    v16f br;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const v4f bv = *(const v4f*)(cst + CST_BRES + c0 + 4 * g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) br[4 * g + j] = bv[j];
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] += br;
                }
                gbf8p wres = (gbf8p)(blk + WP_RES_OFF) + q * 64 + lane;
                gemm_p<MT, 1, P_KSC>(acc, dst, xb1, xb1, xb1, wres);
                char* back = lds + cur * P_IMG_BYTES;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int p = pbase + 32 * m + n;
                    v16f v = acc[m];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = (p < L) ? v[i] : 0.f;
                    bf8 lo, hi;
                    pack16(v, lo, hi);
                    if (p < P_LMAX + P_HALO) {
                        lds_write(back, wa + m * (32 * P_ROW_BYTES), lo);
                        lds_write(back, (wa ^ 16u) + m * (32 * P_ROW_BYTES), hi);
                    }
                }
                __syncthreads();
            } else {
                cur ^= 1;
            }
            if (a.tap && a.tap_layer == l + 1) copy_tap(cur, CPAD);

            // ---- bottleneck h = relu(Wb * y + bb), 128 -> 32 (model.py:774): position tiles dealt over the waves; reads the image
            // the next layer's GEMM reads too, so no barrier follows it
            if (a.has_hw) {
                const char* img = lds + cur * P_IMG_BYTES;
                gbf8p wbot = (gbf8p)(blk + WP_BOT_OFF) + lane;
                uint16_t* hrow = a.h + (size_t)l * a.h_layer_stride + read_idx * (size_t)L * HPAD;
                bf8 wb[P_KSC];
#pragma unroll
                for (int ks = 0; ks < P_KSC; ++ks) wb[ks] = wbot[ks * 64];
                v16f bb;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4f bv = *(const v4f*)(cst + CST_BBOT + 16 * hh + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) bb[4 * g + j] = bv[j];
                }
                for (int tl = wave; tl * 32 < L; tl += NWAVE) {
                    const unsigned xa = cell_addr(P_HALO + 32 * tl + n, hh);
                    bf8 bx[P_KSC];
#pragma unroll
                    for (int ks = 0; ks < P_KSC; ++ks) bx[ks] = lds_read(img, xa ^ (unsigned)(ks << 5));
                    v16f hacc = bb;
#pragma unroll
                    for (int ks = 0; ks < P_KSC; ++ks) hacc = mfma32(wb[ks], bx[ks], hacc);
                    const int p = 32 * tl + n;
                    if (p < L) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) hacc[i] = relu1(hacc[i]);
                        bf8 lo, hi;
                        pack16(hacc, lo, hi);
                        bf8* o = (bf8*)(hrow + (size_t)p * HPAD + 16 * hh);
                        o[0] = lo;
                        o[1] = hi;
                    }
                }
            }
        }
        // ---- the segment's output -> y (bf16, [L][CPAD])
        {
            const char* img = lds + cur * P_IMG_BYTES;
            bf8* ydst = (bf8*)(a.y + read_idx * (size_t)L * CPAD);
            for (int i = tid; i < L * (CPAD / 8); i += SEG_THREADS) ydst[i] = lds_read(img, cell_addr(P_HALO + (i >> 4), i & 15));
        }
        __syncthreads();                                         // the next row re-uses both images
    }
}

bool segmentp_supports(int L, int l_begin, unsigned res_mask, bool has_pool) {
    if (L > P_LMAX) return false;
    // a residual layer that opens a resumed segment takes its residual from y BEFORE the pool add (model.py:732 vs :742); this
    // kernel seeds the residual from the LDS image, which there holds y + pool
    if (l_begin > 0 && has_pool && ((res_mask >> l_begin) & 1u)) return false;
    return true;
}

void launch_segmentp(const SegmentPArgs& a0, int n_sites, int n_cus, hipStream_t s) {
    SegmentPArgs a = a0;
    a.n_rows = n_sites * a.R;
    a.slice_rows = (n_sites + 7) / 8 * a.R;
    int wgs = n_cus > 0 ? n_cus : 256;
    wgs = (wgs + 7) / 8 * 8;
    const int need = ((a.slice_rows + 0) < 1 ? 1 : a.slice_rows) * 8;       // no more workgroups than rows per slice x 8
    if (wgs > need) wgs = need;
    hipLaunchKernelGGL((segmentp_kernel<5>), dim3((unsigned)wgs), dim3(SEG_THREADS), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// bf16-input forms of the HBM-bound reductions (fp32 sums in the same order as the fp32 forms)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void read_mean16_kernel(const bf8* __restrict__ y, v4f* __restrict__ pool, int R, int L,
                                                          const int* __restrict__ row_src) {
    const int n8 = L * (CPAD / 8);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int site = blockIdx.y;
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int* rs = row_src ? row_src + (size_t)site * R : nullptr;
#pragma unroll 8
    for (int r = 0; r < R; ++r) {
        const bf8 v = y[(size_t)(rs ? rs[r] : site * R + r) * n8 + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum[j] += (float)v[j];
    }
    const float inv = 1.f / (float)R;
    v4f o0, o1;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o0[j] = sum[j] / (float)R; o1[j] = sum[4 + j] / (float)R; }
    (void)inv;
    pool[((size_t)site * n8 + i) * 2] = o0;
    pool[((size_t)site * n8 + i) * 2 + 1] = o1;
}

void launch_read_mean16(const uint16_t* y, float* pool, int n_sites, int R, int L, const int* row_src, hipStream_t s) {
    const int n8 = L * (CPAD / 8);
    hipLaunchKernelGGL(read_mean16_kernel, dim3((n8 + 255) / 256, n_sites), dim3(256), 0, s, (const bf8*)y, (v4f*)pool, R, L, row_src);
}

__global__ __launch_bounds__(256) void final_pool16_kernel(const bf8* __restrict__ y, float* __restrict__ feat, long long fs,
                                                           int R, int L, int C, const int* __restrict__ row_src) {
    constexpr int PT = 32, PS = PT + 1;                        // 32 positions per workgroup: 128-byte runs in the feature row
    __shared__ float tmax[CPAD * PS], tavg[CPAD * PS];
    const int pt = blockIdx.x, site = blockIdx.y, tid = threadIdx.x;
    const int c8 = tid & 15, pl = tid >> 4;
    const int n8 = L * (CPAD / 8);
#pragma unroll
    for (int pass = 0; pass < PT / 16; ++pass) {
        const int pp = pass * 16 + pl, p = pt * PT + pp;
        float mx[8], sum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { mx[j] = 0.f; sum[j] = 0.f; }
        if (p < L) {
            const size_t off = (size_t)p * (CPAD / 8) + c8;
            const int* rs = row_src ? row_src + (size_t)site * R : nullptr;
            auto row = [&](int r) { return y[(size_t)(rs ? rs[r] : site * R + r) * n8 + off]; };
            const bf8 v0 = row(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) { mx[j] = (float)v0[j]; sum[j] = mx[j]; }
#pragma unroll 8
            for (int r = 1; r < R; ++r) {
                const bf8 v = row(r);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; mx[j] = fmaxf(mx[j], f); sum[j] += f; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sum[j] = sum[j] / (float)R;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            tmax[(c8 * 8 + j) * PS + pp] = mx[j];
            tavg[(c8 * 8 + j) * PS + pp] = sum[j];
        }
    }
    __syncthreads();
    float* row = feat + (size_t)site * fs;
    for (int idx = tid; idx < CPAD * PT; idx += 256) {
        const int c = idx / PT, pp = idx % PT, p = pt * PT + pp;
        if (c < C && p < L) {
            row[(size_t)c * L + p] = tmax[c * PS + pp];                         // max block first (model.py:833)
            row[(size_t)C * L + (size_t)c * L + p] = tavg[c * PS + pp];
        }
    }
}

void launch_final_pool16(const uint16_t* y, float* feat, long long fs, int n_sites, int R, int L, int C, const int* row_src,
                         hipStream_t s) {
    hipLaunchKernelGGL(final_pool16_kernel, dim3((L + 31) / 32, n_sites), dim3(256), 0, s, (const bf8*)y, feat, fs, R, L, C, row_src);
}

// highway compression from bf16 h: the fp32 kernel's structure (one workgroup = 64 reads x 32 outputs, 8 waves split K, partial
// tiles summed through LDS in wave order), a k-group = ONE position (32 channels): a lane's 16-byte load is channels 8 kk .. 8 kk + 7
// of its read, eight fp32 MFMA k-steps; the weight fragments are packed in that channel order (dan_kernels.h).
constexpr int HW16_WAVES = 8;
constexpr int HW16_RT = 4;
__device__ __forceinline__ v4f mfma16f(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__global__ __launch_bounds__(512) void highway16_kernel(const uint16_t* __restrict__ h, long long hls, const v4f* __restrict__ wc,
                                                        long long wcls, const float* __restrict__ bc, float* __restrict__ feat,
                                                        long long fs, int feat_off, int n_rows, int R, int L, int H,
                                                        const int* __restrict__ row_src) {
    __shared__ float part[HW16_WAVES][HW16_RT][2][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int layer = blockIdx.y;
    const int row0 = blockIdx.x * (16 * HW16_RT);
    const size_t K = (size_t)L * HPAD;
    const int g_lo = (int)((long long)L * wave / HW16_WAVES), g_hi = (int)((long long)L * (wave + 1) / HW16_WAVES);
    const uint16_t* arow[HW16_RT];
#pragma unroll
    for (int i = 0; i < HW16_RT; ++i) {
        const int row = min(row0 + 16 * i + r16, n_rows - 1);
        arow[i] = h + (size_t)layer * hls + (size_t)(row_src ? row_src[row] : row) * K + kk * 8;
    }
    const v4f* wl = wc + (size_t)layer * wcls + (size_t)lane * 2;
    v4f acc[HW16_RT][2];
#pragma unroll
    for (int i = 0; i < HW16_RT; ++i) { acc[i][0] = (v4f){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (v4f){0.f, 0.f, 0.f, 0.f}; }
    constexpr int D = 4;                                    // positions in flight per wave
    bf8 ar[D][HW16_RT];
    v4f b0[D][2], b1[D][2];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int g = min(g_lo + d, g_hi - 1);
#pragma unroll
        for (int i = 0; i < HW16_RT; ++i) ar[d][i] = *(const bf8*)(arow[i] + (size_t)g * HPAD);
        b0[d][0] = wl[((size_t)g * 2) * 128]; b0[d][1] = wl[((size_t)g * 2) * 128 + 1];
        b1[d][0] = wl[((size_t)g * 2 + 1) * 128]; b1[d][1] = wl[((size_t)g * 2 + 1) * 128 + 1];
    }
    for (int g = g_lo; g < g_hi; g += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            bf8 av[HW16_RT];
#pragma unroll
            for (int i = 0; i < HW16_RT; ++i) av[i] = ar[d][i];
            const v4f w00 = b0[d][0], w01 = b0[d][1], w10 = b1[d][0], w11 = b1[d][1];
            const int gn = min(g + d + D, g_hi - 1);
#pragma unroll
            for (int i = 0; i < HW16_RT; ++i) ar[d][i] = *(const bf8*)(arow[i] + (size_t)gn * HPAD);
            b0[d][0] = wl[((size_t)gn * 2) * 128]; b0[d][1] = wl[((size_t)gn * 2) * 128 + 1];
            b1[d][0] = wl[((size_t)gn * 2 + 1) * 128]; b1[d][1] = wl[((size_t)gn * 2 + 1) * 128 + 1];
            if (g + d < g_hi) {
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int i = 0; i < HW16_RT; ++i) {
                        const float x = (float)av[i][s];
                        acc[i][0] = mfma16f(x, s < 4 ? w00[s & 3] : w01[s & 3], acc[i][0]);
                        acc[i][1] = mfma16f(x, s < 4 ? w10[s & 3] : w11[s & 3], acc[i][1]);
                    }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < HW16_RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) part[wave][i][j][lane * 4 + jj] = acc[i][j][jj];
    __syncthreads();
    for (int idx = tid; idx < HW16_RT * 2 * 256; idx += 512) {
        const int i = idx >> 9, j = (idx >> 8) & 1, e = idx & 255;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < HW16_WAVES; ++w) sum += part[w][i][j][e];
        const int ln = e >> 2, jj = e & 3;
        const int row = row0 + 16 * i + 4 * (ln >> 4) + jj, o = 16 * j + (ln & 15);
        if (o < H && row < n_rows) {
            const int site = row / R, r = row - site * R;
            feat[(size_t)site * fs + feat_off + (size_t)layer * H * R + (size_t)o * R + r] = fmaxf(sum + bc[layer * HPAD + o], 0.f);
        }
    }
}

void launch_highway16(const uint16_t* h, long long hls, const float* wc16, long long wcls, const float* bc, float* feat, long long fs,
                      int feat_off, int n_sites, int R, int L, int H, int layers, const int* row_src, hipStream_t s) {
    const int n_rows = n_sites * R;
    hipLaunchKernelGGL(highway16_kernel, dim3((n_rows + 16 * HW16_RT - 1) / (16 * HW16_RT), layers), dim3(512), 0, s, h, hls,
                       (const v4f*)wc16, wcls / 4, bc, feat, fs, feat_off, n_rows, R, L, H, row_src);
}

}  // namespace dan
