// Kernel-side contract shared by dan_kernels.hip (device code + launchers) and dan_capi.cpp (host).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dan {

// ---- fixed geometry of the fp32 path -----------------------------------------------------------
constexpr int CPAD = 128;               // channel capacity of one activation row (c_init, c_final <= 128)
constexpr int NWAVE = 8;                // waves per workgroup, two per SIMD: one covers the other's waits and epilogues
constexpr int SEG_THREADS = NWAVE * 64;
// fp32 kernel: wave = (channel quarter cq = wave & 3, position half ph = wave >> 2).  It owns NT = 2 channel tiles
// (32 channels) x the position tiles [7 ph, 7 ph + 7) -- 7 tiles for ph 0, 6 for ph 1; waves w and w + 4 share a
// SIMD, so every SIMD carries 13 tile-pairs.  One ds_read_b128 then feeds 8 MFMAs (a 1-KiB LDS return costs the
// matrix pipe ~7.4 cycles: tools/ubench/mfma_feed.hip).
constexpr int NT = 2;
constexpr int MTW = 7;                  // position tiles per wave (the upper half uses MT - MTW = 6 of them)
constexpr int MT = 13;                  // 16-position tiles per LDS-resident unit: a read of <= 208 columns, or half of a longer one (SegmentArgs::units)
constexpr int MPOS = MT * 16;           // 208
constexpr int HALO = 4;                 // zero rows either side of the window (dilation <= 4)
constexpr int LDS_S = 136;              // floats per LDS position row: 128 + 8 makes the ds_read_b128 B-fragment reads conflict-free
constexpr int LDS_ROWS = MPOS + 2 * HALO;   // 216
constexpr int HPAD = 32;                // bottleneck channel capacity
constexpr int CIN0 = 48;                // encoded input channels, padded (canonical order, see encode)
constexpr int KG0 = CIN0 / 16;          // k-groups of layer 1
constexpr int KGC = CPAD / 16;          // k-groups of a 128-channel layer
constexpr int EMBED = 20;
constexpr int VOCAB = 10;
constexpr int NHEAD = 27;               // 2 + 3 + 1 + 1 + 10 + 10 head outputs

// Per-layer weight block in HBM (floats), one block per conv layer at a fixed stride so that the
// kernel needs no per-layer descriptor loads:
//   [W_OFF)    conv weights, MFMA A-fragment order  [tap 3][kg 8][tile 8][lane 64][4]   (layer 1 uses kg 3)
//   [WRES_OFF) residual 1x1 weights                 [kg 8][tile 8][lane 64][4]
//   [WBOT_OFF) bottleneck 1x1 weights               [kg 8][tile 2][lane 64][4]
//   [CST_OFF)  bias[128] scale[128] shift[128] bres[128] bbot[32]   (scale/shift = folded eval BatchNorm)
constexpr int W_OFF = 0;
constexpr int WRES_OFF = W_OFF + 3 * KGC * KGC * 256;        // 49152
constexpr int WBOT_OFF = WRES_OFF + KGC * KGC * 256;         // 65536
constexpr int CST_OFF = WBOT_OFF + KGC * 2 * 256;            // 69632
constexpr int CST_BIAS = 0, CST_SCALE = CPAD, CST_SHIFT = 2 * CPAD, CST_BRES = 3 * CPAD, CST_BBOT = 4 * CPAD;
constexpr int CST_FLOATS = 4 * CPAD + HPAD;                  // 544
//   [WW_OFF)   Winograd F(2,3) weights U_k of the conv   [k 4][kg 8][tile 8][lane 64][4]   (layers > 1 only)
constexpr int WW_OFF = CST_OFF + CST_FLOATS + 32;            // 70208
constexpr int LAYER_STRIDE = WW_OFF + 4 * KGC * KGC * 256;   // 135744 floats (16-byte aligned blocks)
constexpr int MAX_LAYERS = 16;

struct SegmentArgs {
    const float* wl;             // [layers][LAYER_STRIDE]
    int l_begin, l_end;          // 0-based [begin, end)
    int n_layers;
    int dil_mid, dil_final;
    unsigned res_mask;           // bit l (0-based): layer has the 1x1 residual branch
    int has_hw;                  // highway bottleneck present
    int R, L;
    // segment that starts at layer 0 encodes from the uint8 planes
    const uint8_t *reads, *qual, *strand, *ref, *ref_mask, *var_mask;
    const float* emb;            // [VOCAB][EMBED]
    const float* pe;             // [L][EMBED]
    // later segments start from y (+ the broadcast read-mean of the previous segment's output)
    float* y;                    // [site][read][L][CPAD]   in/out (a workgroup only touches its own read)
    const float* pool;           // [site][L][CPAD] or nullptr
    float* h;                    // bottleneck outputs [layer][site][read][L][HPAD], or nullptr
    long long h_layer_stride;    // floats between layers of h
    float* tap;                  // [site][read][L][CPAD] or nullptr
    int tap_layer;               // 0 = encoded input, l = after conv layer l (1-based), -1 = none
    int wino;                    // 1: dilation-2 layers after the first run in Winograd F(2,3) form
    int n_rows;                  // filled by launch_segment: n_sites * R
    int xcd_rows;                // filled by launch_segment: rows per XCD slice (whole sites)
    const int* work;             // rows to compute (device list, see launch_row_map); read only when work_count is set
    const int* work_count;       // length of that list (device) or nullptr = every row
    // ---- (may be null) layer 1 by table instead of encode + GEMM (L0_* below; built by dan_finalize)
    const float* l0_tab;
    // ---- windows of 209..304 columns (units == 2; plan_units fills these): a read no longer fits the 208-row LDS image, so it
    // is computed as TWO overlapping units, each an ordinary <= 208-column "read" of the kernel: unit u covers window columns
    // [u_off, u_off + u_len) and STORES (y, h, tap) only its own columns [own_lo, own_hi) (unit-relative).  The overlap is the
    // segment's receptive-field radius (the sum of its layers' dilations): the zero padding the kernel applies at a unit's inner
    // cut is wrong there and leaks one dilation further per layer, never into an owned column.  L is then unused; Lw = the
    // window = the position stride of reads / ref / pe / y / pool / h / tap.  y_out != y: the other unit reads its overlap from
    // y while this one stores (units == 1: Lw == L, y_out == y, in place as ever).  With units == 2, L = the units' common length.
    int units, Lw;
    int u_off[2], u_len[2], own_lo[2], own_hi[2];
    float* y_out;
};

// Windows above MPOS columns: two units per read (see SegmentArgs; SegmentXArgs carries the same fields).  halo = the segment's
// receptive-field radius.  Returns false when a unit would not fit (window > 2 (MPOS - halo)).  For Lw <= MPOS: one unit, the whole
// window.  Both units have the SAME length (for an odd window the second starts one column earlier): the bf16x3 kernel keeps its
// image across rows and relies on the rows past a unit's length staying zero.
// Layer 1 by table (fp32 path).  The network's first conv sees an input that is a SUM of per-column terms -- embedding of the read
// token + positional encoding, embedding of the reference token + the same encoding, q, strand and three mask flags (model.py:
// 450-627) -- so conv1 is linear in each of them: out[p] = bias + sum over taps t of
//     TJ[t][10 tok + ref][.]  +  PE_t[p + t - 1][.]  +  q W_q[t] + strand W_s[t] + m1 W_1[t] + m2 W_2[t] + m3 W_3[t]      (column p + t - 1)
// with columns outside the unit contributing nothing (the conv's zero padding).  ~50 packed vector instructions per four outputs
// instead of building 48 channels per column and a K = 144 GEMM on the matrix cores (4.3 % of the network's MACs, a fifth of the
// first segment's time).  l0_tab (floats): TJ [tap 3][read token x ref token 100, + one all-zero entry][CPAD] | W_sc [channel 5: q,
// strand, refmatch, varmatch, lenmask][tap 3][CPAD] | PE [variant 3: both neighbours, no left neighbour, no right neighbour][window
// column Lw][CPAD] (the positional term of all valid taps, summed on the host in double).  The walk is branch-free: the columns either
// side of the unit are staged as the all-zero table entry with zero scalars.
constexpr int L0_NTJ = 101;                                  // entries of TJ per tap (index 100 = zeros)
constexpr int L0_TJ_OFF = 0;
constexpr int L0_WSC_OFF = L0_TJ_OFF + 3 * L0_NTJ * CPAD;    // 38 784
constexpr int L0_PE_OFF = L0_WSC_OFF + 5 * 3 * CPAD;         // 40 704
inline size_t l0_tab_floats(int Lw) { return (size_t)L0_PE_OFF + (size_t)3 * Lw * CPAD; }
constexpr int L0_TOK = 8;                                    // floats per staged column: table index, q 0.01, strand 0.5, lenmask flag, varmatch flag (entry 0: [5] = the read agrees with the reference allele)
constexpr int L0_MAX_LAYERS = (MAX_LAYERS * CST_FLOATS - (MPOS + 2) * L0_TOK) / CST_FLOATS;   // they live behind the segment's constants: 12 layers leave room

template <class Args>
inline bool plan_units(Args& a, int Lw, int halo) {
    a.Lw = Lw;
    if (Lw <= MPOS) {
        a.units = 1; a.L = Lw;
        a.u_off[0] = 0; a.u_len[0] = Lw; a.own_lo[0] = 0; a.own_hi[0] = Lw;
        a.u_off[1] = 0; a.u_len[1] = 0; a.own_lo[1] = 0; a.own_hi[1] = 0;
        return true;
    }
    const int mid = (Lw + 1) / 2;                                 // unit 0 owns [0, mid), unit 1 owns [mid, Lw)
    const int len = mid + halo;
    if (len > MPOS || len > Lw) return false;
    a.units = 2; a.L = len;
    a.u_off[0] = 0; a.u_len[0] = len; a.own_lo[0] = 0; a.own_hi[0] = mid;
    a.u_off[1] = Lw - len; a.u_len[1] = len; a.own_lo[1] = mid - (Lw - len); a.own_hi[1] = len;
    return true;
}

// max_wgs: workgroups to launch (one per CU: they are persistent and walk the rows with the grid's stride); 0 = one per row
void launch_segment(const SegmentArgs& a, int n_sites, int max_wgs, hipStream_t s);

// ---- bf16 "ping-pong" kernel (dan_kernels_bf16p.hip): plain bf16, v_mfma_f32_32x32x16_bf16, two XOR-swizzled LDS images
// (layer input / layer output), bf16 activations in HBM (y, h), persistent workgroups.  BASELINE config 5 (128 x 301).
constexpr int P_HALO = 4;                 // zero rows either side of the window (dilation <= 4)
constexpr int P_ROW_BYTES = 256;          // one position = 128 bf16 channels, 16 chunks of 16 B, chunk c of row r stored at c ^ (r & 15)
constexpr int P_LMAX = 304;               // window columns the two images hold: 2 x (4 + 304 + 4) rows x 256 B = 159 744 B of the 160 KiB
constexpr int P_ROWS = P_LMAX + 2 * P_HALO;
constexpr int P_IMG_BYTES = P_ROWS * P_ROW_BYTES;
constexpr int P_LDS_BYTES = 163840;
constexpr int P_KS0 = CIN0 / 16;          // 16-channel k-steps of layer 1 (48 encoded channels)
constexpr int P_KSC = CPAD / 16;          // 8
// per-layer weight block (bytes): MFMA 32x32x16 A fragments, 1 KiB each ([lane 64][8 bf16]); the 32 rows of a fragment are the
// output channels of one channel quarter q in the order row m -> channel 32 q + 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3), so
// that a lane's 16 accumulator registers are 16 CONSECUTIVE channels (32 q + 16 (lane >> 5) + reg)
//   [WP_CONV_OFF) conv      [kstep 8][tap 3][q 4] fragments   (layer 1 uses ksteps 0..2: the first nine steps of the walk)
//   [WP_RES_OFF)  residual  [kstep 8][q 4]
//   [WP_BOT_OFF)  bottleneck [kstep 8]                         (32 outputs: one row tile)
//   [WP_CST_OFF)  the fp32 constants of the layer (bias, scale, shift, bres, bbot), as in the other families
constexpr int WP_FRAG = 1024;
constexpr int WP_CONV_OFF = 0;
constexpr int WP_RES_OFF = WP_CONV_OFF + 3 * P_KSC * 4 * WP_FRAG;     // 98304
constexpr int WP_BOT_OFF = WP_RES_OFF + P_KSC * 4 * WP_FRAG;          // 131072
constexpr int WP_CST_OFF = WP_BOT_OFF + P_KSC * WP_FRAG;              // 139264
constexpr int WP_LAYER_BYTES = WP_CST_OFF + (CST_FLOATS + 32) * 4;

struct SegmentPArgs {
    const char* wl;              // [layers][WP_LAYER_BYTES]
    const char* wlr;             // the same blocks in the sixteen-wave form's fragment order (16x16x32 tiles)
    int form;                    // 0 = the eight-wave 32x32x16 form (default), 1 = the sixteen-wave 16x16x32 form (dan_config.bf16_form)
    int l_begin, l_end, n_layers, dil_mid, dil_final;
    unsigned res_mask;
    int has_hw;
    int R, L;
    const uint8_t *reads, *qual, *strand, *ref, *ref_mask, *var_mask;
    const float* emb;
    const float* pe;
    uint16_t* y;                 // bf16 [site][read][L][CPAD]   in/out
    const float* pool;           // fp32 [site][L][CPAD]: conv(read-mean) of the segment's first layer (launch_conv_pool), or nullptr
    uint16_t* h;                 // bf16 [layer][site][read][L][HPAD] or nullptr
    long long h_layer_stride;    // elements between layers of h
    float* tap;                  // fp32 [site][read][L][CPAD] or nullptr
    int tap_layer;
    int n_rows;                  // filled by the launcher
    int slice_rows;              // filled by the launcher: rows per XCD slice (whole sites)
    const int* work;
    const int* work_count;
};
void launch_segmentp(const SegmentPArgs& a, int n_sites, int n_cus, hipStream_t s);
// cp[site][p][o] = sum_{t,c} wpool[o][t * 128 + c] * pool[site][p + (t - 1) dil][c]: the read-mean's share of the convolution
// behind a pool layer, in fp32 (wpool: that layer's bf16-rounded weights; cols: [n_sites * L][384] scratch; zero_bias: 128 zeros)
void launch_conv_pool(const float* pool, const float* wpool, const float* zero_bias, float* cols, float* cp, int n_sites, int L,
                      int dil, hipStream_t s);
// bf16-input forms of the three reductions
void launch_read_mean16(const uint16_t* y, float* pool, int n_sites, int R, int L, const int* row_src, hipStream_t s);
void launch_final_pool16(const uint16_t* y, float* feat, long long feat_stride, int n_sites, int R, int L, int C,
                         const int* row_src, hipStream_t s);
// wc16: compression weights as bf16 hi / lo planes in MFMA 16x16x32 B-fragment order, [layer][pos][n 2][plane 2][lane 64][8 bf16]:
// lane (o = lane & 15, kk = lane >> 4), element s -> Wc[16 n + o][8 kk + s][pos]; wc_layer_stride in floats (= L * 2 * 2 * 64 * 4)
void launch_highway16(const uint16_t* h, long long h_layer_stride, const float* wc16, long long wc_layer_stride,
                      const float* bc, float* feat, long long feat_stride, int feat_off, int n_sites, int R, int L,
                      int H, int layers, const int* row_src, hipStream_t s);

// ---- bf16x3 "split" kernel (dan_kernels_bf16x.hip; dan_config.precision = 1): every GEMM operand as hi + lo bf16, three
// v_mfma_f32_16x16x32_bf16 per product (wh xh + wl xh + wh xl), fp32 sums.  One read = TWO XOR-swizzled planes (hi, lo) of ONE
// image updated in place; y crosses HBM as the same two planes ([row][plane 2][L][128] bf16 -- the resumed segment's image
// arrives by LDS-DMA and the copy-out is a plain copy), h as fp32.
constexpr int X_PT = 7;                   // 16-column tiles per wave: wave = (channel quarter, position half), 2 x 7 x 16 = 224 columns
constexpr int X_COLS = 2 * X_PT * 16;
constexpr int X_LMAX = MPOS;              // 208 columns per LDS-resident unit (the 14th tile is a phantom: skipped when L <= 208); longer windows: two units
constexpr int X_ROWS = X_COLS + 2 * P_HALO;
constexpr int X_ROW_BYTES = 2 * P_ROW_BYTES;              // one image row: the hi plane's 16 chunks, then the lo plane's
constexpr int X_LO = P_ROW_BYTES;
constexpr int X_IMG_BYTES = X_ROWS * X_ROW_BYTES;         // 118 784 B
constexpr int X_LDS_BYTES = X_IMG_BYTES + 4096;           // + two buffers of per-layer constants = 122 880 B
constexpr int X_KS = CPAD / 32;           // 4 k-steps of 32 channels
constexpr int X_KS0 = (CIN0 + 31) / 32;   // 2 for layer 1's 48 encoded channels
// per-layer weight block (bytes): MFMA 16x16x32 A fragments of 1 KiB ([lane 64][8 bf16]) in the sixteen-wave form's row order
// (row r of channel tile ct -> channel 32 (ct >> 1) + 8 (r >> 2) + 4 (ct & 1) + (r & 3): a lane's two tiles of a column are 8
// consecutive channels), hi and lo plane of a tile side by side: fragment ((step * tiles + ct) * 2 + plane)
//   [WX_CONV_OFF) conv       [step = ks * 3 + tap (12)][ct 8][plane 2]      (layer 1 uses the first 6 steps)
//   [WX_RES_OFF)  residual   [ks 4][ct 8][plane 2]
//   [WX_BOT_OFF)  bottleneck [ks 4][ct 2][plane 2]
//   [WX_CST_OFF)  the fp32 constants of the layer, as in the other families
constexpr int WX_CONV_OFF = 0;
constexpr int WX_RES_OFF = WX_CONV_OFF + 3 * X_KS * 8 * 2 * WP_FRAG;     // 196 608
constexpr int WX_BOT_OFF = WX_RES_OFF + X_KS * 8 * 2 * WP_FRAG;          // 262 144
constexpr int WX_CST_OFF = WX_BOT_OFF + X_KS * 2 * 2 * WP_FRAG;          // 278 528
constexpr int WX_LAYER_BYTES = WX_CST_OFF + (CST_FLOATS + 32) * 4;

struct SegmentXArgs {
    const char* wl;              // [layers][WX_LAYER_BYTES]
    int l_begin, l_end, n_layers, dil_mid, dil_final;
    unsigned res_mask;
    int has_hw;
    int R, L;
    const uint8_t *reads, *qual, *strand, *ref, *ref_mask, *var_mask;
    const float* emb;
    const float* pe;
    uint16_t* y;                 // bf16 [row][plane 2][L][CPAD]   in/out
    const float* pool;           // fp32 [site][L][CPAD]: conv(read-mean) of the segment's first layer (launch_conv_pool), or nullptr
    float* h;                    // fp32 [layer][row][L][HPAD] or nullptr
    long long h_layer_stride;    // floats between layers of h
    float* tap;                  // fp32 [row][L][CPAD] or nullptr
    int tap_layer;
    int n_rows, slice_rows;      // filled by the launcher
    int stagger;                 // filled by the launcher: start offset per workgroup index, in units of 256 cycles
    const int* work;
    const int* work_count;
    // windows of 209..304 columns: two units per read, exactly as in SegmentArgs (plan_units); y_out != y then
    int units, Lw;
    int u_off[2], u_len[2], own_lo[2], own_hi[2];
    uint16_t* y_out;
};
void launch_segmentx(const SegmentXArgs& a, int n_sites, int n_cus, hipStream_t s);
// the two reductions over reads from the two-plane y (value = hi + lo, summed in fp32 in read order)
void launch_read_meanx(const uint16_t* y, float* pool, int n_sites, int R, int L, const int* row_src, hipStream_t s);
void launch_final_poolx(const uint16_t* y, float* feat, long long feat_stride, int n_sites, int R, int L, int C,
                        const int* row_src, hipStream_t s);

// Empty-row map: a pileup row whose reads / qual / strand bytes are all zero (padding below the site's coverage) encodes to
// the same activations as every other such row of its site, through every layer.  row_src[site*R + r] = site*R + (first
// empty row of the site) for an empty row, site*R + r otherwise; the segment kernels walk only the rows that are their own
// source and the reductions read a skipped row's y / h through the map (bit-identical results).
// work[0 .. *count) = the rows that are their own source, in row order.
void launch_row_map(const uint8_t* reads, const uint8_t* qual, const uint8_t* strand, int* row_src, int* work, int* count,
                    int n_sites, int R, int L, hipStream_t s);
// pool[site][p][c] = mean over reads of y[site][r][p][c]          (dl4vc/model.py:772)
void launch_read_mean(const float* y, float* pool, int n_sites, int R, int L, const int* row_src, hipStream_t s);
// feat[site][c*L+p] = max_r y, feat[site][C*L + c*L+p] = mean_r y   (dl4vc/model.py:824-839)
void launch_final_pool(const float* y, float* feat, long long feat_stride, int n_sites, int R, int L, int C,
                       const int* row_src, hipStream_t s);
// feat[site][off + l*H*R + o*R + r] = relu(sum_{p,c} Wc[l][o][c][p] h[l][site][r][p][c] + bc[l][o])   (model.py:776-777,859)
void launch_highway(const float* h, long long h_layer_stride, const float* wc_packed, long long wc_layer_stride,
                    const float* bc, float* feat, long long feat_stride, int feat_off, int n_sites, int R, int L,
                    int H, int layers, const int* row_src, hipStream_t s);
// C[M][N] = relu?(A[M][lda] * W[N][ldw]^T + bias)  over K (multiple of 16)
// ws (may be null): ws_floats >= 2 M N lets a long-k product with few tiles split its k range over two co-resident workgroups
void launch_fc(const float* A, long long lda, const float* W, long long ldw, const float* bias, float* C,
               long long ldc, int M, int N, int K, int relu, hipStream_t s, float* ws = nullptr, long long ws_floats = 0);
// the fixed-order sum of launch_fc's split-k partial sums (+ bias, ReLU): ws [parts][M][N] -> C
void launch_fc_combine(const float* ws, int parts, const float* bias, float* C, long long ldc, int M, int N, int relu, hipStream_t s);
// the same product on the bf16 matrix cores with split operands (dan_kernels_bf16x.hip; precision >= 1): W as two bf16 planes
// [2][N][ldw] (hi = bf16(w), lo = bf16(w - hi); w_plane = elements between the planes), A split on the fly, three MFMAs per product
void launch_fcx(const float* A, long long lda, const uint16_t* W, long long ldw, long long w_plane, const float* bias, float* C,
                long long ldc, int M, int N, int K, int relu, hipStream_t s, float* ws = nullptr, long long ws_floats = 0);
// heads + softmax: hidden [B][hid] -> logits/probabilities   (model.py:919-958, trainer.py:609-623)
void launch_heads(const float* hidden, int hid, const float* wh /*[NHEAD][hid]*/, const float* bh, int B,
                  float* bin_logits, float* vt_logits, float* vt_prob, float* bp, float* aux, hipStream_t s);

}  // namespace dan
