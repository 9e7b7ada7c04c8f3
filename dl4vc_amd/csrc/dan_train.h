// Kernel-side contract of the TRAINING path (dan_train.hip <-> dan_train_capi.cpp).  fp32 only.
//
// One optimisation step of the reference's loop (dl4vc/trainer.py:109-439): train-mode forward (BatchNorm on batch
// statistics, dropout), the loss mix, backward, gradient clipping, Adam.  Train-mode BatchNorm needs the statistics of
// ALL B*R*L positions of a layer before the next layer can start (dl4vc/model.py:749-751 under model.train()), so the
// inference design -- one launch carrying a read through a whole segment of layers in LDS -- does not apply: the unit here
// is ONE LAYER per launch, still with one workgroup = one read resident in LDS, activations in HBM between layers
// (row layout [site][read][pos][128], the same as the inference path's y), and a grid-wide reduction (per-read partial
// sums -> fixed-order double-precision reduce) between the launches.  The backward pass mirrors it.
#pragma once
#include "dan_kernels.h"

namespace dan {

constexpr int WG_S = 144;                 // LDS row stride (floats) of the weight-gradient kernel's operand images
constexpr int WG_CH = 104;                // positions per staged chunk (half a read)
constexpr int TRAIN_PARTIAL_WGS = 256;    // workgroups of a weight-gradient launch = slices of the split-K partial buffer

// ---- T1: one layer-shaped step on the LDS-resident read -------------------------------------------------------------------
//   image  = load(mode)                                   encode | rows src1 [affine with src2] [masked by src2 > 0] [+ pool]
//   acc    = w1 ? GEMM(image; taps, kg, dil) : image
//            [+ W3 * (src4 > 0 ? src3 : 0)]               (half-read 3-tap launches only)
//   out1   = [relu](acc + bias1 + add1 + add2 + addb)     rows, CPAD stride;  stats[row] = (sum out1, sum out1 * aux)
//   out2   = relu(W2 * out1 + bias2)                      the 128 -> 32 bottleneck (rows, HPAD stride), optional
struct RowArgs {
    int R, L;
    int mode;                               // 0 = encode from the uint8 planes, 1 = rows
    const uint8_t *reads, *qual, *strand, *ref, *ref_mask, *var_mask;
    const float *emb, *pe;
    const float* src1; int s1_stride;       // floats per position: CPAD or HPAD
    const float* src2;                      // optional second tensor, same stride as src1
    const float* coef;                      // [3][CPAD]: v = A[c] * s1 + B[c] * s2 + C[c]   (nullptr: v = s1)
    int mask_src2;                          // v = (s2 > 0) ? v : 0
    const float* pool_in;                   // [site][L][CPAD] added to every read of the site (model.py:742), or nullptr
    const float* w1;                        // packed A fragments [taps][kg][8][64][4], or nullptr
    int taps, kg, dil;
    int wino;                               // 1: w1 holds the Winograd F(2,3) transform U [4][8][8][64][4] of a 3-tap, dilation-2,
                                            //    128-channel kernel and the launch runs in that form (no second GEMM stage then)
    const float* bias1;                     // [CPAD] or nullptr
    int relu_out;
    const float *add1, *add2;               // row addends (CPAD stride) or nullptr
    const float* add1_coef;                 // [3][CPAD]: the first addend enters as A[c] * add1 + C[c] (a BatchNorm output that was
                                            //    never written: pointwise launches only), or nullptr
    const float* addb;                      // per-site addend [site][L][CPAD] or nullptr
    float* out1;                            // rows (CPAD stride) or nullptr
    float* stats;                           // [row][2][CPAD] or nullptr
    const float* stat_aux;                  // rows (CPAD stride): second statistic = sum out1 * aux; nullptr: sum out1^2
    const float* w3;                        // 3-tap launches on half reads only: a SECOND product added to the accumulators,
    const float *src3, *src4;               //    W3 (packed [2][8][64][4], K = 32) * (src4 > 0 ? src3 : 0), rows of HPAD floats -- the
                                            //    bottleneck's transpose rides on the data-gradient launch, which then writes g_{l-1}
    const float* w2;                        // packed bottleneck fragments [8][2][64][4] or nullptr
    const float* bias2;                     // [HPAD]
    float* out2;                            // rows (HPAD stride)
};
// returns the number of `stats` entries written ([entry][2][CPAD]): reads (the whole-read kernel), half-read units (the 3-tap
// launches of the direct form: train_rowh_kernel), or 64-position tiles (the pointwise 1x1 launches)
// Returns the number of `stats` entries the launch writes.  stat_cap: entries a.stats holds -- a launch that would write more is
// refused BEFORE it is made (TRAIN_ROW_ERR_STATS), as is one whose second product (a.w3) the selected kernel form would ignore
// (TRAIN_ROW_ERR_FORM).  train_row_fuses_second_product(a): true when `a` (with or without w3 / src3 / src4 set) selects the one
// form that honours w3 -- the caller decides the g_{l-1} fusion with it.
constexpr int TRAIN_ROW_ERR_STATS = -2, TRAIN_ROW_ERR_FORM = -3;
int launch_train_row(const RowArgs& a, int n_rows, hipStream_t s, int stat_cap = 1 << 30);
bool train_row_fuses_second_product(const RowArgs& a);

// ---- T2: weight gradient  dW[t][o][c] = sum over positions of  A[p][o] * B[p + (t - 1) dil][c]   (+ bias grad = sum A) ------
struct WgradArgs {
    int R, L, n_rows;
    const float* a1; int a_stride;          // A rows: CPAD or HPAD floats per position
    const float* a2;                        // optional second tensor for the affine / mask (same stride)
    const float* a_coef;                    // [3][CPAD] as RowArgs::coef, or nullptr
    int a_mask;                             // A = (a2 > 0) ? A : 0
    int b_mode;                             // (1 = rows; the encode form is gone: layer 1 goes through launch_l0_backward)
    const uint8_t *reads, *qual, *strand, *ref, *ref_mask, *var_mask;
    const float *emb, *pe;
    const float* b1;                        // B rows, CPAD stride
    const float* b_coef;                    // [3][CPAD]: B = A[c] * b1 + C[c]   (nullptr: B = b1)
    const float* b_pool;                    // [site][L][CPAD] added, or nullptr
    int taps, dil;
    int o_tiles, c_tiles;                   // 16-wide tiles of A (outputs) and B (inputs)
    float* partial;                         // [wgs][taps][o_tiles*16][c_tiles*16]
    float* bias_partial;                    // [wgs][o_tiles*16]
};
int launch_train_wgrad(const WgradArgs& a, hipStream_t s);     // returns the number of workgroups (partial slices) used
// G[(o * n_in + i) * taps + t] = sum_wg partial[wg][t][o][cmap ? cmap[i] : i];  gb[o] = sum_wg bias_partial[wg][o]
void launch_wgrad_reduce(const float* partial, const float* bias_partial, int wgs, int taps, int o_pad, int c_pad, int n_out,
                         int n_in, const int* cmap, float* g_w, float* g_b, hipStream_t s);

// ---- per-channel statistics ---------------------------------------------------------------------------------------------------
// block partials in double: bp[block][2][CPAD] = sum over the block's rows of stats[row][2][CPAD]
void launch_stats_partial(const float* stats, int n_rows, double* bp, int* n_blocks, hipStream_t s);
// forward BatchNorm (training): mean / biased variance over N = n_rows * L positions -> coef = (scale, 0, shift) with
// scale = gamma / sqrt(var + eps), shift = beta - mean * scale; saves mean / invstd; running stats += momentum 0.1 (unbiased)
void launch_bn_forward_finalize(const double* bp, int n_blocks, double n_pos, const float* gamma, const float* beta, int channels,
                                float* coef, float* save_mean, float* save_invstd, float* run_mean, float* run_var, hipStream_t s);
// backward: from sum(dn), sum(dn * a):  dgamma, dbeta and the coefficients of  da = A dn + B a + C  (then masked by a > 0)
void launch_bn_backward_coef(const double* bp, int n_blocks, double n_pos, const float* gamma, const float* save_mean,
                             const float* save_invstd, int channels, int use_bn, float* coef, float* g_gamma, float* g_beta,
                             hipStream_t s);

// ---- weight packing on the device (the weights change every step) ---------------------------------------------------------------
// dst[((t*kg + g)*tiles + n)*64 + lane][s] = W(o = 16n + (lane&15), c = 16g + 4(lane>>4) + s, t):
//   W(o,c,t) = src[omap(o) * so + cmap(c) * sc + (flip ? taps-1-t : t) * st]  (0 outside n_out x n_in or where a map gives -1)
void launch_pack_frag(float* dst, const float* src, int taps, int kg, int tiles, int n_out, int n_in, long long so, long long sc,
                      long long st, int flip, const int* omap, const int* cmap, hipStream_t s);
// U[(o * n_in + c)][4] = Winograd F(2,3) transform of the 3-tap kernel at src[o*so + c*sc + t] (flip: taps reversed)
void launch_wino_u(float* dst, const float* src, int n_out, int n_in, long long so, long long sc, int flip, hipStream_t s);
// highway compression weights Wc[o][c][p] (torch (H,H,1,L)) -> the highway kernel's fragment order [2L][2][64][4]
void launch_pack_wc(float* dst, const float* src, int H, int L, hipStream_t s);
// WcT[p][c][o] = Wc[o][c][p] padded to HPAD x HPAD: the operand of launch_highway_bwd
void launch_pack_wct(float* dst, const float* src, int H, int L, hipStream_t s);
void launch_pad_copy(float* dst, const float* src, int n, int n_pad, hipStream_t s);
// one launch for a table of the four job kinds above (type 0 pack_frag: i = taps, kg, tiles, n_out, n_in, flip; l = so, sc, st;
// 1 pad_copy: i = n, n_pad; 2 pack_wc / 3 pack_wct: i = H, L); first_block = the job's first block of the launch
struct PackJob { int type, first_block; float* dst; const float* src; int i[6]; long long l[3]; const int *omap, *cmap; };
int pack_job_blocks(const PackJob& q);
void launch_pack_jobs(const PackJob* jobs_on_device, int n_jobs, int n_blocks, hipStream_t s);

// ---- pools / highway ---------------------------------------------------------------------------------------------------------------
// g[site][r][p][c] = dfeat[site][C*L + c*L + p] / R + (r == first argmax_r y[.][r][p][c]) * dfeat[site][c*L + p]
void launch_final_pool_bwd(const float* y, const float* dfeat, long long fs, float* g, int n_sites, int R, int L, int C, hipStream_t s);
// dh[layer][row][p][c] = sum_o dhw[row][layer][o] * Wc[layer][o][c][p], dhw = dfeat_hw * (feat_hw > 0)
void launch_highway_bwd(const float* dfeat, const float* feat, long long fs, int feat_off, const float* wc, long long wc_layer,
                        float* dh, long long dh_layer, int n_sites, int R, int L, int H, int layers, hipStream_t s);
// dhw[layer][row][HPAD] = dfeat_hw * (feat_hw > 0): with it both highway products are plain GEMMs (launch_gemm):
//   dh[layer][row][e] = dhw[row][:] . WcT[e][:]            (M = rows, N = L*HPAD, K = HPAD; both K-contiguous)
//   gWcT[o][e]        = sum_rows dhw[row][o] h[row][e]      (M = HPAD, N = L*HPAD, K = rows; both K-slow, split-K)
void launch_highway_dhw(const float* dfeat, const float* feat, long long fs, int feat_off, float* dhw, int n_sites, int R, int H,
                        int layers, hipStream_t s);
void launch_highway_wc_transpose(const float* t, float* g_wc, int L, int H, hipStream_t s);
// the highway compression's backward, all layers per launch: dh[l][row][p][c] = sum_o dhw[l][row][o] WcT[l][p][c][o];
// gWc[l] (torch layout (o, c, p) at g_base + w_off[l]) = sum_row dhw[l][row][o] h[l][row][p][c]
void launch_highway_dh(const float* dhw, const float* wct, float* dh, int n_rows, int L, int layers, hipStream_t s);
void launch_highway_gwc(const float* dhw, const float* h, float* g_base, const long long* w_off, int n_rows, int L, int H, int layers, hipStream_t s);
void launch_highway_bias_grad_all(const float* dfeat, const float* feat, long long fs, int feat_off, float* partial /*[layers][64][HPAD]*/,
                                  float* g_base, const long long* g_off_on_device, int n_sites, int R, int H, int layers, hipStream_t s);
void launch_highway_bias_grad(const float* dfeat, const float* feat, long long fs, int feat_off, float* partial /*[64][HPAD]*/,
                              float* g_bc, int n_sites, int R, int H, hipStream_t s);
// (VALU forms, kept as the reference implementation of the two products)
// gWc[layer][o][c][p] = sum_rows dhw * h[layer][row][p][c];  gbc[layer][o] = sum_rows dhw   (split over row blocks -> partials)
void launch_highway_wgrad(const float* dfeat, const float* feat, long long fs, int feat_off, const float* h, long long h_layer,
                          float* partial, float* g_wc, long long wc_layer, float* g_bc, int n_sites, int R, int L, int H, int layers,
                          int layer_stride_b, hipStream_t s);
// Layer 1's backward by bins (dan_train.hip): conv1's weight and bias gradients and the embedding gradient from ONE pass over
// dz_1 = the transform (a1, a2, a_coef, a_mask) of WgradArgs -- no encode-form GEMM, no data-gradient launch, no embedding launches.
// a: R, L, n_rows, a1, a2, a_coef, a_mask, the six token planes, emb, pe, partial (TRAIN_PARTIAL_WGS slices of 3 CPAD CPAD floats);
// tot: L0B_FLOATS doubles (device scratch); w1: conv1's weights [n_out][n_in][3]; canon: reference channel -> canonical channel
constexpr int L0_BINS_TOTALS = 208 * 128 + 2 * 3 * 10 * 128 + 3 * 5 * 128 + 32;
int l0_bins_lds_bytes();
int launch_l0_backward(const WgradArgs& a, int n_sites, double* tot, const float* w1, const int* canon, int n_out, int n_in, float* g_w,
                        float* g_b, float* g_emb, hipStream_t s);
// Layer 1's forward by table (the inference walk of dan_kernels.h L0_*, tables rebuilt from the step's weights): tab = l0_tab_floats(L)
// floats; inv: canonical channel -> reference channel or -1.  The forward writes a_1 = relu(conv1 + bias) [row][L][CPAD] and, if stats,
// one entry of BatchNorm sums per row; returns the number of entries.
void launch_l0_train_tables(const float* w1, const int* inv, const float* emb, const float* pe, int L, int n_out, int n_in, float* tab, hipStream_t s);
int launch_l0_train_forward(const RowArgs& enc, const float* tab, const float* bias, int n_rows, float* a_out, float* stats, hipStream_t s);
// ---- FC stack ------------------------------------------------------------------------------------------------------------------------
// C[m][n] (+)= sum_k opA(m,k) * opB(n,k);  an operand is K-contiguous (X[i*ld + k]) or K-slow (X[k*ld + i])
// split_ws: workspace for split-K partials (used when the tile grid cannot fill the chip and K is long), or nullptr
void launch_gemm(const float* A, long long lda, int a_kslow, const float* B, long long ldb, int b_kslow, const float* bias,
                 float* C, long long ldc, int M, int N, int K, int relu, float* split_ws, long long split_ws_floats, hipStream_t s);
// y[i] = x[i] * mask[i] * scale  (mask: one byte per element, row-major [rows][cols]; x, y with row stride ld)
void launch_dropout(const float* x, const uint8_t* mask, float scale, float* y, int rows, int cols, long long ld, hipStream_t s);
// dx = dy * mask * scale * (act > 0 || !relu)      (backward of Linear -> ReLU -> Dropout, act = the ReLU output)
void launch_dropout_relu_bwd(const float* dy, const uint8_t* mask, float scale, const float* act, int relu, float* dx, int rows,
                             int cols, long long ld, hipStream_t s);
// column sums: out[c] = sum_rows x[row*ld + c]
void launch_colsum(const float* x, int rows, int cols, long long ld, float* out, hipStream_t s);
// Bernoulli(1 - p) keep masks from a counter-based generator (seed, step, tensor id)
void launch_dropout_mask(uint8_t* mask, long long n, float p, unsigned long long seed, unsigned long long stream_id, hipStream_t s);

// ---- heads, losses ----------------------------------------------------------------------------------------------------------------------
struct LossArgs {
    int B, hid, hid_stride;
    const float* hidden;                    // [B][hid_stride]  (after the last dropout)
    const float *wh, *bh;                   // [27][hid], [27]
    const uint8_t* label;                   // [B] {0 TP, 1 FN, 2 FP}
    const uint8_t* var_type;                // [B]
    const float *allele_freq, *coverage;    // [B]
    const uint8_t *var_base, *var_ref;      // [B]
    const float* weight;                    // [B] example weight
    float label_smoothing, close_window, focal_alpha, focal_gamma, fp_weight, binary_weight, aux_weight, aux_bases_weight,
        aux_allele_weight;
    float* logits;                          // [B][27] raw head outputs
    float* dlogits;                         // [B][27]
    float* losses;                          // [8]: loss, bin, vt, af, cov, vb, vr, (unused)
    uint8_t* close;                         // [B][2]: bin_close, vt_close
    float* site_terms;                      // [B][8] scratch
    // data parallelism: the normalisers of the FULL batch divided by the number of ranks (0 = this rank's own), so that the
    // average of the ranks' gradients is the full-batch gradient nn.DataParallel computes (main.py:117)
    float mean_sites;                       // replaces B in the .mean() terms
    float ce_den[2];                        // replace sum_b w[y_b] of the two weighted cross-entropies (var base, ref base)
};
void launch_heads_loss(const LossArgs& a, hipStream_t s);
// dhidden[b][k] = sum_j dlogits[b][j] wh[j][k];  gwh[j][k] = sum_b dlogits[b][j] hidden[b][k];  gbh[j] = sum_b dlogits[b][j]
void launch_heads_bwd(const float* dlogits, const float* hidden, const float* wh, int B, int hid, int hid_stride, float* dhidden,
                      float* gwh, float* gbh, hipStream_t s);   // hidden, dhidden: row stride hid_stride

// ---- optimiser ---------------------------------------------------------------------------------------------------------------------------
void launch_sumsq(const float* g, long long n, double* block_partials, int* n_blocks, hipStream_t s);
// norm = sqrt(sum partials); coef = clip > 0 ? min(1, clip / (norm + 1e-6)) : 1   -> out[0] = norm, out[1] = coef
void launch_clip_coef(const double* block_partials, int n_blocks, float clip, float* out, hipStream_t s);
void launch_adam(float* p, const float* g, float* m, float* v, long long n, const float* clip_out, float lr, float b1, float b2,
                 float eps, float bc1, float bc2, hipStream_t s);

}  // namespace dan
