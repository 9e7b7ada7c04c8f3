// HIP kernels of the DAN inference forward for gfx950 (MI355X, CDNA4).  fp32 path.
//
// Hot loop: every convolution of the stack is an implicit GEMM  D[out-channel][position] +=
// W[out-channel][k] * X[k][position]  on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).  One
// workgroup (8 waves, two per SIMD) owns ONE READ: its 201 x 128 activation lives in LDS for the
// whole segment of layers (position-major rows of 136 floats, zero halo rows either side), each wave
// owns 32 output channels x 7 (or 6) of the 13 position tiles (56 accumulator registers), weights stream from
// L2 straight into A fragments (host-packed in fragment order, 1 KiB coalesced per wave-load), the
// activation B fragments come from LDS as ds_read_b128 (the K order inside a 16-channel group is
// permuted so that one 16-byte read feeds four MFMA k-steps).  ReLU, folded BatchNorm, the 1x1
// residual GEMM and the 128->32 highway bottleneck GEMM are fused as epilogues on the LDS-resident
// read; only segment boundaries (the site-level read-mean of dl4vc/model.py:766-772 and the final
// max/mean pool) and the 32-channel bottleneck outputs go to HBM.
//
// Reference semantics followed (file:line in /root/reference): dl4vc/model.py:450-627 (encode),
// :728-778 (layer loop), :824-859 (pool + highway concat), :917-958 (FC + heads),
// dl4vc/trainer.py:609-623 (softmax scores).
#include "dan_device.h"

namespace dan {

// Diagnostic build only (tools/seg_probe.hip defines DAN_STAMPS): per-wave s_memtime stamps of the segment
// kernel's phases.  In the shipped library the macro expands to nothing.
#ifdef DAN_STAMPS
__device__ unsigned long long* g_stamps;
constexpr int NSTAMP = 64;
#define STAMP(k)                                                                                      \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                                         \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                           \
        if (lane == 0 && (k) < NSTAMP) g_stamps[((size_t)stamp_row * NWAVE + wave) * NSTAMP + (k)] = t_; \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
// Winograd F(2,3) form of the dilation-2 convolutions (exact-fp32 MFMAs, 4 channel GEMMs per 2 outputs instead of 6):
//   outputs y(P), y(P+2) of one tile come from x(P-2), x(P), x(P+2), x(P+4):
//     V = [x0 - x2, x1 + x2, x2 - x1, x1 - x3],  M_k = U_k V_k  (U = [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2], host-packed),
//     y(P) = M0 + M1 + M2,  y(P+2) = M1 - M2 - M3.
// Mapping: every wave owns ONE 16-channel tile (wave index) and all positions.  MFMA column n of position tile m is the
// Winograd tile with base  P = hf*WHB + 4*(q*MW + m) + c,  c = n&1, hf = (n>>2)&1, q = ((n>>1)&1) + 2*(((n>>3)^(n>>2))&1):
// a lane walks consecutive tiles of one parity class, so two of its four input rows carry over from tile to tile
// (2 ds_read_b128 per 16 MFMAs), and the upper half of the read is tiled from position 102 (== 2 mod 4) so that the 16 lanes
// of every ds_read_b128 group touch rows of all 8 residues mod 8 -- conflict-free with the 136-float row stride.
// Lower-half lanes own positions [0, 102), upper-half lanes [102, 208); tiles outside (and the rows they read, which
// may run past the activation into the constants that follow it in LDS) produce columns nobody stores.
static_assert((HALO + WHB + 4 * (4 * MW - 1) + 1 + 4 + 1) * LDS_S <= LDS_ROWS * LDS_S + MAX_LAYERS * CST_FLOATS, "wino reads stay inside LDS");
static_assert(WHB + 4 * (4 * MW - 1) + 3 >= MPOS - 1 && 4 * (4 * MW - 1) + 3 >= WHB - 1, "wino tiling covers the read");
// ... and the six-tile form of the split kernel's short units (dan_device.h MW_SHORT): [0, 96) + [94, 190)
static_assert(WHB_SHORT + 4 * (4 * MW_SHORT - 1) + 3 >= MPOS_SHORT - 1 && 4 * (4 * MW_SHORT - 1) + 3 >= WHB_SHORT - 1 && WHB_SHORT % 4 == 2,
              "short wino tiling covers its unit");

// 1x1 GEMM in the same column mapping: acc[m][o] += W[own 16 channels][128] * x(P(m) + 2 o)
template <int TW>
__device__ __forceinline__ void gemm1x1_wino(v4f (&acc)[TW][2], const float* xrow, gv4f_ptr wl, v4f a_first) {
    v4f a_nxt = a_first, b[TW][2];
#pragma unroll
    for (int m = 0; m < TW; ++m)
#pragma unroll
        for (int o = 0; o < 2; ++o) b[m][o] = *(const v4f*)(xrow + (4 * m + 2 * o) * LDS_S);
    for (int g = 0; g < KGC; ++g) {
        const v4f a = a_nxt;
        const int gn = (g + 1 < KGC) ? g + 1 : g;
        a_nxt = wl[(size_t)gn * (KGC * 64)];
#pragma unroll
        for (int m = 0; m < TW; ++m) {
            __builtin_amdgcn_s_setprio(3);                      // MFMAs ahead of the other wave's LDS/VALU work, as in conv_gemm_wino
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int o = 0; o < 2; ++o) acc[m][o] = mfma16(a[s], b[m][o][s], acc[m][o]);
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int o = 0; o < 2; ++o) b[m][o] = *(const v4f*)(xrow + (4 * m + 2 * o) * LDS_S + gn * 16);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// segment kernel: one workgroup = one read, layers [l_begin, l_end) with the read resident in LDS
// ------------------------------------------------------------------------------------------------
// PERSIST = false: one workgroup per pileup row (the default: every row is computed).  PERSIST = true: one workgroup per
// CU walking the device-side list of rows to do with the grid's stride -- used with empty-row skipping, where dispatching
// a workgroup per row only to have a quarter of them exit costs the command processor ~0.26 us each, serialised.  The row
// loop costs registers (what is live across rows), so that form re-reads its arguments per row and does without the
// cross-layer weight prefetch; it is ~1 % slower per computed row.
// SPLIT = true: windows of 209..304 columns, every read as two overlapping units (SegmentArgs::units == 2, plan_units): a work
// item is (row, unit), `L` below is the UNIT's length and every position-indexed pointer is offset to the unit's first column;
// what differs from the one-unit form is addressing (window stride Lw), the allele-agreement predicates (taken over the whole
// window, not the unit) and the stores (own columns only, y out of place).  SPLIT = false compiles to the code it always was.
// TW: Winograd tiles per lane -- MW = 7 (one unit of up to 208 columns) or, SPLIT, MW_SHORT = 6 (units of up to 190 columns: every
// unit of a split Winograd read, e.g. 161 columns at a 301-column window; launch_segment).
template <bool WINO, bool PERSIST, bool SPLIT, int TW = MW>
__global__ __launch_bounds__(SEG_THREADS, NWAVE / 4) void segment_kernel(SegmentArgs a_by_value) {
    static_assert((!(SPLIT && WINO) && TW == MW) || (SPLIT && WINO && TW == MW_SHORT), "six tiles per lane: the split Winograd kernel, and only it");
    constexpr int TILES = TW == MW ? MT : (MPOS_SHORT + 15) / 16;           // 16-column tiles of the unit (bottleneck, table walk)
    // one allocation, so that the layout the Winograd tiles past the window rely on (constants right after the activation
    // rows) is explicit
    __shared__ __attribute__((aligned(16))) float lds[LDS_ROWS * LDS_S + MAX_LAYERS * CST_FLOATS];
    float* const xs = lds;
    float* const cst = lds + LDS_ROWS * LDS_S;
    typedef const __attribute__((address_space(4))) SegmentArgs* kernarg_ptr;
    // (the row body sits at function scope with an explicit back edge: wrapped in a lambda, or in a for loop left by a
    // compile-time break, the same code costs the non-persistent form 56 spilled registers)
    constexpr int UNITS = SPLIT ? 2 : 1;
    const int n_work = (a_by_value.work_count ? *a_by_value.work_count : a_by_value.n_rows) * UNITS;
    // XCD-aware order: workgroups b and b + 8 share an XCD (round-robin dealing), so the rows are cut into 8 contiguous
    // slices of `per` rows and workgroup b takes row (b % 8) * per + b / 8 (+ a multiple of the grid's eighth when
    // persistent): the 64 reads of a site, which all read that site's pool image and the same weights, stay in one L2.
    const int per = a_by_value.work_count ? (n_work + 7) >> 3 : a_by_value.xcd_rows * UNITS;
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;                                // position inside the XCD's slice
    int wk = xcd * per + j;
    if (j >= per || wk >= n_work) return;
next_row:                                                   // (PERSIST only: back edge at the bottom)
    {
    // PERSIST: the arguments are re-read from the kernarg segment for every row through a pointer the compiler cannot see
    // through -- kept in SGPRs across the row loop they, and what is derived from them, no longer fit.  The one-row form
    // reads the by-value parameter directly (through the pointer it spills 57 registers).
    kernarg_ptr ap = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    if (PERSIST) asm volatile("" : "+s"(ap));
    auto args = [&]() -> decltype(auto) { if constexpr (PERSIST) return (*ap); else return (a_by_value); };
    const auto& a = args();
    // (SPLIT) the unit of this work item: all wave-uniform
    [[maybe_unused]] const int unit = wk & 1;
    const int L = SPLIT ? (unit ? a.u_len[1] : a.u_len[0]) : a.L;
    const int Lw = SPLIT ? a.Lw : L;                             // position stride of the tensors
    const int u_off = SPLIT ? (unit ? a.u_off[1] : a.u_off[0]) : 0;
    [[maybe_unused]] const int own_lo = SPLIT ? (unit ? a.own_lo[1] : a.own_lo[0]) : 0;
    [[maybe_unused]] const int own_hi = SPLIT ? (unit ? a.own_hi[1] : a.own_hi[0]) : L;
    [[maybe_unused]] const int stamp_row = wk;
    int tid = threadIdx.x;
    if (PERSIST) asm volatile("" : "+v"(tid));              // per-row: nothing derived from the thread index is hoisted out of
    const int lane = tid & 63;                              // the row loop (it would live through every GEMM and spill)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = SPLIT ? wk >> 1 : wk;
    const int row_index = __builtin_amdgcn_readfirstlane(a.work_count ? a.work[wrow] : wrow);     // uniform: everything derived stays scalar
    const int site = (int)((unsigned)row_index / (unsigned)a.R);
    const int r = row_index - site * a.R;
    const size_t read_idx = (size_t)site * a.R + r;
    float* yrow = a.y + (read_idx * (size_t)Lw + u_off) * CPAD;                   // this unit's first column of the segment's input
    float* yout = SPLIT ? a.y_out + (read_idx * (size_t)Lw + u_off) * CPAD : yrow;  // ... and of its output (in place unless SPLIT)
    const int pos = lane & 15, kk = lane >> 4;
    const int cq = wave & 3, ph = wave >> 2;                  // channel quarter, position half
    const int m_base = ph * MTW, cnt = ph ? MT - MTW : MTW;     // this wave's position tiles [m_base, m_base + cnt)
    int chb[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) chb[n] = (cq * NT + n) * 16 + kk * 4;

    STAMP(0);
    // Winograd column mapping of this lane (used by the WINO instantiation on its dilation-2 layers)
    [[maybe_unused]] const int wP0 = wino_base<TW>(lane);
    [[maybe_unused]] const int wlim = ((lane >> 2) & 1) ? wino_cover(TW) : wino_half_base(TW);      // positions this lane's tiling owns: p < wlim
    [[maybe_unused]] const int chw = wave * 16 + kk * 4;
    auto dil_of = [&](int l) { return (l == 0) ? 1 : (l + 1 < a.n_layers ? a.dil_mid : a.dil_final); };
    auto wino_layer = [&](int l) { return WINO && l > 0; };
    // First weight fragments of a conv GEMM, requested a stage ahead of it (eight plain vector variables: an aggregate, or an
    // array, captured by reference in the layer lambdas ends up in scratch memory).  The PERSIST Winograd form has no
    // registers to carry them across stages: its conv GEMM fetches its own.
    constexpr bool CARRY = !(WINO && PERSIST);
    v4f pc0 = splat(0.f), pc1 = pc0, pc2 = pc0, pc3 = pc0, pn0 = pc0, pn1 = pc0, pn2 = pc0, pn3 = pc0;
    auto first_frags = [&](int l, v4f& f0, v4f& f1, v4f& f2, v4f& f3) {
        const float* blk = a.wl + (size_t)l * LAYER_STRIDE;
        if (wino_layer(l)) {
            gv4f_ptr w0 = (gv4f_ptr)(blk + WW_OFF) + wave * 64 + lane;
            constexpr size_t KS = (size_t)KGC * (KGC * 64);
            f0 = w0[0]; f1 = w0[KS]; f2 = w0[2 * KS]; f3 = w0[3 * KS];
        } else {
            gv4f_ptr w0 = (gv4f_ptr)(blk + W_OFF) + (cq * NT) * 64 + lane;
            f0 = w0[0]; f1 = w0[64];
        }
    };
    // layer 1 by table (dan_kernels.h L0_*): no encoded image, no GEMM -- unless the tap asks for the encoded input itself
    const bool l0_lookup = a.l_begin == 0 && a.l0_tab != nullptr && !(a.tap && a.tap_layer == 0) && (a.l_end - a.l_begin) <= L0_MAX_LAYERS;
    float* const tokf = cst + (a.l_end - a.l_begin) * CST_FLOATS;           // [L + 2][L0_TOK] (column p at entry p + 1), behind the segment's constants
    if ((!WINO || a.l_begin == 0) && !l0_lookup) first_frags(a.l_begin, pc0, pc1, pc2, pc3);
    auto stage_constants = [&]() {
        for (int i = tid; i < (a.l_end - a.l_begin) * CST_FLOATS; i += SEG_THREADS) {
            const int l = i / CST_FLOATS, j = i - l * CST_FLOATS;
            cst[i] = a.wl[(size_t)(a.l_begin + l) * LAYER_STRIDE + CST_OFF + j];
        }
    };

    if (a.l_begin == 0) {
        if (l0_lookup) {                                     // (every row of the window is written by the table walk: only the others are cleared)
            for (int i = tid; i < (LDS_ROWS - L) * (LDS_S / 4); i += SEG_THREADS) {
                const int rr = i / (LDS_S / 4), c4 = i - rr * (LDS_S / 4);
                const int row = rr < HALO ? rr : rr + L;
                *(v4f*)(xs + row * LDS_S + c4 * 4) = splat(0.f);
            }
        } else {
            for (int i = tid; i < LDS_ROWS * LDS_S / 4; i += SEG_THREADS) ((v4f*)xs)[i] = splat(0.f);
        }
        stage_constants();
        __syncthreads();
        // ---- encode (dl4vc/model.py:450-627): canonical 48-channel order
        //      [read emb+pe (20) | ref emb+pe (20) | q*0.01 | strand*0.5 | refmatch | varmatch | lenmask | 0 0 0]
        const size_t rbase = read_idx * (size_t)Lw + u_off, sbase = (size_t)site * Lw + u_off;
        const int p = tid;
        const bool in = p < L;
        int tok = 0, q = 0, st = 0, rf = 0, rm = 0, vm = 0;
        if (in) {
            tok = a.reads[rbase + p]; q = a.qual[rbase + p]; st = a.strand[rbase + p];
            rf = a.ref[sbase + p]; rm = a.ref_mask[sbase + p]; vm = a.var_mask[sbase + p];
        }
        // a read agrees with an allele iff it equals the mask wherever the mask is non-zero (model.py:592-593) -- over the WHOLE
        // window: a unit of a split read looks at the columns of the other unit too (Lw <= 304 < SEG_THREADS)
        int tok_w = tok, rm_w = rm, vm_w = vm;
        if constexpr (SPLIT) {
            tok_w = rm_w = vm_w = 0;
            if (tid < Lw) { tok_w = a.reads[rbase - u_off + tid]; rm_w = a.ref_mask[sbase - u_off + tid]; vm_w = a.var_mask[sbase - u_off + tid]; }
        }
        const int agree_ref = __syncthreads_and((rm_w == 0) || (tok_w == rm_w));
        const int agree_var = __syncthreads_and((vm_w == 0) || (tok_w == vm_w));
        if (l0_lookup) {
            // the column's table index and scalar channels; the columns either side of the unit: the all-zero table entry, zero scalars
            // (entry 0 also carries the read's agreement with the reference allele: refmatch = agreement x lenmask, one weight per read)
            if (p <= L + 1) {
                const int pc = p - 1;                            // thread e stages column e - 1
                int tk = 0, qq = 0, ss = 0, rr = 0, mm = 0, vv = 0;
                const bool inside = pc >= 0 && pc < L;
                if (inside) {
                    tk = a.reads[rbase + pc]; qq = a.qual[rbase + pc]; ss = a.strand[rbase + pc];
                    rr = a.ref[sbase + pc]; mm = a.ref_mask[sbase + pc]; vv = a.var_mask[sbase + pc];
                }
                float* t = tokf + p * L0_TOK;
                const v4f t0 = {__builtin_bit_cast(float, inside ? min(tk, VOCAB - 1) * 10 + min(rr, VOCAB - 1) : 100), (float)qq * 0.01f, (float)ss * 0.5f,
                                (inside && mm != 0) ? 1.f : 0.f};
                const v4f t1 = {(inside && vv != 0 && agree_var) ? 1.f : 0.f, (p == 0 && agree_ref) ? 1.f : 0.f, 0.f, 0.f};
                *(v4f*)t = t0; *(v4f*)(t + 4) = t1;
            }
        } else if (in) {
            float* row = xs + (HALO + p) * LDS_S;
            const float* er = a.emb + min(tok, VOCAB - 1) * EMBED;
            const float* ef = a.emb + min(rf, VOCAB - 1) * EMBED;
            const float* pp = a.pe + (u_off + p) * EMBED;
#pragma unroll
            for (int e = 0; e < EMBED; ++e) {
                const float pv = pp[e];
                row[e] = er[e] + pv;
                row[EMBED + e] = ef[e] + pv;
            }
            row[40] = (float)q * 0.01f;                      // Q_SCORE_SCALE_FACTOR, model.py:24
            row[41] = (float)st * 0.5f;                      // STRAND_ENCODE_FACTOR, model.py:16
            row[42] = (rm != 0 && agree_ref) ? 1.f : 0.f;
            row[43] = (vm != 0 && agree_var) ? 1.f : 0.f;
            row[44] = (rm != 0) ? 1.f : 0.f;                 // length mask comes from ref_masks, model.py:578-584
        }
    } else {
        // ---- resume from the previous segment's output, adding the broadcast read-mean (model.py:734-742)
        const v4f* src = (const v4f*)yrow;
        const v4f* pl = a.pool ? (const v4f*)(a.pool + ((size_t)site * Lw + u_off) * CPAD) : nullptr;
        // a single CU streams at (bytes in flight) / latency: put the whole read (and the pool image) in flight
        // at once -- 26 + 26 sixteen-byte loads per lane -- instead of a few loads per round trip
        const int n4 = L * (CPAD / 4);
        constexpr int NP = TILES * 16 * (CPAD / 4) / SEG_THREADS;
        v4f vy[NP], vp[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = tid + k * SEG_THREADS;
            vy[k] = (i < n4) ? src[i] : splat(0.f);
        }
        if (pl) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int i = tid + k * SEG_THREADS;
                vp[k] = (i < n4) ? pl[i] : splat(0.f);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NP; ++k) vp[k] = splat(0.f);
        }
        // while they fly: zero the rows the read does not cover (halo rows and rows >= L; the 8 pad floats of a row are
        // never read) and stage the constants
        for (int i = tid; i < (LDS_ROWS - L) * (LDS_S / 4); i += SEG_THREADS) {
            const int rr = i / (LDS_S / 4), c4 = i - rr * (LDS_S / 4);
            const int row = rr < HALO ? rr : rr + L;
            *(v4f*)(xs + row * LDS_S + c4 * 4) = splat(0.f);
        }
        stage_constants();
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = tid + k * SEG_THREADS;
            if (i < n4) *(v4f*)(xs + (HALO + (i >> 5)) * LDS_S + (i & 31) * 4) = vy[k] + vp[k];
        }
        if (WINO && CARRY) first_frags(a.l_begin, pc0, pc1, pc2, pc3);   // (after the 52-load prologue: no registers to spare before)
    }
    __syncthreads();
    STAMP(1);
    if (a.tap && a.tap_layer == 0 && a.l_begin == 0) copy_out(xs, a.tap + (read_idx * (size_t)Lw + u_off) * CPAD, own_lo, own_hi, tid);

    // per-layer prologue/tail shared by both forms
    v4f wbot[KGC];
    auto layer_tail = [&](int l, const float* lc, bool late_prefetch) {
        [[maybe_unused]] const int sb = 2 + (l - a.l_begin) * 8;
        __syncthreads();
        STAMP(sb + 6);
        if (a.tap && a.tap_layer == l + 1) {                // (diagnostic path: its addresses must not be hoisted into registers)
            int t = tid;
            asm volatile("" : "+v"(t));
            copy_out(xs, a.tap + (read_idx * (size_t)Lw + u_off) * CPAD, own_lo, own_hi, t);
        }
        if (WINO && CARRY && late_prefetch && l + 1 < a.l_end) first_frags(l + 1, pn0, pn1, pn2, pn3);   // (direct first layer of the Winograd form)
        // WINO instantiation, not the segment's last layer: the bottleneck GEMM of this layer is deferred into the next
        // layer's conv stage, where the four older waves run it on the same LDS-resident input while the younger wave of
        // each SIMD is still in its conv GEMM (the arbiter serves the older wave first, so it finishes the conv early and
        // would only wait at the barrier)
        if (a.has_hw && !(WINO && l + 1 < a.l_end))
            bottleneck<NWAVE, SPLIT, TILES>(xs, wbot, lc + CST_BBOT, a.h + (size_t)l * a.h_layer_stride + (read_idx * (size_t)Lw + u_off) * HPAD,
                                     own_hi, wave, lane, own_lo);
        STAMP(sb + 7);
        if (CARRY) { pc0 = pn0; pc1 = pn1; pc2 = pn2; pc3 = pn3; }
    };

    // ---- direct form: wave = (channel quarter, position half), 3-tap implicit GEMM
    auto direct_layer = [&](int l) {
        const float* wblk = a.wl + (size_t)l * LAYER_STRIDE;
        const float* lc = cst + (l - a.l_begin) * CST_FLOATS;
        const bool residual = (a.res_mask >> l) & 1u;
        const int kg = (l == 0) ? KG0 : KGC;
        const int dil = dil_of(l);
        gv4f_ptr w_conv = (gv4f_ptr)(wblk + W_OFF) + (cq * NT) * 64 + lane;
        gv4f_ptr w_res = (gv4f_ptr)(wblk + WRES_OFF) + (cq * NT) * 64 + lane;
        gv4f_ptr w_bot = (gv4f_ptr)(wblk + WBOT_OFF) + lane;
        [[maybe_unused]] const int sb = 2 + (l - a.l_begin) * 8;
        // first fragments of the later GEMM stages of this layer and of the next conv: loaded now, used after
        // the conv GEMM, so their L2 latency is never exposed
        if (!WINO && l + 1 < a.l_end) first_frags(l + 1, pn0, pn1, pn2, pn3);
        // x_in of a residual layer is the layer input BEFORE the pool add (model.py:732): for the first layer of a
        // pooled segment it is re-read from HBM
        const bool from_global = (l == a.l_begin) && (a.l_begin != 0) && (a.pool != nullptr);
        v4f pre_res[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) pre_res[n] = residual ? w_res[n * 64] : splat(0.f);
        static_assert(NT == 2, "two direct-form fragments are carried");
        const v4f pre_dir[NT] = {pc0, pc1};

        v4f acc[MTW][NT];
        {
            v4f bias[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bias[n] = *(const v4f*)(lc + CST_BIAS + chb[n]);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = bias[n];
        }
        STAMP(sb + 0);
        conv_gemm(acc, xs, w_conv, pre_dir, kg, 3, dil, lane, m_base, cnt);
        STAMP(sb + 1);
        // this wave's eight bottleneck weight fragments of the layer: issued now, consumed after the epilogue
        // (and the residual GEMM), so the ~1.5k-cycle loaded-L2 latency is off the critical path
        // (for residual layers they are issued after the residual GEMM instead, to keep 32 registers free in it)
        const bool bot_here = a.has_hw && !(WINO && l + 1 < a.l_end);     // else deferred into the next layer's conv stage
        if (bot_here && !residual) {
#pragma unroll
            for (int g = 0; g < KGC; ++g) wbot[g] = w_bot[(g * 2 + (wave & 1)) * 64];
        }
        // ---- epilogue: ReLU then eval-mode BatchNorm as one affine (model.py:749-751); rows >= L stay zero.
        // Tile m of this wave is position tile m_base + m; the upper half's seventh slot (m = 6) does not exist.
        {
            v4f sc[NT], sh[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) { sc[n] = *(const v4f*)(lc + CST_SCALE + chb[n]); sh[n] = *(const v4f*)(lc + CST_SHIFT + chb[n]); }
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const bool live = ((m_base + m) * 16 + pos) < L;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    v4f v = acc[m][n];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = live ? relu1(v[j]) * sc[n][j] + sh[n][j] : 0.f;
                    acc[m][n] = v;
                }
            }
        }
        STAMP(sb + 2);
        __syncthreads();                                    // every wave has finished reading the layer input
        STAMP(sb + 3);
        if (residual) {
            // y = W1x1 * bn(relu(conv(x))) + b + x_in   (model.py:753-761).  x_in is the layer input BEFORE
            // the pool add (model.py:732): for the first layer of a pooled segment it is re-read from HBM.
            v4f bres[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bres[n] = *(const v4f*)(lc + CST_BRES + chb[n]);
            if (!from_global) {
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    if (m < cnt) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            v4f* cell = (v4f*)(xs + (HALO + (m_base + m) * 16 + pos) * LDS_S + chb[n]);
                            const v4f old = *cell;
                            *cell = acc[m][n];
                            acc[m][n] = old + bres[n];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const int p = (m_base + m) * 16 + pos;
                    if (m < cnt) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const v4f old = (p < L) ? *(const v4f*)(yrow + (size_t)p * CPAD + chb[n]) : splat(0.f);
                            *(v4f*)(xs + (HALO + p) * LDS_S + chb[n]) = acc[m][n];
                            acc[m][n] = old + bres[n];
                        }
                    }
                }
            }
            __syncthreads();
            STAMP(sb + 4);
            conv_gemm(acc, xs, w_res, pre_res, KGC, 1, 0, lane, m_base, cnt);
            STAMP(sb + 5);
            if (bot_here) {
#pragma unroll
                for (int g = 0; g < KGC; ++g) wbot[g] = w_bot[(g * 2 + (wave & 1)) * 64];
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int p = (m_base + m) * 16 + pos;
                if (m < cnt) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) *(v4f*)(xs + (HALO + p) * LDS_S + chb[n]) = (p < L) ? acc[m][n] : splat(0.f);
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                if (m < cnt) {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        *(v4f*)(xs + (HALO + (m_base + m) * 16 + pos) * LDS_S + chb[n]) = acc[m][n];
                }
            }
        }
        layer_tail(l, lc, true);
    };

    // ---- Winograd form (dilation-2 layers of the WINO instantiation): wave = one 16-channel tile, all positions
    [[maybe_unused]] auto wino_layer_body = [&](int l) {
        if constexpr (WINO) {
        const float* wblk = a.wl + (size_t)l * LAYER_STRIDE;
        const float* lc = cst + (l - a.l_begin) * CST_FLOATS;
        const bool residual = (a.res_mask >> l) & 1u;
        gv4f_ptr w_bot = (gv4f_ptr)(wblk + WBOT_OFF) + lane;
        [[maybe_unused]] const int sb = 2 + (l - a.l_begin) * 8;
        const bool from_global = (l == a.l_begin) && (a.l_begin != 0) && (a.pool != nullptr);
        gv4f_ptr w_w = (gv4f_ptr)(wblk + WW_OFF) + wave * 64 + lane;
        gv4f_ptr w_r1 = (gv4f_ptr)(wblk + WRES_OFF) + wave * 64 + lane;
        float* xw = xs + (HALO + wP0) * LDS_S;          // row of x(P(0)) (channel 0)
        v4f out[TW][2], pre_r1;
        int wp = wP0;
        float* xq;
        {
            v4f acc[TW][4];
            const v4f bias = *(const v4f*)(lc + CST_BIAS + chw);
#pragma unroll
            for (int m = 0; m < TW; ++m) { acc[m][0] = splat(0.f); acc[m][1] = bias; acc[m][2] = splat(0.f); acc[m][3] = splat(0.f); }
            STAMP(sb + 0);
            if constexpr (CARRY) {
                const v4f pre_w[4] = {pc0, pc1, pc2, pc3};
                conv_gemm_wino(acc, xw - 2 * LDS_S + kk * 4, w_w, pre_w);
            } else {
                gv4f_ptr w0 = w_w;
                const v4f pre_w[4] = {w0[0], w0[(size_t)KGC * (KGC * 64)], w0[(size_t)2 * KGC * (KGC * 64)], w0[(size_t)3 * KGC * (KGC * 64)]};
                conv_gemm_wino(acc, xw - 2 * LDS_S + kk * 4, w_w, pre_w);
            }
            STAMP(sb + 1);
            asm volatile("" : "+v"(wp));       // keeps everything derived from it (addresses, masks) out of the GEMM's live set
            xq = xs + (HALO + wp) * LDS_S;
            // output transform, then ReLU and the folded BatchNorm (model.py:749-751); positions >= L stay zero
            // (on register pairs: packed fp32 adds and multiply-adds, the same operations in the same order per element.  Columns
            // at or past L are not masked here: they are never stored -- `wl` below -- and their LDS rows keep the prologue's zeros.)
            const v4f sc = *(const v4f*)(lc + CST_SCALE + chw), sh = *(const v4f*)(lc + CST_SHIFT + chw);
            const v2f sc_lo = {sc[0], sc[1]}, sc_hi = {sc[2], sc[3]}, sh_lo = {sh[0], sh[1]}, sh_hi = {sh[2], sh[3]};
            auto half = [](v4f v, int h) { return (v2f){v[2 * h], v[2 * h + 1]}; };
            v2f neg1 = {-1.f, -1.f};
            asm volatile("" : "+v"(neg1));                 // (opaque, or the subtraction comes back)
            auto act = [](v2f y, v2f s, v2f t) {
                const v2f r = {relu1(y[0]), relu1(y[1])};
                return __builtin_elementwise_fma(r, s, t);
            };
#pragma unroll
            for (int m = 0; m < TW; ++m) {
                const v2f y0l = half(acc[m][0], 0) + half(acc[m][1], 0) + half(acc[m][2], 0), y0h = half(acc[m][0], 1) + half(acc[m][1], 1) + half(acc[m][2], 1);
                // (a - b as fma(b, -1, a): exact, and a packed instruction -- hipcc splits a subtraction of pairs into two scalar ones)
                const v2f y1l = __builtin_elementwise_fma(half(acc[m][3], 0), neg1, __builtin_elementwise_fma(half(acc[m][2], 0), neg1, half(acc[m][1], 0)));
                const v2f y1h = __builtin_elementwise_fma(half(acc[m][3], 1), neg1, __builtin_elementwise_fma(half(acc[m][2], 1), neg1, half(acc[m][1], 1)));
                const v2f o0l = act(y0l, sc_lo, sh_lo), o0h = act(y0h, sc_hi, sh_hi), o1l = act(y1l, sc_lo, sh_lo), o1h = act(y1h, sc_hi, sh_hi);
                out[m][0] = (v4f){o0l[0], o0l[1], o0h[0], o0h[1]};
                out[m][1] = (v4f){o1l[0], o1l[1], o1h[0], o1h[1]};
            }
        }
        int wl = min(wlim, L);                           // columns this lane stores: its tiling's, inside the window
        asm volatile("" : "+v"(wl));                    // (opaque: hipcc, knowing p < wl implies p < L, merges the residual's two loads below into one through a selected POINTER and fails on the LDS one's cast)
        // the accumulators fill the register file through the GEMM and the output transform: the later stages' first
        // fragments are requested only now (the barriers, the write-back and the residual GEMM cover their latency)
        __builtin_amdgcn_sched_barrier(0);
        pre_r1 = residual ? w_r1[0] : splat(0.f);
        if (CARRY && !residual && l + 1 < a.l_end) first_frags(l + 1, pn0, pn1, pn2, pn3);      // (residual layers: after their 1x1 GEMM)
        // the previous layer's bottleneck GEMM (deferred by its layer_tail): the LDS image is still that layer's output
        if (a.has_hw && l > a.l_begin && wave < NWAVE / 2) {
            gv4f_ptr w_bp = (gv4f_ptr)(wblk - LAYER_STRIDE + WBOT_OFF) + lane;
#pragma unroll
            for (int g = 0; g < KGC; ++g) wbot[g] = w_bp[(g * 2 + (wave & 1)) * 64];
            bottleneck<NWAVE / 2, SPLIT, TILES>(xs, wbot, lc - CST_FLOATS + CST_BBOT,
                                         a.h + (size_t)(l - 1) * a.h_layer_stride + (read_idx * (size_t)Lw + u_off) * HPAD, own_hi, wave, lane, own_lo);
        }
        const bool bot_here = a.has_hw && !(l + 1 < a.l_end);
        if (bot_here && !residual) {
#pragma unroll
            for (int g = 0; g < KGC; ++g) wbot[g] = w_bot[(g * 2 + (wave & 1)) * 64];
        }
        STAMP(sb + 2);
        __syncthreads();                                // every wave has finished reading the layer input
        STAMP(sb + 3);
        if (residual) {
            const v4f bres = *(const v4f*)(lc + CST_BRES + chw);
            // the 1x1 GEMM's input (this layer's BatchNorm output) goes into the image, the layer input it replaces seeds the accumulators.
            // Two loops under one uniform branch (as one loop with the choice inside, every cell carried both paths and their masks).
            if (from_global) {
#pragma unroll
                for (int m = 0; m < TW; ++m)
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        const int p = wp + 4 * m + 2 * o;
                        if (p < wl) {
                            // (uniform base + 32-bit lane offset: one VGPR per address instead of a 64-bit pair)
                            const v4f old = *(const v4f*)((const char*)yrow + (unsigned)(p * CPAD + chw) * 4u);
                            *(v4f*)(xq + (4 * m + 2 * o) * LDS_S + chw) = out[m][o];
                            out[m][o] = old + bres;
                        }
                    }
            } else {
#pragma unroll
                for (int m = 0; m < TW; ++m)
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        const int p = wp + 4 * m + 2 * o;
                        if (p < wl) {
                            v4f* cell = (v4f*)(xq + (4 * m + 2 * o) * LDS_S + chw);
                            const v4f old = *cell;
                            *cell = out[m][o];
                            out[m][o] = old + bres;
                        }
                    }
            }
            __syncthreads();
            STAMP(sb + 4);
            gemm1x1_wino(out, xq + kk * 4, w_r1, pre_r1);
            STAMP(sb + 5);
            if (CARRY && l + 1 < a.l_end) first_frags(l + 1, pn0, pn1, pn2, pn3);
            if (bot_here) {
#pragma unroll
                for (int g = 0; g < KGC; ++g) wbot[g] = w_bot[(g * 2 + (wave & 1)) * 64];
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < TW; ++m)
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int p = wp + 4 * m + 2 * o;
                    if (p < wl) *(v4f*)(xq + (4 * m + 2 * o) * LDS_S + chw) = out[m][o];
                }
        } else {
#pragma unroll
            for (int m = 0; m < TW; ++m)
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int p = wp + 4 * m + 2 * o;
                    if (p < wl) *(v4f*)(xq + (4 * m + 2 * o) * LDS_S + chw) = out[m][o];
                }
        }
        layer_tail(l, lc, false);
        }
    };

    // ---- layer 1 by table (dan_kernels.h L0_*): thread = (four channels c4, position p = pr + 16 k); the image rows are written directly
    auto lookup_layer0 = [&]() {
        const float* lc = cst;
        [[maybe_unused]] const int sb = 2;
        if (!WINO && 1 < a.l_end) first_frags(1, pn0, pn1, pn2, pn3);
        STAMP(sb + 0);
        const int c4 = tid & 31, pr = tid >> 5;
        const v4f bias = *(const v4f*)(lc + CST_BIAS + c4 * 4), sc = *(const v4f*)(lc + CST_SCALE + c4 * 4), sh = *(const v4f*)(lc + CST_SHIFT + c4 * 4);
        // per tap: q, strand, lenmask (+ refmatch where the read agrees with the reference allele: one combined weight), varmatch
        v4f wq[3], wst[3], wlen[3], wvar[3];
        {
            const float agree = tokf[5];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                auto wv = [&](int k) { return *(const v4f*)(a.l0_tab + L0_WSC_OFF + (k * 3 + t) * CPAD + c4 * 4); };
                wq[t] = wv(0); wst[t] = wv(1); wvar[t] = wv(3);
                wlen[t] = wv(4) + agree * wv(2);
            }
        }
        const v4f* tj = (const v4f*)(a.l0_tab + L0_TJ_OFF) + c4;
        const v4f* pe = (const v4f*)(a.l0_tab + L0_PE_OFF) + (size_t)u_off * (CPAD / 4) + c4;
#pragma unroll 4
        for (int k = 0; k < TILES; ++k) {
            const int p = min(pr + 16 * k, L - 1);                   // (clamped: the last sweep's extra threads recompute column L - 1)
            const int var = (p == 0) ? 1 : (p == L - 1) ? 2 : 0;         // which neighbours the unit has at this column
            const v4f pe0 = pe[((size_t)var * Lw + p) * (CPAD / 4)];
            v4f s0[3], tv[3];
            v2f s1[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                s0[t] = *(const v4f*)(tokf + (p + t) * L0_TOK);          // column p + t - 1 at entry p + t
                s1[t] = *(const v2f*)(tokf + (p + t) * L0_TOK + 4);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) tv[t] = tj[(size_t)(t * L0_NTJ + __builtin_bit_cast(int, s0[t][0])) * (CPAD / 4)];
            v2f alo = {bias[0] + pe0[0], bias[1] + pe0[1]}, ahi = {bias[2] + pe0[2], bias[3] + pe0[3]};   // (packed fp32: two lanes per instruction)
            auto fma4 = [&](float sv, const v4f& w) {
                const v2f ss = {sv, sv};
                alo = __builtin_elementwise_fma(ss, (v2f){w[0], w[1]}, alo);
                ahi = __builtin_elementwise_fma(ss, (v2f){w[2], w[3]}, ahi);
            };
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                alo += (v2f){tv[t][0], tv[t][1]}; ahi += (v2f){tv[t][2], tv[t][3]};
                fma4(s0[t][1], wq[t]);
                fma4(s0[t][2], wst[t]);
                fma4(s0[t][3], wlen[t]);
                fma4(s1[t][0], wvar[t]);
            }
            v4f v;
            v[0] = relu1(alo[0]) * sc[0] + sh[0]; v[1] = relu1(alo[1]) * sc[1] + sh[1];
            v[2] = relu1(ahi[0]) * sc[2] + sh[2]; v[3] = relu1(ahi[1]) * sc[3] + sh[3];
            *(v4f*)(xs + (HALO + p) * LDS_S + c4 * 4) = v;
        }
        STAMP(sb + 2);
        const bool bot_here = a.has_hw && !(WINO && 1 < a.l_end);
        if (bot_here) {
            gv4f_ptr w_bot = (gv4f_ptr)(a.wl + WBOT_OFF) + lane;
#pragma unroll
            for (int g = 0; g < KGC; ++g) wbot[g] = w_bot[(g * 2 + (wave & 1)) * 64];
        }
        layer_tail(0, lc, true);
    };

    if constexpr (WINO) {
        // host contract (launch_segment): every layer after the network's first has dilation 2
        int l = a.l_begin;
        if (l == 0) { if (l0_lookup) lookup_layer0(); else direct_layer(0); l = 1; }
        for (; l < a.l_end; ++l) wino_layer_body(l);
    } else {
        int l = a.l_begin;
        if (l == 0 && l0_lookup) { lookup_layer0(); l = 1; }
        for (; l < a.l_end; ++l) direct_layer(l);
    }
    STAMP(62);
    copy_out(xs, yout, own_lo, own_hi, tid);
    STAMP(63);
    }
    if constexpr (PERSIST) {
        j += gridDim.x >> 3;
        wk = xcd * per + j;
        __syncthreads();                                    // the next row re-uses the LDS image
        if (j < per && wk < n_work) goto next_row;
    }
}

void launch_segment(const SegmentArgs& a0, int n_sites, int max_wgs, hipStream_t s) {
    SegmentArgs a = a0;
    a.n_rows = n_sites * a.R;
    a.xcd_rows = ((n_sites + 7) / 8) * a.R;                 // whole sites per XCD slice
    const bool split = a.units == 2;
    if (!split) { a.units = 1; a.Lw = a.L; a.y_out = a.y; }   // (callers that never heard of units: one unit, the whole window, in place)
    const int units = split ? 2 : 1;
    const bool persist = a.work_count != nullptr && max_wgs >= 8 && max_wgs < a.n_rows * units;
    const dim3 grid((unsigned)(persist ? (max_wgs & ~7) : 8 * a.xcd_rows * units)), blk(SEG_THREADS);
    if (split) {
        // The Winograd form of a split read always runs six tiles per lane: its layers have dilation 2, so a unit is at most
        // ceil(304 / 2) + 1 + 2 (MAX_LAYERS - 1) = 183 columns <= MPOS_SHORT.  (A longer unit -- not constructible today -- takes the
        // direct form, which tiles the whole 208-column image.)
        static_assert((304 + 1) / 2 + 1 + 2 * (MAX_LAYERS - 1) <= MPOS_SHORT, "a split Winograd unit fits the six-tile form");
        const bool wino = a.wino && a.L <= MPOS_SHORT;
        a.wino = wino;
        if (persist) {
            if (wino) hipLaunchKernelGGL((segment_kernel<true, true, true, MW_SHORT>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((segment_kernel<false, true, true>), grid, blk, 0, s, a);
        } else {
            if (wino) hipLaunchKernelGGL((segment_kernel<true, false, true, MW_SHORT>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((segment_kernel<false, false, true>), grid, blk, 0, s, a);
        }
    } else if (persist) {
        if (a.wino) hipLaunchKernelGGL((segment_kernel<true, true, false>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((segment_kernel<false, true, false>), grid, blk, 0, s, a);
    } else {
        if (a.wino) hipLaunchKernelGGL((segment_kernel<true, false, false>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((segment_kernel<false, false, false>), grid, blk, 0, s, a);
    }
}

// ------------------------------------------------------------------------------------------------
// site-level reductions over the read axis (sequential r = 0..R-1 fp32 sums, like AvgPool2d on CPU)
// ------------------------------------------------------------------------------------------------
// the empty-row map of one site (see dan_kernels.h): one workgroup per site, one wave per row in turn
__global__ __launch_bounds__(256) void row_map_kernel(const uint8_t* __restrict__ reads, const uint8_t* __restrict__ qual,
                                                      const uint8_t* __restrict__ strand, int* __restrict__ row_src, int R, int L) {
    __shared__ int empty[1024];
    const int site = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < R; r += 4) {
        const size_t base = ((size_t)site * R + r) * L;
        unsigned any = 0;
        for (int p = lane; p < L; p += 64) any |= (unsigned)reads[base + p] | (unsigned)qual[base + p] | (unsigned)strand[base + p];
        const bool none = __ballot(any != 0) == 0ull;
        if (lane == 0 && r < 1024) empty[r] = none;
    }
    __syncthreads();
    int first = -1;
    for (int r = 0; r < R && r < 1024; ++r)
        if (empty[r]) { first = r; break; }
    for (int r = tid; r < R; r += 256) row_src[(size_t)site * R + r] = site * R + ((r < 1024 && empty[r]) ? first : r);
}

// rows that are their own source, in row order: work[0 .. *count)
__global__ __launch_bounds__(1024) void work_list_kernel(const int* __restrict__ row_src, int* __restrict__ work,
                                                         int* __restrict__ count, int n_sites, int R) {
    __shared__ int offs[4096 + 1];
    const int tid = threadIdx.x;
    int total = 0;
    for (int base = 0; base < n_sites; base += 4096) {                      // (one pass unless a chunk exceeds 4096 sites)
        const int n = min(4096, n_sites - base);
        for (int sidx = tid; sidx < n; sidx += 1024) {
            int c = 0;
            const int* rs = row_src + (size_t)(base + sidx) * R;
            for (int r = 0; r < R; ++r) c += rs[r] == (base + sidx) * R + r;
            offs[sidx + 1] = c;
        }
        __syncthreads();
        if (tid == 0) {
            offs[0] = total;
            for (int i = 0; i < n; ++i) offs[i + 1] += offs[i];
        }
        __syncthreads();
        for (int sidx = tid; sidx < n; sidx += 1024) {
            int o = offs[sidx];
            const int* rs = row_src + (size_t)(base + sidx) * R;
            for (int r = 0; r < R; ++r)
                if (rs[r] == (base + sidx) * R + r) work[o++] = (base + sidx) * R + r;
        }
        total = offs[n];
        __syncthreads();
    }
    if (tid == 0) *count = total;
}

void launch_row_map(const uint8_t* reads, const uint8_t* qual, const uint8_t* strand, int* row_src, int* work, int* count,
                    int n_sites, int R, int L, hipStream_t s) {
    hipLaunchKernelGGL(row_map_kernel, dim3(n_sites), dim3(256), 0, s, reads, qual, strand, row_src, R, L);
    hipLaunchKernelGGL(work_list_kernel, dim3(1), dim3(1024), 0, s, row_src, work, count, n_sites, R);
}

__global__ __launch_bounds__(256) void read_mean_kernel(const v4f* __restrict__ y, v4f* __restrict__ pool, int R, int L,
                                                        const int* __restrict__ row_src) {
    const int n4 = L * (CPAD / 4);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int site = blockIdx.y;
    v4f sum = splat(0.f);
    if (row_src) {                                          // same rows, same order: skipped rows are read through the map
        const int* rs = row_src + (size_t)site * R;
#pragma unroll 8
        for (int r = 0; r < R; ++r) sum += y[(size_t)rs[r] * n4 + i];
    } else {
        const v4f* src = y + (size_t)site * R * n4 + i;
#pragma unroll 8
        for (int r = 0; r < R; ++r) sum += src[(size_t)r * n4];
    }
    pool[(size_t)site * n4 + i] = sum / splat((float)R);
}

void launch_read_mean(const float* y, float* pool, int n_sites, int R, int L, const int* row_src, hipStream_t s) {
    const int n4 = L * (CPAD / 4);
    hipLaunchKernelGGL(read_mean_kernel, dim3((n4 + 255) / 256, n_sites), dim3(256), 0, s, (const v4f*)y, (v4f*)pool, R, L,
                       row_src);
}

__global__ __launch_bounds__(256) void final_pool_kernel(const v4f* __restrict__ y, float* __restrict__ feat,
                                                         long long fs, int R, int L, int C, const int* __restrict__ row_src) {
    constexpr int PT = 32, PS = PT + 1;                        // 32 positions per workgroup: 128-byte runs in the feature row
    __shared__ float tmax[CPAD * PS], tavg[CPAD * PS];
    const int pt = blockIdx.x, site = blockIdx.y, tid = threadIdx.x;
    const int c4 = tid & 31, pl = tid >> 5;
    const int n4 = L * (CPAD / 4);
#pragma unroll
    for (int half = 0; half < PT / 8; ++half) {
        const int pp = half * 8 + pl, p = pt * PT + pp;
        v4f mx = splat(0.f), av = splat(0.f);
        if (p < L) {
            const size_t off = (size_t)p * (CPAD / 4) + c4;
            const int* rs = row_src ? row_src + (size_t)site * R : nullptr;
            auto row = [&](int r) { return rs ? y[(size_t)rs[r] * n4 + off] : y[((size_t)site * R + r) * n4 + off]; };
            mx = row(0);
            v4f sum = mx;
#pragma unroll 8
            for (int r = 1; r < R; ++r) {
                const v4f v = row(r);
#pragma unroll
                for (int j = 0; j < 4; ++j) mx[j] = fmaxf(mx[j], v[j]);
                sum += v;
            }
            av = sum / splat((float)R);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tmax[(c4 * 4 + j) * PS + pp] = mx[j];
            tavg[(c4 * 4 + j) * PS + pp] = av[j];
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

void launch_final_pool(const float* y, float* feat, long long fs, int n_sites, int R, int L, int C, const int* row_src,
                       hipStream_t s) {
    hipLaunchKernelGGL(final_pool_kernel, dim3((L + 31) / 32, n_sites), dim3(256), 0, s, (const v4f*)y, feat, fs, R, L, C,
                       row_src);
}

// ------------------------------------------------------------------------------------------------
// highway compression: per layer a GEMM  [reads of the chunk] x [L*32] x [32]   (model.py:776-777,859)
// Round 5 form (see highway16_kernel in dan_kernels_bf16p.hip for the measurements behind it): one workgroup = 128 reads, wave = 16
// reads x ALL positions (no cross-wave sum), so the 4 KiB of weights of a position are fetched once per workgroup -- wave w loads
// those of position 8 ph + w -- into a two-phase LDS ring that all eight waves read; HW_DH positions (two 16-byte loads each) of h in
// flight per wave.  Rounds 1-4: eight waves splitting the positions of 64 reads, each streaming its own weights from L2 -- as
// many bytes as half its h -- and summing partial tiles through LDS: 4.3 TB/s of h.  The sums of a read are formed position by
// position in index order (deterministic, independent of the chunking).
// ------------------------------------------------------------------------------------------------
constexpr int HW_DH = 16;                                   // positions of h in flight per wave (a multiple of 8)
__global__ __launch_bounds__(512) void highway_kernel(const float* __restrict__ h, long long hls,
                                                      const v4f* __restrict__ wc, long long wcls,
                                                      const float* __restrict__ bc, float* __restrict__ feat,
                                                      long long fs, int feat_off, int n_rows, int R, int L, int H,
                                                      const int* __restrict__ row_src) {
    __shared__ __attribute__((aligned(16))) char ring[2][8][4][1024];      // [phase parity][position of the phase][k-group 2 x n 2][lane * 16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int layer = blockIdx.y;
    const int row0 = blockIdx.x * 128 + wave * 16;
    const size_t K = (size_t)L * HPAD;
    const int row = min(row0 + r16, n_rows - 1);
    const float* arow = h + (size_t)layer * hls + (size_t)(row_src ? row_src[row] : row) * K + kk * 4;   // skipped rows: their source's h
    const v4f* wl = wc + (size_t)layer * wcls + lane;       // [k-group g = 2 p + (c >> 4)][n 2][lane 64]
    v4f acc0 = splat(0.f), acc1 = splat(0.f);
    const int n_ph = (L + 7) >> 3;
    v4f ar[HW_DH][2];
#pragma unroll
    for (int d = 0; d < HW_DH; ++d) {
        const float* src = arow + (size_t)min(d, L - 1) * HPAD;
        ar[d][0] = *(const v4f*)src; ar[d][1] = *(const v4f*)(src + 16);
    }
    v4f wq[4];
    {
        const int p = min(wave, L - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[j] = wl[((size_t)p * 4 + j) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) *(v4f*)(&ring[0][wave][j][lane * 16]) = wq[j];
    }
    __syncthreads();
    for (int ph0 = 0; ph0 < n_ph; ph0 += HW_DH / 8) {
#pragma unroll
        for (int q = 0; q < HW_DH / 8; ++q) {                    // (unrolled: the slot of ar a position lives in is a compile-time index)
            const int ph = ph0 + q;
            if (ph < n_ph) {                                     // wave-uniform and the same for every wave: all reach the barrier below
                if (ph + 1 < n_ph) {
                    const int pn = min(8 * (ph + 1) + wave, L - 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) wq[j] = wl[((size_t)pn * 4 + j) * 64];
                }
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const int p = 8 * ph + d;
                    const v4f a0 = ar[q * 8 + d][0], a1 = ar[q * 8 + d][1];
                    const float* src = arow + (size_t)min(p + HW_DH, L - 1) * HPAD;
                    ar[q * 8 + d][0] = *(const v4f*)src; ar[q * 8 + d][1] = *(const v4f*)(src + 16);
                    if (p < L) {
                        const char* wr = &ring[ph & 1][d][0][lane * 16];
                        const v4f w00 = *(const v4f*)wr, w01 = *(const v4f*)(wr + 1024), w10 = *(const v4f*)(wr + 2048), w11 = *(const v4f*)(wr + 3072);
#pragma unroll
                        for (int sI = 0; sI < 4; ++sI) { acc0 = mfma16(a0[sI], w00[sI], acc0); acc1 = mfma16(a0[sI], w01[sI], acc1); }
#pragma unroll
                        for (int sI = 0; sI < 4; ++sI) { acc0 = mfma16(a1[sI], w10[sI], acc0); acc1 = mfma16(a1[sI], w11[sI], acc1); }
                    }
                }
                if (ph + 1 < n_ph) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *(v4f*)(&ring[(ph + 1) & 1][wave][j][lane * 16]) = wq[j];
                }
                __syncthreads();
            }
        }
    }
    // C fragment: read 4 kk + jj of the wave's sixteen, output o = 16 j + r16
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int rr = row0 + 4 * kk + jj;
        if (rr < n_rows) {
            const int site = rr / R, r = rr - site * R;
            float* dst = feat + (size_t)site * fs + feat_off + (size_t)layer * H * R + r;
            if (r16 < H) dst[(size_t)r16 * R] = fmaxf(acc0[jj] + bc[layer * HPAD + r16], 0.f);
            if (16 + r16 < H) dst[(size_t)(16 + r16) * R] = fmaxf(acc1[jj] + bc[layer * HPAD + 16 + r16], 0.f);
        }
    }
}

void launch_highway(const float* h, long long hls, const float* wc, long long wcls, const float* bc, float* feat,
                    long long fs, int feat_off, int n_sites, int R, int L, int H, int layers, const int* row_src,
                    hipStream_t s) {
    const int n_rows = n_sites * R;
    hipLaunchKernelGGL(highway_kernel, dim3((n_rows + 127) / 128, layers), dim3(512), 0, s, h, hls, (const v4f*)wc,
                       wcls / 4, bc, feat, fs, feat_off, n_rows, R, L, H, row_src);
}

// ------------------------------------------------------------------------------------------------
// FC:  C[M][N] = act(A[M][K] * W[N][K]^T + bias)   -- both operands K-contiguous (torch Linear layout), model.py:362-377,917
// Workgroup tile 128 x 128, k-tiles of 32 staged through LDS in full 128-byte lines (8 lanes x 16 B per row; fragment-shaped
// loads straight from L2 -- 16 rows x 64 B per instruction -- made the 64 x 64 kernel of round 1 load-path bound at 67 TF),
// double-buffered: the next k-tile's global loads are issued before the MFMAs of the current one and written to the other
// LDS buffer after them, one barrier per k-tile.  8 waves = 2 (m) x 4 (n), each 64 x 32 (8 accumulator tiles); fragments
// are ds_read_b128 with the k order inside a 16-group permuted as everywhere else (one 16-byte read feeds four k-steps);
// row stride 40 dwords: the four 16-lane groups of a ds_read_b128 ({0-3,12-15,20-27}, ...) then touch rows of all 8
// residues mod 8 at two k offsets -- conflict-free.  XCD-aware tile order: workgroups b and b + 8 share an XCD, so tiles are
// dealt in contiguous runs per XCD (n fastest): an XCD's 32 tiles share 4 row blocks of A and all of W's 8 column blocks,
// A streams from HBM once and W once per XCD.
// ------------------------------------------------------------------------------------------------
// FC_S = 36: conflict-free b128 reads (rows land 4 banks apart, 16 rows cover all 64), and two stages of A + W tiles are 72 KB, so
// TWO workgroups fit a CU: with `ksplit` = 2 every tile's k range is halved between two workgroups that end up co-resident
// (256 tiles of FC1 = one per CU otherwise), their partial sums go to `ws` and fc_combine_kernel adds them in a fixed order.
constexpr int FC_BM = 128, FC_BN = 128, FC_BK = 32, FC_S = FC_BK + 4, FC_THREADS = 512;
__global__ __launch_bounds__(FC_THREADS, 2) void fc_kernel(const float* __restrict__ A, long long lda,
                                                           const float* __restrict__ W, long long ldw,
                                                           const float* __restrict__ bias, float* __restrict__ C, long long ldc,
                                                           int M, int N, int K, int relu, int tiles_n, int n_tiles, int ksplit, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) float sa[2][FC_BM * FC_S], sw[2][FC_BN * FC_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    // tile of this workgroup (XCD-contiguous order)
    const int n_work = n_tiles * ksplit;
    const int per = (n_work + 7) >> 3;
    const int qa = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || qa >= n_work) return;
    const int q = qa / ksplit, part = qa - q * ksplit;          // the parts of one tile are neighbours: same XCD, same L2 lines of C
    const int bm = (q / tiles_n) * FC_BM, bn = (q % tiles_n) * FC_BN;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32;
    // staging: thread -> rows (tid >> 3) + 64 i, 16-byte column tid & 7
    const int srow = tid >> 3, sc4 = (tid & 7) * 4;
    const float *ga[2], *gw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ga[i] = A + (size_t)min(bm + srow + 64 * i, M - 1) * lda + sc4;
        gw[i] = W + (size_t)min(bn + srow + 64 * i, N - 1) * ldw + sc4;
    }
    v4f ra[2], rw[2];
    auto gload = [&](int k0) {
        const bool in = k0 + sc4 < K;                           // K is a multiple of 4 (16, in fact): whole vectors
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = in ? *(const v4f*)(ga[i] + k0) : splat(0.f);
            rw[i] = in ? *(const v4f*)(gw[i] + k0) : splat(0.f);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(v4f*)(&sa[buf][(srow + 64 * i) * FC_S + sc4]) = ra[i];
            *(v4f*)(&sw[buf][(srow + 64 * i) * FC_S + sc4]) = rw[i];
        }
    };
    v4f acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = splat(0.f);
    const int KT_all = (K + FC_BK - 1) / FC_BK;
    const int kts = (KT_all + ksplit - 1) / ksplit;
    const int kt_lo = part * kts, KT = min(KT_all, kt_lo + kts);
    gload(kt_lo * FC_BK);
    sstore(kt_lo & 1);
    __syncthreads();
    for (int kt = kt_lo; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) gload((kt + 1) * FC_BK);
        const float* pa = &sa[cur][(wm + r16) * FC_S + kk * 4];
        const float* pw = &sw[cur][(wn + r16) * FC_S + kk * 4];
#pragma unroll
        for (int g = 0; g < FC_BK / 16; ++g) {
            v4f a[4], w[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *(const v4f*)(pa + i * 16 * FC_S + g * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) w[j] = *(const v4f*)(pw + j * 16 * FC_S + g * 16);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[i][s], w[j][s], acc[i][j]);
        }
        if (kt + 1 < KT) sstore(cur ^ 1);
        __syncthreads();
    }
    if (ksplit > 1) {                                            // raw partial sums; bias / ReLU in fc_combine_kernel
        float* wp = ws + (size_t)part * M * N;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = bn + wn + j * 16 + r16;
            if (n >= N) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int m = bm + wm + i * 16 + kk * 4 + jj;
                    if (m < M) wp[(size_t)m * N + n] = acc[i][j][jj];
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = bn + wn + j * 16 + r16;
        if (n >= N) continue;
        const float b = bias[n];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = bm + wm + i * 16 + kk * 4 + jj;
                if (m < M) {
                    float v = acc[i][j][jj] + b;
                    if (relu) v = fmaxf(v, 0.f);
                    C[(size_t)m * ldc + n] = v;
                }
            }
    }
}

__global__ __launch_bounds__(256) void fc_combine_kernel(const float* __restrict__ ws, int parts, const float* __restrict__ bias,
                                                         float* __restrict__ C, long long ldc, int M, int N, int relu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * N) return;
    const int m = (int)(i / N), n = (int)(i - (long long)m * N);
    float v = ws[i];
    for (int p = 1; p < parts; ++p) v += ws[(size_t)p * M * N + i];
    v += bias[n];
    if (relu) v = fmaxf(v, 0.f);
    C[(size_t)m * ldc + n] = v;
}

void launch_fc(const float* A, long long lda, const float* W, long long ldw, const float* bias, float* C,
               long long ldc, int M, int N, int K, int relu, hipStream_t s, float* ws, long long ws_floats) {
    const int tiles_m = (M + FC_BM - 1) / FC_BM, tiles_n = (N + FC_BN - 1) / FC_BN, n_tiles = tiles_m * tiles_n;
    // split k in two when there is a workspace, the k range is long and the tiles alone leave CU slots empty
    const int ksplit = (ws && ws_floats >= 2LL * M * N && K >= 64 * FC_BK && n_tiles <= 384) ? 2 : 1;
    const int grid = ((n_tiles * ksplit + 7) / 8) * 8;
    hipLaunchKernelGGL(fc_kernel, dim3(grid), dim3(FC_THREADS), 0, s, A, lda, W, ldw, bias, C, ldc, M, N, K, relu, tiles_n, n_tiles, ksplit, ws);
    if (ksplit > 1) launch_fc_combine(ws, ksplit, bias, C, ldc, M, N, relu, s);
}

void launch_fc_combine(const float* ws, int parts, const float* bias, float* C, long long ldc, int M, int N, int relu, hipStream_t s) {
    const long long total = (long long)M * N;
    hipLaunchKernelGGL(fc_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws, parts, bias, C, ldc, M, N, relu);
}

// ------------------------------------------------------------------------------------------------
// heads + score post-processing: one 64-lane block per site
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void heads_kernel(const float* __restrict__ hidden, int hid,
                                                   const float* __restrict__ wh, const float* __restrict__ bh,
                                                   float* bin_logits, float* vt_logits, float* vt_prob, float* bp,
                                                   float* aux) {
    __shared__ float o[NHEAD];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < NHEAD) {
        const float* x = hidden + (size_t)b * hid;
        const float* w = wh + (size_t)t * hid;
        float acc = 0.f;
        for (int k = 0; k < hid; ++k) acc = fmaf(x[k], w[k], acc);
        o[t] = acc + bh[t];
    }
    __syncthreads();
    if (t == 0) {
        if (bin_logits) { bin_logits[b * 2] = o[0]; bin_logits[b * 2 + 1] = o[1]; }
        if (vt_logits) { vt_logits[b * 3] = o[2]; vt_logits[b * 3 + 1] = o[3]; vt_logits[b * 3 + 2] = o[4]; }
        if (bp) {                                            // 1 - softmax(bin)[0]   trainer.py:620-621
            const float m = fmaxf(o[0], o[1]);
            const float e0 = expf(o[0] - m), e1 = expf(o[1] - m);
            bp[b] = 1.f - e0 / (e0 + e1);
        }
        if (vt_prob) {                                       // softmax(vt) = (NV, HV, OV)   trainer.py:623
            const float m = fmaxf(o[2], fmaxf(o[3], o[4]));
            const float e0 = expf(o[2] - m), e1 = expf(o[3] - m), e2 = expf(o[4] - m);
            const float s = e0 + e1 + e2;
            vt_prob[b * 3] = e0 / s; vt_prob[b * 3 + 1] = e1 / s; vt_prob[b * 3 + 2] = e2 / s;
        }
    }
    if (aux && t < 22) {
        float v = o[5 + t];
        if (t == 0) v = 1.f / (1.f + expf(-v));             // sigmoid(AF)        model.py:954
        else if (t == 1) v = v > 0.f ? v : 0.01f * v;       // leaky_relu(cov)    model.py:956
        aux[(size_t)b * 22 + t] = v;
    }
}

void launch_heads(const float* hidden, int hid, const float* wh, const float* bh, int B, float* bin_logits,
                  float* vt_logits, float* vt_prob, float* bp, float* aux, hipStream_t s) {
    hipLaunchKernelGGL(heads_kernel, dim3(B), dim3(64), 0, s, hidden, hid, wh, bh, bin_logits, vt_logits, vt_prob, bp, aux);
}

}  // namespace dan
