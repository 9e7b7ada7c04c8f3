// Plain-bf16 segment kernel, "two workgroups per CU" form (dan_config.precision = 2, every row computed).
//
// The eight-wave bf16 kernel (dan_kernels_bf16.hip) keeps the matrix pipe busy a quarter of the time at 128 x 301: one read per
// CU means its conv GEMM (7.3 k cycles of MFMA issue per SIMD and layer) alternates with stages that issue next to no MFMA --
// the in-place LDS write-back (4.9 k: LDS stores move ~80 B/clk), the residual layers' swap / second write-back, barriers, the
// resumed-segment prologue -- and nothing else is resident to fill the pipe.  Here a workgroup is FOUR waves (one per SIMD) and
// the LDS image is 256-byte rows with no padding (312 rows = 78 KB at 19 tiles; the per-layer constants come from L2 into
// registers, the 35-KB LDS constant block is gone), so TWO workgroups -- two reads -- are resident per CU and one read's
// write-back / swap / prologue runs under the other's GEMM.
//
// Registers: a wave owns 32 output channels x all position tiles, which at 19 tiles would be 152 accumulators -- measured: it does
// not allocate (hundreds of spills).  So a layer runs in TWO CHANNEL PASSES per wave (16 channels x all tiles = 76 accumulators
// each): pass 0's outputs wait, already activated and rounded to bf16 (2 registers per tile -- exactly the bytes the image will
// hold, so nothing changes numerically), while pass 1 runs; after the barrier both are written.  A residual layer swaps those
// packed registers with the image cells (the old x comes back packed), runs the 1x1 GEMM in two passes from them and writes
// the result.  Outputs are bit-identical to the eight-wave kernel's.
//
// LDS image: row r (position + HALO) = 128 bf16 = 16 chunks of 16 B; chunk c is stored at chunk (c ^ (r & 15)).  A B fragment
// read (16 consecutive rows x the four 16-byte chunks of one 32-channel k-group) then touches every 16-byte bank slot once per
// 16-lane group for even tap shifts (the dilation-2 layers; odd shifts cost a 2-way conflict on one lane pair).  The row's low
// four bits do not change with the position tile (16 rows apart), so a lane's swizzle is one XOR per (tap, k-group).
//
// Weight blocks, HBM formats and the work list are those of dan_kernels_bf16.hip: wave w of four uses channel tiles 2w, 2w + 1.
//
// Measured (MI355X): 128 x 301: 15.8 k -> 16.5 k sites/s (+4 %), 64 x 201: 43.9 k -> 46.5 k (+6 %), outputs bit-identical.  Far
// from the 2x a free overlap would give: a four-wave workgroup takes 1.9x as long per read as the eight-wave one -- every stage,
// not only the GEMM, is bound by what ONE wave per SIMD can issue (twice the stores, converts and fragment reads per wave), so
// two reads per CU buy little; starting the second workgroup of a CU half a read late changed nothing (so it is not lock step).
#include "dan_kernels.h"

namespace dan {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) bf8* gbf8_ptr;

constexpr int W_THREADS = 256;
constexpr int W_WAVES = 4;
constexpr int W_NT = 2;                  // MFMA row tiles (16 channels each) per wave, one per channel pass
constexpr int W_ROWB = 256;              // bytes per LDS row

__device__ __forceinline__ v4f mfma_w(bf8 a, bf8 b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4f splat_w(float x) { return (v4f){x, x, x, x}; }
__device__ __forceinline__ float relu_w(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()); }

// byte offset of the 8-byte cell holding channels [ch, ch + 4) of image row `row`
__device__ __forceinline__ int cell_w(int row, int ch) { return row * W_ROWB + ((((ch >> 3) ^ (row & 15))) << 4) + ((ch & 4) << 1); }

__device__ __forceinline__ void store_cell_w(unsigned char* xs, int off, v4f v) {
    bf4 hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) hi[j] = (__bf16)v[j];
    *(bf4*)(xs + off) = hi;
}
__device__ __forceinline__ v4f load_cell_w(const unsigned char* xs, int off) {
    const bf4 hi = *(const bf4*)(xs + off);
    v4f v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)hi[j];
    return v;
}

// implicit GEMM of ONE 16-channel tile over taps x 32-channel k-groups; B fragments in a ring of 8 registers (dan_kernels_bf16.hip)
template <int MT>
__device__ __forceinline__ void gemm_w(v4f (&acc)[MT], const unsigned char* xs, gbf8_ptr wl, int kg, int ntaps, int dil, int lane) {
    asm volatile("" : "+v"(lane));                               // (no fragment address is formed ahead of the layer loop)
    const int pos = lane & 15, kq = lane >> 4;
    const int total = ntaps * kg;
    const int t0 = (ntaps == 3) ? -dil : 0;
    constexpr int RING = MT < 8 ? MT : 8;
    auto frag = [&](int shift, int g) {                          // byte offset of tile 0's fragment for this lane
        const int r0 = HALO + pos + shift;
        return r0 * W_ROWB + ((((g << 2) | kq) ^ (r0 & 15)) << 4);
    };
    bf8 a_nxt = wl[0], b[RING];
    int xc = frag(t0, 0);
#pragma unroll
    for (int m = 0; m < RING; ++m) b[m] = *(const bf8*)(xs + xc + m * 16 * W_ROWB);
    int t = 0, g = 0;
    for (int it = 0; it < total; ++it) {
        const bf8 a = a_nxt;
        const int nx = (it + 1 < total) ? it + 1 : it;
        a_nxt = wl[(size_t)nx * (KGC * 64)];
        int tn = t, gn = g + 1;
        if (gn == kg) { gn = 0; ++tn; }
        if (it + 1 == total) { tn = t; gn = g; }
        const int xn = frag(t0 + tn * dil, gn);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int slot = m % RING;
            acc[m] = mfma_w(a, b[slot], acc[m]);
            const int src = (m + RING < MT) ? xc + (m + RING) * 16 * W_ROWB : xn + slot * 16 * W_ROWB;
            b[slot] = *(const bf8*)(xs + src);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        t = tn; g = gn;
        xc = xn;
    }
}

// 128 -> 32 highway bottleneck: wave w takes position tiles w, w + 4, ... and both 16-channel output tiles
template <int MT>
__device__ __forceinline__ void bottleneck_w(const unsigned char* xs, gbf8_ptr wb, const float* bbot, float* hrow, int L, int wave, int lane) {
    constexpr int NB = (MT + W_WAVES - 1) / W_WAVES;
    constexpr int KG = CPAD / 32;
    asm volatile("" : "+v"(lane));
    const int pos = lane & 15, kq = lane >> 4;
    bf8 ah[KG][2];
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int n = 0; n < 2; ++n) ah[g][n] = wb[(g * 2 + n) * 64];
    bf8 bh[KG][NB];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        const int r0 = HALO + pos;
        const int base = r0 * W_ROWB + ((((g << 2) | kq) ^ (r0 & 15)) << 4);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int pt = min(wave + W_WAVES * i, MT - 1);
            bh[g][i] = *(const bf8*)(xs + base + pt * 16 * W_ROWB);
        }
    }
    v4f acc[NB][2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const v4f bb = *(const v4f*)(bbot + n * 16 + kq * 4);
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[i][n] = bb;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[i][n] = mfma_w(ah[g][n], bh[g][i], acc[i][n]);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int po = pos;
        asm volatile("" : "+v"(po));
        const int pt = wave + W_WAVES * i, p = pt * 16 + po;
        if (pt < MT && p < L) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                v4f v = acc[i][n];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = relu_w(v[j]);
                *(v4f*)(hrow + (size_t)p * HPAD + n * 16 + kq * 4) = v;
            }
        }
    }
}

__device__ __forceinline__ void copy_out_w(const unsigned char* xs, float* dst, int L, int tid) {
    asm volatile("" : "+v"(tid));
    for (int i = tid; i < L * (CPAD / 4); i += W_THREADS) {
        const int p = i >> 5, c4 = i & 31;
        ((v4f*)dst)[i] = load_cell_w(xs, cell_w(HALO + p, c4 * 4));
    }
}

template <int MT>
__global__ __launch_bounds__(W_THREADS, 2) void segment16w_kernel(Segment16Args a) {
    constexpr int MPOS_ = MT * 16, ROWS = MPOS_ + 2 * HALO;
    __shared__ __attribute__((aligned(16))) unsigned char xs[ROWS * W_ROWB];
    const int wk = blockIdx.x;
    if (a.work_count && wk >= *a.work_count) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row_index = __builtin_amdgcn_readfirstlane(a.work_count ? a.work[wk] : wk);
    const int site = row_index / a.R, r = row_index - site * a.R;
    const int L = a.L;
    const size_t read_idx = (size_t)site * a.R + r;
    float* yrow = a.y + read_idx * (size_t)L * CPAD;
    const int pos = lane & 15, kq = lane >> 4;
    int chb[W_NT];
#pragma unroll
    for (int n = 0; n < W_NT; ++n) chb[n] = (wave * W_NT + n) * 16 + kq * 4;
    auto block = [&](int l) { return a.wl + (size_t)l * W16_LAYER_BYTES; };
    auto consts = [&](int l) { return (const float*)(block(l) + W16_CST_OFF); };

    if (a.l_begin == 0) {
        for (int i = tid; i < ROWS * W_ROWB / 16; i += W_THREADS) ((v4f*)xs)[i] = splat_w(0.f);
        __syncthreads();
        // ---- encode (dl4vc/model.py:450-627), canonical 48-channel order, rounded to bf16
        const size_t rbase = read_idx * (size_t)L, sbase = (size_t)site * L;
        int ok_ref = 1, ok_var = 1;
        for (int p = tid; p < L; p += W_THREADS) {
            const int tok = a.reads[rbase + p], rm = a.ref_mask[sbase + p], vm = a.var_mask[sbase + p];
            ok_ref &= (rm == 0) || (tok == rm);
            ok_var &= (vm == 0) || (tok == vm);
        }
        const int agree_ref = __syncthreads_and(ok_ref);
        const int agree_var = __syncthreads_and(ok_var);
        for (int p = tid; p < L; p += W_THREADS) {
            const int tok = a.reads[rbase + p], q = a.qual[rbase + p], st = a.strand[rbase + p];
            const int rf = a.ref[sbase + p], rm = a.ref_mask[sbase + p], vm = a.var_mask[sbase + p];
            const float* er = a.emb + min(tok, VOCAB - 1) * EMBED;
            const float* ef = a.emb + min(rf, VOCAB - 1) * EMBED;
            const float* pp = a.pe + p * EMBED;
            float row[CIN0];
#pragma unroll
            for (int e = 0; e < EMBED; ++e) { const float pv = pp[e]; row[e] = er[e] + pv; row[EMBED + e] = ef[e] + pv; }
            row[40] = (float)q * 0.01f;
            row[41] = (float)st * 0.5f;
            row[42] = (rm != 0 && agree_ref) ? 1.f : 0.f;
            row[43] = (vm != 0 && agree_var) ? 1.f : 0.f;
            row[44] = (rm != 0) ? 1.f : 0.f;
            row[45] = row[46] = row[47] = 0.f;
#pragma unroll
            for (int c = 0; c < CIN0; c += 4) store_cell_w(xs, cell_w(HALO + p, c), (v4f){row[c], row[c + 1], row[c + 2], row[c + 3]});
        }
    } else {
        const v4f* src = (const v4f*)yrow;
        const v4f* pl = a.pool ? (const v4f*)(a.pool + (size_t)site * L * CPAD) : nullptr;
        const int n4 = L * (CPAD / 4);
        // rows the read does not cover (halo rows, rows >= L) are zero
        constexpr int RV = W_ROWB / 16;
        for (int i = tid; i < (ROWS - L) * RV; i += W_THREADS) {
            const int rr = i / RV, c = i - rr * RV;
            const int row = rr < HALO ? rr : rr + L;
            *(v4f*)(xs + row * W_ROWB + c * 16) = splat_w(0.f);
        }
        // the read and its site's pool image, half of the window at a time (all loads of a half in flight together)
        constexpr int NPF = (MPOS_ * (CPAD / 4) + W_THREADS - 1) / W_THREADS;
        constexpr int NH = (NPF + 1) / 2;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            v4f vy[NH], vp[NH];
#pragma unroll
            for (int k = 0; k < NH; ++k) {
                const int i = tid + (half * NH + k) * W_THREADS;
                vy[k] = (i < n4) ? src[i] : splat_w(0.f);
            }
#pragma unroll
            for (int k = 0; k < NH; ++k) {
                const int i = tid + (half * NH + k) * W_THREADS;
                vp[k] = (pl && i < n4) ? pl[i] : splat_w(0.f);
            }
#pragma unroll
            for (int k = 0; k < NH; ++k) {
                const int i = tid + (half * NH + k) * W_THREADS;
                if (i < n4) store_cell_w(xs, cell_w(HALO + (i >> 5), (i & 31) * 4), vy[k] + vp[k]);
            }
        }
    }
    __syncthreads();
    if (a.tap && a.tap_layer == 0 && a.l_begin == 0) copy_out_w(xs, a.tap + read_idx * (size_t)L * CPAD, L, tid);

    for (int l = a.l_begin; l < a.l_end; ++l) {
        const float* lc = consts(l);
        const bool residual = (a.res_mask >> l) & 1u;
        const int kg = (l == 0) ? KG16_0 : KG16_C;
        const int dil = (l == 0) ? 1 : (l + 1 < a.n_layers ? a.dil_mid : a.dil_final);
        // The lane's position goes through an opaque copy per layer: otherwise the row masks and cell addresses of every tile
        // are formed ahead of the layer loop and live through every GEMM.
        int pos_l = pos;
        asm volatile("" : "+v"(pos_l));
        const int cellb[W_NT] = {cell_w(HALO + pos_l, chb[0]), cell_w(HALO + pos_l, chb[1])};
        auto pack = [&](int m, v4f v) {                          // rounded to bf16 as the image holds it; rows >= L stay zero
            const float keep = (m * 16 + pos_l) < L ? 1.f : 0.f;
            bf4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (__bf16)(v[j] * keep);
            return o;
        };
        bf4 held[W_NT][MT];                                      // the layer's outputs between their pass and the write-back
#pragma unroll
        for (int p = 0; p < W_NT; ++p) {
            v4f acc[MT];
            {
                const v4f bias = *(const v4f*)(lc + CST_BIAS + chb[p]);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = bias;
            }
            gemm_w<MT>(acc, xs, (gbf8_ptr)(block(l) + W16_CONV_OFF) + (wave * W_NT + p) * 64 + lane, kg, 3, dil, lane);
            const v4f sc = *(const v4f*)(lc + CST_SCALE + chb[p]), sh = *(const v4f*)(lc + CST_SHIFT + chb[p]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                v4f v = acc[m];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = relu_w(v[j]) * sc[j] + sh[j];
                held[p][m] = pack(m, v);
            }
        }
        __syncthreads();                                         // every wave has finished reading the layer's input
        if (residual) {
            const bool from_global = (l == a.l_begin) && (a.l_begin != 0) && (a.pool != nullptr);
            // swap: the image takes n (this layer's conv output), the registers take the old x
#pragma unroll
            for (int p = 0; p < W_NT; ++p)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    bf4* cell = (bf4*)(xs + cellb[p] + m * 16 * W_ROWB);
                    const bf4 old = *cell;
                    *cell = held[p][m];
                    held[p][m] = old;
                }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < W_NT; ++p) {
                v4f acc[MT];
                const v4f bres = *(const v4f*)(lc + CST_BRES + chb[p]);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    v4f old;
#pragma unroll
                    for (int j = 0; j < 4; ++j) old[j] = (float)held[p][m][j];
                    if (from_global) {                           // the segment's input before the pool image was added to it
                        const int pg = m * 16 + pos_l;
                        old = (pg < L) ? *(const v4f*)(yrow + (size_t)pg * CPAD + chb[p]) : splat_w(0.f);
                    }
                    acc[m] = old + bres;
                }
                gemm_w<MT>(acc, xs, (gbf8_ptr)(block(l) + W16_RES_OFF) + (wave * W_NT + p) * 64 + lane, KG16_C, 1, 0, lane);
#pragma unroll
                for (int m = 0; m < MT; ++m) held[p][m] = pack(m, acc[m]);
            }
            __syncthreads();                                     // every wave has finished reading n
        }
#pragma unroll
        for (int p = 0; p < W_NT; ++p)
#pragma unroll
            for (int m = 0; m < MT; ++m) *(bf4*)(xs + cellb[p] + m * 16 * W_ROWB) = held[p][m];
        __syncthreads();
        if (a.tap && a.tap_layer == l + 1) copy_out_w(xs, a.tap + read_idx * (size_t)L * CPAD, L, tid);
        asm volatile("" ::: "memory");                           // (the bottleneck's weight loads stay behind the write-back)
        if (a.has_hw)
            bottleneck_w<MT>(xs, (gbf8_ptr)(block(l) + W16_BOT_OFF) + lane, lc + CST_BBOT,
                             a.h + (size_t)l * a.h_layer_stride + read_idx * (size_t)L * HPAD, L, wave, lane);
    }
    copy_out_w(xs, yrow, L, tid);
}

void launch_segment16w(const Segment16Args& a0, int n_sites, hipStream_t s) {
    Segment16Args a = a0;
    a.n_rows = n_sites * a.R;
    const dim3 grid((unsigned)a.n_rows), blk(W_THREADS);
    if (a.L <= 13 * 16) hipLaunchKernelGGL((segment16w_kernel<13>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((segment16w_kernel<19>), grid, blk, 0, s, a);
}

}  // namespace dan
