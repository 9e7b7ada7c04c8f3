// bf16-MFMA family of the conv-stack segment kernel for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
//   SPLIT = true  ("bf16x3", dan_config.precision = 1): every fp32 operand is carried as hi + lo bf16 and a
//       product is three MFMAs  wh*xh + wl*xh + wh*xl  -- ~16 mantissa bits per operand, fp32 sums; softmax
//       scores stay within 1e-4 of the fp32 reference (tests) at ~1/3 of the bf16 matrix rate, i.e. ~5x the
//       fp32-MFMA rate.
//   SPLIT = false ("bf16", precision = 2): plain bf16 operands; BASELINE config 5 (128 reads x 301 bp): the
//       read still fits one CU's LDS (312 rows x 288 B), MT = 19 position tiles.
//
// Same structure as the fp32 kernel (dan_kernels.hip): one workgroup = one read resident in LDS for a whole
// segment of layers, each wave owns 32 output channels x all position tiles, weights stream from L2 in MFMA
// A-fragment order, B fragments are single-buffered ds_read_b128 (8 consecutive channels of one position)
// re-filled right after their last use.  HBM formats are unchanged (fp32 y / pool / h), so the pooling,
// highway and FC kernels are shared with the fp32 path.
#include "dan_kernels.h"
#include <cstdlib>

namespace dan {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) bf8* gbf8_ptr;

#ifdef DAN_STAMPS
extern __device__ unsigned long long* g_stamps;
#define STAMP16(k)                                                                                    \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                                         \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                           \
        if (lane == 0 && (k) < 64) g_stamps[((size_t)stamp_row * NWAVE + wave) * 64 + (k)] = t_;     \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#else
#define STAMP16(k) do {} while (0)
#endif

__device__ __forceinline__ v4f mfma_bf16(bf8 a, bf8 b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4f splat4(float x) { return (v4f){x, x, x, x}; }
// one compiler-visible instruction (an inline-asm v_max reading an MFMA result gets no hazard wait states)
__device__ __forceinline__ float relu1f(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()); }

template <bool SPLIT, int MT>
struct Geo {
    static constexpr int MPOS_ = MT * 16;
    static constexpr int ROWS = MPOS_ + 2 * HALO;
    static constexpr int PLANE = ROWS * S16;                 // bf16 elements per plane
    static constexpr int PLANES = SPLIT ? 2 : 1;
    static constexpr int NP = SPLIT ? 2 : 1;                 // fragments per operand (hi [, lo])
};

// fp32 quad -> bf16 hi (+ lo) stored at one LDS cell (4 consecutive channels of one position)
template <bool SPLIT, int PLANE>
__device__ __forceinline__ void store_cell(__bf16* cell, v4f v) {
    bf4 hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) hi[j] = (__bf16)v[j];
    *(bf4*)cell = hi;
    if (SPLIT) {
        bf4 lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) lo[j] = (__bf16)(v[j] - (float)hi[j]);
        *(bf4*)(cell + PLANE) = lo;
    }
}

template <bool SPLIT, int PLANE>
__device__ __forceinline__ v4f load_cell(const __bf16* cell) {
    const bf4 hi = *(const bf4*)cell;
    v4f v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)hi[j];
    if (SPLIT) {
        const bf4 lo = *(const bf4*)(cell + PLANE);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += (float)lo[j];
    }
    return v;
}

// implicit GEMM over taps x 32-channel k-groups.  wl points at this wave's first hi fragment; the lo plane of the
// same block lies lo_off fragments further.  Fragment index of (step, tile n): step*(tiles*64) + n*64.
template <bool SPLIT, int MT, int TILES>
__device__ __forceinline__ void gemm16(v4f (&acc)[MT][NT16], const __bf16* xs, gbf8_ptr wl, int lo_off,
                                       const bf8 (&a_first)[NT16][2], int kg, int ntaps, int dil, int lane) {
    typedef Geo<SPLIT, MT> G;
    const int pos = lane & 15, kq = lane >> 4;
    const int total = ntaps * kg;
    const int t0 = (ntaps == 3) ? -dil : 0;
    const __bf16* xrow = xs + (HALO + pos) * S16 + kq * 8;
    // B fragments live in a ring of RING registers, not one per position tile: tile m sits in slot m % RING and is
    // replaced, right after its MFMAs, by tile m + RING of the same k-step or -- for the last RING tiles -- by tile
    // m % RING of the NEXT k-step (any RING consecutive m cover every slot, so all indices stay compile-time).  At 19
    // position tiles that is 32 instead of 76 registers (the fragments were what spilled), with 8 MFMAs of load lead.
    constexpr int RING = MT < 8 ? MT : 8;
    bf8 a_nxt[NT16][2], bh[RING], bl[SPLIT ? RING : 1];
#pragma unroll
    for (int n = 0; n < NT16; ++n) { a_nxt[n][0] = a_first[n][0]; a_nxt[n][1] = a_first[n][1]; }
    const __bf16* xc = xrow + t0 * S16;
#pragma unroll
    for (int m = 0; m < RING; ++m) {
        bh[m] = *(const bf8*)(xc + m * 16 * S16);
        if (SPLIT) bl[m] = *(const bf8*)(xc + m * 16 * S16 + G::PLANE);
    }
    int t = 0, g = 0;
    for (int it = 0; it < total; ++it) {
        bf8 a[NT16][2];
#pragma unroll
        for (int n = 0; n < NT16; ++n) { a[n][0] = a_nxt[n][0]; a[n][1] = a_nxt[n][1]; }
        const int nx = (it + 1 < total) ? it + 1 : it;
#pragma unroll
        for (int n = 0; n < NT16; ++n) {
            a_nxt[n][0] = wl[(size_t)nx * (TILES * 64) + n * 64];
            if (SPLIT) a_nxt[n][1] = wl[(size_t)nx * (TILES * 64) + n * 64 + lo_off];
        }
        int tn = t, gn = g + 1;
        if (gn == kg) { gn = 0; ++tn; }
        if (it + 1 == total) { tn = t; gn = g; }
        const __bf16* xn = xrow + (t0 + tn * dil) * S16 + gn * 32;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int slot = m % RING;
#pragma unroll
            for (int n = 0; n < NT16; ++n) acc[m][n] = mfma_bf16(a[n][0], bh[slot], acc[m][n]);
            if (SPLIT) {
#pragma unroll
                for (int n = 0; n < NT16; ++n) acc[m][n] = mfma_bf16(a[n][1], bh[slot], acc[m][n]);
#pragma unroll
                for (int n = 0; n < NT16; ++n) acc[m][n] = mfma_bf16(a[n][0], bl[slot], acc[m][n]);
            }
            const __bf16* src = (m + RING < MT) ? xc + (m + RING) * 16 * S16 : xn + slot * 16 * S16;
            bh[slot] = *(const bf8*)src;
            if (SPLIT) bl[slot] = *(const bf8*)(src + G::PLANE);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, NT16 * G::NP, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, NT16 * (SPLIT ? 3 : 1), 0);
            __builtin_amdgcn_sched_group_barrier(0x100, G::NP, 0);
        }
        t = tn; g = gn;
        xc = xn;
    }
}

// 128 -> 32 highway bottleneck.  Unit u = 2*pt + n is owned by wave u % NWAVE: every wave has a fixed channel
// tile n = wave & 1 and the position tiles pt = (wave >> 1) mod (NWAVE/2).
// (its weight fragments are requested by the caller right after the conv GEMM, a barrier and the write-back ahead of their use:
// loaded here they cost the stage one L2 latency)
template <bool SPLIT>
__device__ __forceinline__ void bottleneck16_weights(bf8 (&ah)[CPAD / 32], bf8 (&al)[SPLIT ? CPAD / 32 : 1], gbf8_ptr wb, int lo_off, int wave) {
    const int n = wave & 1;
#pragma unroll
    for (int g = 0; g < CPAD / 32; ++g) {
        ah[g] = wb[(g * 2 + n) * 64];
        if (SPLIT) al[g] = wb[(g * 2 + n) * 64 + lo_off];
    }
}

// NWV = 8: a stage of its own, all waves.  NWV = 4: the DEFERRED form -- the four older waves of the workgroup (the arbiter serves
// them first, so they reach the barrier behind the conv GEMM ~3 k cycles before the younger four) run the PREVIOUS layer's
// bottleneck inside this layer's stage, from the same LDS image the conv GEMM just read, in that wait.
template <bool SPLIT, int MT, int NWV>
__device__ __forceinline__ void bottleneck16(const __bf16* xs, const bf8 (&ah)[CPAD / 32], const bf8 (&al)[SPLIT ? CPAD / 32 : 1],
                                             const float* bbot, float* hrow, int L, int wave, int lane) {
    typedef Geo<SPLIT, MT> G;
    constexpr int PS = NWV / 2;
    constexpr int NB = (MT + PS - 1) / PS;
    constexpr int KG = CPAD / 32;
    const int pos = lane & 15, kq = lane >> 4;
    const int n = wave & 1, p0 = wave >> 1;
    const __bf16* xrow = xs + (HALO + pos) * S16 + kq * 8;
    v4f acc[NB];
    {
        const v4f b = *(const v4f*)(bbot + n * 16 + kq * 4);
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[i] = b;
    }
    // B fragments are requested in batches ahead of their MFMAs (as load -> MFMA pairs the stage was one LDS latency per MFMA:
    // 3.5 k cycles for 0.3 k of MFMA).  As a stage of its own the conv accumulators are dead and the whole stage is one batch;
    // the deferred form runs beside live conv accumulators: one k-group per batch.
    constexpr int GB = (NWV == NWAVE) ? KG : 1;                  // k-groups per batch
#pragma unroll
    for (int g0 = 0; g0 < KG; g0 += GB) {
        bf8 bh[GB][NB], bl[SPLIT ? GB : 1][SPLIT ? NB : 1];
#pragma unroll
        for (int g = 0; g < GB; ++g)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int pt = min(p0 + PS * i, MT - 1);
                bh[g][i] = *(const bf8*)(xrow + pt * 16 * S16 + (g0 + g) * 32);
                if (SPLIT) bl[g][i] = *(const bf8*)(xrow + pt * 16 * S16 + (g0 + g) * 32 + G::PLANE);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < GB; ++g) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                acc[i] = mfma_bf16(ah[g0 + g], bh[g][i], acc[i]);
                if (SPLIT) {
                    acc[i] = mfma_bf16(al[g0 + g], bh[g][i], acc[i]);
                    acc[i] = mfma_bf16(ah[g0 + g], bl[g][i], acc[i]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int po = pos;
        asm volatile("" : "+v"(po));                             // (keeps the NB store addresses from living through the layer loop)
        const int pt = p0 + PS * i, p = pt * 16 + po;
        if (pt < MT && p < L) {
            v4f v = acc[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = relu1f(v[j]);
            *(v4f*)(hrow + (size_t)p * HPAD + n * 16 + kq * 4) = v;
        }
    }
}

template <bool SPLIT, int MT>
__device__ __forceinline__ void copy_out16(const __bf16* xs, float* dst, int L, int tid) {
    asm volatile("" : "+v"(tid));                               // (the first store address is formed here, not ahead of the layer loop)
    for (int i = tid; i < L * (CPAD / 4); i += SEG_THREADS) {
        const int p = i >> 5, c4 = i & 31;
        ((v4f*)dst)[i] = load_cell<SPLIT, Geo<SPLIT, MT>::PLANE>(xs + (HALO + p) * S16 + c4 * 4);
    }
}

// PERSIST: one workgroup per CU walking the row list (used with empty-row skipping, where the rows to do are a device-side
// list); otherwise one workgroup per row -- the row loop costs these kernels registers they do not have (spills, -10 %).
template <bool SPLIT, int MT, bool PERSIST>
__global__ __launch_bounds__(SEG_THREADS, NWAVE / 4) void segment16_kernel(Segment16Args a_by_value) {
    typedef Geo<SPLIT, MT> G;
    __shared__ __attribute__((aligned(16))) __bf16 xs[G::PLANES * G::PLANE];
    __shared__ __attribute__((aligned(16))) float cst[MAX_LAYERS * CST_FLOATS];
    // persistent workgroups walking the row list, as in the fp32 kernel (dan_kernels.hip)
    typedef const __attribute__((address_space(4))) Segment16Args* kernarg_ptr;
    auto row_body = [&](int wk, const auto& a) {
    [[maybe_unused]] const int stamp_row = wk;
    int tid = threadIdx.x;
    if (PERSIST) asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row_index = __builtin_amdgcn_readfirstlane(a.work_count ? a.work[wk] : wk);     // uniform: everything derived stays scalar
    const int site = row_index / a.R, r = row_index - site * a.R;
    const int L = a.L;
    const size_t read_idx = (size_t)site * a.R + r;
    float* yrow = a.y + read_idx * (size_t)L * CPAD;
    const int pos = lane & 15, kq = lane >> 4;
    int chb[NT16];
#pragma unroll
    for (int n = 0; n < NT16; ++n) chb[n] = (wave * NT16 + n) * 16 + kq * 4;

    STAMP16(0);
    auto block = [&](int l) { return a.wl + (size_t)l * W16_LAYER_BYTES; };
    auto conv_ptr = [&](int l) { return (gbf8_ptr)(block(l) + W16_CONV_OFF) + (wave * NT16) * 64 + lane; };
    bf8 pre_conv[NT16][2];
    {
        gbf8_ptr w0 = conv_ptr(a.l_begin);
#pragma unroll
        for (int n = 0; n < NT16; ++n) { pre_conv[n][0] = w0[n * 64]; pre_conv[n][1] = w0[n * 64 + (SPLIT ? W16_CONV_FRAGS : 0)]; }
    }
    auto stage_constants = [&]() {
        for (int i = tid; i < (a.l_end - a.l_begin) * CST_FLOATS; i += SEG_THREADS) {
            const int l = i / CST_FLOATS, j = i - l * CST_FLOATS;
            cst[i] = *(const float*)(block(a.l_begin + l) + W16_CST_OFF + (size_t)j * 4);
        }
    };

    if (a.l_begin == 0) {
        for (int i = tid; i < G::PLANES * G::PLANE / 8; i += SEG_THREADS) ((v4f*)xs)[i] = splat4(0.f);
        stage_constants();
        __syncthreads();
        // ---- encode (dl4vc/model.py:450-627), canonical 48-channel order, rounded to bf16 (hi [+ lo])
        const size_t rbase = read_idx * (size_t)L, sbase = (size_t)site * L;
        int ok_ref = 1, ok_var = 1;
        for (int p = tid; p < L; p += SEG_THREADS) {
            const int tok = a.reads[rbase + p], rm = a.ref_mask[sbase + p], vm = a.var_mask[sbase + p];
            ok_ref &= (rm == 0) || (tok == rm);
            ok_var &= (vm == 0) || (tok == vm);
        }
        const int agree_ref = __syncthreads_and(ok_ref);
        const int agree_var = __syncthreads_and(ok_var);
        for (int p = tid; p < L; p += SEG_THREADS) {
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
            __bf16* cell = xs + (HALO + p) * S16;
#pragma unroll
            for (int c = 0; c < CIN0; c += 4) store_cell<SPLIT, G::PLANE>(cell + c, (v4f){row[c], row[c + 1], row[c + 2], row[c + 3]});
        }
    } else {
        const v4f* src = (const v4f*)yrow;
        const v4f* pl = a.pool ? (const v4f*)(a.pool + (size_t)site * L * CPAD) : nullptr;
        // whole read (and pool image) in flight at once, then convert + store (see the fp32 kernel's prologue)
        const int n4 = L * (CPAD / 4);
        constexpr int NPF = (G::MPOS_ * (CPAD / 4) + SEG_THREADS - 1) / SEG_THREADS;
        v4f vy[NPF], vp[NPF];                                    // the read AND its site's pool image in flight together
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * SEG_THREADS;
            vy[k] = (i < n4) ? src[i] : splat4(0.f);
        }
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * SEG_THREADS;
            vp[k] = (pl && i < n4) ? pl[i] : splat4(0.f);
        }
        // while the read is in flight: zero the rows it does not cover (halo rows and rows >= L, both planes) and stage
        // the constants
        constexpr int RV = S16 / 8;                               // 16-byte vectors per row
        for (int i = tid; i < G::PLANES * (G::ROWS - L) * RV; i += SEG_THREADS) {
            const int pl_i = i / ((G::ROWS - L) * RV), j = i - pl_i * ((G::ROWS - L) * RV);
            const int rr = j / RV, c8 = j - rr * RV;
            const int row = rr < HALO ? rr : rr + L;
            *(v4f*)(xs + pl_i * G::PLANE + row * S16 + c8 * 8) = splat4(0.f);
        }
        stage_constants();
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * SEG_THREADS;
            if (i < n4) store_cell<SPLIT, G::PLANE>(xs + (HALO + (i >> 5)) * S16 + (i & 31) * 4, vy[k] + vp[k]);
        }
    }
    __syncthreads();
    STAMP16(1);
    if (a.tap && a.tap_layer == 0 && a.l_begin == 0) copy_out16<SPLIT, MT>(xs, a.tap + read_idx * (size_t)L * CPAD, L, tid);

    for (int l = a.l_begin; l < a.l_end; ++l) {
        const float* lc = cst + (l - a.l_begin) * CST_FLOATS;
        const bool residual = (a.res_mask >> l) & 1u;
        const int kg = (l == 0) ? KG16_0 : KG16_C;
        const int dil = (l == 0) ? 1 : (l + 1 < a.n_layers ? a.dil_mid : a.dil_final);
        gbf8_ptr w_conv = conv_ptr(l);
        gbf8_ptr w_res = (gbf8_ptr)(block(l) + W16_RES_OFF) + (wave * NT16) * 64 + lane;
        gbf8_ptr w_bot = (gbf8_ptr)(block(l) + W16_BOT_OFF) + lane;
        bf8 pre_res[NT16][2], pre_next[NT16][2];
#pragma unroll
        for (int n = 0; n < NT16; ++n) {
            pre_res[n][0] = w_res[n * 64];
            pre_res[n][1] = w_res[n * 64 + (SPLIT ? W16_RES_FRAGS : 0)];
            gbf8_ptr wn = (l + 1 < a.l_end) ? conv_ptr(l + 1) : w_conv;
            pre_next[n][0] = wn[n * 64];
            pre_next[n][1] = wn[n * 64 + (SPLIT ? W16_CONV_FRAGS : 0)];
        }

        v4f acc[MT][NT16];
        {
            v4f bias[NT16];
#pragma unroll
            for (int n = 0; n < NT16; ++n) bias[n] = *(const v4f*)(lc + CST_BIAS + chb[n]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT16; ++n) acc[m][n] = bias[n];
        }
        [[maybe_unused]] const int sb = 2 + (l - a.l_begin) * 8;
        STAMP16(sb + 0);
        gemm16<SPLIT, MT, KGC>(acc, xs, w_conv, W16_CONV_FRAGS, pre_conv, kg, 3, dil, lane);
        STAMP16(sb + 1);
        // bottleneck schedule: the bf16 kernels defer layer l's 128 -> 32 GEMM into layer l + 1's stage (older four waves, in
        // their wait at the barrier below); the segment's last layer keeps a stage of its own.  (The bf16x3 kernels have no
        // registers for it.)
        constexpr bool DEFER = !SPLIT;
        bf8 bot_h[CPAD / 32], bot_l[SPLIT ? CPAD / 32 : 1];
        if (DEFER) {
            if (a.has_hw && l > a.l_begin && wave < NWAVE / 2) {
                bottleneck16_weights<SPLIT>(bot_h, bot_l, (gbf8_ptr)(block(l - 1) + W16_BOT_OFF) + lane, W16_BOT_FRAGS, wave);
                bottleneck16<SPLIT, MT, NWAVE / 2>(xs, bot_h, bot_l, lc - CST_FLOATS + CST_BBOT,
                                                   a.h + (size_t)(l - 1) * a.h_layer_stride + read_idx * (size_t)L * HPAD, L, wave, lane);
            }
            if (a.has_hw && l + 1 == a.l_end) bottleneck16_weights<SPLIT>(bot_h, bot_l, w_bot, W16_BOT_FRAGS, wave);
        } else if (a.has_hw) {
            bottleneck16_weights<SPLIT>(bot_h, bot_l, w_bot, W16_BOT_FRAGS, wave);
        }
        {
            v4f sc[NT16], sh[NT16];
#pragma unroll
            for (int n = 0; n < NT16; ++n) { sc[n] = *(const v4f*)(lc + CST_SCALE + chb[n]); sh[n] = *(const v4f*)(lc + CST_SHIFT + chb[n]); }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int n = 0; n < NT16; ++n) {
                    v4f v = acc[m][n];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = relu1f(v[j]) * sc[n][j] + sh[n][j];
                    acc[m][n] = v;
                }
                if ((m + 1) * 16 > L) {
                    const bool live = (m * 16 + pos) < L;
#pragma unroll
                    for (int n = 0; n < NT16; ++n) acc[m][n] = live ? acc[m][n] : splat4(0.f);
                }
            }
        }
        STAMP16(sb + 2);
        __syncthreads();
        STAMP16(sb + 3);
        if (residual) {
            v4f bres[NT16];
#pragma unroll
            for (int n = 0; n < NT16; ++n) bres[n] = *(const v4f*)(lc + CST_BRES + chb[n]);
            const bool from_global = (l == a.l_begin) && (a.l_begin != 0) && (a.pool != nullptr);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int p = m * 16 + pos;
#pragma unroll
                for (int n = 0; n < NT16; ++n) {
                    __bf16* cell = xs + (HALO + p) * S16 + chb[n];
                    v4f old = load_cell<SPLIT, G::PLANE>(cell);
                    if (from_global) {
                        // (an opaque copy of the position: otherwise all MT row addresses are formed ahead of the layer
                        // loop and live -- spilled -- through every GEMM for a branch one layer per segment takes)
                        int po = pos;
                        asm volatile("" : "+v"(po));
                        const int pg = m * 16 + po;
                        old = (pg < L) ? *(const v4f*)(yrow + (size_t)pg * CPAD + chb[n]) : splat4(0.f);
                    }
                    store_cell<SPLIT, G::PLANE>(cell, acc[m][n]);
                    acc[m][n] = old + bres[n];
                }
            }
            __syncthreads();
            STAMP16(sb + 4);
            gemm16<SPLIT, MT, KGC>(acc, xs, w_res, W16_RES_FRAGS, pre_res, KG16_C, 1, 0, lane);
            STAMP16(sb + 5);
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if ((m + 1) * 16 > L) {
                    const bool live = (m * 16 + pos) < L;
#pragma unroll
                    for (int n = 0; n < NT16; ++n) acc[m][n] = live ? acc[m][n] : splat4(0.f);
                }
#pragma unroll
                for (int n = 0; n < NT16; ++n) store_cell<SPLIT, G::PLANE>(xs + (HALO + m * 16 + pos) * S16 + chb[n], acc[m][n]);
            }
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT16; ++n) store_cell<SPLIT, G::PLANE>(xs + (HALO + m * 16 + pos) * S16 + chb[n], acc[m][n]);
        }
        __syncthreads();
        STAMP16(sb + 6);
        if (a.tap && a.tap_layer == l + 1) copy_out16<SPLIT, MT>(xs, a.tap + read_idx * (size_t)L * CPAD, L, tid);
        if (a.has_hw && (!DEFER || l + 1 == a.l_end))
            bottleneck16<SPLIT, MT, NWAVE>(xs, bot_h, bot_l, lc + CST_BBOT,
                                           a.h + (size_t)l * a.h_layer_stride + read_idx * (size_t)L * HPAD, L, wave, lane);
        STAMP16(sb + 7);
#pragma unroll
        for (int n = 0; n < NT16; ++n) { pre_conv[n][0] = pre_next[n][0]; pre_conv[n][1] = pre_next[n][1]; }
    }
    STAMP16(62);
    copy_out16<SPLIT, MT>(xs, yrow, L, tid);
    STAMP16(63);
    };   // row_body
    if constexpr (PERSIST) {
        const int n_work = a_by_value.work_count ? *a_by_value.work_count : a_by_value.n_rows;
        for (int wk = blockIdx.x; wk < n_work; wk += gridDim.x) {
            // arguments re-read per row through an opaque kernarg pointer (see the fp32 kernel)
            kernarg_ptr ap = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ap));
            row_body(wk, *ap);
            __syncthreads();                                // the next row re-uses the LDS image
        }
    } else {
        // one workgroup per row; with a row list there may be fewer rows than workgroups
        if (!a_by_value.work_count || (int)blockIdx.x < *a_by_value.work_count) row_body((int)blockIdx.x, a_by_value);
    }
}

void launch_segment16(const Segment16Args& a0, int n_sites, int precision, int max_wgs, hipStream_t s) {
    Segment16Args a = a0;
    a.n_rows = n_sites * a.R;
    const bool persist = a.work_count != nullptr && max_wgs > 0 && max_wgs < a.n_rows;
    const char* form_env = getenv("DAN_BF16_FORM");           // 8: the eight-wave kernel for plain bf16 too (A/B runs, tests)
    const int form = form_env ? atoi(form_env) : 4;
    if (precision == 2 && !persist && form == 4) { launch_segment16w(a0, n_sites, s); return; }
    const dim3 grid((unsigned)(persist ? max_wgs : a.n_rows)), blk(SEG_THREADS);
    if (persist) {
        if (precision == 1) hipLaunchKernelGGL((segment16_kernel<true, 13, true>), grid, blk, 0, s, a);
        else if (a.L <= 13 * 16) hipLaunchKernelGGL((segment16_kernel<false, 13, true>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((segment16_kernel<false, 19, true>), grid, blk, 0, s, a);
    } else {
        if (precision == 1) hipLaunchKernelGGL((segment16_kernel<true, 13, false>), grid, blk, 0, s, a);
        else if (a.L <= 13 * 16) hipLaunchKernelGGL((segment16_kernel<false, 13, false>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((segment16_kernel<false, 19, false>), grid, blk, 0, s, a);
    }
}

}  // namespace dan
