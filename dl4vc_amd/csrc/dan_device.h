// Device-side building blocks shared by the inference kernels (dan_kernels.hip) and the training kernels
// (dan_train.hip): MFMA wrappers, the LDS-resident implicit-GEMM core of one read, the 128 -> 32 bottleneck GEMM.
#pragma once
#include "dan_kernels.h"

namespace dan {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4f mfma16(float a, float b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ v4f splat(float x) { return (v4f){x, x, x, x}; }

// ------------------------------------------------------------------------------------------------
// implicit-GEMM core:  acc[m][n] (+)= sum_{tap,g,s} Wfrag[tap][g][n][s] * X[pos + shift(tap)][16g + 4kk + s]
// ------------------------------------------------------------------------------------------------
// Software pipeline: one continuous MFMA stream.  The B fragment of position tile m is single-buffered:
// right after the eight MFMAs that consume it, the ds_read_b128 of the SAME tile for the next k-group is
// issued, so it has twelve tiles of MFMAs (~3000 cycles) to land; the next k-group's weight fragments
// (two 1-KiB global loads per wave) are issued at the top of the step.  sched_group_barrier pins the
// {8 MFMA, 1 ds_read} interleave so that LDS issue never drains the matrix pipe.  The first k-group's
// weight fragments are loaded by the caller well ahead of the call (a_first).
typedef const __attribute__((address_space(1))) v4f* gv4f_ptr;     // global (not flat) loads

// ReLU as ONE compiler-visible instruction (v_med3_f32 v, 0, +inf).  Not inline asm: an asm block that reads an
// MFMA result is invisible to hipcc's hazard recognizer (no wait states are inserted between the MFMA and the asm),
// and fmaxf costs two instructions (canonicalise + max).
__device__ __forceinline__ float relu1(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()); }

// One wave's share: acc[MTW][NT] = its 32 output channels x its position tiles [m_base, m_base + cnt), cnt = 7 or 6.
__device__ __forceinline__ void gemm_tile(v4f (&acc)[NT], const v4f (&a)[NT], v4f& b, const float* next) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = mfma16(a[n][s], b[s], acc[n]);
    b = *(const v4f*)next;
}

__device__ __forceinline__ void conv_gemm(v4f (&acc)[MTW][NT], const float* xs, gv4f_ptr wl, const v4f (&a_first)[NT],
                                          int kg, int ntaps, int dil, int lane, int m_base, int cnt) {
    const int pos = lane & 15, kk = lane >> 4;
    const int total = ntaps * kg;
    const int t0 = (ntaps == 3) ? -dil : 0;
    const float* xrow = xs + (HALO + m_base * 16 + pos) * LDS_S + kk * 4;
    const bool full = cnt == MTW;                              // wave-uniform
    v4f a_nxt[NT], b[MTW];
#pragma unroll
    for (int n = 0; n < NT; ++n) a_nxt[n] = a_first[n];
    {
        const float* xb = xrow + t0 * LDS_S;
#pragma unroll
        for (int m = 0; m < MTW - 1; ++m) b[m] = *(const v4f*)(xb + m * 16 * LDS_S);
        b[MTW - 1] = full ? *(const v4f*)(xb + (MTW - 1) * 16 * LDS_S) : splat(0.f);
    }
    int t = 0, g = 0;
    for (int it = 0; it < total; ++it) {
        v4f a[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) a[n] = a_nxt[n];
        const int nx = (it + 1 < total) ? it + 1 : it;
#pragma unroll
        for (int n = 0; n < NT; ++n) a_nxt[n] = wl[(size_t)nx * (KGC * 64) + n * 64];
        int tn = t, gn = g + 1;
        if (gn == kg) { gn = 0; ++tn; }
        if (it + 1 == total) { tn = t; gn = g; }             // last step: harmless re-read
        const float* xn = xrow + (t0 + tn * dil) * LDS_S + gn * 16;
#pragma unroll
        for (int m = 0; m < MTW - 1; ++m) gemm_tile(acc[m], a, b[m], xn + m * 16 * LDS_S);
        __builtin_amdgcn_sched_group_barrier(0x020, NT, 0);         // the weight loads first
#pragma unroll
        for (int m = 0; m < MTW - 1; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);  // one tile's MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
        }
        if (full) gemm_tile(acc[MTW - 1], a, b[MTW - 1], xn + (MTW - 1) * 16 * LDS_S);
        t = tn; g = gn;
    }
}

// 128 -> 32 highway bottleneck (1x1 conv + ReLU) of the LDS-resident read, written to HBM.  Output unit
// u = 2*pt + n (13 position tiles x 2 channel tiles) is owned by wave u % NWAVE: every wave has ONE channel tile
// n = wave & 1 (so one weight fragment per k-group, no selects) and the position tiles pt = (wave >> 1) mod 4
// -- 4/4/3/3/3/3/3/3 units, i.e. 7/7/6/6 per SIMD.  All eight weight fragments are preloaded by the caller (wf).
// NW = participating waves: all 8 (standalone stage), or only the 4 OLDER waves (0..3, one per SIMD) when the GEMM is
// deferred into the next layer's conv stage (see the Winograd layer body).
// RANGE: only positions p_lo <= p < L are stored (a unit of a split read stores its own columns; dan_kernels.hip SPLIT).
// TILES: 16-column tiles the image holds for this instantiation (MT; 12 for the split kernel's short units, dan_kernels.hip TW = 6).
template <int NW, bool RANGE = false, int TILES = MT>
__device__ __forceinline__ void bottleneck(const float* xs, const v4f (&wf)[KGC], const float* bbot, float* hrow, int L,
                                           int wave, int lane, int p_lo = 0) {
    constexpr int PSTEP = NW / 2;                         // position-tile stride of one wave
    constexpr int NBT = (TILES + PSTEP - 1) / PSTEP;
    // everything below is recomputed per call from an opaque copy of the lane index: hoisted out of the layer loop, the
    // per-tile offsets and 64-bit store addresses would sit in registers through the conv GEMMs (and spill)
    lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));     // (not even the lane index is kept)
    asm volatile("" : "+v"(lane));
    const int pos = lane & 15, kk = lane >> 4;
    const int n = wave & 1, p0 = wave >> 1;
    const float* xrow = xs + (HALO + pos) * LDS_S + kk * 4;
    int roff[NBT];
#pragma unroll
    for (int i = 0; i < NBT; ++i) roff[i] = min(p0 + PSTEP * i, TILES - 1) * 16 * LDS_S;
    v4f acc[NBT], b[NBT];
    {
        const v4f bias = *(const v4f*)(bbot + n * 16 + kk * 4);
#pragma unroll
        for (int i = 0; i < NBT; ++i) acc[i] = bias;
    }
#pragma unroll
    for (int i = 0; i < NBT; ++i) b[i] = *(const v4f*)(xrow + roff[i]);
#pragma unroll
    for (int g = 0; g < KGC; ++g) {
        const int gn = (g + 1 < KGC) ? g + 1 : g;
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[i] = mfma16(wf[g][s], b[i][s], acc[i]);
            b[i] = *(const v4f*)(xrow + roff[i] + gn * 16);
        }
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < NBT; ++i) {
        const int pt = p0 + PSTEP * i, p = pt * 16 + pos;
        if (pt < TILES && p < L && (!RANGE || p >= p_lo)) {
            v4f v = acc[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = relu1(v[j]);
            *(v4f*)((char*)hrow + (unsigned)(p * HPAD + n * 16 + kk * 4) * 4u) = v;
        }
    }
}

// image rows [lo, hi) -> dst rows [lo, hi)   (lo = 0, hi = L: the whole read)
__device__ __forceinline__ void copy_out(const float* xs, float* dst, int lo, int hi, int tid) {
    for (int i = lo * (CPAD / 4) + tid; i < hi * (CPAD / 4); i += SEG_THREADS) {
        const int p = i >> 5, c4 = i & 31;
        ((v4f*)dst)[i] = *(const v4f*)(xs + (HALO + p) * LDS_S + c4 * 4);
    }
}

// ------------------------------------------------------------------------------------------------
// Winograd F(2,3) core over the dilated positions (see the comment block above its use in dan_kernels.hip)
// ------------------------------------------------------------------------------------------------
constexpr int MW = 7;                   // Winograd tiles per lane
constexpr int WHB = 102;                // first position of the upper half's tiling
// The split kernel's short units (windows of 209..304 columns, two units of <= 190 columns each: 161 at 301 columns) take SIX tiles per
// lane: the lower-half lanes tile [0, 96), the upper-half lanes [94, 190) -- 94 == 2 mod 4 like 102, so that the sixteen lanes of a
// ds_read_b128 group still touch rows of all eight residues mod 8.  6 / 7 of the matrix work of a layer for such a unit.
constexpr int MW_SHORT = 6;
constexpr int WHB_SHORT = 94;
constexpr int MPOS_SHORT = WHB_SHORT + 16 * MW_SHORT;    // 190: the longest unit the six-tile form covers
constexpr int wino_half_base(int tw) { return tw == MW ? WHB : WHB_SHORT; }
constexpr int wino_cover(int tw) { return tw == MW ? MPOS : MPOS_SHORT; }
template <int TW = MW>
__device__ __forceinline__ int wino_base(int lane) {
    const int n = lane & 15;
    const int c = n & 1, hf = (n >> 2) & 1, q = ((n >> 1) & 1) + 2 * (((n >> 3) ^ (n >> 2)) & 1);
    return hf * wino_half_base(TW) + 4 * q * TW + c;
}

// acc[m][k] += U_k[own 16 channels][all 128 in-channels] * V_k[tile m]
typedef float v2f __attribute__((ext_vector_type(2)));
// a - b as two v_pk_fma_f32 (b * m1 + a with m1 = -1.0 in a register the compiler cannot see through: hipcc turns a
// v2f32 fsub, or an fma by a literal -1, into two scalar v_sub_f32; every VALU instruction costs the SIMD 3-5 cycles of
// MFMA issue -- tools/ubench/mfma_valu.hip -- so the packed form halves the price of the input transform)
// (round 4 measured the one-float form of the same transform, sixteen v_sub/v_add per tile: 71.85 ms per launch against 71.02 --
// HISTORY.md, round 4, item 4a; the losing variant is not kept in the source)
__device__ __forceinline__ v4f pk_sub(v4f a, v4f b, v2f m1) {
    const v2f lo = __builtin_elementwise_fma((v2f){b[0], b[1]}, m1, (v2f){a[0], a[1]});
    const v2f hi = __builtin_elementwise_fma((v2f){b[2], b[3]}, m1, (v2f){a[2], a[3]});
    return (v4f){lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ v4f pk_add(v4f a, v4f b) { return a + b; }
template <int TW>
__device__ __forceinline__ void conv_gemm_wino(v4f (&acc)[TW][4], const float* xrow, gv4f_ptr wl, const v4f (&a_first)[4]) {
    v4f a_nxt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a_nxt[k] = a_first[k];
    v4f xa = *(const v4f*)(xrow), xb = *(const v4f*)(xrow + 2 * LDS_S), xc = *(const v4f*)(xrow + 4 * LDS_S),
        xd = *(const v4f*)(xrow + 6 * LDS_S);
    float neg1 = -1.f;
    asm volatile("" : "+v"(neg1));
    const v2f m1 = {neg1, neg1};
    for (int g = 0; g < KGC; ++g) {
        v4f a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = a_nxt[k];
        const int gn = (g + 1 < KGC) ? g + 1 : g;            // last step: harmless re-read
#pragma unroll
        for (int k = 0; k < 4; ++k) a_nxt[k] = wl[(size_t)(k * KGC + gn) * (KGC * 64)];
        const float* xg = xrow + g * 16;
        const float* xn = xrow + gn * 16;
#pragma unroll
        for (int m = 0; m < TW; ++m) {
            v4f v[4];
            v[0] = pk_sub(xa, xc, m1); v[1] = pk_add(xb, xc); v[2] = pk_sub(xc, xb, m1); v[3] = pk_sub(xb, xd, m1);
            if (m + 1 < TW) {
                xa = xc; xb = xd;
                xc = *(const v4f*)(xg + (4 * m + 8) * LDS_S);
                xd = *(const v4f*)(xg + (4 * m + 10) * LDS_S);
            } else {
                xa = *(const v4f*)(xn); xb = *(const v4f*)(xn + 2 * LDS_S);
                xc = *(const v4f*)(xn + 4 * LDS_S); xd = *(const v4f*)(xn + 6 * LDS_S);
            }
            // MFMAs at s_setprio 3, the VALU/LDS stretch that forms the next V at 0: the arbiter then serves the other
            // wave's MFMAs ahead of this wave's VALU stream (by default the OLDER wave's instructions of either kind come
            // first, and its VALU batches stall the younger wave's MFMA issue).  tools/ubench/wino_loop.hip: 37.6 -> 36.4
            // cycles per MFMA at the SIMD.
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[m][k] = mfma16(a[k][s], v[k][s], acc[m][k]);
            __builtin_amdgcn_s_setprio(0);
        }
    }
}

// LDS rows a Winograd GEMM may read: tiles past the window read beyond the activation rows (into whatever follows them)
constexpr int WINO_ROWS_READ = HALO + WHB + 4 * (4 * MW - 1) + 1 + 4 + 1;

}  // namespace dan
