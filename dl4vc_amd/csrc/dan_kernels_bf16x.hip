// bf16x3 ("split bf16") conv-stack segment kernel for gfx950 (dan_config.precision = 1): the throughput mode that stays inside
// the 1e-4 score bar of the fp32 reference.  dl4vc/model.py:728-778 for one read resident in LDS, on the design of the plain-bf16
// ping-pong kernel (dan_kernels_bf16p.hip) -- XOR-swizzled unpadded images, persistent XCD-sliced workgroups, LDS-DMA prologue,
// constants staged a layer ahead, the deferred bottleneck -- with every GEMM operand carried as TWO bf16 (hi = bf16(v),
// lo = bf16(v - hi): 16 mantissa bits) and a product as three MFMAs, wh xh + wl xh + wh xl, summed in fp32:
//
//   * v_mfma_f32_16x16x32_bf16, wave = (channel quarter q = wave & 3) x (position half = wave >> 2): 32 output channels x 7
//     tiles of 16 columns = 56 accumulator registers; an activation fragment (1 KiB ds_read_b128) feeds four (hi plane) or two
//     (lo plane) MFMAs, a weight fragment seven;
//   * ONE image of the read in LDS, two planes (hi, lo) of 232 rows x 256 B, updated IN PLACE: a layer's outputs wait in
//     registers -- already split and packed -- for the barrier behind the GEMM, are stored, and a second barrier publishes them
//     (two images of two planes do not fit 160 KiB);
//   * weight rows permuted on the host so that a lane's 2 x 4 accumulators of a column are 8 CONSECUTIVE channels: the epilogue
//     is one 16-byte LDS store per tile and plane;
//   * y crosses HBM as the same two planes ([row][plane][L][128] bf16, the bytes of an fp32 y): the resumed segment's image arrives
//     by LDS-DMA and the copy-out needs no arithmetic; the read-mean enters the layer behind it as conv(pool), one fp32 GEMM
//     per site that seeds the accumulators (conv(y + pool) = conv(y) + conv(pool)); h stays fp32 (the highway kernel is shared
//     with the fp32 path).
//
// Numerics: products of bf16 pairs are exact in fp32, sums fp32; dropped: wl xl (2^-18 of a product) and what the two-piece
// split loses of an operand (2^-17).  Bias / ReLU / BatchNorm / residual add in fp32.  Scores within 1e-4 of the reference
// (tests/test_hip_bf16.py, all golden cases and layer taps).
#include "dan_kernels.h"

namespace dan {
namespace x3 {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) bf8* gbf8;

#define XFENCE() __builtin_amdgcn_sched_barrier(0)

#ifdef DAN_STAMPS
// diagnostic build only (tools/segx_probe.hip): s_memtime stamps of the THIRD row a workgroup walks, per wave
__device__ unsigned long long* g_xstamps;
#define XSTAMP(k_)                                                                                    \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                                         \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                           \
        if (lane == 0 && k == jw + 2 * nj && (k_) < 64) g_xstamps[((size_t)blockIdx.x * NWAVE + wave) * 64 + (k_)] = t_; \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#else
#define XSTAMP(k) do {} while (0)
#endif

__device__ __forceinline__ v4f mfma16(bf8 a, bf8 b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// ReLU as one integer maximum on the bit pattern (see dan_kernels_bf16p.hip)
__device__ __forceinline__ float relu1(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ bf8 lds_read(const char* lds, unsigned addr) { return *(const bf8*)(lds + addr); }
__device__ __forceinline__ void lds_write(char* lds, unsigned addr, bf8 v) { *(bf8*)(lds + addr) = v; }
// one image row = 512 B: the hi plane's 16 chunks, then the lo plane's (X_LO bytes on); chunk c of row r is stored at c ^ (r & 15)
__device__ __forceinline__ unsigned cell_addr(int row, int chunk) { return (unsigned)row * X_ROW_BYTES + (unsigned)((chunk ^ row) & 15) * 16; }
__device__ __forceinline__ void lds8(float (&v)[8], const float* p) {
    const v4f t0 = *(const v4f*)p, t1 = *(const v4f*)(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = t0[j]; v[4 + j] = t1[j]; }
}
// 8 fp32 -> hi = bf16(v) (ties to even), lo = bf16(v - hi).  Written on PAIRS: five instructions per two values (v_cvt_pk_bf16_f32,
// the two halves widened again by one shift and one mask, v_pk_add_f32 with negated operand, v_cvt_pk_bf16_f32); element by
// element hipcc packed only some of the pairs (263 vector instructions per 56-value epilogue instead of 224).
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float (&v)[8], bf8& hi, bf8& lo) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const v2f x = {v[j], v[j + 1]};
        const bf2 h = __builtin_convertvector(x, bf2);
        const v2f d = x - __builtin_convertvector(h, v2f);
        const bf2 l = __builtin_convertvector(d, bf2);
        hi[j] = h[0]; hi[j + 1] = h[1];
        lo[j] = l[0]; lo[j + 1] = l[1];
    }
}
// LDS-DMA (see dan_kernels_bf16p.hip::glds16): 64 lanes x 16 bytes from per-lane global addresses to lds_base + 16 * lane
__device__ __forceinline__ void glds16(const void* src, char* lds_base) {
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

// acc[e][t] += W(channel tile 2 q + e) x X(16 columns of tile t) over TAPS x 4 k-steps of 32 channels, three MFMAs per product.
//   lds:    the image (a row = hi chunks, then lo chunks X_LO bytes on);  xb0..2: this lane's byte address of chunk g = lane >> 4 of
//           its row for tap t (tile 0); chunk 4 ks + g lies at xb ^ (ks << 6), tile t a further t * 16 rows on.
//   w:      this wave's first fragment (+ lane); channel-group-major walk (step = ks * TAPS + tap); a step's four fragments
//           (tile e, plane) = w[(step * 16 + 2 e + plane) * 64].  first[] = steps 0 and 1 (requested a stage ahead by the caller).
// Weights: three steps in flight per wave (a step is 6 PT MFMAs per wave, two waves per SIMD: ~1.3 k cycles; an L2 round trip
// ~1.5 k).  Activations: a ring of RING tiles, the read of tile i + RING - 1 rides among the MFMAs of tile i.
// (Tried: the previous layer's bottleneck folded into this walk -- its B operands ARE the centre tap's fragments.  The units
// cannot be dealt evenly over four waves with one code path; per-tile tests of the wave's share cut the steps into blocks that
// hipcc schedules one by one, per-wave copies of the whole walk tripled the spills: 18 k -> 23 k cycles per layer either way.)
// skip_last (wave-uniform): the wave's last column tile lies past the window entirely (the second half's seventh tile when
// L <= 208) -- its MFMAs are jumped over, 13 tiles per SIMD instead of 14.  (One branch per step; two copies of the whole walk,
// chosen once per layer, cost 114 spilled registers where the copies' register assignments meet.)
// fresh (wave-uniform): the accumulators hold nothing yet -- the first MFMA of each takes the literal 0 as its C operand instead of a
// register the caller cleared (28 register-pair moves per layer and wave).
template <int PT, int TAPS>
__device__ __forceinline__ void gemm_x(v4f (&acc)[2][PT], const char* lds, unsigned xb0, unsigned xb1, unsigned xb2, gbf8 w,
                                       const bf8 (&first)[2][4], bool k_short, bool skip_last, bool fresh = false) {
    constexpr int S = TAPS * X_KS, NA = 3, RING = 4, N = S * PT;
    bf8 a[NA][4], bh[RING], bl[RING];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[0][j] = first[0][j]; a[1][j] = first[1][j]; }
    if (S > 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[2][j] = w[(size_t)(2 * 16 + j) * 64];
    }
    auto xaddr = [&](int i) {
        const int s = i / PT, t = i % PT;
        const unsigned xt = (s % TAPS == 0) ? xb0 : (s % TAPS == 1) ? xb1 : xb2;
        return (xt ^ (unsigned)((s / TAPS) << 6)) + (unsigned)(t * (16 * X_ROW_BYTES));
    };
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) { bh[i] = lds_read(lds, xaddr(i)); bl[i] = lds_read(lds + X_LO, xaddr(i)); }
    XFENCE();
    // (s_setprio 1 or 3 around the walk -- MFMAs of the younger wave ahead of the older wave's epilogue -- measured -1.8 %)
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (TAPS == 3 && s == TAPS * X_KS0) {
            if (k_short) break;
            XFENCE();
        }
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int i = s * PT + t, in = i + RING - 1;
            if (in < N) { bh[in % RING] = lds_read(lds, xaddr(in)); bl[in % RING] = lds_read(lds + X_LO, xaddr(in)); }
            const bf8 xh = bh[i % RING], xl = bl[i % RING];
            if (t + 1 < PT || !skip_last) {
                if (s == 0 && fresh) {
                    acc[0][t] = mfma16(a[s % NA][0], xh, (v4f){0.f, 0.f, 0.f, 0.f});
                    acc[1][t] = mfma16(a[s % NA][2], xh, (v4f){0.f, 0.f, 0.f, 0.f});
                } else {
                    acc[0][t] = mfma16(a[s % NA][0], xh, acc[0][t]);
                    acc[1][t] = mfma16(a[s % NA][2], xh, acc[1][t]);
                }
                acc[0][t] = mfma16(a[s % NA][1], xh, acc[0][t]);
                acc[1][t] = mfma16(a[s % NA][3], xh, acc[1][t]);
                acc[0][t] = mfma16(a[s % NA][0], xl, acc[0][t]);
                acc[1][t] = mfma16(a[s % NA][2], xl, acc[1][t]);
            }
        }
        if (s + NA < S) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[s % NA][j] = w[(size_t)((s + NA) * 16 + j) * 64];
        }
        // (a fence per tile -- reads for the tile RING - 1 ahead, six MFMAs, fence -- gave the intended ISA for the 1x1 walk and a
        // 3-tap walk sixty times slower: 464 spilled scalar registers, reloaded through memory inside the loop)
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            if (s * PT + t + RING - 1 < N) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            if (s * PT + t + RING - 1 < N) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (s + NA < S) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
    }
    // Eight wait states between the walk's last MFMA and whatever reads an accumulator next (the caller's epilogue).  gfx950 has no
    // interlock for a VALU read of a register an MFMA is still writing; hipcc covers it with s_nop -- but not on every path: the last
    // step's `skip_last` branch jumps from the MFMAs of tile PT - 2 straight to the join, and ROCm 7.2's hazard recognizer, which walks
    // a block's predecessors with one visited-set for all paths, can miss that edge (it reaches the branch block first THROUGH the
    // skipped block).  With hipcc's default and max-ilp schedules the tiles read first behind the join are finished long before it, so
    // nothing is wrong today; under -amdgpu-sched-strategy=iterative-ilp their last MFMA is the instruction in front of the branch and
    // the second-half waves read a stale accumulator now and then (one 16-column tile of one row in ~16 k: HISTORY.md section 14.2,
    // tools/isa_hazard_check.py, tests/test_isa_hazards.py).  The nop makes the join safe whatever the schedule: 8 of ~18 k cycles.
    asm volatile("s_nop 7");
}

__device__ __forceinline__ void load_first(bf8 (&f)[2][4], gbf8 w) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) f[s][j] = w[(size_t)(s * 16 + j) * 64];
}

// h = relu(Wb y + bb), 128 -> 32 (model.py:774) from the image, 16-column tiles dealt over waves 0 .. NW-1.  fp32 out.
//   wbot: the layer's bottleneck fragments (+ lane), (ks, 2 e + plane) = wbot[(ks * 4 + ..) * 64];  bbp: its 32 biases.
// Channel-group major: the four weight fragments of a group are held for all of the wave's tiles (two groups in flight: 32
// registers, not the 64 of the whole matrix), the activation fragments of the next (group, tile) are requested under the MFMAs of
// the current one.  A tile index past the image is clamped (the wave recomputes its last tile; nothing is stored twice).
// NW = 8: a stage of its own.  NW = 4: the deferred form -- the SIMD arbiter serves the older wave first, so waves 0-3 leave the
// conv GEMM ~7 k cycles before waves 4-7 and would wait at the barrier: they run the PREVIOUS layer's bottleneck in that wait,
// from the image the GEMM has just read (as the fp32 kernel does).
// p_lo, p_hi: the columns that are stored (a unit of a split read stores its own columns only; default: the whole window)
template <int NW, int XPT = X_PT>
__device__ __forceinline__ void bottleneck_x(const char* lds, gbf8 wbot, const float* bbp, float* hrow, int L, int wave, int lane,
                                             int p_lo = 0, int p_hi = 1 << 30) {
    constexpr int NT = 2 * XPT, NTL = (NT + NW - 1) / NW, N = X_KS * NTL;
    asm volatile("" : "+v"(lane));                               // (addresses formed here, not ahead of the layer loop)
    const int n = lane & 15, g = lane >> 4;
    const unsigned xa0 = cell_addr(P_HALO + n, g);
    const int nt_live = min(NT, (L + 15) >> 4);                  // tiles that hold window columns
    bf8 a[2][4], bh[2], bl[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[0][j] = wbot[(size_t)j * 64];
    const v4f bb0 = *(const v4f*)(bbp + 8 * g), bb1 = *(const v4f*)(bbp + 8 * g + 4);
    v4f h[NTL][2];
#pragma unroll
    for (int i = 0; i < NTL; ++i) { h[i][0] = bb0; h[i][1] = bb1; }
    auto xaddr = [&](int k) {                                    // k = ks * NTL + i
        const int ks = k / NTL, i = k % NTL;
        const int tl = min(wave + NW * i, nt_live - 1);
        return (unsigned)(tl * (16 * X_ROW_BYTES)) + (xa0 ^ (unsigned)(ks << 6));
    };
    bh[0] = lds_read(lds, xaddr(0)); bl[0] = lds_read(lds + X_LO, xaddr(0));
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int ks = k / NTL, i = k % NTL;
        if (i == 0 && ks + 1 < X_KS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[(ks + 1) & 1][j] = wbot[(size_t)((ks + 1) * 4 + j) * 64];
        }
        if (k + 1 < N) { bh[(k + 1) & 1] = lds_read(lds, xaddr(k + 1)); bl[(k + 1) & 1] = lds_read(lds + X_LO, xaddr(k + 1)); }
        const bf8 xh = bh[k & 1], xl = bl[k & 1];
        h[i][0] = mfma16(a[ks & 1][0], xh, h[i][0]);
        h[i][1] = mfma16(a[ks & 1][2], xh, h[i][1]);
        h[i][0] = mfma16(a[ks & 1][1], xh, h[i][0]);
        h[i][1] = mfma16(a[ks & 1][3], xh, h[i][1]);
        h[i][0] = mfma16(a[ks & 1][0], xl, h[i][0]);
        h[i][1] = mfma16(a[ks & 1][2], xl, h[i][1]);
    }
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int tl = wave + NW * i, p = 16 * tl + n;
        if (tl < NT && p < L && p >= p_lo && p < p_hi) {
            v4f h0 = h[i][0], h1 = h[i][1];
#pragma unroll
            for (int j = 0; j < 4; ++j) { h0[j] = relu1(h0[j]); h1[j] = relu1(h1[j]); }
            float* o = hrow + (size_t)p * HPAD + 8 * g;
            *(v4f*)o = h0;
            *(v4f*)(o + 4) = h1;
        }
    }
}

// SPLIT = true: windows of 209..304 columns, every read as two overlapping units of the SAME length a.L (SegmentXArgs::units == 2,
// dan_kernels.h plan_units -- the fp32 kernel's scheme): a work item is (row, unit), position-indexed pointers are offset to the
// unit's first column of a window of Lw columns, the agreement predicates look at the whole window, a unit stores (y, h, tap) its
// own columns only and y crosses segments out of place.  SPLIT = false is the kernel as it was.
struct XUnit { int off, lo, hi; };                              // first window column, own columns [lo, hi) (unit-relative)
// PT: 16-column tiles per wave -- X_PT = 7 (units of up to 208 columns: 7 + 6 tiles per SIMD, the fourteenth is the phantom), or 6 for
// the split kernel's units of up to 192 columns (6 + 5 tiles per SIMD at the 161 columns of a 301-column window: 11 / 13 of the work).
template <bool SPLIT, int PT = X_PT>
__global__ __launch_bounds__(SEG_THREADS, NWAVE / 4) void segmentx_kernel(SegmentXArgs a) {
    static_assert(PT == X_PT || (SPLIT && PT == X_PT - 1), "the six-tile form exists for split units only");
    __shared__ __attribute__((aligned(16))) char lds[X_LDS_BYTES];
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int q = wave & 3, half = wave >> 2;
    const int L = a.L;                                           // columns of one LDS-resident unit (the whole window unless SPLIT)
    const int Lw = SPLIT ? a.Lw : L;                             // the window: position stride of every tensor
    const int pbase = half * (PT * 16);
    const bool phantom = pbase + 16 * (PT - 1) >= L;             // (wave-uniform) the wave's last tile lies past the window entirely
    // behind the two planes: the per-channel constants (bias, scale, shift, bres: 512 floats) of the current layer and of the
    // next one, staged a layer ahead
    auto cbuf = [&](int l) { return (float*)(lds + X_IMG_BYTES + (l & 1) * 2048); };

    // zeroed once: the halo rows and the rows past the window are never written with anything but zeros afterwards
    for (int i = tid0; i < X_LDS_BYTES / 16; i += SEG_THREADS) *(v4f*)(lds + (size_t)i * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // persistent walk over XCD-contiguous slices of whole sites (see dan_kernels_bf16p.hip)
    constexpr int UNITS = SPLIT ? 2 : 1;
    const int n_work = (a.work_count ? *a.work_count : a.n_rows) * UNITS;
    const int slice = a.work_count ? (n_work + 7) / 8 : a.slice_rows * UNITS;
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, nj = gridDim.x >> 3;
    auto row_of = [&](int k) {
        const int wk = xcd * slice + k;
        if (k >= slice || wk >= n_work) return -1;
        const int wr = SPLIT ? wk >> 1 : wk;
        return a.work_count ? a.work[wr] : wr;
    };
    auto unit_of = [&](int k) -> XUnit {                         // (wave-uniform)
        if (!SPLIT) return XUnit{0, 0, L};
        const int u = (xcd * slice + k) & 1;
        return u ? XUnit{a.u_off[1], a.own_lo[1], a.own_hi[1]} : XUnit{a.u_off[0], a.own_lo[0], a.own_hi[0]};
    };
    const bool resumed = a.l_begin > 0;
    const size_t y_row = (size_t)2 * Lw * CPAD;                 // bf16 elements of one read's two planes
    // a resumed segment's input: both planes of the read by LDS-DMA (1-KiB pieces of 4 rows; the chunk swizzle goes on the
    // per-lane SOURCE address, the destination is lane-linear)
    auto dma_read = [&](int row_index, int u_off, int lane) {
        const char* ysrc = (const char*)(a.y + (size_t)row_index * y_row) + (size_t)u_off * P_ROW_BYTES;
        char* img = lds + P_HALO * X_ROW_BYTES;
        // a 1-KiB piece = two image rows: lane -> row 2 kb + (lane >> 5), plane (lane >> 4) & 1, stored chunk lane & 15
        const int pl = (lane >> 4) & 1;
        for (int kb = wave; kb * 2 < L; kb += NWAVE) {
            const int p = 2 * kb + (lane >> 5), r = P_HALO + p;
            if (p < L) glds16(ysrc + ((size_t)pl * Lw + p) * P_ROW_BYTES + (((lane ^ r) & 15) << 4), img + kb * 1024);
        }
    };
    // ... and the seed of its first layer's accumulators: conv(pool) of the read's site (launch_conv_pool, model.py:742)
    v4f acc[2][PT];
    auto seed_request = [&](int row_index, int u_off, int lane) {
        const int n = lane & 15, g = lane >> 4;
        const float* cp = a.pool + ((size_t)(row_index / a.R) * (size_t)Lw + u_off) * CPAD + 32 * q + 8 * g;
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int p = pbase + 16 * t + n;
            if (p < L) { acc[0][t] = *(const v4f*)(cp + (size_t)p * CPAD); acc[1][t] = *(const v4f*)(cp + (size_t)p * CPAD + 4); }
            else { acc[0][t] = (v4f){0.f, 0.f, 0.f, 0.f}; acc[1][t] = (v4f){0.f, 0.f, 0.f, 0.f}; }
        }
    };
    auto blk_of = [&](int l) { return a.wl + (size_t)l * WX_LAYER_BYTES; };
    // What a row needs from global memory at its start is requested a row AHEAD, in front of the previous row's copy-out stores
    // (vector-memory operations of a wave retire in order: a load behind 13 stores waits for them): the first layer's constants
    // and first weight fragments (the same for every row), the row's token bytes (encode) or image + accumulator seed (resumed)
    v4f creq;
    bf8 pre_a[2][4];
    int tk_tok = 0, tk_q = 0, tk_st = 0, tk_rf = 0, tk_rm = 0, tk_vm = 0;
    [[maybe_unused]] int tw_tok = 0, tw_rm = 0, tw_vm = 0;         // SPLIT: the WINDOW's column tid0 (the agreement predicates look at all of it)
    auto cst_request = [&](int l, int tid) { if (tid < 128) creq = *(const v4f*)((const float*)(blk_of(l) + WX_CST_OFF) + tid * 4); };
    auto first_request = [&](int tid) {                          // the segment's first layer, for the next row
        cst_request(a.l_begin, tid);
        load_first(pre_a, (gbf8)(blk_of(a.l_begin) + WX_CONV_OFF) + 4 * q * 64 + (tid & 63));
    };
    auto token_request = [&](int row_index, int u_off, int tid) {  // one column per thread (L <= 208, Lw <= 304 < 512)
        const size_t rb0 = (size_t)row_index * Lw, sb0 = (size_t)(row_index / a.R) * Lw;
        if (tid < L) {
            const size_t rb = rb0 + u_off + tid, sbs = sb0 + u_off + tid;
            tk_tok = a.reads[rb]; tk_q = a.qual[rb]; tk_st = a.strand[rb];
            tk_rf = a.ref[sbs]; tk_rm = a.ref_mask[sbs]; tk_vm = a.var_mask[sbs];
        }
        if constexpr (SPLIT) {
            tw_tok = tw_rm = tw_vm = 0;
            if (tid < Lw) { tw_tok = a.reads[rb0 + tid]; tw_rm = a.ref_mask[sb0 + tid]; tw_vm = a.var_mask[sb0 + tid]; }
        }
    };
    {
        const int r0 = __builtin_amdgcn_readfirstlane(row_of(jw));
        if (r0 >= 0) {
            const XUnit u0 = unit_of(jw);
            first_request(tid0);
            if (resumed) {
                dma_read(r0, u0.off, tid0 & 63);
                if (a.pool) seed_request(r0, u0.off, tid0 & 63);
            } else {
                token_request(r0, u0.off, tid0);
            }
        }
    }
    // every workgroup of a launch has the same work per row: left alone they stay in step and take turns at HBM in bursts (all
    // 256 copy-outs, then all 256 image loads).  A start offset of stagger x 256 cycles per workgroup spreads the phases.
    for (int i = 0; i < (int)blockIdx.x * a.stagger; ++i) __builtin_amdgcn_s_sleep(4);

    for (int k = jw; k < slice; k += nj) {
        const int row_index = __builtin_amdgcn_readfirstlane(row_of(k));
        if (row_index < 0) break;
        const int next_row = __builtin_amdgcn_readfirstlane(row_of(k + nj));
        const XUnit cu = unit_of(k), nu = unit_of(k + nj);         // this work item's unit and the next one's
        // (an opaque copy of the thread index per row: hipcc otherwise forms every per-lane address of the row body ahead of the
        // row loop and keeps them -- spilled -- through all of it)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int n = lane & 15, g = lane >> 4;
        const size_t read_idx = (size_t)row_index;
        auto cst_put = [&](int l) { if (tid < 128) *(v4f*)(cbuf(l) + tid * 4) = creq; };
        XSTAMP(0);
        const int c0 = 32 * q + 8 * g;                            // this lane's 8 output channels
        const int row0 = P_HALO + pbase + n;                      // its row in tile 0
        const unsigned wa = cell_addr(row0, 4 * q + g);           // its output chunk
        if (!resumed) {
            // ---- encode (dl4vc/model.py:450-627), canonical 48-channel order, split into the two planes; thread p = column p
            const int tok = tk_tok, qv = tk_q, st = tk_st, rf = tk_rf, rm = tk_rm, vm = tk_vm;
            const bool col = tid < L;
            int ok_ref = !col || (rm == 0) || (tok == rm), ok_var = !col || (vm == 0) || (tok == vm);
            if constexpr (SPLIT) {                               // ... over the whole window, not the unit
                ok_ref = (tw_rm == 0) || (tw_tok == tw_rm);
                ok_var = (tw_vm == 0) || (tw_tok == tw_vm);
            }
            // workgroup-wide AND through sixteen flag words in the constants buffer that is not in use at a row's start
            int* flags = (int*)cbuf(a.l_begin + 1);
            {
                const int w_ref = __all(ok_ref), w_var = __all(ok_var);
                if (lane == 0) { flags[wave] = w_ref; flags[NWAVE + wave] = w_var; }
            }
            __syncthreads();
            int agree_ref = 1, agree_var = 1;
#pragma unroll
            for (int w8 = 0; w8 < NWAVE; ++w8) { agree_ref &= flags[w8]; agree_var &= flags[NWAVE + w8]; }
            if (col) {
                const int p = tid;
                const float* er = a.emb + min(tok, VOCAB - 1) * EMBED;
                const float* ef = a.emb + min(rf, VOCAB - 1) * EMBED;
                const float* pp = a.pe + (cu.off + p) * EMBED;
                float row[64];
#pragma unroll
                for (int e = 0; e < EMBED; ++e) { const float pv = pp[e]; row[e] = er[e] + pv; row[EMBED + e] = ef[e] + pv; }
                row[40] = (float)qv * 0.01f;
                row[41] = (float)st * 0.5f;
                row[42] = (rm != 0 && agree_ref) ? 1.f : 0.f;
                row[43] = (vm != 0 && agree_var) ? 1.f : 0.f;
                row[44] = (rm != 0) ? 1.f : 0.f;
#pragma unroll
                for (int c = 45; c < 64; ++c) row[c] = 0.f;      // (layer 1's second 32-channel step reads chunks 4..7: 6, 7 as zeros)
                const int r = P_HALO + p;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = row[c * 8 + j];
                    bf8 vh, vl;
                    split8(v, vh, vl);
                    lds_write(lds, cell_addr(r, c), vh);
                    lds_write(lds + X_LO, cell_addr(r, c), vl);
                }
            }
        }
        cst_put(a.l_begin);
        if (resumed) {
            // this wave's pieces of the DMA'd image have landed.  Behind them in the queue are only the previous row's copy-out stores
            // (at least L / 16 per wave): they may stay in flight.  vmcnt(12) proves the DMA done only if at least TWELVE operations
            // were issued behind it -- true of the copy-out (L >= 192: six full sweeps per plane for every lane), not of a workgroup's
            // FIRST row, whose DMA is followed by the seed loads alone: a second-half wave issues 2 per tile with a column inside the
            // window, ten at L = 192 exactly (found by reading the count's assumptions in round 6; the first row needs its seed at
            // once anyway, so waiting for everything there costs nothing)
            if (!SPLIT && L >= 192 && k != jw) __builtin_amdgcn_s_waitcnt(0x0F70 | 12);   // vmcnt(12)
            else __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0)  (a split unit stores fewer than twelve chunks per thread)
        }
        __syncthreads();
        XSTAMP(1);

        auto copy_tap = [&](int nch) {
            // image -> fp32 [L][CPAD] (debug tap)
            float* dst = a.tap + (read_idx * (size_t)Lw + cu.off) * CPAD;
            for (int i = cu.lo * (CPAD / 8) + tid; i < cu.hi * (CPAD / 8); i += SEG_THREADS) {
                const int p = i >> 4, c = i & 15;
                const bf8 vh = lds_read(lds, cell_addr(P_HALO + p, c)), vl = lds_read(lds + X_LO, cell_addr(P_HALO + p, c));
                v4f o0, o1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o0[j] = (c * 8 + j < nch) ? (float)vh[j] + (float)vl[j] : 0.f;
                    o1[j] = (c * 8 + 4 + j < nch) ? (float)vh[4 + j] + (float)vl[4 + j] : 0.f;
                }
                *(v4f*)(dst + (size_t)i * 8) = o0;
                *(v4f*)(dst + (size_t)i * 8 + 4) = o1;
            }
        };
        if (a.tap && a.tap_layer == 0 && !resumed) copy_tap(CIN0);

        for (int l = a.l_begin; l < a.l_end; ++l) {
            const char* blk = blk_of(l);
            const float* lc = cbuf(l);
            const bool residual = (a.res_mask >> l) & 1u;
            const bool last_layer = l + 1 == a.l_end;
            // the bottleneck of layer l-1 runs behind this layer's GEMM on the four older waves (the image is intact until the barrier);
            // the segment's last layer has no GEMM behind it: a stage of its own, all eight waves
            const bool defer = a.has_hw && l > a.l_begin && wave < NWAVE / 2;
            const int dil = (l == 0) ? 1 : (l + 1 < a.n_layers ? a.dil_mid : a.dil_final);
            [[maybe_unused]] const int sb = 2 + (l - a.l_begin) * 8;
            XSTAMP(sb + 0);
            XFENCE();
            if (!last_layer) cst_request(l + 1, tid);
            // The conv bias is NOT in the accumulators: relu(acc + b) * sc + sh = max(acc, -b) * sc + (sh + b * sc), and the host
            // stores -b and sh + b * sc in this family's constants (dan_capi.cpp) -- one v_maximum3_f32 per value where ReLU was one
            // integer maximum, and no 28 register-pair moves per layer to spread the bias over the accumulators.  A resumed
            // segment's first layer starts from its seed, conv(pool) of the site (requested a row ahead).
            const bool fresh = !(l == a.l_begin && resumed && a.pool);
            const unsigned xb0 = cell_addr(row0 - dil, g), xb1 = cell_addr(row0, g), xb2 = cell_addr(row0 + dil, g);
            gbf8 wconv = (gbf8)(blk + WX_CONV_OFF) + 4 * q * 64 + lane;
            gemm_x<PT, 3>(acc, lds, xb0, xb1, xb2, wconv, pre_a, l == 0, phantom, fresh);
            XFENCE();
            XSTAMP(sb + 1);

            // ---- the deferred bottleneck first (older waves), before the epilogue needs registers for the packed outputs
            // (Tried in round 5: all sixteen weight fragments of this stage in ONE round trip -- the older waves' stage fell from 6.6 k to
            // 4.7 k cycles under the stamps, but the 64 registers beside the live accumulators spilled 76 and the bench lost 10 %.)
            if (defer)
                bottleneck_x<NWAVE / 2, PT>(lds, (gbf8)(blk_of(l - 1) + WX_BOT_OFF) + lane, (const float*)(blk_of(l - 1) + WX_CST_OFF) + CST_BBOT,
                                        a.h + (size_t)(l - 1) * a.h_layer_stride + (read_idx * (size_t)Lw + cu.off) * HPAD, L, wave, lane, cu.lo, cu.hi);
            XFENCE();
            XSTAMP(sb + 7);
            // ---- epilogue: ReLU, BatchNorm (folded), columns past the window forced to zero, split; the packed outputs wait in
            // registers for the barrier (the image is updated in place)
            bf8 oh[PT], ol[PT];
            // (the empty asm pins the packed values HERE, ahead of the barrier: they are stored under a lane predicate, and hipcc
            // sinks the whole split into those predicated blocks behind the barrier -- where all eight waves then do their vector
            // work at once instead of each in its own wait)
            auto pack_tile = [&](int t, const float (&v)[8]) {
                split8(v, oh[t], ol[t]);
                asm volatile("" : "+v"(oh[t]), "+v"(ol[t]));
            };
            // (columns past the window are never stored: their rows stay zero from the start -- a lane predicate on the two stores of
            // the tile that straddles L instead of eight v_cndmask per tile)
            auto store_tiles = [&]() {
#pragma unroll
                for (int t = 0; t < PT; ++t) {
                    if (pbase + 16 * t + n < L) {
                        lds_write(lds, wa + t * (16 * X_ROW_BYTES), oh[t]);
                        lds_write(lds + X_LO, wa + t * (16 * X_ROW_BYTES), ol[t]);
                    }
                }
            };
            {
                float nb[8], sc[8], sh[8];
                lds8(nb, lc + CST_BIAS + c0);                    // -bias
                lds8(sc, lc + CST_SCALE + c0);
                lds8(sh, lc + CST_SHIFT + c0);                   // shift + bias * scale
#pragma unroll
                for (int t = 0; t < PT; ++t) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const v4f& src = j < 4 ? acc[0][t] : acc[1][t];
                        const v2f x = {src[j & 3], src[(j & 3) + 1]};
                        const v2f m = __builtin_elementwise_maximum(x, (v2f){nb[j], nb[j + 1]});
                        const v2f r = __builtin_elementwise_fma(m, (v2f){sc[j], sc[j + 1]}, (v2f){sh[j], sh[j + 1]});
                        v[j] = r[0]; v[j + 1] = r[1];
                    }
                    pack_tile(t, v);
                }
            }
            // the first fragments of the next GEMM (this layer's residual 1x1, else the next layer's conv) ride under the barrier wait
            if (residual) load_first(pre_a, (gbf8)(blk + WX_RES_OFF) + 4 * q * 64 + lane);
            else if (!last_layer) load_first(pre_a, (gbf8)(blk_of(l + 1) + WX_CONV_OFF) + 4 * q * 64 + lane);
            if (!last_layer) cst_put(l + 1);
            XFENCE();
            XSTAMP(sb + 2);
            __syncthreads();                                     // every read of the layer input is done
            XFENCE();
            XSTAMP(sb + 3);
            if (residual) {
                // y = Wr t + bres + x   (model.py:753-761): x = this lane's own cells of the layer input, replaced by t = the outputs held
                {
                    float br[8];
                    lds8(br, lc + CST_BRES + c0);
#pragma unroll
                    for (int t = 0; t < PT; ++t) {
                        const bf8 xh = lds_read(lds, wa + t * (16 * X_ROW_BYTES)), xl = lds_read(lds + X_LO, wa + t * (16 * X_ROW_BYTES));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[0][t][j] = ((float)xh[j] + (float)xl[j]) + br[j];
                            acc[1][t][j] = ((float)xh[4 + j] + (float)xl[4 + j]) + br[4 + j];
                        }
                    }
                }
                store_tiles();
                __syncthreads();
                XFENCE();
                gbf8 wres = (gbf8)(blk + WX_RES_OFF) + 4 * q * 64 + lane;
                gemm_x<PT, 1>(acc, lds, xb1, xb1, xb1, wres, pre_a, false, phantom);
                XFENCE();
                XSTAMP(sb + 4);
#pragma unroll
                for (int t = 0; t < PT; ++t) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = acc[0][t][j]; v[4 + j] = acc[1][t][j]; }
                    pack_tile(t, v);
                }
                if (!last_layer) load_first(pre_a, (gbf8)(blk_of(l + 1) + WX_CONV_OFF) + 4 * q * 64 + lane);
                __syncthreads();                                 // every read of t is done
                XFENCE();
            }
            store_tiles();
            // (Tried in round 5: the last layer's own bottleneck with its sixteen weight fragments requested here, ahead of the barrier --
            // 9.0 k -> 5.5 k cycles for the stage under the stamps, but 31 spilled registers: -1.5 % in a same-box A/B.  The stage is
            // bound by the address unit anyway: eight waves x 16 KiB of the same matrix.)
            if (last_layer && next_row >= 0) {
                // requests for the NEXT row (see above): issued here, behind the layer's last use of the accumulators, they travel under
                // the barrier, the segment's last bottleneck and the copy-out reads.
                // (Opaque copies of the thread index and the row: without them hipcc forms every address of these requests -- the
                // site index's division, three token pointers, three site pointers, seven seed pointers -- at the HEAD of the layer
                // loop, in every layer, and parks the row's store predicates in VGPR lanes to make room: 150 of a plain layer's ~520
                // vector instructions per wave, for loads that are issued once per row.)
                int t2 = tid, nr = next_row;
                asm volatile("" : "+v"(t2), "+s"(nr));
                first_request(t2);
                int noff = nu.off;
                asm volatile("" : "+s"(noff));
                if (resumed) { if (a.pool) seed_request(nr, noff, t2 & 63); }
                else token_request(nr, noff, t2);
            }
            __syncthreads();
            XFENCE();
            XSTAMP(sb + 5);
            if (a.tap && a.tap_layer == l + 1) copy_tap(CPAD);
            if (a.has_hw && last_layer)
                bottleneck_x<NWAVE, PT>(lds, (gbf8)(blk + WX_BOT_OFF) + lane, (const float*)(blk + WX_CST_OFF) + CST_BBOT,
                                    a.h + (size_t)l * a.h_layer_stride + (read_idx * (size_t)Lw + cu.off) * HPAD, L, wave, lane, cu.lo, cu.hi);
            XSTAMP(sb + 6);
        }
        XSTAMP(62);
        // ---- row end: the segment's output -> y: every chunk of the thread is read
        // from the image (both planes, coalesced 1-KiB stores), a barrier hands the image to the next row's DMA, the DMA is issued
        // and only then the stores.
        {
            // Plane by plane, 512 chunks (32 image rows) per sweep: chunk tid & 15 of row (tid >> 4) + 32 k2 -- the swizzle term
            // (chunk ^ row) & 15 does not change with k2, so the LDS address is ONE base + k2 * 16 KiB (immediate offsets) and the
            // destination one base + k2 * 8 KiB.  (As one flat walk over both planes, with the plane and a clamp derived per element,
            // the 13 reads cost 179 vector instructions of address arithmetic per row.)
            char* ydst = (char*)((SPLIT ? a.y_out : a.y) + read_idx * y_row) + (size_t)cu.off * P_ROW_BYTES;
            constexpr int NCP = (X_LMAX * (CPAD / 8) + SEG_THREADS - 1) / SEG_THREADS;        // 7 sweeps per plane
            const int n8 = Lw * (CPAD / 8);                          // chunks of one plane of the window
            const unsigned src0 = cell_addr(P_HALO + (tid >> 4), tid & 15);
            bf8 v[2][NCP];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int k2 = 0; k2 < NCP; ++k2)
                    v[pl][k2] = lds_read(lds + pl * X_LO + k2 * (32 * X_ROW_BYTES), src0);       // (rows past the window: zeros or stale, never stored)
            __syncthreads();                                     // every read of the image is done: the next row may land in it
            if (resumed && next_row >= 0) dma_read(next_row, nu.off, lane);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int k2 = 0; k2 < NCP; ++k2) {
                    const int i = tid + k2 * SEG_THREADS;              // chunk i of the unit = column i >> 4
                    if (i >= cu.lo * (CPAD / 8) && i < cu.hi * (CPAD / 8)) *(bf8*)(ydst + (size_t)(unsigned)((pl * n8 + i) * 16)) = v[pl][k2];
                }
        }
        XSTAMP(63);
    }
}

}  // namespace x3

void launch_segmentx(const SegmentXArgs& a0, int n_sites, int n_cus, hipStream_t s) {
    SegmentXArgs a = a0;
    a.n_rows = n_sites * a.R;
    a.slice_rows = (n_sites + 7) / 8 * a.R;
    const bool split = a.units == 2;
    if (!split) { a.units = 1; a.Lw = a.L; a.y_out = a.y; }       // (callers that never heard of units: one unit, the whole window, in place)
    int wgs = n_cus > 0 ? n_cus : 256;
    wgs = (wgs + 7) / 8 * 8;
    const int need = (a.slice_rows < 1 ? 1 : a.slice_rows) * 8 * (split ? 2 : 1);   // no more workgroups than work items per slice x 8
    if (wgs > need) wgs = need;
    // one row takes ~30 us (layers 1-2) / ~100 us (layers 3-7): the offsets spread the workgroups over about one row
    if (a.stagger < 0) a.stagger = (a.l_end - a.l_begin) <= 2 ? 1 : 3;
    if (split && a.L <= 2 * (X_PT - 1) * 16) hipLaunchKernelGGL((x3::segmentx_kernel<true, X_PT - 1>), dim3((unsigned)wgs), dim3(SEG_THREADS), 0, s, a);
    else if (split) hipLaunchKernelGGL(x3::segmentx_kernel<true>, dim3((unsigned)wgs), dim3(SEG_THREADS), 0, s, a);
    else hipLaunchKernelGGL(x3::segmentx_kernel<false>, dim3((unsigned)wgs), dim3(SEG_THREADS), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// The two reductions over reads from the two-plane y: value = hi + lo (exact in fp32), fp32 sums in read order, as the fp32 forms
// ------------------------------------------------------------------------------------------------
namespace x3 {

__global__ __launch_bounds__(256) void read_meanx_kernel(const bf8* __restrict__ y, v4f* __restrict__ pool, int R, int L,
                                                         const int* __restrict__ row_src) {
    const int n8 = L * (CPAD / 8);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int site = blockIdx.y;
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int* rs = row_src ? row_src + (size_t)site * R : nullptr;
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
        const bf8* row = y + (size_t)(rs ? rs[r] : site * R + r) * (2 * n8);
        const bf8 vh = row[i], vl = row[n8 + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum[j] += (float)vh[j] + (float)vl[j];
    }
    v4f o0, o1;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o0[j] = sum[j] / (float)R; o1[j] = sum[4 + j] / (float)R; }
    pool[((size_t)site * n8 + i) * 2] = o0;
    pool[((size_t)site * n8 + i) * 2 + 1] = o1;
}

__global__ __launch_bounds__(256) void final_poolx_kernel(const bf8* __restrict__ y, float* __restrict__ feat, long long fs,
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
            auto rowp = [&](int r) { return y + (size_t)(rs ? rs[r] : site * R + r) * (2 * n8) + off; };
            {
                const bf8* r0 = rowp(0);
                const bf8 vh = r0[0], vl = r0[n8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { mx[j] = (float)vh[j] + (float)vl[j]; sum[j] = mx[j]; }
            }
#pragma unroll 4
            for (int r = 1; r < R; ++r) {
                const bf8* rp = rowp(r);
                const bf8 vh = rp[0], vl = rp[n8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)vh[j] + (float)vl[j]; mx[j] = fmaxf(mx[j], f); sum[j] += f; }
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


// ------------------------------------------------------------------------------------------------
// FC on the bf16 matrix cores, split operands (dan_config.precision >= 1):  C = relu?(A W^T + bias), A fp32 [M][lda] (split into
// hi / lo bf16 while it is staged), W as two bf16 planes [2][N][ldw] (split once at dan_finalize), three MFMAs per product
// (ah wh + al wh + ah wl), fp32 sums -- the arithmetic of the conv stack's split kernel, ~2^-17 per operand.  With the fp32 MFMA
// kernel (dan_kernels.hip::fc_kernel, 0.8 of ITS peak) FC1 was 4.4 ms per 4096 x 65 792 x 1024 macro-batch (7.1 ms at config 5's
// 105 728 features) beside a conv stack on bf16 matrix cores.
// The fp32 kernel's structure: 128 x 128 tiles, k-tiles of 32 staged through LDS (80-byte rows: conflict-free ds_read_b128),
// double-buffered, XCD-contiguous tile order, the k range of a tile split over two co-resident workgroups (2 x 80 KiB of LDS)
// whose partial sums fc_combine_kernel adds in a fixed order.
// ------------------------------------------------------------------------------------------------
constexpr int FX_BM = 128, FX_BN = 128, FX_BK = 32, FX_THREADS = 512;
constexpr int FX_ROW = 80;                                   // bytes per LDS row: 32 bf16 + 16 bytes of padding
constexpr int FX_PLANE = FX_BM * FX_ROW;                     // 10 240 B
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(FX_THREADS, 2) void fcx_kernel(const float* __restrict__ A, long long lda, const uint16_t* __restrict__ W,
                                                            long long ldw, long long w_plane, const float* __restrict__ bias,
                                                            float* __restrict__ C, long long ldc, int M, int N, int K, int relu,
                                                            int tiles_n, int n_tiles, int ksplit, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) char lds[2][4 * FX_PLANE];   // per stage: A hi, A lo, W hi, W lo
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int n_work = n_tiles * ksplit;
    const int per = (n_work + 7) >> 3;
    const int qa = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || qa >= n_work) return;
    const int q = qa / ksplit, part = qa - q * ksplit;
    const int bm = (q / tiles_n) * FX_BM, bn = (q % tiles_n) * FX_BN;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32;
    // staging: A: thread -> rows (tid >> 3) + 64 i, floats 4 (tid & 7) ..;  W: thread -> row tid >> 2, 8 bf16 at 8 (tid & 3), both planes
    const int arow = tid >> 3, ac4 = (tid & 7) * 4, wrow = tid >> 2, wc8 = (tid & 3) * 8;
    const float* ga[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) ga[i] = A + (size_t)min(bm + arow + 64 * i, M - 1) * lda + ac4;
    const uint16_t* gw = W + (size_t)min(bn + wrow, N - 1) * ldw + wc8;
    v4f ra[2];
    bf8 rwh, rwl;
    auto gload = [&](int k0) {
        const bool ina = k0 + ac4 < K, inw = k0 + wc8 < K;       // K is a multiple of 16: whole vectors
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = ina ? *(const v4f*)(ga[i] + k0) : (v4f){0.f, 0.f, 0.f, 0.f};
        const bf8 z = {};
        rwh = inw ? *(const bf8*)(gw + k0) : z;
        rwl = inw ? *(const bf8*)(gw + w_plane + k0) : z;
    };
    auto sstore = [&](int buf) {
        char* st = lds[buf];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) { hi[j] = (__bf16)ra[i][j]; lo[j] = (__bf16)(ra[i][j] - (float)hi[j]); }
            *(bf4*)(st + (arow + 64 * i) * FX_ROW + ac4 * 2) = hi;
            *(bf4*)(st + FX_PLANE + (arow + 64 * i) * FX_ROW + ac4 * 2) = lo;
        }
        *(bf8*)(st + 2 * FX_PLANE + wrow * FX_ROW + wc8 * 2) = rwh;
        *(bf8*)(st + 3 * FX_PLANE + wrow * FX_ROW + wc8 * 2) = rwl;
    };
    v4f acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int KT_all = (K + FX_BK - 1) / FX_BK;
    const int kts = (KT_all + ksplit - 1) / ksplit;
    const int kt_lo = part * kts, KT = min(KT_all, kt_lo + kts);
    gload(kt_lo * FX_BK);
    sstore(kt_lo & 1);
    __syncthreads();
    for (int kt = kt_lo; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) gload((kt + 1) * FX_BK);
        const char* st = lds[cur];
        const unsigned oa = (unsigned)((wm + r16) * FX_ROW + kk * 16), ow = (unsigned)((wn + r16) * FX_ROW + kk * 16);
        bf8 ah[4], al[4], wh[2], wl[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *(const bf8*)(st + oa + i * 16 * FX_ROW);
            al[i] = *(const bf8*)(st + FX_PLANE + oa + i * 16 * FX_ROW);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wh[j] = *(const bf8*)(st + 2 * FX_PLANE + ow + j * 16 * FX_ROW);
            wl[j] = *(const bf8*)(st + 3 * FX_PLANE + ow + j * 16 * FX_ROW);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = mfma16(ah[i], wh[j], acc[i][j]);
                acc[i][j] = mfma16(al[i], wh[j], acc[i][j]);
                acc[i][j] = mfma16(ah[i], wl[j], acc[i][j]);
            }
        if (kt + 1 < KT) sstore(cur ^ 1);
        __syncthreads();
    }
    float* wp = ksplit > 1 ? ws + (size_t)part * M * N : nullptr;  // raw partial sums; bias / ReLU in fc_combine_kernel
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = bn + wn + j * 16 + r16;
        if (n >= N) continue;
        const float b = wp ? 0.f : bias[n];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = bm + wm + i * 16 + kk * 4 + jj;
                if (m >= M) continue;
                if (wp) { wp[(size_t)m * N + n] = acc[i][j][jj]; continue; }
                float v = acc[i][j][jj] + b;
                if (relu) v = fmaxf(v, 0.f);
                C[(size_t)m * ldc + n] = v;
            }
    }
}

}  // namespace x3

void launch_fcx(const float* A, long long lda, const uint16_t* W, long long ldw, long long w_plane, const float* bias, float* C,
                long long ldc, int M, int N, int K, int relu, hipStream_t s, float* ws, long long ws_floats) {
    const int tiles_m = (M + x3::FX_BM - 1) / x3::FX_BM, tiles_n = (N + x3::FX_BN - 1) / x3::FX_BN, n_tiles = tiles_m * tiles_n;
    const int ksplit = (ws && ws_floats >= 2LL * M * N && K >= 64 * x3::FX_BK && n_tiles <= 384) ? 2 : 1;
    const int grid = ((n_tiles * ksplit + 7) / 8) * 8;
    hipLaunchKernelGGL(x3::fcx_kernel, dim3(grid), dim3(x3::FX_THREADS), 0, s, A, lda, W, ldw, w_plane, bias, C, ldc, M, N, K, relu, tiles_n,
                       n_tiles, ksplit, ws);
    if (ksplit > 1) launch_fc_combine(ws, ksplit, bias, C, ldc, M, N, relu, s);
}

void launch_read_meanx(const uint16_t* y, float* pool, int n_sites, int R, int L, const int* row_src, hipStream_t s) {
    const int n8 = L * (CPAD / 8);
    hipLaunchKernelGGL(x3::read_meanx_kernel, dim3((n8 + 255) / 256, n_sites), dim3(256), 0, s, (const x3::bf8*)y, (x3::v4f*)pool, R, L,
                       row_src);
}

void launch_final_poolx(const uint16_t* y, float* feat, long long fs, int n_sites, int R, int L, int C, const int* row_src,
                        hipStream_t s) {
    hipLaunchKernelGGL(x3::final_poolx_kernel, dim3((L + 31) / 32, n_sites), dim3(256), 0, s, (const x3::bf8*)y, feat, fs, R, L, C, row_src);
}

}  // namespace dan
