// Plain-bf16 conv-stack segment kernel for gfx950, "ping-pong" form (dan_config.precision = 2; BASELINE config 5:
// 128 reads x 301 bp).  dl4vc/model.py:728-778 for one read resident in LDS, as in the other families, but built around
// what bounded those at 8x the fp32 matrix rate (HISTORY.md section 4):
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

// compile-time fence: nothing is scheduled across it.  Without one behind each GEMM hipcc hoists the epilogue's LDS reads (32
// registers of BatchNorm constants) above the GEMM and spills them inside it
#define PFENCE() __builtin_amdgcn_sched_barrier(0)

#ifdef DAN_STAMPS
// diagnostic build only (tools/segp_probe.hip): s_memtime stamps of the THIRD row a workgroup walks (steady state), per wave
__device__ unsigned long long* g_pstamps;
#define PSTAMP(k_)                                                                                    \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                                         \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                           \
        if (lane == 0 && k == jw + 2 * nj && (k_) < 64) g_pstamps[((size_t)blockIdx.x * NWAVE + wave) * 64 + (k_)] = t_; \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#define RSTAMP(k_) do { if (wave < NWAVE) PSTAMP(k_); } while (0)
#else
#define PSTAMP(k) do {} while (0)
#define RSTAMP(k) do {} while (0)
#endif

__device__ __forceinline__ v16f mfma32(bf8 a, bf8 b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
// ReLU as ONE integer maximum on the bit pattern (negative floats, -0 included, are negative integers): through fmed3 / fmaxf
// hipcc emits a canonicalising v_max_f32 v, v first -- two vector instructions per value in an epilogue bounded by vector issue.
// (Not inline assembly: the hazard recogniser does not see an asm operand, and a ReLU right behind an MFMA read stale registers.)
__device__ __forceinline__ float relu1(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
// v * s + t on register pairs (v_pk_fma_f32: half the issue slots of sixteen v_fma_f32; left to itself hipcc packs about a third)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void scale_shift16(v16f& v, const v16f& s, const v16f& t) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const v2f r = __builtin_elementwise_fma((v2f){v[i], v[i + 1]}, (v2f){s[i], s[i + 1]}, (v2f){t[i], t[i + 1]});
        v[i] = r[0]; v[i + 1] = r[1];
    }
}

__device__ __forceinline__ bf8 lds_read(const char* lds, unsigned addr) { return *(const bf8*)(lds + addr); }
__device__ __forceinline__ void lds_write(char* lds, unsigned addr, bf8 v) { *(bf8*)(lds + addr) = v; }
__device__ __forceinline__ unsigned cell_addr(int row, int chunk) { return (unsigned)row * P_ROW_BYTES + (unsigned)((chunk ^ row) & 15) * 16; }

// acc[m] += W(this wave's 32 channels) x X(32 positions of tile m) over TAPS x KS k-steps of 16 channels.
//   xb0..2: this lane's byte address of chunk (lane >> 5) of its row for tap t in the source image (tile 0); chunk
//           2 ks + (lane >> 5) lies at xb ^ (ks << 5), tile m a further m * 32 rows on.
//   w:      this wave's first fragment (+ lane); the walk is CHANNEL-GROUP major: step s = ks * TAPS + t lies s * 4 fragments on;
//           first[] = steps 0..3.  k_short (wave-uniform): only the first P_KS0 channel groups exist (layer 1's 48 encoded
//           channels): the walk leaves after TAPS x P_KS0 steps -- one exit from the one unrolled shape, 45 MFMAs instead of 120.
// A run-time loop over chunks of NA = 4 k-steps: the four weight fragments of the NEXT chunk are requested slot by slot as
// the current chunk's are consumed (an L2 round trip = 4 x MT MFMAs ahead), the activations of the next k-step while the
// current one's MFMAs issue (double-buffered registers, one ds_read_b128 behind each MFMA).
template <int MT, int TAPS>
__device__ __forceinline__ void gemm_p(v16f (&acc)[MT], const char* lds, unsigned xb0, unsigned xb1, unsigned xb2, gbf8p w,
                                       const bf8 (&first)[4], bool k_short = false) {
    // Fully unrolled over TAPS x 8 k-steps (hipcc then counts its vmcnt waits exactly: a run-time chunk loop drained every
    // outstanding weight fragment once per chunk).  ONE shape serves layer 1 too -- a second conv instance kept a second set of
    // 80 accumulator registers alive -- which leaves the walk early (k_short).
    // Two waves computing side by side cover each other's waits: one activation read behind each MFMA.
    // NA weight fragments in flight per wave (8 measured the same as 4: the L2 round trip is covered)
#ifndef DAN_P_NA
#define DAN_P_NA 4
#endif
    constexpr int NA = DAN_P_NA, S = TAPS * P_KSC;
    bf8 a[NA], b[2][MT];
#pragma unroll
    for (int j = 0; j < NA; ++j) a[j] = (j < 4) ? first[j] : w[(size_t)j * 4 * 64];   // (first four: requested a stage ahead by the caller)
#pragma unroll
    for (int m = 0; m < MT; ++m) b[0][m] = lds_read(lds, xb0 + m * (32 * P_ROW_BYTES));
    // (a hard fence: left in the region below, the first step's reads of tiles 1 .. MT-1 are the first "DS read"s the scheduling
    // groups find -- every read of the whole walk then sits ONE MFMA ahead of its use instead of MT, each MFMA behind an
    // s_waitcnt lgkmcnt(0) on a read issued 8 cycles earlier)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int sn = s + 1;
        const unsigned xt = (sn % TAPS == 0) ? xb0 : (sn % TAPS == 1) ? xb1 : xb2;
        const unsigned xa = xt ^ (unsigned)((sn / TAPS) << 5);
        if (TAPS == 3 && s == TAPS * P_KS0) {
            if (k_short) break;
            __builtin_amdgcn_sched_barrier(0);
        }
        {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = mfma32(a[s % NA], b[s & 1][m], acc[m]);
                if (sn < S) b[sn & 1][m] = lds_read(lds, xa + m * (32 * P_ROW_BYTES));
            }
            if (s + NA < S) a[s % NA] = w[(size_t)(s + NA) * 4 * 64];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (sn < S) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (s + NA < S) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    }
}

__device__ __forceinline__ void load_first(bf8 (&f)[4], gbf8p w) {
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = w[j * 4 * 64];
}

__device__ __forceinline__ void load16(v16f& v, const float* p) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const v4f t = *(const v4f*)(p + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * g + j] = t[j];
    }
}

// 16 fp32 values of one lane (16 consecutive channels of one position) -> two 16-byte chunks of bf16
__device__ __forceinline__ void pack16(const v16f& v, bf8& lo, bf8& hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo[j] = (__bf16)v[j]; hi[j] = (__bf16)v[8 + j]; }
}

// h = relu(Wb * y + bb), 128 -> 32 (model.py:774) from image `img`, position tiles of 32 dealt over waves 0 .. NW-1.
//   a: the layer's eight weight fragments, one per 16-channel k-step, bb: this lane's 16 biases -- requested by the caller a stage
//   ahead (streamed inside the stage, two steps ahead, every second step waited an L2 round trip: 5.6 k cycles for 24 MFMAs).
// Channel-group major (round 4; the sums per output keep their order: bit-identical to the tile-major form): a k-step's ONE
// weight fragment serves all of the wave's tiles, the activation fragments of the next three (step, tile) pairs are in flight
// under the current MFMA -- 16 NTL accumulators + 16 registers of activations, where the tile-major form held two tiles'
// fragments (64 registers) and could not run beside live conv accumulators.
// NW = 8: a stage of its own (the segment's last layer).  NW = 4: the deferred form -- the SIMD arbiter serves the older wave
// first, so waves 0-3 leave the conv GEMM well before waves 4-7 and would wait at the barrier: they run the PREVIOUS layer's
// bottleneck in that wait, from the image the GEMM has just read (as the fp32 and bf16x3 kernels do).
template <int MT, int NW>
__device__ __forceinline__ void bottleneck_ps(const char* img, const bf8 (&a)[P_KSC], const v16f& bb, uint16_t* hrow, int L, int wave, int lane) {
    constexpr int NT = 2 * MT, NTL = (NT + NW - 1) / NW, N = P_KSC * NTL;
    asm volatile("" : "+v"(lane));                               // (addresses formed here, not ahead of the layer loop: they spilled)
    const int n = lane & 31, hh = lane >> 5;
    const unsigned xa0 = cell_addr(P_HALO + n, hh);              // 32 rows further on the swizzle repeats: tile tl lies tl * 8 KiB on
    const int nt_live = min(NT, (L + 31) >> 5);                  // tiles that hold window columns (an index past them is clamped)
    constexpr int RB = 4;                                        // activation fragments in flight: an MFMA is 32 cycles, an LDS round trip > 100
    bf8 bx[RB];
    v16f hacc[NTL];
#pragma unroll
    for (int i = 0; i < NTL; ++i) hacc[i] = bb;
    auto xaddr = [&](int k) {                                    // k = ks * NTL + i
        const int ks = k / NTL, i = k % NTL;
        return (unsigned)(min(wave + NW * i, nt_live - 1) * (32 * P_ROW_BYTES)) + (xa0 ^ (unsigned)(ks << 5));
    };
#pragma unroll
    for (int k = 0; k < RB - 1 && k < N; ++k) bx[k] = lds_read(img, xaddr(k));
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int ks = k / NTL, i = k % NTL;
        if (k + RB - 1 < N) bx[(k + RB - 1) % RB] = lds_read(img, xaddr(k + RB - 1));
        hacc[i] = mfma32(a[ks], bx[k % RB], hacc[i]);
    }
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int tl = wave + NW * i, p = 32 * tl + n;
        if (tl < NT && p < L) {
            v16f hv = hacc[i];
#pragma unroll
            for (int j = 0; j < 16; ++j) hv[j] = relu1(hv[j]);
            bf8 lo, hi;
            pack16(hv, lo, hi);
            bf8* o = (bf8*)(hrow + (size_t)p * HPAD + 16 * hh);
            o[0] = lo;
            o[1] = hi;
        }
    }
}

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to lds_base + 16 * lane (no registers, asynchronous; counted in vmcnt)
// Issued as inline assembly, not through __builtin_amdgcn_global_load_lds: with the builtin in flight hipcc orders every later
// ds_read behind it (it cannot see that the DMA fills the OTHER image) and waits vmcnt(0) -- an HBM round trip -- at the first
// LDS read of the stage that was meant to run under the transfer.  hipcc does not count an asm load (its own vmcnt waits only
// become more conservative: memory returns in order); the consumer waits vmcnt(0) explicitly before the barrier that
// publishes the image.  M0 = the wave-uniform LDS byte address; saved and restored around the instruction.
__device__ __forceinline__ void glds16(const void* src, char* lds_base) {
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

__device__ __forceinline__ void lds16(v16f& v, const float* p) {        // 16 consecutive floats from LDS
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const v4f t = *(const v4f*)(p + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * g + j] = t[j];
    }
}

template <int MT>
__global__ __launch_bounds__(SEG_THREADS, NWAVE / 4) void segmentp_kernel(SegmentPArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[P_LDS_BYTES];
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int q = wave & 3, half = wave >> 2;
    const int L = a.L;
    const int pbase = half * (MT * 32);
    // behind the two images: the per-channel constants (bias, scale, shift, bres: 512 floats) of the current layer and of the
    // next one, staged a layer ahead by waves 0-1 -- a constant costs an LDS read where it is needed, not an L2 round trip
    auto cbuf = [&](int l) { return (float*)(lds + 2 * P_IMG_BYTES + (l & 1) * 2048); };

    // zero both images and the tail once: the halo rows, the rows past the window and the bytes a last tile reads beyond its
    // image are never written with anything but zeros (or, in the tail, finite constants) afterwards (see the epilogue's mask)
    for (int i = tid0; i < P_LDS_BYTES / 16; i += SEG_THREADS) *(v4f*)(lds + (size_t)i * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // persistent walk: workgroups b and b + 8 share an XCD (round-robin dealing), so slice x = b & 7 of the rows -- whole sites --
    // belongs to the workgroups b = x, x + 8, ...: the reads of a site and the weights stay in one L2
    const int n_work = a.work_count ? *a.work_count : a.n_rows;
    const int slice = a.work_count ? (n_work + 7) / 8 : a.slice_rows;
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, nj = gridDim.x >> 3;
    auto row_of = [&](int k) {                                   // row index of this workgroup's k-th slice entry, -1 past the end
        const int wk = xcd * slice + k;
        if (k >= slice || wk >= n_work) return -1;
        return a.work_count ? a.work[wk] : wk;
    };
    const bool resumed = a.l_begin > 0;
    // a resumed segment's input: the read's bf16 image y, copied by LDS-DMA into the image that is free (each 1-KiB piece = 4
    // rows; the chunk swizzle goes on the per-lane SOURCE address, the destination is lane-linear), requested a row ahead
    auto dma_read = [&](int row_index, int img_i, int lane) {
        const char* ysrc = (const char*)(a.y + (size_t)row_index * (size_t)L * CPAD);
        char* img = lds + img_i * P_IMG_BYTES + P_HALO * P_ROW_BYTES;
        for (int kb = wave; kb * 4 < L; kb += NWAVE) {
            const int p = 4 * kb + (lane >> 4), r = P_HALO + p;
            if (p < L) glds16(ysrc + (size_t)p * P_ROW_BYTES + (((lane ^ r) & 15) << 4), img + kb * 1024);
        }
    };
    // ... and the seed of its first layer's accumulators: conv(pool) of the read's site (launch_conv_pool), the read-mean's share
    // of conv(y + pool) (model.py:742) -- the kernel convolves y alone
    v16f acc[MT];
    auto seed_request = [&](int row_index, int lane) {
        const int n = lane & 31, hh = lane >> 5;
        const float* cp = a.pool + (size_t)(row_index / a.R) * (size_t)L * CPAD + 32 * q + 16 * hh;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int p = pbase + 32 * m + n;
            if (p < L) load16(acc[m], cp + (size_t)p * CPAD);
            else acc[m] = (v16f)(0.f);
        }
    };
    int dma_img = 0;
    if (resumed) {
        const int r0 = __builtin_amdgcn_readfirstlane(row_of(jw));
        if (r0 >= 0) {
            dma_read(r0, 0, tid0 & 63);
            if (a.pool) seed_request(r0, tid0 & 63);
        }
    }

    for (int k = jw; k < slice; k += nj) {
        const int row_index = __builtin_amdgcn_readfirstlane(row_of(k));
        if (row_index < 0) break;
        const int next_row = __builtin_amdgcn_readfirstlane(row_of(k + nj));
        // (an opaque copy of the thread index per row: hipcc otherwise forms every per-lane address of the row body ahead of the
        // row loop and keeps them -- spilled -- through all of it)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int n = lane & 31, hh = lane >> 5;
        const int site = row_index / a.R;
        const size_t read_idx = (size_t)row_index;
        int cur = resumed ? dma_img : 0;                         // image holding the current layer input
        auto blk_of = [&](int l) { return a.wl + (size_t)l * WP_LAYER_BYTES; };
        // constants of layer l: requested by threads 0..127 (one 16-byte load each), put into cbuf(l) a stage later
        v4f creq;
        auto cst_request = [&](int l) { if (tid < 128) creq = *(const v4f*)((const float*)(blk_of(l) + WP_CST_OFF) + tid * 4); };
        auto cst_put = [&](int l) { if (tid < 128) *(v4f*)(cbuf(l) + tid * 4) = creq; };
        PSTAMP(0);
        cst_request(a.l_begin);
        const int c0 = 32 * q + 16 * hh;                          // this lane's 16 output channels
        const int row0 = P_HALO + pbase + n;                      // its row in tile 0
        const unsigned wa = cell_addr(row0, 4 * q + 2 * hh);      // its two output chunks: wa, wa ^ 16
        bf8 pre_a[4];
        load_first(pre_a, (gbf8p)(blk_of(a.l_begin) + WP_CONV_OFF) + q * 64 + lane);
        if (!resumed) {
            char* img = lds;                                     // image 0
            // ---- encode (dl4vc/model.py:450-627), canonical 48-channel order, rounded to bf16
            const size_t rbase = read_idx * (size_t)L, sbase = (size_t)site * L;
            int ok_ref = 1, ok_var = 1;
            for (int p = tid; p < L; p += SEG_THREADS) {
                const int tok = a.reads[rbase + p], rm = a.ref_mask[sbase + p], vm = a.var_mask[sbase + p];
                ok_ref &= (rm == 0) || (tok == rm);
                ok_var &= (vm == 0) || (tok == vm);
            }
            // workgroup-wide AND through sixteen flag words (__syncthreads_and would add its own LDS on top of the 160 KiB this
            // kernel declares); they sit in the constants buffer that is not in use at a row's start
            int* flags = (int*)cbuf(a.l_begin + 1);
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
                for (int c = 0; c < CIN0 / 8; ++c) {             // (layer 1's walk reads these six chunks of a row only)
                    bf8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (__bf16)row[c * 8 + j];
                    lds_write(img, cell_addr(r, c), v);
                }
            }
        }
        cst_put(a.l_begin);
        if (resumed) __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0): this wave's pieces of the DMA'd image have landed
        __syncthreads();
        PSTAMP(1);

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
        if (a.tap && a.tap_layer == 0 && !resumed) copy_tap(0, CIN0);

        for (int l = a.l_begin; l < a.l_end; ++l) {
            const char* blk = blk_of(l);
            const float* lc = cbuf(l);
            const bool residual = (a.res_mask >> l) & 1u;
            const bool last_layer = l + 1 == a.l_end;
            // layer l-1's bottleneck runs behind this layer's GEMM on the four older waves, in their wait at the barrier
            const bool defer = a.has_hw && l > a.l_begin && wave < NWAVE / 2;
            const int dil = (l == 0) ? 1 : (l + 1 < a.n_layers ? a.dil_mid : a.dil_final);
            const char* src = lds + cur * P_IMG_BYTES;
            char* dst = lds + (cur ^ 1) * P_IMG_BYTES;
            [[maybe_unused]] const int sb = 2 + (l - a.l_begin) * 8;
            PSTAMP(sb + 0);
            PFENCE();
            if (!last_layer) cst_request(l + 1);
            // the deferred bottleneck's operands (layer l-1, older waves): requested here, they arrive under the conv GEMM
            bf8 wbq[P_KSC];
            v16f bbq;
            auto bottleneck_request = [&](int lb) {
                gbf8p wbot = (gbf8p)(blk_of(lb) + WP_BOT_OFF) + lane;
#pragma unroll
                for (int ks = 0; ks < P_KSC; ++ks) wbq[ks] = wbot[ks * 64];
                load16(bbq, (const float*)(blk_of(lb) + WP_CST_OFF) + CST_BBOT + 16 * hh);
            };
            if (defer) bottleneck_request(l - 1);

            {
                v16f bias;
                lds16(bias, lc + CST_BIAS + c0);
                if (l == a.l_begin && resumed && a.pool) {       // seeded with conv(pool) of the site (requested a row ahead)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] += bias;
                } else {
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = bias;
                }
            }
            const unsigned xb0 = cell_addr(row0 - dil, hh), xb1 = cell_addr(row0, hh), xb2 = cell_addr(row0 + dil, hh);
            gbf8p wconv = (gbf8p)(blk + WP_CONV_OFF) + q * 64 + lane;
            gemm_p<MT, 3>(acc, src, xb0, xb1, xb2, wconv, pre_a, l == 0);
            PFENCE();
            PSTAMP(sb + 1);

            if (defer) bottleneck_ps<MT, NWAVE / 2>(src, wbq, bbq, a.h + (size_t)(l - 1) * a.h_layer_stride + read_idx * (size_t)L * HPAD, L, wave, lane);
            PFENCE();
            PSTAMP(sb + 7);
            // ---- epilogue: ReLU, BatchNorm (folded), rows past the window forced to zero, bf16, two 16-byte stores per tile
            {
                v16f sc, sh;
                lds16(sc, lc + CST_SCALE + c0);
                lds16(sh, lc + CST_SHIFT + c0);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int p = pbase + 32 * m + n;
                    v16f v = acc[m];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = relu1(v[i]);
                    scale_shift16(v, sc, sh);
                    if (pbase + 32 * m + 32 > L) {               // (uniform) the tile reaches past the window
#pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] = (p < L) ? v[i] : 0.f;
                    }
                    bf8 lo, hi;
                    pack16(v, lo, hi);
                    if (p < P_LMAX + P_HALO) {
                        lds_write(dst, wa + m * (32 * P_ROW_BYTES), lo);
                        lds_write(dst, (wa ^ 16u) + m * (32 * P_ROW_BYTES), hi);
                    }
                }
            }
            // the first fragments of the next GEMM (this layer's residual 1x1, else the next layer's conv) ride under the barrier
            // wait; the next layer's constants go to their buffer
            {
                const char* nb = residual ? blk : blk_of(last_layer ? l : l + 1);
                load_first(pre_a, (gbf8p)(nb + (residual ? WP_RES_OFF : WP_CONV_OFF)) + q * 64 + lane);
            }
            if (!last_layer) cst_put(l + 1);
            if (a.has_hw && last_layer && !residual) bottleneck_request(l);   // (the segment's last bottleneck: a stage of its own below)
            PFENCE();
            PSTAMP(sb + 2);
            __syncthreads();
            PFENCE();
            PSTAMP(sb + 3);
            if (residual) {
                // y = Wr * t + bres + x   (model.py:753-761): t = the image just written, x = this lane's own cells of the input image
                {
                    v16f br;
                    lds16(br, lc + CST_BRES + c0);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const bf8 xl = lds_read(src, wa + m * (32 * P_ROW_BYTES)), xh = lds_read(src, (wa ^ 16u) + m * (32 * P_ROW_BYTES));
#pragma unroll
                        for (int j = 0; j < 8; ++j) { acc[m][j] = (float)xl[j] + br[j]; acc[m][8 + j] = (float)xh[j] + br[8 + j]; }
                    }
                }
                gbf8p wres = (gbf8p)(blk + WP_RES_OFF) + q * 64 + lane;
                PFENCE();
                gemm_p<MT, 1>(acc, dst, xb1, xb1, xb1, wres, pre_a);
                PFENCE();
                PSTAMP(sb + 4);
                load_first(pre_a, (gbf8p)(blk_of(last_layer ? l : l + 1) + WP_CONV_OFF) + q * 64 + lane);
                if (a.has_hw && last_layer) bottleneck_request(l);
                char* back = lds + cur * P_IMG_BYTES;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int p = pbase + 32 * m + n;
                    v16f v = acc[m];
                    if (pbase + 32 * m + 32 > L) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] = (p < L) ? v[i] : 0.f;
                    }
                    bf8 lo, hi;
                    pack16(v, lo, hi);
                    if (p < P_LMAX + P_HALO) {
                        lds_write(back, wa + m * (32 * P_ROW_BYTES), lo);
                        lds_write(back, (wa ^ 16u) + m * (32 * P_ROW_BYTES), hi);
                    }
                }
                __syncthreads();
                PFENCE();
                PSTAMP(sb + 5);
            } else {
                cur ^= 1;
            }
            if (a.tap && a.tap_layer == l + 1) copy_tap(cur, CPAD);
            if (last_layer) {
                // ---- the segment's last layer has nobody to defer to: its bottleneck is a stage of its own, all eight waves (its
                // weights were requested ahead of the barrier above).  First (a resumed segment) the next row's image is requested
                // into the image that is now free: nothing the bottleneck waits for is behind it in the queue.
                // Every load hipcc tracks is retired HERE (they were requested ahead of the barrier: no wait in practice), so that
                // it places no vmcnt wait inside the stage that runs under the DMA -- one there would wait for the DMA as well
                __builtin_amdgcn_s_waitcnt(0x0F70);
                if (resumed && next_row >= 0) { dma_img = cur ^ 1; dma_read(next_row, dma_img, lane); }
                if (a.has_hw)
                    bottleneck_ps<MT, NWAVE>(lds + cur * P_IMG_BYTES, wbq, bbq, a.h + (size_t)l * a.h_layer_stride + read_idx * (size_t)L * HPAD, L, wave, lane);
            }
            PSTAMP(sb + 6);
        }
        PSTAMP(62);
        // ---- the segment's output -> y (bf16, [L][CPAD]); behind it the next row's accumulator seed is requested
        {
            const char* img = lds + cur * P_IMG_BYTES;
            bf8* ydst = (bf8*)(a.y + read_idx * (size_t)L * CPAD);
            // every chunk of the thread read before the first is stored (the accumulators are dead: registers are free) -- as a
            // loop of read -> wait -> store the stage was ten LDS round trips long.  (Storing y from the last epilogue's registers
            // instead, with no copy-out stage at all, was tried: 25 spilled registers, reloads inside the conv GEMM, -6 %.  Round 5:
            // this block moved in FRONT of the last bottleneck, so that its stores drain under that stage: 56 spilled registers at
            // five tiles, 21.3 ms per launch against 18.96 in a same-box A/B -- it stays behind.)
            constexpr int NC = (P_LMAX * (CPAD / 8) + SEG_THREADS - 1) / SEG_THREADS;     // 10
            bf8 v[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int i = tid + k * SEG_THREADS;
                v[k] = lds_read(img, cell_addr(P_HALO + min(i >> 4, P_LMAX - 1), i & 15));
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int i = tid + k * SEG_THREADS;
                if (i < L * (CPAD / 8)) ydst[i] = v[k];
            }
        }
        if (resumed && a.pool && next_row >= 0) seed_request(next_row, lane);
        PSTAMP(63);
        __syncthreads();                                         // the next row re-uses both images
    }
}

// ================================================================================================
// Sixteen-wave form (dan_config.bf16_form = 1): the same two images, stages and barriers as segmentp_kernel, on FOUR waves per SIMD.
// v_mfma_f32_16x16x32_bf16: a 16-column tile costs a wave 4 accumulator registers per 16 channels, so a wave of
// (channel quarter q, position quarter) = 32 channels x PT tiles of 16 columns holds 8 PT accumulators (40 at PT = 5) and the
// whole wave fits 128 registers.  Why: the lockstep form's SIMDs idle ~25 % of the time with BOTH waves in a wait (L2 round
// trips of the bottleneck weights, LDS store drains, pipeline fills); four waves leave fewer such moments.  The instruction
// counts per SIMD are the same.  Weight rows are permuted (dan_capi.cpp::pack_fragr) so that a lane's 2 x 4 accumulators of a
// column are 8 CONSECUTIVE channels: one 16-byte LDS store per tile.
//   lane (n = lane & 15, g = lane >> 4):  B fragment = chunk 4 ks + g of row n (k = 32 ks + 8 g + j),
//   C: column n, rows 4 g + i of channel tile e  ->  channel 32 q + 8 g + 4 e + i.
// ================================================================================================
constexpr int R_WAVES = 16, R_THREADS = R_WAVES * 64;
constexpr int R_KS = CPAD / 32;               // 4 k-steps of 32 channels
constexpr int R_KS0 = (CIN0 + 31) / 32;       // 2 for layer 1's 48 encoded channels (the upper 16 of the second step read zeros)
typedef const __attribute__((address_space(1))) bf8* gbf8r;
__device__ __forceinline__ v4f mfma16r(bf8 a, bf8 b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// acc[e][t] += W(channel tile 2 q + e) x X(16 columns of tile t) over TAPS x 4 k-steps; channel-group-major walk (step = ks * TAPS + tap),
// fragment (step * 8 + channel tile) of the layer block; two steps of weights in flight, the next step's activations one read
// behind every second MFMA.
template <int PT, int TAPS>
__device__ __forceinline__ void gemm_r(v4f (&acc)[2][PT], const char* lds, unsigned xb0, unsigned xb1, unsigned xb2, gbf8r w,
                                       const bf8 (&first)[2][2], bool k_short = false) {
    // weight steps in flight: two, or one at five tiles (40 accumulators + two activation sets of 20 leave no room for 16 more
    // registers in a 128-register wave; a step is 10 MFMAs x 4 waves = ~640 cycles, about an L2 round trip)
    constexpr int NA = PT >= 5 ? 1 : 2, S = TAPS * R_KS;
    bf8 a[NA][2], b[2][PT];
#pragma unroll
    for (int j = 0; j < NA; ++j) { a[j][0] = first[j][0]; a[j][1] = first[j][1]; }
#pragma unroll
    for (int t = 0; t < PT; ++t) b[0][t] = lds_read(lds, xb0 + t * (16 * P_ROW_BYTES));
    __builtin_amdgcn_sched_barrier(0);                          // (see gemm_p: the up-front reads must not join the groups below)
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int sn = s + 1;
        const unsigned xt = (sn % TAPS == 0) ? xb0 : (sn % TAPS == 1) ? xb1 : xb2;
        const unsigned xa = xt ^ (unsigned)((sn / TAPS) << 6);
        if (TAPS == 3 && s == TAPS * R_KS0) {
            if (k_short) break;
            __builtin_amdgcn_sched_barrier(0);
        }
        bf8 an[2];
        if (NA == 1 && sn < S) { an[0] = w[(size_t)sn * 8 * 64]; an[1] = w[(size_t)sn * 8 * 64 + 64]; }   // (requested ahead of the step's MFMAs)
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            acc[0][t] = mfma16r(a[s % NA][0], b[s & 1][t], acc[0][t]);
            acc[1][t] = mfma16r(a[s % NA][1], b[s & 1][t], acc[1][t]);
            if (sn < S) b[sn & 1][t] = lds_read(lds, xa + t * (16 * P_ROW_BYTES));
        }
        if (NA == 1) {
            if (sn < S) { a[0][0] = an[0]; a[0][1] = an[1]; }
        } else if (s + NA < S) {
            a[s % NA][0] = w[(size_t)(s + NA) * 8 * 64];
            a[s % NA][1] = w[(size_t)(s + NA) * 8 * 64 + 64];
        }
        if (NA == 1 && sn < S) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (sn < S) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (NA != 1 && s + NA < S) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    }
}

__device__ __forceinline__ void load_first_r(bf8 (&f)[2][2], gbf8r w) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { f[j][0] = w[(size_t)j * 8 * 64]; f[j][1] = w[(size_t)j * 8 * 64 + 64]; }
}
__device__ __forceinline__ void lds8(float (&v)[8], const float* p) {        // 8 consecutive floats from LDS
    const v4f t0 = *(const v4f*)p, t1 = *(const v4f*)(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = t0[j]; v[4 + j] = t1[j]; }
}
__device__ __forceinline__ bf8 pack8(const float (&v)[8]) {
    bf8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
    return o;
}

// h = relu(Wb y + bb), 128 -> 32, 16-column tiles dealt over the sixteen waves (tile tl lies tl * 4 KiB into the image: 16 rows
// on the swizzle repeats); a lane's two channel tiles of a column are 8 consecutive outputs: one 16-byte store
template <int PT>
__device__ __forceinline__ void bottleneck_r(const char* img, const char* blk, uint16_t* hrow, int L, int wave, int lane) {
    constexpr int NT = 4 * PT, NTL = (NT + R_WAVES - 1) / R_WAVES;
    asm volatile("" : "+v"(lane));
    const int n = lane & 15, g = lane >> 4;
    // the weights are requested HERE, not a stage ahead as in the eight-wave form: 32 registers held through the epilogue do not
    // fit a 128-register wave (tried three ways: 72-177 spilled registers, reloaded one per MFMA)
    gbf8r wbot = (gbf8r)(blk + WP_BOT_OFF) + lane;
    bf8 wb[R_KS][2];
#pragma unroll
    for (int ks = 0; ks < R_KS; ++ks) { wb[ks][0] = wbot[(ks * 2) * 64]; wb[ks][1] = wbot[(ks * 2 + 1) * 64]; }
    const float* bp = (const float*)(blk + WP_CST_OFF) + CST_BBOT + 8 * g;
    const v4f bb0 = *(const v4f*)bp, bb1 = *(const v4f*)(bp + 4);
    const unsigned xa0 = cell_addr(P_HALO + n, g);
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int tl = wave + R_WAVES * i;
        if (tl >= NT) break;                                     // (wave-uniform)
        const char* tile = img + tl * (16 * P_ROW_BYTES);
        bf8 bx[R_KS];
#pragma unroll
        for (int ks = 0; ks < R_KS; ++ks) bx[ks] = lds_read(tile, xa0 ^ (unsigned)(ks << 6));
        v4f h0 = bb0, h1 = bb1;
#pragma unroll
        for (int ks = 0; ks < R_KS; ++ks) { h0 = mfma16r(wb[ks][0], bx[ks], h0); h1 = mfma16r(wb[ks][1], bx[ks], h1); }
        const int p = 16 * tl + n;
        if (p < L) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = relu1(h0[j]); v[4 + j] = relu1(h1[j]); }
            *(bf8*)(hrow + (size_t)p * HPAD + 8 * g) = pack8(v);
        }
    }
}

// ... the same with the weights requested by the caller (the deferred bottleneck of the layer before: see the epilogue)
template <int PT>
__device__ __forceinline__ void bottleneck_rw(const char* img, const bf8 (&wb)[R_KS][2], v4f bb0, v4f bb1, uint16_t* hrow, int L,
                                              int wave, int lane) {
    constexpr int NT = 4 * PT, NTL = (NT + R_WAVES - 1) / R_WAVES;
    asm volatile("" : "+v"(lane));
    const int n = lane & 15, g = lane >> 4;
    const unsigned xa0 = cell_addr(P_HALO + n, g);
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int tl = wave + R_WAVES * i;
        if (tl >= NT) break;
        const char* tile = img + tl * (16 * P_ROW_BYTES);
        bf8 bx[R_KS];
#pragma unroll
        for (int ks = 0; ks < R_KS; ++ks) bx[ks] = lds_read(tile, xa0 ^ (unsigned)(ks << 6));
        v4f h0 = bb0, h1 = bb1;
#pragma unroll
        for (int ks = 0; ks < R_KS; ++ks) { h0 = mfma16r(wb[ks][0], bx[ks], h0); h1 = mfma16r(wb[ks][1], bx[ks], h1); }
        const int p = 16 * tl + n;
        if (p < L) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = relu1(h0[j]); v[4 + j] = relu1(h1[j]); }
            *(bf8*)(hrow + (size_t)p * HPAD + 8 * g) = pack8(v);
        }
    }
}

template <int PT>
__global__ __launch_bounds__(R_THREADS, 1) void segmentr_kernel(SegmentPArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[P_LDS_BYTES];
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int q = wave & 3, pq = wave >> 2;
    const int L = a.L;
    const int pbase = pq * (PT * 16);
    auto cbuf = [&](int l) { return (float*)(lds + 2 * P_IMG_BYTES + (l & 1) * 2048); };
    for (int i = tid0; i < P_LDS_BYTES / 16; i += R_THREADS) *(v4f*)(lds + (size_t)i * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const int n_work = a.work_count ? *a.work_count : a.n_rows;
    const int slice = a.work_count ? (n_work + 7) / 8 : a.slice_rows;
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, nj = gridDim.x >> 3;
    auto row_of = [&](int k) {
        const int wk = xcd * slice + k;
        if (k >= slice || wk >= n_work) return -1;
        return a.work_count ? a.work[wk] : wk;
    };
    const bool resumed = a.l_begin > 0;
    auto dma_read = [&](int row_index, int img_i, int lane) {
        const char* ysrc = (const char*)(a.y + (size_t)row_index * (size_t)L * CPAD);
        char* img = lds + img_i * P_IMG_BYTES + P_HALO * P_ROW_BYTES;
        for (int kb = wave; kb * 4 < L; kb += R_WAVES) {
            const int p = 4 * kb + (lane >> 4), r = P_HALO + p;
            if (p < L) glds16(ysrc + (size_t)p * P_ROW_BYTES + (((lane ^ r) & 15) << 4), img + kb * 1024);
        }
    };
    v4f acc[2][PT];
    auto seed_request = [&](int row_index, int lane) {
        const int n = lane & 15, g = lane >> 4;
        const float* cp = a.pool + (size_t)(row_index / a.R) * (size_t)L * CPAD + 32 * q + 8 * g;
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int p = pbase + 16 * t + n;
            if (p < L) { acc[0][t] = *(const v4f*)(cp + (size_t)p * CPAD); acc[1][t] = *(const v4f*)(cp + (size_t)p * CPAD + 4); }
            else { acc[0][t] = (v4f){0.f, 0.f, 0.f, 0.f}; acc[1][t] = (v4f){0.f, 0.f, 0.f, 0.f}; }
        }
    };
    int dma_img = 0;
    if (resumed) {
        const int r0 = __builtin_amdgcn_readfirstlane(row_of(jw));
        if (r0 >= 0) {
            dma_read(r0, 0, tid0 & 63);
            if (a.pool) seed_request(r0, tid0 & 63);
        }
    }
    for (int k = jw; k < slice; k += nj) {
        const int row_index = __builtin_amdgcn_readfirstlane(row_of(k));
        if (row_index < 0) break;
        const int next_row = __builtin_amdgcn_readfirstlane(row_of(k + nj));
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int n = lane & 15, g = lane >> 4;
        const int site = row_index / a.R;
        const size_t read_idx = (size_t)row_index;
        int cur = resumed ? dma_img : 0;
        auto blk_of = [&](int l) { return a.wlr + (size_t)l * WP_LAYER_BYTES; };
        v4f creq;
        auto cst_request = [&](int l) { if (tid < 128) creq = *(const v4f*)((const float*)(blk_of(l) + WP_CST_OFF) + tid * 4); };
        auto cst_put = [&](int l) { if (tid < 128) *(v4f*)(cbuf(l) + tid * 4) = creq; };
        RSTAMP(0);
        cst_request(a.l_begin);
        const int c0 = 32 * q + 8 * g;                            // this lane's 8 output channels
        const int row0 = P_HALO + pbase + n;                      // its row in tile 0
        const unsigned wa = cell_addr(row0, 4 * q + g);           // its output chunk
        bf8 pre_a[2][2];
        load_first_r(pre_a, (gbf8r)(blk_of(a.l_begin) + WP_CONV_OFF) + 2 * q * 64 + lane);
        if (!resumed) {
            char* img = lds;
            const size_t rbase = read_idx * (size_t)L, sbase = (size_t)site * L;
            int ok_ref = 1, ok_var = 1;
            for (int p = tid; p < L; p += R_THREADS) {
                const int tok = a.reads[rbase + p], rm = a.ref_mask[sbase + p], vm = a.var_mask[sbase + p];
                ok_ref &= (rm == 0) || (tok == rm);
                ok_var &= (vm == 0) || (tok == vm);
            }
            int* flags = (int*)cbuf(a.l_begin + 1);
            {
                const int w_ref = __all(ok_ref), w_var = __all(ok_var);
                if (lane == 0) { flags[wave] = w_ref; flags[R_WAVES + wave] = w_var; }
            }
            __syncthreads();
            int agree_ref = 1, agree_var = 1;
#pragma unroll
            for (int w8 = 0; w8 < R_WAVES; ++w8) { agree_ref &= flags[w8]; agree_var &= flags[R_WAVES + w8]; }
            for (int p = tid; p < L; p += R_THREADS) {
                const int tok = a.reads[rbase + p], qv = a.qual[rbase + p], st = a.strand[rbase + p];
                const int rf = a.ref[sbase + p], rm = a.ref_mask[sbase + p], vm = a.var_mask[sbase + p];
                const float* er = a.emb + min(tok, VOCAB - 1) * EMBED;
                const float* ef = a.emb + min(rf, VOCAB - 1) * EMBED;
                const float* pp = a.pe + p * EMBED;
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
                    bf8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (__bf16)row[c * 8 + j];
                    lds_write(img, cell_addr(r, c), v);
                }
            }
        }
        cst_put(a.l_begin);
        if (resumed) __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        RSTAMP(1);
        auto copy_tap = [&](int img_i, int nch) {
            float* dst = a.tap + read_idx * (size_t)L * CPAD;
            const char* img = lds + img_i * P_IMG_BYTES;
            for (int i = tid; i < L * (CPAD / 8); i += R_THREADS) {
                const int p = i >> 4, c = i & 15;
                const bf8 v = lds_read(img, cell_addr(P_HALO + p, c));
                v4f o0, o1;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o0[j] = (c * 8 + j < nch) ? (float)v[j] : 0.f; o1[j] = (c * 8 + 4 + j < nch) ? (float)v[4 + j] : 0.f; }
                *(v4f*)(dst + (size_t)i * 8) = o0;
                *(v4f*)(dst + (size_t)i * 8 + 4) = o1;
            }
        };
        if (a.tap && a.tap_layer == 0 && !resumed) copy_tap(0, CIN0);

        for (int l = a.l_begin; l < a.l_end; ++l) {
            const char* blk = blk_of(l);
            const float* lc = cbuf(l);
            const bool residual = (a.res_mask >> l) & 1u;
            const bool last_layer = l + 1 == a.l_end;
            const bool defer = a.has_hw && l > a.l_begin;
            const int dil = (l == 0) ? 1 : (l + 1 < a.n_layers ? a.dil_mid : a.dil_final);
            const char* src = lds + cur * P_IMG_BYTES;
            char* dst = lds + (cur ^ 1) * P_IMG_BYTES;
            [[maybe_unused]] const int sb = 2 + (l - a.l_begin) * 8;
            RSTAMP(sb + 0);
            PFENCE();
            if (!last_layer) cst_request(l + 1);
            {
                float bias[8];
                lds8(bias, lc + CST_BIAS + c0);
                const v4f b0 = {bias[0], bias[1], bias[2], bias[3]}, b1 = {bias[4], bias[5], bias[6], bias[7]};
                if (l == a.l_begin && resumed && a.pool) {
#pragma unroll
                    for (int t = 0; t < PT; ++t) { acc[0][t] += b0; acc[1][t] += b1; }
                } else {
#pragma unroll
                    for (int t = 0; t < PT; ++t) { acc[0][t] = b0; acc[1][t] = b1; }
                }
            }
            const unsigned xb0 = cell_addr(row0 - dil, g), xb1 = cell_addr(row0, g), xb2 = cell_addr(row0 + dil, g);
            gbf8r wconv = (gbf8r)(blk + WP_CONV_OFF) + 2 * q * 64 + lane;
            gemm_r<PT, 3>(acc, src, xb0, xb1, xb2, wconv, pre_a, l == 0);
            PFENCE();
            RSTAMP(sb + 1);
            auto store_tile = [&](char* img, int t, const float (&v)[8]) {
                const int p = pbase + 16 * t + n;
                bf8 o = pack8(v);
                if (pbase + 16 * t + 16 > L) {                    // (uniform) the tile reaches past the window
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    u4 w4 = __builtin_bit_cast(u4, o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) w4[j] = (p < L) ? w4[j] : 0u;
                    o = __builtin_bit_cast(bf8, w4);
                }
                if (p < P_LMAX + P_HALO) lds_write(img, wa + t * (16 * P_ROW_BYTES), o);
            };
            bf8 wb[R_KS][2];
            v4f bb0, bb1;
            {
                float sc[8], sh[8];
                lds8(sc, lc + CST_SCALE + c0);
                lds8(sh, lc + CST_SHIFT + c0);
#pragma unroll
                for (int t = 0; t < PT; ++t) {
                    // the deferred bottleneck's weights (layer l-1; for the segment's first layer a harmless request of its own):
                    // requested UNCONDITIONALLY (a conditional definition would keep 32 registers live through the GEMM) and only
                    // once most accumulators are stored -- beside all of them they do not fit a 128-register wave
                    if (t == (PT >= 4 ? PT - 2 : PT - 1)) {
                        gbf8r wbot = (gbf8r)(blk_of(l > a.l_begin ? l - 1 : l) + WP_BOT_OFF) + lane;
#pragma unroll
                        for (int ks = 0; ks < R_KS; ++ks) { wb[ks][0] = wbot[(ks * 2) * 64]; wb[ks][1] = wbot[(ks * 2 + 1) * 64]; }
                        const float* bp = (const float*)(blk_of(l > a.l_begin ? l - 1 : l) + WP_CST_OFF) + CST_BBOT + 8 * g;
                        bb0 = *(const v4f*)bp; bb1 = *(const v4f*)(bp + 4);
                    }
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = relu1(acc[0][t][j]); v[4 + j] = relu1(acc[1][t][j]); }
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const v2f r2 = __builtin_elementwise_fma((v2f){v[j], v[j + 1]}, (v2f){sc[j], sc[j + 1]}, (v2f){sh[j], sh[j + 1]});
                        v[j] = r2[0]; v[j + 1] = r2[1];
                    }
                    store_tile(dst, t, v);
                }
            }
            {
                const char* nb = residual ? blk : blk_of(last_layer ? l : l + 1);
                load_first_r(pre_a, (gbf8r)(nb + (residual ? WP_RES_OFF : WP_CONV_OFF)) + 2 * q * 64 + lane);
            }
            if (!last_layer) cst_put(l + 1);
            PFENCE();
            RSTAMP(sb + 2);
            if (defer) bottleneck_rw<PT>(src, wb, bb0, bb1, a.h + (size_t)(l - 1) * a.h_layer_stride + read_idx * (size_t)L * HPAD, L, wave, lane);
            RSTAMP(sb + 7);
            __syncthreads();
            PFENCE();
            RSTAMP(sb + 3);
            if (residual) {
                {
                    float br[8];
                    lds8(br, lc + CST_BRES + c0);
#pragma unroll
                    for (int t = 0; t < PT; ++t) {
                        const bf8 x = lds_read(src, wa + t * (16 * P_ROW_BYTES));
#pragma unroll
                        for (int j = 0; j < 4; ++j) { acc[0][t][j] = (float)x[j] + br[j]; acc[1][t][j] = (float)x[4 + j] + br[4 + j]; }
                    }
                }
                gbf8r wres = (gbf8r)(blk + WP_RES_OFF) + 2 * q * 64 + lane;
                PFENCE();
                gemm_r<PT, 1>(acc, dst, xb1, xb1, xb1, wres, pre_a);
                PFENCE();
                RSTAMP(sb + 4);
                load_first_r(pre_a, (gbf8r)(blk_of(last_layer ? l : l + 1) + WP_CONV_OFF) + 2 * q * 64 + lane);
                char* back = lds + cur * P_IMG_BYTES;
#pragma unroll
                for (int t = 0; t < PT; ++t) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = acc[0][t][j]; v[4 + j] = acc[1][t][j]; }
                    store_tile(back, t, v);
                }
                __syncthreads();
                PFENCE();
                RSTAMP(sb + 5);
            } else {
                cur ^= 1;
            }
            if (a.tap && a.tap_layer == l + 1) copy_tap(cur, CPAD);
            if (last_layer) {
                __builtin_amdgcn_s_waitcnt(0x0F70);
                if (resumed && next_row >= 0) { dma_img = cur ^ 1; dma_read(next_row, dma_img, lane); }
                if (a.has_hw)
                    bottleneck_r<PT>(lds + cur * P_IMG_BYTES, blk, a.h + (size_t)l * a.h_layer_stride + read_idx * (size_t)L * HPAD, L, wave, lane);
            }
            RSTAMP(sb + 6);
        }
        RSTAMP(62);
        {
            const char* img = lds + cur * P_IMG_BYTES;
            bf8* ydst = (bf8*)(a.y + read_idx * (size_t)L * CPAD);
            constexpr int NC = (P_LMAX * (CPAD / 8) + R_THREADS - 1) / R_THREADS;         // 5
            bf8 v[NC];
#pragma unroll
            for (int k2 = 0; k2 < NC; ++k2) {
                const int i = tid + k2 * R_THREADS;
                v[k2] = lds_read(img, cell_addr(P_HALO + min(i >> 4, P_LMAX - 1), i & 15));
            }
#pragma unroll
            for (int k2 = 0; k2 < NC; ++k2) {
                const int i = tid + k2 * R_THREADS;
                if (i < L * (CPAD / 8)) ydst[i] = v[k2];
            }
        }
        if (resumed && a.pool && next_row >= 0) seed_request(next_row, lane);
        RSTAMP(63);
        __syncthreads();
    }
}

void launch_segmentp(const SegmentPArgs& a0, int n_sites, int n_cus, hipStream_t s) {
    SegmentPArgs a = a0;
    a.n_rows = n_sites * a.R;
    a.slice_rows = (n_sites + 7) / 8 * a.R;
    int wgs = n_cus > 0 ? n_cus : 256;
    wgs = (wgs + 7) / 8 * 8;
    const int need = ((a.slice_rows + 0) < 1 ? 1 : a.slice_rows) * 8;       // no more workgroups than rows per slice x 8
    if (wgs > need) wgs = need;
    // form 1: the sixteen-wave form (16x16x32 tiles, four waves per SIMD) -- an independently written second implementation held to
    // the same layer-by-layer oracle tests; measured 7 % slower end to end (the chip clocks the denser form lower: HISTORY.md 11.1).
    // (Round 3 also carried a staggered form -- position halves one phase apart -- measured 12 % slower and deleted in round 4: a vector
    // instruction of the SIMD's other wave delays an MFMA walk by its full issue time, so an epilogue beside a GEMM costs what it costs
    // behind it, and that form paid 20 % of redundant tiles on top.)
    if (a.form == 1) {
        if (a.L <= 4 * 3 * 16) hipLaunchKernelGGL((segmentr_kernel<3>), dim3((unsigned)wgs), dim3(R_THREADS), 0, s, a);
        else if (a.L <= 4 * 4 * 16) hipLaunchKernelGGL((segmentr_kernel<4>), dim3((unsigned)wgs), dim3(R_THREADS), 0, s, a);
        else hipLaunchKernelGGL((segmentr_kernel<5>), dim3((unsigned)wgs), dim3(R_THREADS), 0, s, a);
    }
    // each position half owns MT tiles of 32 columns: the narrowest tiling that covers the window (201 columns on 2 x 5 tiles
    // would spend 37 % of the MFMAs past column 201; 2 x 4 tiles 22 %)
    else if (a.L <= 2 * 3 * 32)
        hipLaunchKernelGGL((segmentp_kernel<3>), dim3((unsigned)wgs), dim3(SEG_THREADS), 0, s, a);
    else if (a.L <= 2 * 4 * 32)
        hipLaunchKernelGGL((segmentp_kernel<4>), dim3((unsigned)wgs), dim3(SEG_THREADS), 0, s, a);
    else
        hipLaunchKernelGGL((segmentp_kernel<5>), dim3((unsigned)wgs), dim3(SEG_THREADS), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// conv(pool): the read-mean's share of the layer after a pool layer, once per site (model.py:742 adds the mean to every read
// before the convolution; conv(y + pool) = conv(y) + conv(pool), and the second term is the same for all reads of a site).
// cols[site * L + p][t * 128 + c] = pool[site][p + (t - 1) dil][c] (zero outside the window); the product with the layer's
// bf16-rounded weights [128][384] runs on the fp32 FC GEMM (dan_kernels.hip).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_cols_kernel(const v4f* __restrict__ pool, v4f* __restrict__ cols, int L, int dil, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)(i % 32), t = (int)((i / 32) % 3);
    const long long sp = i / 96;                               // site * L + p
    const int p = (int)(sp % L), ps = p + (t - 1) * dil;
    cols[i] = (ps >= 0 && ps < L) ? pool[(sp - p + ps) * 32 + c4] : (v4f){0.f, 0.f, 0.f, 0.f};
}

void launch_conv_pool(const float* pool, const float* wpool, const float* zero_bias, float* cols, float* cp, int n_sites, int L,
                      int dil, hipStream_t s) {
    const long long n4 = (long long)n_sites * L * 96;
    hipLaunchKernelGGL(pool_cols_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const v4f*)pool, (v4f*)cols, L, dil, n4);
    launch_fc(cols, 3 * CPAD, wpool, 3 * CPAD, zero_bias, cp, CPAD, n_sites * L, CPAD, 3 * CPAD, 0, s);
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

// highway compression from bf16 h on the bf16 matrix cores (model.py:776-777,859).  A k-group = ONE position (32 channels) = one
// v_mfma_f32_16x16x32_bf16 per (read tile, output tile, weight plane): a lane's 16-byte load of h IS its A fragment (read
// lane & 15, channels 8 (lane >> 4) ..), the compression weights come as two bf16 planes (hi = bf16(w), lo = bf16(w - hi): 16
// mantissa bits, products exact, fp32 sums).
// Round 5 form: the eight waves of a workgroup walk the SAME positions on different reads (wave = 16 reads x all positions: no
// cross-wave sum, no 64-KiB partial buffer), so a position's four weight fragments (4 KiB) are fetched ONCE per workgroup -- wave w
// loads those of position 8 ph + w -- into a two-phase LDS ring and read from there by all eight waves; 32 positions of h in flight
// per wave.  Why (tools/highway_probe.hip, 128 x 301): rounds 3-4's form -- eight waves splitting the positions of 64 reads, every
// wave streaming its own weights from L2, as many bytes as its h -- read h at 3.92 TB/s; without the weight loads 5.11, without the
// MFMAs 3.93 (they are free), with coalesced kilobyte runs instead of 16 rows per instruction 5.3 (+4 %): the weight stream was the
// bound.  This form: 5.6 TB/s.
constexpr int HW16_DH = 32;                                   // positions of h in flight per wave (a multiple of 8)
typedef __bf16 hbf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v4f mfma16b(hbf8 a, hbf8 b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

__global__ __launch_bounds__(512) void highway16_kernel(const uint16_t* __restrict__ h, long long hls, const hbf8* __restrict__ wc,
                                                        long long wcls, const float* __restrict__ bc, float* __restrict__ feat,
                                                        long long fs, int feat_off, int n_rows, int R, int L, int H,
                                                        const int* __restrict__ row_src) {
    __shared__ __attribute__((aligned(16))) char ring[2][8][4][1024];      // [phase parity][position of the phase][fragment][lane * 16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int layer = blockIdx.y;
    const int row0 = blockIdx.x * 128 + wave * 16;
    const size_t K = (size_t)L * HPAD;
    const int row = min(row0 + r16, n_rows - 1);
    const uint16_t* arow = h + (size_t)layer * hls + (size_t)(row_src ? row_src[row] : row) * K + kk * 8;   // skipped rows: their source's h
    const hbf8* wl = wc + (size_t)layer * wcls + lane;       // [pos][n 2][plane 2][lane 64]
    v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const int n_ph = (L + 7) >> 3;
    hbf8 ar[HW16_DH];
#pragma unroll
    for (int d = 0; d < HW16_DH; ++d) ar[d] = *(const hbf8*)(arow + (size_t)min(d, L - 1) * HPAD);
    hbf8 wq[4];
    {
        const int p = min(wave, L - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[j] = wl[((size_t)p * 4 + j) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) *(hbf8*)(&ring[0][wave][j][lane * 16]) = wq[j];
    }
    __syncthreads();
    for (int ph0 = 0; ph0 < n_ph; ph0 += HW16_DH / 8) {
#pragma unroll
        for (int q = 0; q < HW16_DH / 8; ++q) {                  // (unrolled: the slot of ar a position lives in is a compile-time index)
            const int ph = ph0 + q;
            if (ph < n_ph) {                                     // wave-uniform, and the same for every wave: the barrier below is reached by all
                if (ph + 1 < n_ph) {
                    const int pn = min(8 * (ph + 1) + wave, L - 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) wq[j] = wl[((size_t)pn * 4 + j) * 64];
                }
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const int p = 8 * ph + d;
                    const hbf8 av = ar[q * 8 + d];
                    ar[q * 8 + d] = *(const hbf8*)(arow + (size_t)min(p + HW16_DH, L - 1) * HPAD);
                    if (p < L) {
                        const char* wr = &ring[ph & 1][d][0][lane * 16];
                        const hbf8 w0 = *(const hbf8*)wr, w1 = *(const hbf8*)(wr + 1024), w2 = *(const hbf8*)(wr + 2048), w3 = *(const hbf8*)(wr + 3072);
                        acc0 = mfma16b(av, w0, acc0);
                        acc0 = mfma16b(av, w1, acc0);
                        acc1 = mfma16b(av, w2, acc1);
                        acc1 = mfma16b(av, w3, acc1);
                    }
                }
                if (ph + 1 < n_ph) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *(hbf8*)(&ring[(ph + 1) & 1][wave][j][lane * 16]) = wq[j];
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

void launch_highway16(const uint16_t* h, long long hls, const float* wc16, long long wcls, const float* bc, float* feat, long long fs,
                      int feat_off, int n_sites, int R, int L, int H, int layers, const int* row_src, hipStream_t s) {
    const int n_rows = n_sites * R;
    hipLaunchKernelGGL(highway16_kernel, dim3((n_rows + 127) / 128, layers), dim3(512), 0, s, h, hls,
                       (const hbf8*)wc16, wcls / 4, bc, feat, fs, feat_off, n_rows, R, L, H, row_src);
}

}  // namespace dan
