// HIP kernels of ONE TRAINING STEP of the DAN network for gfx950 (MI355X).  fp32, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Reference semantics (file:line in /root/reference): train-mode forward dl4vc/model.py:434-961 under model.train()
// (trainer.py:69): BatchNorm on batch statistics (model.py:749-751), dropout in conv2hidden (model.py:369-377); losses
// trainer.py:82-96,134-172,221-224,309-313,425-427 and objectives.py:49-112; clip + Adam trainer.py:435-439, main.py:116;
// embedding gradient rules model.py:143-145 (padding_idx, scale_grad_by_freq).
//
// Layout and schedule: see dan_train.h.  One workgroup (8 waves) = one read resident in LDS for ONE layer-shaped step
// (train_row_kernel; the 3-tap launches of the direct form run on half reads, two workgroups per CU: train_rowh_kernel): the convolution, its transpose (data gradient), the 1x1 residual and bottleneck GEMMs and their
// transposes are all the same implicit GEMM over the LDS image (conv_gemm of dan_device.h) with weights re-packed on the
// device every step.  Weight gradients contract over positions (train_wgrad_kernel): persistent workgroups, both operands
// staged per half read, split-K partials reduced in a fixed order.  Every reduction is deterministic (no atomics).
#include "dan_device.h"
#include "dan_train.h"
#include <type_traits>

#include <algorithm>

namespace dan {

// ------------------------------------------------------------------------------------------------
// The token planes of a batch (dl4vc/model.py:450-627).  Nothing in the training step builds the encoded 48-channel input any more:
// layer 1 runs forward from tables and backward as binned sums (the end of this file); canonical channel order of its weights
//   [read emb+pe (20) | ref emb+pe (20) | q*0.01 | strand*0.5 | refmatch | varmatch | lenmask | 0 0 0]
// ------------------------------------------------------------------------------------------------
struct EncodeSrc {
    const uint8_t *reads, *qual, *strand, *ref, *ref_mask, *var_mask;
    const float *emb, *pe;
};

// The same with the three constants of the thread's channels already in registers.  Every staging loop below walks positions
// with a stride that is a multiple of the row's vector count, so a thread meets the SAME four channels in every iteration:
// fetched per element (three 16-byte loads beside every 16-byte row load) the constants cost the half-read conv launch 60 us
// of its 1.1 ms.
struct Coef3 { v4f A, B, C; };
__device__ __forceinline__ Coef3 load_coef(const float* coef, int c) {
    Coef3 k{splat(1.f), splat(0.f), splat(0.f)};
    if (coef) { k.A = *(const v4f*)(coef + c); k.B = *(const v4f*)(coef + CPAD + c); k.C = *(const v4f*)(coef + 2 * CPAD + c); }
    return k;
}
__device__ __forceinline__ v4f apply_transform(v4f s1, v4f s2, const Coef3& k, bool has_coef, int mask) {
    v4f v = s1;
    if (has_coef) v = k.A * s1 + k.B * s2 + k.C;
    if (mask) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = s2[j] > 0.f ? v[j] : 0.f;
    }
    return v;
}

// v = A s1 + B s2 + C, then masked by s2 > 0
__device__ __forceinline__ v4f load_transform(v4f s1, v4f s2, const float* coef, int c, int mask) {
    v4f v = s1;
    if (coef) {
        const v4f A = *(const v4f*)(coef + c), B = *(const v4f*)(coef + CPAD + c), C = *(const v4f*)(coef + 2 * CPAD + c);
        v = A * s1 + B * s2 + C;
    }
    if (mask) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = s2[j] > 0.f ? v[j] : 0.f;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// T1: one layer-shaped step on the LDS-resident read
// ------------------------------------------------------------------------------------------------
constexpr int TR_LDS_ROWS = (WINO_ROWS_READ > LDS_ROWS ? WINO_ROWS_READ : LDS_ROWS);      // Winograd tiles past the window read a few rows on
__global__ __launch_bounds__(SEG_THREADS, NWAVE / 4) void train_row_kernel(RowArgs a, int n_rows) {
    __shared__ __attribute__((aligned(16))) float xs[TR_LDS_ROWS * LDS_S];
    __shared__ float sred[2][2][CPAD];
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int L = a.L;
    // PERSISTENT: workgroup b walks reads b, b + grid, ... (one workgroup per CU: the image is 119 KB) -- no workgroup launch and
    // no image clear per read, and the NEXT read's tensors (13 x 16 bytes per thread each) are requested behind this read's GEMM:
    // their round trip passes under the epilogue's stores and barriers instead of in front of idle matrix pipes.  (Requested
    // AHEAD of the GEMM they cost 148 spilled registers beside the Winograd accumulators.)  The image is cleared ONCE: a read's stores cover rows [HALO, HALO + L) x all 128 channels (encode: its 45 of the 48 that
    // layer 1 reads), the halo rows and the rows past the window are never written again.
    for (int i = tid0; i < TR_LDS_ROWS * LDS_S / 4; i += SEG_THREADS) ((v4f*)xs)[i] = splat(0.f);
    __syncthreads();
    constexpr int NP = (MPOS * (CPAD / 4) + SEG_THREADS - 1) / SEG_THREADS;     // 13
    const int vpr = a.s1_stride >> 2;                           // 16-byte vectors per position
    const int vsh = vpr == 32 ? 5 : 3;                          // (rows are CPAD or HPAD floats wide)
    const int n4 = L * vpr;
    v4f r1[NP], r2[NP];
    auto request = [&](int rw, int tid) {                        // a read's first and second tensor
        const v4f* s1 = (const v4f*)(a.src1 + (size_t)rw * L * a.s1_stride);
        const v4f* s2 = a.src2 ? (const v4f*)(a.src2 + (size_t)rw * L * a.s1_stride) : nullptr;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = tid + k * SEG_THREADS;
            const bool ok = i < n4;
            r1[k] = ok ? s1[i] : splat(0.f);
            r2[k] = (ok && s2) ? s2[i] : splat(0.f);
        }
    };
    if ((int)blockIdx.x < n_rows) request(blockIdx.x, tid0);
    for (int row = blockIdx.x; row < n_rows; row += gridDim.x) {
    // (an opaque copy of the thread index per read: hipcc otherwise forms every per-lane address of the body ahead of the loop
    // and keeps them -- spilled -- through the GEMMs)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int site = row / a.R;
    const int next = row + (int)gridDim.x;
    {
        const v4f* pl = a.pool_in ? (const v4f*)(a.pool_in + (size_t)site * L * CPAD) : nullptr;
        if (vpr < CPAD / 4) {                                    // (a narrower tensor: the columns it does not cover must read zero)
            for (int i = tid; i < TR_LDS_ROWS * LDS_S / 4; i += SEG_THREADS) ((v4f*)xs)[i] = splat(0.f);
            __syncthreads();
        }
        v4f r3[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = tid + k * SEG_THREADS;
            r3[k] = (i < n4 && pl) ? pl[i] : splat(0.f);
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = tid + k * SEG_THREADS;
            if (i < n4) {
                const int p = i >> vsh, c4 = i & (vpr - 1);          // (vpr is 32 or 8: a shift, not twenty instructions of integer division)
                *(v4f*)(xs + (HALO + p) * LDS_S + c4 * 4) = load_transform(r1[k], r2[k], a.coef, c4 * 4, a.mask_src2) + r3[k];
            }
        }
    }
    __syncthreads();

    const int pos = lane & 15, kk = lane >> 4;
    const size_t rbase = (size_t)row * L * CPAD, sbase = (size_t)site * L * CPAD;
    if (a.wino) {
        // ---- 3 taps at dilation 2 in Winograd F(2,3) form (conv_gemm_wino of dan_device.h, the inference kernel's core: 4
        // channel GEMMs per 2 outputs instead of 6).  Wave = one 16-channel output tile x all positions; MFMA column n of
        // tile step m is the Winograd tile with base P = wino_base(lane) + 4 m, outputs y(P), y(P + 2).
        const int wP0 = wino_base(lane);
        const int wlim = ((lane >> 2) & 1) ? MPOS : WHB;            // positions this lane's tiling owns
        const int chw = wave * 16 + kk * 4;
        gv4f_ptr w_w = (gv4f_ptr)(a.w1) + wave * 64 + lane;
        constexpr size_t KS = (size_t)KGC * (KGC * 64);
        const v4f pre_w[4] = {w_w[0], w_w[KS], w_w[2 * KS], w_w[3 * KS]};
        v4f acc4[MW][4];
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc4[m][k] = splat(0.f);
        conv_gemm_wino(acc4, xs + (HALO + wP0 - 2) * LDS_S + kk * 4, w_w, pre_w);
        // the next read's tensors travel under this read's epilogue (always redefined -- the last read re-requests itself --
        // or the old values would count as live through the GEMM: 104 registers)
        request(min(next, n_rows - 1), tid);
        const v4f bias = a.bias1 ? *(const v4f*)(a.bias1 + chw) : splat(0.f);
        v4f s0 = splat(0.f), s1 = splat(0.f);
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int p = wP0 + 4 * m + 2 * o;
                if (p < wlim && p < L) {
                    const size_t off = (size_t)p * CPAD + chw;
                    v4f v = (o == 0 ? acc4[m][0] + acc4[m][1] + acc4[m][2] : acc4[m][1] - acc4[m][2] - acc4[m][3]) + bias;
                    if (a.add1) v += *(const v4f*)(a.add1 + rbase + off);
                    if (a.add2) v += *(const v4f*)(a.add2 + rbase + off);
                    if (a.addb) v += *(const v4f*)(a.addb + sbase + off);
                    if (a.relu_out) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    if (a.stats) {
                        const v4f x = a.stat_aux ? *(const v4f*)(a.stat_aux + rbase + off) : v;
                        s0 += v;
                        s1 += v * x;
                    }
                    if (a.out1) *(v4f*)(a.out1 + rbase + off) = v;
                }
            }
        if (a.stats) {                                           // every channel belongs to exactly one (wave, k-quarter, j)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = s0[j], y = s1[j];
#pragma unroll
                for (int msk = 1; msk < 16; msk <<= 1) { x += __shfl_xor(x, msk); y += __shfl_xor(y, msk); }
                if (pos == 0) {
                    a.stats[((size_t)row * 2 + 0) * CPAD + chw + j] = x;
                    a.stats[((size_t)row * 2 + 1) * CPAD + chw + j] = y;
                }
            }
        }
        __syncthreads();                                         // every wave has finished reading the image
        continue;                                                // (no bottleneck stage rides on the 3-tap launches)
    }
    const int cq = wave & 3, ph = wave >> 2;
    const int m_base = ph * MTW, cnt = ph ? MT - MTW : MTW;
    int chb[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) chb[n] = (cq * NT + n) * 16 + kk * 4;
    v4f acc[MTW][NT];
    if (a.w1) {
        gv4f_ptr w = (gv4f_ptr)(a.w1) + (cq * NT) * 64 + lane;
        const v4f first[NT] = {w[0], w[64]};
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = splat(0.f);
        conv_gemm(acc, xs, w, first, a.kg, a.taps, a.dil, lane, m_base, cnt);
        request(min(next, n_rows - 1), tid);
    } else {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[m][n] = (m < cnt) ? *(const v4f*)(xs + (HALO + (m_base + m) * 16 + pos) * LDS_S + chb[n]) : splat(0.f);
        request(min(next, n_rows - 1), tid);
    }
    // ---- epilogue
    v4f s0[NT], s1v[NT], bias[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        s0[n] = splat(0.f); s1v[n] = splat(0.f);
        bias[n] = a.bias1 ? *(const v4f*)(a.bias1 + chb[n]) : splat(0.f);
    }
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int p = (m_base + m) * 16 + pos;
        const bool live = (m < cnt) && (p < L);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            v4f v = splat(0.f);
            if (live) {
                const size_t o = (size_t)p * CPAD + chb[n];
                v = acc[m][n] + bias[n];
                if (a.add1) v += *(const v4f*)(a.add1 + rbase + o);
                if (a.add2) v += *(const v4f*)(a.add2 + rbase + o);
                if (a.addb) v += *(const v4f*)(a.addb + sbase + o);
                if (a.relu_out) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                if (a.stats) {
                    const v4f x = a.stat_aux ? *(const v4f*)(a.stat_aux + rbase + o) : v;
                    s0[n] += v;
                    s1v[n] += v * x;
                }
                if (a.out1) *(v4f*)(a.out1 + rbase + o) = v;
            }
            acc[m][n] = v;
        }
    }
    if (a.stats) {
        // sum over the 16 position lanes of a k-quarter, then over the two position halves (waves cq and cq + 4)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = s0[n][j], y = s1v[n][j];
#pragma unroll
                for (int msk = 1; msk < 16; msk <<= 1) { x += __shfl_xor(x, msk); y += __shfl_xor(y, msk); }
                if (pos == 0) { sred[ph][0][chb[n] + j] = x; sred[ph][1][chb[n] + j] = y; }
            }
    }
    if (a.w2 || a.stats) __syncthreads();                        // (also: every wave has finished reading the image)
    if (a.stats && tid < 2 * CPAD) {
        const int st = tid >> 7, ch = tid & (CPAD - 1);
        a.stats[((size_t)row * 2 + st) * CPAD + ch] = sred[0][st][ch] + sred[1][st][ch];
    }
    if (a.w2) {
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            if (m < cnt) {
#pragma unroll
                for (int n = 0; n < NT; ++n) *(v4f*)(xs + (HALO + (m_base + m) * 16 + pos) * LDS_S + chb[n]) = acc[m][n];
            }
        }
        gv4f_ptr wb = (gv4f_ptr)(a.w2) + lane;
        v4f wbot[KGC];
#pragma unroll
        for (int g = 0; g < KGC; ++g) wbot[g] = wb[(g * 2 + (wave & 1)) * 64];
        __syncthreads();
        bottleneck<NWAVE>(xs, wbot, a.bias2, a.out2 + (size_t)row * L * HPAD, L, wave, lane);
    }
    __syncthreads();                                             // the image (and sred) are free for the next read
    }
}

// ------------------------------------------------------------------------------------------------
// T1h: the 3-tap launches of the direct form (forward convolution, data gradient) on HALF reads, two workgroups per CU.
// A whole-read workgroup owns its CU alone (the image is 118 KB), so its memory phases -- 100-200 KB of rows in, 100 KB out, at
// a CU's 20 GB/s share of HBM -- and its GEMM take turns: every CU loads, then every CU multiplies, HBM idle under the GEMMs and
// the matrix pipes idle under the copies (1.25 ms per launch where the MFMAs alone are 0.87 ms and the rows 0.35 ms).  One layer
// per launch needs no whole read: a unit here is 7 (or 6) position tiles of a read plus 4 halo rows either side, 64 KB of LDS,
// four waves (wave = 32 output channels x the unit's tiles: the same wave tile as above), and the two workgroups of a CU run
// half a unit apart (the second one sleeps first), one's rows travelling under the other's GEMM.  No prefetch registers, no
// second GEMM stage, no addends: what the two 3-tap launch kinds need and nothing else.  stats entries are per unit.
// tools/rowh_probe.hip, 6400 reads: whole read 1.235 / 1.21 ms (forward / data gradient), half reads 1.10 / 1.14 ms, of which
// 1.03 ms remain with the loads and stores switched off (LDS staging, barriers, GEMM, epilogue arithmetic: the GEMM core's own
// 0.84 of the MFMA rate); one workgroup per CU 1.29 ms; the start offset is worth 4 % on the forward launches, nothing on the
// data gradient (upper half of the grid or odd workgroups, one to five sleeps: all within 1 %).
// ------------------------------------------------------------------------------------------------
constexpr int RH_THREADS = 256, RH_ROWS = MTW * 16 + 2 * HALO;                    // 120 image rows
// SRC2 / POOL / F2: what the launch carries is known when it is made -- a second source tensor with its mask (the data gradient),
// the per-site addend (the pooled layer's convolution), the second product (see below) -- and as template parameters the parts a
// launch does not use cost no instruction: every vector instruction beside the MFMAs is paid for in MFMA issue time (as
// run-time flags: 1 950 vector instructions per unit beside 1 344 MFMAs).
template <bool SRC2, bool POOL, bool F2>
__global__ __launch_bounds__(RH_THREADS, 2) void train_rowh_kernel(RowArgs a, int n_rows, int stagger) {
    __shared__ __attribute__((aligned(16))) float xs[RH_ROWS * LDS_S];
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);                    // = the channel quarter
    const int L = a.L;
    const int n_units = 2 * n_rows;
    if ((stagger & 255) && ((stagger & 256) ? (blockIdx.x & 1) : ((int)blockIdx.x >= (int)gridDim.x / 2))) {
        for (int i = 0; i < (stagger & 255); ++i) __builtin_amdgcn_s_sleep(127);
    }
    int k_it = 0;
    for (int u = blockIdx.x; u < n_units; u += gridDim.x, ++k_it) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));                           // (per-lane addresses are formed per unit, not kept across the GEMM)
        const int lane = tid & 63;
        const int row = u >> 1, half = (u ^ k_it) & 1;          // a workgroup alternates between the 7-tile and the 6-tile half
        const int site = row / a.R;
        const int p0 = half * (MTW * 16), cnt = half ? MT - MTW : MTW;
        // ---- the image: rows [p0 - HALO, p0 + 112 + HALO) of the read, zero outside the window
        {
            constexpr int NP = RH_ROWS * (CPAD / 4) / RH_THREADS;                   // 15
            static_assert(NP * RH_THREADS == RH_ROWS * (CPAD / 4), "the staging covers the image exactly");
            const v4f* s1 = (const v4f*)(a.src1 + (size_t)row * L * CPAD);
            const v4f* s2 = (const v4f*)(a.src2 + (size_t)row * L * CPAD);
            const v4f* pl = (const v4f*)(a.pool_in + (size_t)site * L * CPAD);
            const Coef3 ck = load_coef(a.coef, (tid & 31) * 4);     // (i & 31 == tid & 31 in every iteration below)
            v4f r1[NP], r2[SRC2 ? NP : 1], r3[POOL ? NP : 1];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int i = tid + k * RH_THREADS;
                const int p = p0 - HALO + (i >> 5);
                const bool ok = p >= 0 && p < L;
                const int g = p * (CPAD / 4) + (i & 31);
                r1[k] = ok ? s1[g] : splat(0.f);
                if constexpr (SRC2) r2[k] = ok ? s2[g] : splat(0.f);
                if constexpr (POOL) r3[k] = ok ? pl[g] : splat(0.f);
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int i = tid + k * RH_THREADS;
                const int pr = i >> 5, c4 = i & 31;
                // (a row outside the window was loaded as zeros: it must be stored as zero, not as the transform's constant term)
                v4f v = r1[k];
                if constexpr (SRC2) v = apply_transform(r1[k], r2[k], ck, a.coef != nullptr, a.mask_src2);
                else if (a.coef) {
                    const int p = p0 - HALO + pr;
                    v = (p >= 0 && p < L) ? ck.A * r1[k] + ck.C : splat(0.f);
                }
                if constexpr (POOL) v += r3[k];
                *(v4f*)(xs + pr * LDS_S + c4 * 4) = v;
            }
        }
        __syncthreads();
        const int pos = lane & 15, kk = lane >> 4;
        int chb[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) chb[n] = (wave * NT + n) * 16 + kk * 4;
        v4f acc[MTW][NT];
        {
            gv4f_ptr w = (gv4f_ptr)(a.w1) + (wave * NT) * 64 + lane;
            const v4f first[NT] = {w[0], w[64]};
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = splat(0.f);
            conv_gemm(acc, xs, w, first, a.kg, a.taps, a.dil, lane, 0, cnt);
        }
        if constexpr (F2) {
            // ---- the data gradient's consumer adds W_b^T (dh * (h > 0)) to it (and the residual skip, and takes the BatchNorm
            // statistics): done here, du never crosses HBM (658 MB out, 658 MB in per layer at 64 sites) and the pointwise launch
            // that did it is gone.  The masked dh rows of the unit take the image's place (channels 0..31 of rows HALO..), the
            // product is a 1-tap K = 32 walk of the same GEMM core on top of the accumulators.
            gv4f_ptr w3 = (gv4f_ptr)(a.w3) + (wave * NT) * 64 + lane;
            const v4f f3[NT] = {w3[0], w3[64]};
            constexpr int NQ = (MTW * 16 * (HPAD / 4) + RH_THREADS - 1) / RH_THREADS;            // 4 (the last one half used)
            const v4f* d1 = (const v4f*)(a.src3 + (size_t)row * L * HPAD);
            const v4f* d2 = (const v4f*)(a.src4 + (size_t)row * L * HPAD);
            v4f q1[NQ], q2[NQ];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int i = tid + k * RH_THREADS;
                const int p = p0 + (i >> 3);
                const bool ok = i < MTW * 16 * (HPAD / 4) && p < L;
                const int g = p * (HPAD / 4) + (i & 7);
                q1[k] = ok ? d1[g] : splat(0.f);
                q2[k] = ok ? d2[g] : splat(0.f);
            }
            __syncthreads();                                     // every wave has finished reading the image
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int i = tid + k * RH_THREADS;
                if (i < MTW * 16 * (HPAD / 4)) {
                    v4f v = q1[k];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = q2[k][j] > 0.f ? v[j] : 0.f;
                    *(v4f*)(xs + (HALO + (i >> 3)) * LDS_S + (i & 7) * 4) = v;
                }
            }
            __syncthreads();
            conv_gemm(acc, xs, w3, f3, HPAD / 16, 1, 0, lane, 0, cnt);
        }
        // ---- epilogue: out1 = [relu](acc + bias1 [+ add2]); stats[u] = (sum out1, sum out1 * (aux or out1)) over the unit's positions
        const size_t rbase = (size_t)row * L * CPAD;
        v4f s0[NT], s1v[NT], bias[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            s0[n] = splat(0.f); s1v[n] = splat(0.f);
            bias[n] = a.bias1 ? *(const v4f*)(a.bias1 + chb[n]) : splat(0.f);
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int p = p0 + m * 16 + pos;
            if (m < cnt && p < L) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const size_t off = rbase + (size_t)p * CPAD + chb[n];
                    v4f v = acc[m][n] + bias[n];
                    if constexpr (F2) { if (a.add2) v += *(const v4f*)(a.add2 + off); }
                    if (a.relu_out) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    if (a.stats) {
                        v4f x = v;
                        if constexpr (F2) { if (a.stat_aux) x = *(const v4f*)(a.stat_aux + off); }
                        s0[n] += v;
                        s1v[n] += v * x;
                    }
                    *(v4f*)(a.out1 + off) = v;
                }
            }
        }
        if (a.stats) {                                           // every channel belongs to exactly one (wave, tile, k-quarter, j)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = s0[n][j], y = s1v[n][j];
#pragma unroll
                    for (int msk = 1; msk < 16; msk <<= 1) { x += __shfl_xor(x, msk); y += __shfl_xor(y, msk); }
                    if (pos == 0) {
                        a.stats[((size_t)u * 2 + 0) * CPAD + chb[n] + j] = x;
                        a.stats[((size_t)u * 2 + 1) * CPAD + chb[n] + j] = y;
                    }
                }
        }
        __syncthreads();                                         // every wave has finished reading the image
    }
}

// ------------------------------------------------------------------------------------------------
// T1p: the POINTWISE launches (1x1 GEMMs: BN-apply [+ residual] + bottleneck, the accumulation of g_l with the bottleneck's
// transpose, the residual's transpose).  Nothing reaches sideways, so the unit is a tile of 64 positions of the flat
// [rows * L] position axis instead of a whole read: 35 KB of LDS and <= 128 registers, FOUR workgroups (16 waves) per CU, and
// the HBM-bound streaming of one tile overlaps the GEMM of the others (as whole-read workgroups, one per CU, these launches
// ran at ~55 % of the HBM rate).  Wave = 32 output channels x the 4 position tiles.  Same RowArgs, same epilogue; `stats`
// entries are per tile.
// ------------------------------------------------------------------------------------------------
constexpr int TP_POS = 64, TP_THREADS = 256, TP_MT = TP_POS / 16;
__global__ __launch_bounds__(TP_THREADS, 4) void train_point_kernel(RowArgs a, long long n_pos) {
    __shared__ __attribute__((aligned(16))) float xs[TP_POS * LDS_S];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long base = (long long)blockIdx.x * TP_POS;
    const int L = a.L;
    const int n_here = (int)min((long long)TP_POS, n_pos - base);
    // ---- load the tile (positions beyond the tensor are zero)
    {
        const int vpr = a.s1_stride >> 2;
        const int vsh = vpr == 32 ? 5 : 3;                      // (rows are CPAD or HPAD floats wide)
        const v4f* s1 = (const v4f*)a.src1 + base * vpr;
        const v4f* s2 = a.src2 ? (const v4f*)a.src2 + base * vpr : nullptr;
        constexpr int NP = TP_POS * (CPAD / 4) / TP_THREADS;        // 8
        const Coef3 ck = load_coef(a.coef, (tid & (vpr - 1)) * 4);  // vpr is 32 or 8: i mod vpr == tid mod vpr in every iteration
        v4f r1[NP], r2[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = tid + k * TP_THREADS;
            const bool ok = i < n_here * vpr;
            r1[k] = ok ? s1[i] : splat(0.f);
            r2[k] = (ok && s2) ? s2[i] : splat(0.f);
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = tid + k * TP_THREADS;
            if (i < TP_POS * vpr) {
                const int pr = i >> vsh, c4 = i & (vpr - 1);          // (vpr is 32 or 8: shifts -- as a division by a run-time number this line was a third of the kernel's vector instructions)
                *(v4f*)(xs + pr * LDS_S + c4 * 4) = (pr < n_here) ? apply_transform(r1[k], r2[k], ck, a.coef != nullptr, a.mask_src2) : splat(0.f);
            }
        }
        if (vpr < CPAD / 4) {                                    // 32-channel input: the GEMM reads only its kg groups, nothing to clear
        }
    }
    __syncthreads();
    const int pos = lane & 15, kk = lane >> 4;
    const int cq = wave;
    int chb[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) chb[n] = (cq * NT + n) * 16 + kk * 4;
    v4f acc[TP_MT][NT];
    if (a.w1) {
        gv4f_ptr w = (gv4f_ptr)(a.w1) + (cq * NT) * 64 + lane;
        const float* xrow = xs + pos * LDS_S + kk * 4;
        v4f a_nxt[NT], b[TP_MT];
#pragma unroll
        for (int n = 0; n < NT; ++n) a_nxt[n] = w[n * 64];
#pragma unroll
        for (int m = 0; m < TP_MT; ++m) { b[m] = *(const v4f*)(xrow + m * 16 * LDS_S); acc[m][0] = splat(0.f); acc[m][1] = splat(0.f); }
        for (int g = 0; g < a.kg; ++g) {
            v4f av[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) av[n] = a_nxt[n];
            const int gn = (g + 1 < a.kg) ? g + 1 : g;
#pragma unroll
            for (int n = 0; n < NT; ++n) a_nxt[n] = w[(size_t)gn * (KGC * 64) + n * 64];
#pragma unroll
            for (int m = 0; m < TP_MT; ++m) {
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = mfma16(av[n][sidx], b[m][sidx], acc[m][n]);
                b[m] = *(const v4f*)(xrow + m * 16 * LDS_S + gn * 16);
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < TP_MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = *(const v4f*)(xs + (m * 16 + pos) * LDS_S + chb[n]);
    }
    // ---- epilogue (flat position q = base + m 16 + pos; its site for the per-site addend)
    v4f s0[NT], s1v[NT], bias[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        s0[n] = splat(0.f); s1v[n] = splat(0.f);
        bias[n] = a.bias1 ? *(const v4f*)(a.bias1 + chb[n]) : splat(0.f);
    }
    const long long per_site = (long long)a.R * L;
    // The addends of ALL eight (m, n) cells are requested before the first is used, tensor by tensor: one memory round trip per
    // tensor and workgroup instead of one per cell (the cells' loads sat behind each other's uses, eight exposed round trips
    // in a workgroup that lives for about twenty).  Same operations in the same order per element.
    auto cell_off = [&](int m, int n) { return (size_t)(base + m * 16 + pos) * CPAD + chb[n]; };
    auto fetch = [&](const float* src, v4f (&dst)[TP_MT][NT]) {
#pragma unroll
        for (int m = 0; m < TP_MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) dst[m][n] = (m * 16 + pos < n_here) ? *(const v4f*)(src + cell_off(m, n)) : splat(0.f);
    };
    v4f aux[TP_MT][NT];
    const bool own_aux = a.stats && a.stat_aux;
    if (own_aux) fetch(a.stat_aux, aux);
#pragma unroll
    for (int m = 0; m < TP_MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] += bias[n];
    if (a.add1) {
        v4f t[TP_MT][NT];
        fetch(a.add1, t);
        if (a.add1_coef) {                                       // the addend is a BatchNorm output that exists only as its input
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const v4f A = *(const v4f*)(a.add1_coef + chb[n]), C = *(const v4f*)(a.add1_coef + 2 * CPAD + chb[n]);
#pragma unroll
                for (int m = 0; m < TP_MT; ++m) t[m][n] = (m * 16 + pos < n_here) ? A * t[m][n] + C : splat(0.f);
            }
        }
#pragma unroll
        for (int m = 0; m < TP_MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] += t[m][n];
    }
    if (a.add2) {
        v4f t[TP_MT][NT];
        fetch(a.add2, t);
#pragma unroll
        for (int m = 0; m < TP_MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] += t[m][n];
    }
    if (a.addb) {
        v4f t[TP_MT][NT];
#pragma unroll
        for (int m = 0; m < TP_MT; ++m) {
            const long long q = base + m * 16 + pos;
            const long long site = q / per_site;
            const int p = (int)(q % L);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                t[m][n] = (m * 16 + pos < n_here) ? *(const v4f*)(a.addb + ((size_t)site * L + p) * CPAD + chb[n]) : splat(0.f);
        }
#pragma unroll
        for (int m = 0; m < TP_MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] += t[m][n];
    }
#pragma unroll
    for (int m = 0; m < TP_MT; ++m) {
        const int pl = m * 16 + pos;
        const bool live = pl < n_here;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            v4f v = splat(0.f);
            if (live) {
                v = acc[m][n];
                if (a.relu_out) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                if (a.stats) {
                    const v4f x = own_aux ? aux[m][n] : v;
                    s0[n] += v;
                    s1v[n] += v * x;
                }
                if (a.out1) *(v4f*)(a.out1 + cell_off(m, n)) = v;
            }
            acc[m][n] = v;
        }
    }
    if (a.stats) {                                               // every channel belongs to exactly one (wave, tile, k-quarter, j)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = s0[n][j], y = s1v[n][j];
#pragma unroll
                for (int msk = 1; msk < 16; msk <<= 1) { x += __shfl_xor(x, msk); y += __shfl_xor(y, msk); }
                if (pos == 0) {
                    a.stats[((size_t)blockIdx.x * 2 + 0) * CPAD + chb[n] + j] = x;
                    a.stats[((size_t)blockIdx.x * 2 + 1) * CPAD + chb[n] + j] = y;
                }
            }
    }
    if (a.w2) {
        __syncthreads();                                         // every wave has finished reading the tile
#pragma unroll
        for (int m = 0; m < TP_MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) *(v4f*)(xs + (m * 16 + pos) * LDS_S + chb[n]) = acc[m][n];
        // 128 -> 32 bottleneck: wave = (channel tile n = wave & 1, position tiles (wave >> 1), (wave >> 1) + 2)
        gv4f_ptr wb = (gv4f_ptr)(a.w2) + lane;
        const int nb = wave & 1, p0 = wave >> 1;
        v4f wbot[KGC];
#pragma unroll
        for (int g = 0; g < KGC; ++g) wbot[g] = wb[(g * 2 + nb) * 64];
        __syncthreads();
        v4f hacc[2];
        {
            const v4f bb = *(const v4f*)(a.bias2 + nb * 16 + kk * 4);
            hacc[0] = bb; hacc[1] = bb;
        }
        const float* xrow = xs + pos * LDS_S + kk * 4;
#pragma unroll
        for (int g = 0; g < KGC; ++g)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const v4f bv = *(const v4f*)(xrow + (p0 + 2 * i) * 16 * LDS_S + g * 16);
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) hacc[i] = mfma16(wbot[g][sidx], bv[sidx], hacc[i]);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pl = (p0 + 2 * i) * 16 + pos;
            if (pl < n_here) {
                v4f v = hacc[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = relu1(v[j]);
                *(v4f*)(a.out2 + (size_t)(base + pl) * HPAD + nb * 16 + kk * 4) = v;
            }
        }
    }
}

// The half-read form's predicate.  With f2 (a.w3 set) the launch must also carry a second source.
static bool row_half_units(const RowArgs& a, bool f2) {
    const bool src2 = a.src2 != nullptr, pool = a.pool_in != nullptr;
    return !a.wino && a.w1 && a.taps == 3 && a.out1 && !a.w2 && !a.add1 && !a.addb && (f2 || (!a.add2 && !a.stat_aux)) &&
           a.s1_stride == CPAD && a.L <= RH_THREADS && !(pool && (src2 || f2)) && !(f2 && !src2) &&
           !(a.mask_src2 && !src2);
}
bool train_row_fuses_second_product(const RowArgs& a) {
    const bool pointwise = !a.pool_in && !a.wino && (a.w1 == nullptr || a.taps == 1);
    return !pointwise && row_half_units(a, true);
}

// Returns the number of `stats` entries the launch writes (reads for the whole-read form, half-read units for the half-read form,
// 64-position tiles for the pointwise one), or TRAIN_ROW_ERR_* without launching.
int launch_train_row(const RowArgs& a, int n_rows, hipStream_t s, int stat_cap) {
    if (a.mode == 0) return TRAIN_ROW_ERR_FORM;                  // (the encode form is gone: layer 1 runs from tables, launch_l0_train_forward)
    const bool pointwise = !a.pool_in && !a.wino && (a.w1 == nullptr || a.taps == 1);
    static const int n_cus_h = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    const bool f2 = a.w3 != nullptr, src2 = a.src2 != nullptr, pool = a.pool_in != nullptr;
    const bool half_units = !pointwise && row_half_units(a, f2);
    if (f2 && !half_units) return TRAIN_ROW_ERR_FORM;           // only train_rowh_kernel<., ., true> forms the second product
    {
        const long long entries = pointwise ? ((long long)n_rows * a.L + TP_POS - 1) / TP_POS : half_units ? 2LL * n_rows : n_rows;
        if (a.stats && entries > stat_cap) return TRAIN_ROW_ERR_STATS;
    }
    if (pointwise) {
        const long long n_pos = (long long)n_rows * a.L;
        const int tiles = (int)((n_pos + TP_POS - 1) / TP_POS);
        hipLaunchKernelGGL(train_point_kernel, dim3((unsigned)tiles), dim3(TP_THREADS), 0, s, a, n_pos);
        return tiles;
    }
    if (half_units) {
        // as many workgroups as give every one the same number of units (2 000 units at 10 sites: 500 workgroups of 4, not 512 of
        // which 464 run a fourth round at 94 % idle), an even count (the half a workgroup takes alternates with its round)
        const int n_units = 2 * n_rows, rounds = (n_units + 2 * n_cus_h - 1) / (2 * n_cus_h);
        const int wgs = std::min(n_units, ((n_units + rounds - 1) / rounds + 1) & ~1);
        // the second workgroup of a CU starts half a unit late (three s_sleep 127: ~10 us) when there is a second one per CU
        const dim3 grid((unsigned)wgs), blk(RH_THREADS);
        const int st = wgs > n_cus_h ? 3 : 0;
        if (f2) hipLaunchKernelGGL((train_rowh_kernel<true, false, true>), grid, blk, 0, s, a, n_rows, st);
        else if (src2) hipLaunchKernelGGL((train_rowh_kernel<true, false, false>), grid, blk, 0, s, a, n_rows, st);
        else if (pool) hipLaunchKernelGGL((train_rowh_kernel<false, true, false>), grid, blk, 0, s, a, n_rows, st);
        else hipLaunchKernelGGL((train_rowh_kernel<false, false, false>), grid, blk, 0, s, a, n_rows, st);
        return n_units;
    }
    static const int n_cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    hipLaunchKernelGGL(train_row_kernel, dim3((unsigned)std::min(n_rows, n_cus)), dim3(SEG_THREADS), 0, s, a, n_rows);
    return n_rows;
}

// ------------------------------------------------------------------------------------------------
// T2: weight gradients (contraction over positions)
// ------------------------------------------------------------------------------------------------
// Wave w owns output tile w of A (OWN_O) and every input tile of B and tap, or input tile w of B and both output tiles of
// A (bottleneck: A is 32 wide).  MFMA k index = position: lane group kk supplies position 4 k4 + kk, so the two k-groups
// of a 32-lane half read rows one apart: with the 144-float row stride those are 16 banks apart -- conflict-free ds_read_b32.
// CT: input tiles of B an OWN_O wave walks (8, or 3 for the 48-channel encoded input) -- a compile-time count, so that the
// k-step body is straight-line code: all of a step's LDS reads issued, then its MFMAs (with the count at run time every tile
// was a branch and every MFMA waited for its own read)
template <int TAPS, bool OWN_O, int CT>
__global__ __launch_bounds__(SEG_THREADS, NWAVE / 4) void train_wgrad_kernel(WgradArgs a) {
    constexpr int BROWS = WG_CH + 2 * HALO;
    constexpr int NB = OWN_O ? CT : 1, NA = OWN_O ? 1 : 2;
    __shared__ __attribute__((aligned(16))) float sa[WG_CH * WG_S];
    __shared__ __attribute__((aligned(16))) float sb[BROWS * WG_S];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.L;
    const int i16 = lane & 15, kk = lane >> 4;
    const int vpa = a.a_stride >> 2;
    const int ash = vpa == 32 ? 5 : 3;                          // (A rows are CPAD or HPAD floats wide: shifts, not integer divisions, in the staging loops)
    v4f acc[TAPS][NA][NB];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int x = 0; x < NA; ++x)
#pragma unroll
            for (int y = 0; y < NB; ++y) acc[t][x][y] = splat(0.f);
    v4f bsum = splat(0.f);
    for (int i = tid; i < WG_CH * WG_S / 4; i += SEG_THREADS) ((v4f*)sa)[i] = splat(0.f);
    const bool active = OWN_O ? (wave < a.o_tiles) : (wave < a.c_tiles);
    for (int row = blockIdx.x; row < a.n_rows; row += gridDim.x) {
        const int site = row / a.R;
        for (int p0 = 0; p0 < L; p0 += WG_CH) {
            __syncthreads();                                     // the previous chunk's MFMAs are done with the images
            // ---- A rows [p0, p0 + WG_CH) and B rows [p0 - HALO, p0 + WG_CH + HALO): every load of the chunk -- both operands -- in
            // flight before the first is consumed (a staging loop of dependent load -> transform -> LDS store round trips was half of
            // this kernel's time); in the 1x1 forms both operands travel together, ONE memory round trip per chunk.  The per-site
            // addend of the one pooled layer is a trip of its own.
            {
                const v4f* s1 = (const v4f*)(a.a1 + (size_t)row * L * a.a_stride);
                const v4f* s2 = a.a2 ? (const v4f*)(a.a2 + (size_t)row * L * a.a_stride) : nullptr;
                const v4f* b1 = (const v4f*)(a.b1 + (size_t)row * L * CPAD);
                constexpr int NIT = (WG_CH * (CPAD / 4) + SEG_THREADS - 1) / SEG_THREADS;       // 7 at 128 channels
                constexpr int NITB = BROWS * (CPAD / 4) / SEG_THREADS;                           // 7
                static_assert(NITB * SEG_THREADS == BROWS * (CPAD / 4), "B staging covers the image exactly");
                const Coef3 ck = load_coef(a.a_coef, (tid & (vpa - 1)) * 4);                    // vpa is 32 or 8: i mod vpa == tid mod vpa
                // (the same four pieces of code in two orders, written out: behind lambdas hipcc spilled 42 registers of the 3-tap form)
                if constexpr (TAPS == 1) {
                    v4f r1[NIT], r2[NIT], rb[NITB];
#pragma unroll
                    for (int k = 0; k < NIT; ++k) {
                        const int i = tid + k * SEG_THREADS;
                        const int pl = i >> ash, c4 = i & (vpa - 1), p = p0 + pl;
                        const bool ok = i < WG_CH * vpa && p < L;
                        const size_t g = (size_t)p * vpa + c4;
                        r1[k] = ok ? s1[g] : splat(0.f);
                        r2[k] = (ok && s2) ? s2[g] : splat(0.f);
                    }
#pragma unroll
                    for (int k = 0; k < NITB; ++k) {
                        const int i = tid + k * SEG_THREADS;
                        const int p = p0 - HALO + (i >> 5);
                        rb[k] = (p >= 0 && p < L) ? b1[(size_t)p * (CPAD / 4) + (i & 31)] : splat(0.f);
                    }
#pragma unroll
                    for (int k = 0; k < NIT; ++k) {
                        const int i = tid + k * SEG_THREADS;
                        if (i < WG_CH * vpa) {
                            const int pl = i >> ash, c4 = i & (vpa - 1), p = p0 + pl;
                            const v4f v = (p < L) ? apply_transform(r1[k], r2[k], ck, a.a_coef != nullptr, a.a_mask) : splat(0.f);
                            *(v4f*)(sa + pl * WG_S + c4 * 4) = v;
                            bsum += v;
                        }
                    }
                    {
                        const v4f* pl4 = a.b_pool ? (const v4f*)(a.b_pool + (size_t)site * L * CPAD) : nullptr;
                        const Coef3 cb = load_coef(a.b_coef, (tid & 31) * 4);
#pragma unroll
                        for (int k = 0; k < NITB; ++k) {
                            const int i = tid + k * SEG_THREADS;
                            const int pl = i >> 5, c4 = i & 31, p = p0 - HALO + pl;
                            const bool in = p >= 0 && p < L;
                            v4f v = rb[k];
                            if (a.b_coef && in) v = cb.A * v + cb.C;
                            if (pl4 && in) v = v + pl4[(size_t)p * (CPAD / 4) + c4];
                            *(v4f*)(sb + pl * WG_S + c4 * 4) = v;
                        }
                    }
                } else {
                    // (beside the 96 accumulator registers of the 3-tap form the 21 vectors in flight spill 33 registers and the
                    // launch takes 1.37 ms instead of 1.19: there B is requested once A is in LDS)
                    {
                        v4f r1[NIT], r2[NIT];
#pragma unroll
                        for (int k = 0; k < NIT; ++k) {
                            const int i = tid + k * SEG_THREADS;
                            const int pl = i >> ash, c4 = i & (vpa - 1), p = p0 + pl;
                            const bool ok = i < WG_CH * vpa && p < L;
                            const size_t g = (size_t)p * vpa + c4;
                            r1[k] = ok ? s1[g] : splat(0.f);
                            r2[k] = (ok && s2) ? s2[g] : splat(0.f);
                        }
#pragma unroll
                        for (int k = 0; k < NIT; ++k) {
                            const int i = tid + k * SEG_THREADS;
                            if (i < WG_CH * vpa) {
                                const int pl = i >> ash, c4 = i & (vpa - 1), p = p0 + pl;
                                const v4f v = (p < L) ? apply_transform(r1[k], r2[k], ck, a.a_coef != nullptr, a.a_mask) : splat(0.f);
                                *(v4f*)(sa + pl * WG_S + c4 * 4) = v;
                                bsum += v;
                            }
                        }
                    }
                    {
                        const v4f* pl4 = a.b_pool ? (const v4f*)(a.b_pool + (size_t)site * L * CPAD) : nullptr;
                        const Coef3 cb = load_coef(a.b_coef, (tid & 31) * 4);
                        v4f rb[NITB], rp[NITB];
#pragma unroll
                        for (int k = 0; k < NITB; ++k) {
                            const int i = tid + k * SEG_THREADS;
                            const int pl = i >> 5, c4 = i & 31, p = p0 - HALO + pl;
                            const bool ok = p >= 0 && p < L;
                            const size_t g = (size_t)p * (CPAD / 4) + c4;
                            rb[k] = ok ? b1[g] : splat(0.f);
                            rp[k] = (ok && pl4) ? pl4[g] : splat(0.f);
                        }
#pragma unroll
                        for (int k = 0; k < NITB; ++k) {
                            const int i = tid + k * SEG_THREADS;
                            const int pl = i >> 5, c4 = i & 31, p = p0 - HALO + pl;
                            v4f v = rb[k];
                            if (a.b_coef && p >= 0 && p < L) v = cb.A * v + cb.C;
                            *(v4f*)(sb + pl * WG_S + c4 * 4) = v + rp[k];
                        }
                    }
                }
            }
            __syncthreads();
            if (active) {
                const float* pa = sa + kk * WG_S + i16;
                const float* pb = sb + (HALO + kk) * WG_S + i16;
                const int dstep = a.dil * WG_S;
                // Software-pipelined over the k-steps: step k4 + 1's operands (NA + TAPS x NB one-float LDS reads) are requested
                // while step k4's MFMAs issue, into the other register set; one read behind each MFMA, pinned by scheduling
                // groups.  (Left to itself hipcc keeps ONE register pair for the B operands: ds_read2_b32, s_waitcnt lgkmcnt(0),
                // two MFMAs, again -- every pair of MFMAs behind an exposed LDS round trip: MFMA busy 0.67, waits 0.42.)
                constexpr int NSTEP = WG_CH / 4;
                static_assert(NSTEP % 2 == 0, "the two register sets alternate over pairs of k-steps");
                float av[2][NA], bv[2][TAPS][NB];
                auto fetch = [&](int k4, float (&A)[NA], float (&B)[TAPS][NB]) {
#pragma unroll
                    for (int x = 0; x < NA; ++x) A[x] = pa[4 * k4 * WG_S + 16 * (OWN_O ? wave : x)];
#pragma unroll
                    for (int t = 0; t < TAPS; ++t)
#pragma unroll
                        for (int y = 0; y < NB; ++y) B[t][y] = pb[4 * k4 * WG_S + (t - TAPS / 2) * dstep + 16 * (OWN_O ? y : wave)];
                };
                auto mfmas = [&](const float (&A)[NA], const float (&B)[TAPS][NB]) {
#pragma unroll
                    for (int t = 0; t < TAPS; ++t)
#pragma unroll
                        for (int y = 0; y < NB; ++y)
#pragma unroll
                            for (int x = 0; x < NA; ++x) acc[t][x][y] = mfma16(A[x], B[t][y], acc[t][x][y]);
                };
                auto pipeline = [&]() {
#pragma unroll
                    for (int i = 0; i < TAPS * NB * NA; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, NA + TAPS * NB, 0);      // (the reads no MFMA carried)
                };
                fetch(0, av[0], bv[0]);
                __builtin_amdgcn_sched_barrier(0);
                for (int k4 = 0; k4 < NSTEP; k4 += 2) {
                    fetch(k4 + 1, av[1], bv[1]);
                    mfmas(av[0], bv[0]);
                    pipeline();
                    __builtin_amdgcn_sched_barrier(0);
                    fetch(min(k4 + 2, NSTEP - 1), av[0], bv[0]);                        // (last pair: a harmless re-read)
                    mfmas(av[1], bv[1]);
                    pipeline();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    // ---- partial results of this workgroup
    const int OP = a.o_tiles * 16, CP = a.c_tiles * 16;
    if (active) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int x = 0; x < NA; ++x)
#pragma unroll
                for (int y = 0; y < NB; ++y) {
                    const int ot = OWN_O ? wave : x, ct = OWN_O ? y : wave;
                    if (ot >= a.o_tiles || ct >= a.c_tiles) continue;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int o = 16 * ot + 4 * kk + jj, c = 16 * ct + i16;
                        a.partial[(((size_t)blockIdx.x * TAPS + t) * OP + o) * CP + c] = acc[t][x][y][jj];
                    }
                }
    }
    // bias gradient: per-thread channel sums -> fixed-order sum over the threads that staged the same channels
    __syncthreads();
    float* scratch = sb;                                         // [SEG_THREADS][4]
    *(v4f*)(scratch + tid * 4) = bsum;
    __syncthreads();
    if (tid < OP) {
        const int c4 = tid >> 2, j = tid & 3;
        float sum = 0.f;
        for (int t = c4; t < SEG_THREADS; t += vpa) sum += scratch[t * 4 + j];
        a.bias_partial[(size_t)blockIdx.x * OP + tid] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// T2b: the 1x1 weight gradients (residual 128 x 128, bottleneck 32 x 128) with TWO workgroups per CU.  One tap means 8 (or 2)
// accumulator tiles per wave instead of 24 and no halo: with chunks of 52 positions both images take 60 KB and a thread stages
// at most twelve vectors, so the kernel fits 128 registers and two workgroups share a CU -- one stages (these forms are closer
// to their HBM bound than to their MFMA bound: 1.0-1.3 GB per launch) while the other multiplies.  Unlike the half-output split
// tried for the 3-tap form nothing is staged twice.  Same arithmetic per chunk as train_wgrad_kernel<1, ...>; the chunk
// boundaries differ, so the position sums associate differently (fp32, bounded as before by the split-K reduction's own).
// ------------------------------------------------------------------------------------------------
constexpr int W1_CH = 52;
template <bool OWN_O, int CT>
__global__ __launch_bounds__(SEG_THREADS, 4) void train_wgrad1_kernel(WgradArgs a) {
    constexpr int NSTEP = W1_CH / 4;                          // 13
    static_assert(NSTEP % 2 == 1, "the k-step pipeline below runs pairs of steps and one last step");
    constexpr int NB = OWN_O ? CT : 1, NA = OWN_O ? 1 : 2;
    __shared__ __attribute__((aligned(16))) float sa[W1_CH * WG_S];
    __shared__ __attribute__((aligned(16))) float sb[W1_CH * WG_S];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.L;
    const int i16 = lane & 15, kk = lane >> 4;
    const int vpa = a.a_stride >> 2;
    const int ash = vpa == 32 ? 5 : 3;                          // (A rows are CPAD or HPAD floats wide: shifts, not integer divisions, in the staging loops)
    v4f acc[NA][NB];
#pragma unroll
    for (int x = 0; x < NA; ++x)
#pragma unroll
        for (int y = 0; y < NB; ++y) acc[x][y] = splat(0.f);
    v4f bsum = splat(0.f);
    for (int i = tid; i < W1_CH * WG_S / 4; i += SEG_THREADS) ((v4f*)sa)[i] = splat(0.f);
    const bool active = OWN_O ? (wave < a.o_tiles) : (wave < a.c_tiles);
    for (int row = blockIdx.x; row < a.n_rows; row += gridDim.x) {
        for (int p0 = 0; p0 < L; p0 += W1_CH) {
            __syncthreads();                                     // the previous chunk's MFMAs are done with the images
            {
                const v4f* s1 = (const v4f*)(a.a1 + (size_t)row * L * a.a_stride);
                const v4f* s2 = a.a2 ? (const v4f*)(a.a2 + (size_t)row * L * a.a_stride) : nullptr;
                const v4f* b1 = (const v4f*)(a.b1 + (size_t)row * L * CPAD);
                constexpr int NIT = (W1_CH * (CPAD / 4) + SEG_THREADS - 1) / SEG_THREADS;       // 4 (the last one a quarter used)
                v4f r1[NIT], r2[NIT], rb[NIT];
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int i = tid + k * SEG_THREADS;
                    const int pl = i >> ash, c4 = i & (vpa - 1), p = p0 + pl;
                    const bool ok = i < W1_CH * vpa && p < L;
                    const size_t g = (size_t)p * vpa + c4;
                    r1[k] = ok ? s1[g] : splat(0.f);
                    r2[k] = (ok && s2) ? s2[g] : splat(0.f);
                }
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int i = tid + k * SEG_THREADS;
                    const int p = p0 + (i >> 5);
                    rb[k] = (i < W1_CH * (CPAD / 4) && p < L) ? b1[(size_t)p * (CPAD / 4) + (i & 31)] : splat(0.f);
                }
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int i = tid + k * SEG_THREADS;
                    if (i < W1_CH * vpa) {
                        const int pl = i >> ash, c4 = i & (vpa - 1);
                        v4f v = r1[k];                           // (rows past the window were loaded as zeros; these forms have no affine on A)
                        if (a.a_mask) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = r2[k][j] > 0.f ? v[j] : 0.f;
                        }
                        *(v4f*)(sa + pl * WG_S + c4 * 4) = v;
                        bsum += v;
                    }
                }
                v4f cbA = splat(1.f), cbC = splat(0.f);
                if (a.b_coef) { cbA = *(const v4f*)(a.b_coef + (tid & 31) * 4); cbC = *(const v4f*)(a.b_coef + 2 * CPAD + (tid & 31) * 4); }
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int i = tid + k * SEG_THREADS;
                    if (i < W1_CH * (CPAD / 4)) {
                        const int pl = i >> 5, c4 = i & 31, p = p0 + pl;
                        v4f v = rb[k];
                        if (a.b_coef && p < L) v = cbA * v + cbC;
                        *(v4f*)(sb + pl * WG_S + c4 * 4) = v;
                    }
                }
            }
            __syncthreads();
            if (active) {
                const float* pa = sa + kk * WG_S + i16;
                const float* pb = sb + kk * WG_S + i16;
                float av[2][NA], bv[2][NB];
                auto fetch = [&](int k4, float (&A)[NA], float (&B)[NB]) {
#pragma unroll
                    for (int x = 0; x < NA; ++x) A[x] = pa[4 * k4 * WG_S + 16 * (OWN_O ? wave : x)];
#pragma unroll
                    for (int y = 0; y < NB; ++y) B[y] = pb[4 * k4 * WG_S + 16 * (OWN_O ? y : wave)];
                };
                auto mfmas = [&](const float (&A)[NA], const float (&B)[NB]) {
#pragma unroll
                    for (int y = 0; y < NB; ++y)
#pragma unroll
                        for (int x = 0; x < NA; ++x) acc[x][y] = mfma16(A[x], B[y], acc[x][y]);
                };
                fetch(0, av[0], bv[0]);
                for (int k4 = 0; k4 < NSTEP - 1; k4 += 2) {
                    fetch(k4 + 1, av[1], bv[1]);
                    mfmas(av[0], bv[0]);
                    fetch(k4 + 2, av[0], bv[0]);
                    mfmas(av[1], bv[1]);
                }
                mfmas(av[0], bv[0]);                             // the last (odd) step
            }
        }
    }
    // ---- partial results of this workgroup
    const int OP = a.o_tiles * 16, CP = a.c_tiles * 16;
    if (active) {
#pragma unroll
        for (int x = 0; x < NA; ++x)
#pragma unroll
            for (int y = 0; y < NB; ++y) {
                const int ot = OWN_O ? wave : x, ct = OWN_O ? y : wave;
                if (ot >= a.o_tiles || ct >= a.c_tiles) continue;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int o = 16 * ot + 4 * kk + jj, c = 16 * ct + i16;
                    a.partial[((size_t)blockIdx.x * OP + o) * CP + c] = acc[x][y][jj];
                }
            }
    }
    __syncthreads();
    float* scratch = sb;                                         // [SEG_THREADS][4]
    *(v4f*)(scratch + tid * 4) = bsum;
    __syncthreads();
    if (tid < OP) {
        const int c4 = tid >> 2, j = tid & 3;
        float sum = 0.f;
        for (int t = c4; t < SEG_THREADS; t += vpa) sum += scratch[t * 4 + j];
        a.bias_partial[(size_t)blockIdx.x * OP + tid] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// T2c: the 3-tap weight gradient with the NEXT chunk's rows in flight under the current chunk's MFMAs.  Same eight waves, same
// wave tile (one output tile x eight input tiles x three taps: 96 accumulator registers) as train_wgrad_kernel, but chunks of
// 52 positions in a double-buffered pair of images (2 x 63 KB), so that a thread's share of a chunk is twelve vectors -- few
// enough to stay in registers through a GEMM that itself touches no global memory (in-order vmcnt: nothing retires them
// early).  Per chunk: request chunk j + 1 -> multiply chunk j -> transform and store chunk j + 1 into the other image ->
// barrier.  What stays exposed is the transform and the LDS stores, not the memory round trip.
// ------------------------------------------------------------------------------------------------
constexpr int W3_CH = 52;
template <int CT>
__global__ __launch_bounds__(SEG_THREADS, NWAVE / 4) void train_wgrad3_kernel(WgradArgs a) {
    constexpr int TAPS = 3;
    constexpr int BROWS = W3_CH + 2 * HALO;                    // 60
    constexpr int NSTEP = W3_CH / 4;                          // 13
    static_assert(NSTEP % 2 == 1, "the k-step pipeline below runs pairs of steps and one last step");
    __shared__ __attribute__((aligned(16))) float sa[2][W3_CH * WG_S];
    __shared__ __attribute__((aligned(16))) float sb[2][BROWS * WG_S];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.L;
    const int i16 = lane & 15, kk = lane >> 4;
    v4f acc[TAPS][CT];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int y = 0; y < CT; ++y) acc[t][y] = splat(0.f);
    v4f bsum = splat(0.f);
    const bool active = wave < a.o_tiles;
    const int n_chunks = (L + W3_CH - 1) / W3_CH;
    const int my_rows = (a.n_rows - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int n_items = my_rows * n_chunks;
    constexpr int NIT = (W3_CH * (CPAD / 4) + SEG_THREADS - 1) / SEG_THREADS;                   // 4 (the last one a quarter used)
    constexpr int NITB = (BROWS * (CPAD / 4) + SEG_THREADS - 1) / SEG_THREADS;                  // 4 (the last one three quarters used)
    v4f r1[NIT], r2[NIT], rb[NITB];
    auto request = [&](int j) {
        const int row = (int)blockIdx.x + (j / n_chunks) * (int)gridDim.x, p0 = (j % n_chunks) * W3_CH;
        const v4f* s1 = (const v4f*)(a.a1 + (size_t)row * L * CPAD);
        const v4f* s2 = (const v4f*)(a.a2 + (size_t)row * L * CPAD);
        const v4f* b1 = (const v4f*)(a.b1 + (size_t)row * L * CPAD);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + k * SEG_THREADS;
            const int p = p0 + (i >> 5);
            const bool ok = i < W3_CH * (CPAD / 4) && p < L;
            const size_t g = (size_t)p * (CPAD / 4) + (i & 31);
            r1[k] = ok ? s1[g] : splat(0.f);
            r2[k] = ok ? s2[g] : splat(0.f);
        }
#pragma unroll
        for (int k = 0; k < NITB; ++k) {
            const int i = tid + k * SEG_THREADS;
            const int p = p0 - HALO + (i >> 5);
            rb[k] = (i < BROWS * (CPAD / 4) && p >= 0 && p < L) ? b1[(size_t)p * (CPAD / 4) + (i & 31)] : splat(0.f);
        }
    };
    auto stage = [&](int j) {
        const int p0 = (j % n_chunks) * W3_CH;
        float* da = sa[j & 1];
        float* db = sb[j & 1];
        const Coef3 ck = load_coef(a.a_coef, (tid & 31) * 4);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + k * SEG_THREADS;
            if (i < W3_CH * (CPAD / 4)) {
                const int pl = i >> 5, c4 = i & 31, p = p0 + pl;
                const v4f v = (p < L) ? apply_transform(r1[k], r2[k], ck, a.a_coef != nullptr, a.a_mask) : splat(0.f);
                *(v4f*)(da + pl * WG_S + c4 * 4) = v;
                bsum += v;
            }
        }
        v4f cbA = splat(1.f), cbC = splat(0.f);
        if (a.b_coef) { cbA = *(const v4f*)(a.b_coef + (tid & 31) * 4); cbC = *(const v4f*)(a.b_coef + 2 * CPAD + (tid & 31) * 4); }
#pragma unroll
        for (int k = 0; k < NITB; ++k) {
            const int i = tid + k * SEG_THREADS;
            if (i < BROWS * (CPAD / 4)) {
                const int pl = i >> 5, c4 = i & 31, p = p0 - HALO + pl;
                v4f v = rb[k];
                if (a.b_coef && p >= 0 && p < L) v = cbA * v + cbC;
                *(v4f*)(db + pl * WG_S + c4 * 4) = v;
            }
        }
    };
    if (n_items > 0) { request(0); stage(0); }
    __syncthreads();
    for (int j = 0; j < n_items; ++j) {
        if (j + 1 < n_items) request(j + 1);                     // in flight under the MFMAs below
        if (active) {
            const float* pa = sa[j & 1] + kk * WG_S + i16 + 16 * wave;
            const float* pb = sb[j & 1] + (HALO + kk) * WG_S + i16;
            const int dstep = a.dil * WG_S;
            float av[2], bv[2][TAPS][CT];
            auto fetch = [&](int k4, float& A, float (&B)[TAPS][CT]) {
                A = pa[4 * k4 * WG_S];
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int y = 0; y < CT; ++y) B[t][y] = pb[4 * k4 * WG_S + (t - TAPS / 2) * dstep + 16 * y];
            };
            auto mfmas = [&](const float& A, const float (&B)[TAPS][CT]) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int y = 0; y < CT; ++y) acc[t][y] = mfma16(A, B[t][y], acc[t][y]);
            };
            auto pipeline = [&]() {
#pragma unroll
                for (int i = 0; i < TAPS * CT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 1 + TAPS * CT, 0);
            };
            fetch(0, av[0], bv[0]);
            __builtin_amdgcn_sched_barrier(0);
            for (int k4 = 0; k4 < NSTEP - 1; k4 += 2) {
                fetch(k4 + 1, av[1], bv[1]);
                mfmas(av[0], bv[0]);
                pipeline();
                __builtin_amdgcn_sched_barrier(0);
                fetch(k4 + 2, av[0], bv[0]);
                mfmas(av[1], bv[1]);
                pipeline();
                __builtin_amdgcn_sched_barrier(0);
            }
            mfmas(av[0], bv[0]);                                 // the last (odd) step
        }
        if (j + 1 < n_items) stage(j + 1);                       // into the other image: nobody reads it during chunk j
        __syncthreads();
    }
    // ---- partial results of this workgroup
    const int OP = a.o_tiles * 16, CP = a.c_tiles * 16;
    if (active) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int y = 0; y < CT; ++y) {
                if (y >= a.c_tiles) continue;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int o = 16 * wave + 4 * kk + jj, c = 16 * y + i16;
                    a.partial[(((size_t)blockIdx.x * TAPS + t) * OP + o) * CP + c] = acc[t][y][jj];
                }
            }
    }
    float* scratch = sb[0];                                      // [SEG_THREADS][4]  (the loop's last barrier freed the images)
    *(v4f*)(scratch + tid * 4) = bsum;
    __syncthreads();
    if (tid < OP) {
        const int c4 = tid >> 2, j = tid & 3;
        float sum = 0.f;
        for (int t = c4; t < SEG_THREADS; t += CPAD / 4) sum += scratch[t * 4 + j];
        a.bias_partial[(size_t)blockIdx.x * OP + tid] = sum;
    }
}

int launch_train_wgrad(const WgradArgs& a, hipStream_t s) {
    // (balanced over the rounds: 1 000 rows at 10 sites are 250 workgroups of 4 rows, not 256 of which 232 run a fourth round)
    auto balanced = [](int n, int cap) { const int rounds = (n + cap - 1) / cap; return (n + rounds - 1) / rounds; };
    const int wgs = balanced(a.n_rows, TRAIN_PARTIAL_WGS);
    const dim3 grid((unsigned)wgs), blk(SEG_THREADS);
    if (a.taps == 1 && !a.b_pool && !a.a_coef) {   // 1x1 forms: two workgroups per CU (train_wgrad1_kernel)
        const int wgs2 = balanced(a.n_rows, 2 * TRAIN_PARTIAL_WGS);
        if (a.o_tiles <= 2) hipLaunchKernelGGL((train_wgrad1_kernel<false, 1>), dim3((unsigned)wgs2), blk, 0, s, a);
        else hipLaunchKernelGGL((train_wgrad1_kernel<true, KGC>), dim3((unsigned)wgs2), blk, 0, s, a);
        return wgs2;
    }
    if (a.taps == 3 && !a.b_pool && a.a2 && a.a_stride == CPAD && a.c_tiles == KGC && a.o_tiles == KGC) {
        hipLaunchKernelGGL((train_wgrad3_kernel<KGC>), grid, blk, 0, s, a);                    // the next chunk in flight under the MFMAs
        return wgs;
    }
    if (a.o_tiles <= 2 && a.taps == 1) hipLaunchKernelGGL((train_wgrad_kernel<1, false, 1>), grid, blk, 0, s, a);   // A 32 wide: both its tiles per wave
    else if (a.taps == 1) hipLaunchKernelGGL((train_wgrad_kernel<1, true, KGC>), grid, blk, 0, s, a);
    else if (a.c_tiles <= KG0) hipLaunchKernelGGL((train_wgrad_kernel<3, true, KG0>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((train_wgrad_kernel<3, true, KGC>), grid, blk, 0, s, a);
    return wgs;
}

// sum of load(0) .. load(n-1) IN INDEX ORDER (the result is bit-for-bit that of the plain loop): the loads are requested sixteen
// at a time, so a reduction over a few hundred partials waits for ~n/16 memory round trips instead of n
template <typename T, typename F>
__device__ __forceinline__ T ordered_sum(int n, F load) {
    T sum = (T)0;
    int i = 0;
    for (; i + 16 <= n; i += 16) {
        T v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = load(i + j);
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += v[j];
    }
    for (; i < n; ++i) sum += load(i);
    return sum;
}

// two sums over the same index range, their loads in flight together (each sum in index order)
template <typename T, typename F0, typename F1>
__device__ __forceinline__ void ordered_sum2(int n, F0 load0, F1 load1, T& out0, T& out1) {
    T s0 = (T)0, s1 = (T)0;
    int i = 0;
    for (; i + 16 <= n; i += 16) {
        T v[16], u[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { v[j] = load0(i + j); u[j] = load1(i + j); }
#pragma unroll
        for (int j = 0; j < 16; ++j) { s0 += v[j]; s1 += u[j]; }
    }
    for (; i < n; ++i) { s0 += load0(i); s1 += load1(i); }
    out0 = s0; out1 = s1;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias_partial,
                                                           int wgs, int taps, int OP, int CP, int n_out, int n_in,
                                                           const int* __restrict__ cmap, float* g_w, float* g_b) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = taps * n_out * n_in;
    if (idx < total) {
        const int i = idx % n_in, o = (idx / n_in) % n_out, t = idx / (n_in * n_out);
        const int c = cmap ? cmap[i] : i;
        const float* src = partial + ((size_t)t * OP + o) * CP + c;
        const size_t stride = (size_t)taps * OP * CP;
        g_w[((size_t)o * n_in + i) * taps + t] = ordered_sum<float>(wgs, [&](int w) { return src[(size_t)w * stride]; });
    } else if (g_b && idx < total + n_out) {
        const int o = idx - total;
        g_b[o] = ordered_sum<float>(wgs, [&](int w) { return bias_partial[(size_t)w * OP + o]; });
    }
}

void launch_wgrad_reduce(const float* partial, const float* bias_partial, int wgs, int taps, int o_pad, int c_pad, int n_out,
                         int n_in, const int* cmap, float* g_w, float* g_b, hipStream_t s) {
    const int total = taps * n_out * n_in + n_out;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, partial, bias_partial, wgs, taps, o_pad,
                       c_pad, n_out, n_in, cmap, g_w, g_b);
}

// ------------------------------------------------------------------------------------------------
// per-channel statistics: [row][2][CPAD] fp32 partials -> block partials in double -> BatchNorm coefficients
// ------------------------------------------------------------------------------------------------
// Entries per block: at least 32, and few enough blocks (<= 128) that the single-workgroup kernels behind this one -- which add
// the block partials in index order, sixteen loads at a time -- wait for eight round trips instead of forty (20 100 tile entries
// of a pointwise launch: 30 us -> 8 us per layer).
__global__ __launch_bounds__(256) void stats_partial_kernel(const float* __restrict__ stats, int n_rows, int per_block, double* __restrict__ bp) {
    const int lo = blockIdx.x * per_block, hi = min(n_rows, lo + per_block);
    double sum = 0.0;
    int r = lo;
    for (; r + 8 <= hi; r += 8) {                               // (eight loads in flight; the additions stay in index order)
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = stats[(size_t)(r + j) * 2 * CPAD + threadIdx.x];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += (double)v[j];
    }
    for (; r < hi; ++r) sum += (double)stats[(size_t)r * 2 * CPAD + threadIdx.x];
    bp[(size_t)blockIdx.x * 2 * CPAD + threadIdx.x] = sum;
}

void launch_stats_partial(const float* stats, int n_rows, double* bp, int* n_blocks, hipStream_t s) {
    const int per_block = std::max(32, (n_rows + 127) / 128);
    const int nb = (n_rows + per_block - 1) / per_block;
    *n_blocks = nb;
    hipLaunchKernelGGL(stats_partial_kernel, dim3(nb), dim3(2 * CPAD), 0, s, stats, n_rows, per_block, bp);
}

__global__ __launch_bounds__(CPAD) void bn_forward_finalize_kernel(const double* __restrict__ bp, int nb, double n_pos,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   int channels, float* coef, float* save_mean, float* save_invstd,
                                                                   float* run_mean, float* run_var) {
    const int ch = threadIdx.x;
    float scale = 0.f, shift = 0.f;
    if (ch < channels) {
        double s0, s1;
        ordered_sum2<double>(nb, [&](int b) { return bp[(size_t)b * 2 * CPAD + ch]; }, [&](int b) { return bp[(size_t)b * 2 * CPAD + CPAD + ch]; }, s0, s1);
        const double mean = s0 / n_pos;
        double var = s1 / n_pos - mean * mean;                  // biased variance normalises (nn.BatchNorm2d, training)
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + 1e-5);
        scale = (float)((double)gamma[ch] * invstd);
        shift = (float)((double)beta[ch] - mean * (double)gamma[ch] * invstd);
        save_mean[ch] = (float)mean;
        save_invstd[ch] = (float)invstd;
        // running statistics: momentum 0.1, UNBIASED batch variance
        run_mean[ch] = (float)(0.9 * (double)run_mean[ch] + 0.1 * mean);
        run_var[ch] = (float)(0.9 * (double)run_var[ch] + 0.1 * var * (n_pos / (n_pos - 1.0)));
    }
    coef[ch] = scale;
    coef[CPAD + ch] = 0.f;
    coef[2 * CPAD + ch] = shift;
}

void launch_bn_forward_finalize(const double* bp, int n_blocks, double n_pos, const float* gamma, const float* beta, int channels,
                                float* coef, float* save_mean, float* save_invstd, float* run_mean, float* run_var, hipStream_t s) {
    hipLaunchKernelGGL(bn_forward_finalize_kernel, dim3(1), dim3(CPAD), 0, s, bp, n_blocks, n_pos, gamma, beta, channels, coef,
                       save_mean, save_invstd, run_mean, run_var);
}

// da = gamma r (dn - dbeta/N - nhat dgamma/N),  nhat = (a - mean) r   =>   da = A dn + B a + C
__global__ __launch_bounds__(CPAD) void bn_backward_coef_kernel(const double* __restrict__ bp, int nb, double n_pos,
                                                                const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                                const float* __restrict__ save_invstd, int channels, int use_bn,
                                                                float* coef, float* g_gamma, float* g_beta) {
    const int ch = threadIdx.x;
    float A = 0.f, B = 0.f, C = 0.f;
    if (ch < channels) {
        if (use_bn) {
            double s0, s1;
            ordered_sum2<double>(nb, [&](int b) { return bp[(size_t)b * 2 * CPAD + ch]; }, [&](int b) { return bp[(size_t)b * 2 * CPAD + CPAD + ch]; }, s0, s1);
            const double mu = save_mean[ch], r = save_invstd[ch], g = gamma[ch];
            const double dbeta = s0, dgamma = (s1 - mu * s0) * r;
            g_gamma[ch] = (float)dgamma;
            g_beta[ch] = (float)dbeta;
            A = (float)(g * r);
            const double Bd = -g * r * r * dgamma / n_pos;
            B = (float)Bd;
            C = (float)(-g * r * dbeta / n_pos - Bd * mu);
        } else {
            A = 1.f;
        }
    }
    coef[ch] = A;
    coef[CPAD + ch] = B;
    coef[2 * CPAD + ch] = C;
}

void launch_bn_backward_coef(const double* bp, int n_blocks, double n_pos, const float* gamma, const float* save_mean,
                             const float* save_invstd, int channels, int use_bn, float* coef, float* g_gamma, float* g_beta,
                             hipStream_t s) {
    hipLaunchKernelGGL(bn_backward_coef_kernel, dim3(1), dim3(CPAD), 0, s, bp, n_blocks, n_pos, gamma, save_mean, save_invstd,
                       channels, use_bn, coef, g_gamma, g_beta);
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_frag_body(long long idx, float* __restrict__ dst, const float* __restrict__ src, int taps, int kg,
                                               int tiles, int n_out, int n_in, long long so, long long sc, long long st,
                                               int flip, const int* __restrict__ omap, const int* __restrict__ cmap) {
    const long long total = (long long)taps * kg * tiles * 256;
    if (idx >= total) return;
    const int sidx = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    long long rest = idx >> 8;
    const int n = (int)(rest % tiles); rest /= tiles;
    const int g = (int)(rest % kg);
    const int t = (int)(rest / kg);
    int o = 16 * n + (lane & 15), c = 16 * g + 4 * (lane >> 4) + sidx;
    float v = 0.f;
    if (omap) o = (o < n_out) ? omap[o] : -1; else if (o >= n_out) o = -1;
    if (cmap) c = (c < n_in) ? cmap[c] : -1; else if (c >= n_in) c = -1;
    if (o >= 0 && c >= 0) v = src[(long long)o * so + (long long)c * sc + (long long)(flip ? taps - 1 - t : t) * st];
    dst[idx] = v;
}
__global__ __launch_bounds__(256) void pack_frag_kernel(float* __restrict__ dst, const float* __restrict__ src, int taps, int kg,
                                                        int tiles, int n_out, int n_in, long long so, long long sc, long long st,
                                                        int flip, const int* __restrict__ omap, const int* __restrict__ cmap) {
    pack_frag_body((long long)blockIdx.x * 256 + threadIdx.x, dst, src, taps, kg, tiles, n_out, n_in, so, sc, st, flip, omap, cmap);
}

void launch_pack_frag(float* dst, const float* src, int taps, int kg, int tiles, int n_out, int n_in, long long so, long long sc,
                      long long st, int flip, const int* omap, const int* cmap, hipStream_t s) {
    const long long total = (long long)taps * kg * tiles * 256;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dst, src, taps, kg, tiles, n_out, n_in,
                       so, sc, st, flip, omap, cmap);
}

// Winograd F(2,3) weight transform of a 3-tap kernel g = (g0, g1, g2):  U = [g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2],
// formed in double.  dst[(o * n_in + c) * 4 + k];  g_t = src[o * so + c * sc + (flip ? 2 - t : t)]  (the data gradient uses the
// flipped kernel with the roles of o and c exchanged through so / sc)
__global__ __launch_bounds__(256) void wino_u_kernel(float* __restrict__ dst, const float* __restrict__ src, int n_out, int n_in,
                                                     long long so, long long sc, int flip) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_out * n_in) return;
    const int c = idx % n_in, o = idx / n_in;
    const float* g = src + (long long)o * so + (long long)c * sc;
    const double g0 = g[flip ? 2 : 0], g1 = g[1], g2 = g[flip ? 0 : 2];
    float* u = dst + (size_t)idx * 4;
    u[0] = (float)g0;
    u[1] = (float)((g0 + g1 + g2) * 0.5);
    u[2] = (float)((g0 - g1 + g2) * 0.5);
    u[3] = (float)g2;
}

void launch_wino_u(float* dst, const float* src, int n_out, int n_in, long long so, long long sc, int flip, hipStream_t s) {
    hipLaunchKernelGGL(wino_u_kernel, dim3((n_out * n_in + 255) / 256), dim3(256), 0, s, dst, src, n_out, n_in, so, sc, flip);
}

// highway kernel fragment order: dst[g = 2p + (c>>4)][n][lane][s] = Wc[o = 16n + (lane&15)][c = 16(g&1) + 4(lane>>4) + s][p]
__device__ __forceinline__ void pack_wc_body(int idx, float* __restrict__ dst, const float* __restrict__ src, int H, int L) {
    if (idx >= 2 * L * 2 * 256) return;
    const int sidx = idx & 3, lane = (idx >> 2) & 63, n = (idx >> 8) & 1, g = idx >> 9;
    const int o = 16 * n + (lane & 15), c = 16 * (g & 1) + 4 * (lane >> 4) + sidx, p = g >> 1;
    dst[idx] = (o < H && c < H) ? src[((size_t)o * H + c) * L + p] : 0.f;
}
__global__ __launch_bounds__(256) void pack_wc_kernel(float* __restrict__ dst, const float* __restrict__ src, int H, int L) {
    pack_wc_body(blockIdx.x * 256 + threadIdx.x, dst, src, H, L);
}

// WcT[p][c][o] = Wc[o][c][p], zero-padded to HPAD x HPAD (the highway backward reads 32 consecutive o per (p, c))
__device__ __forceinline__ void pack_wct_body(int idx, float* __restrict__ dst, const float* __restrict__ src, int H, int L) {
    if (idx >= L * HPAD * HPAD) return;
    const int o = idx & 31, c = (idx >> 5) & 31, p = idx >> 10;
    dst[idx] = (o < H && c < H) ? src[((size_t)o * H + c) * L + p] : 0.f;
}
__global__ __launch_bounds__(256) void pack_wct_kernel(float* __restrict__ dst, const float* __restrict__ src, int H, int L) {
    pack_wct_body(blockIdx.x * 256 + threadIdx.x, dst, src, H, L);
}

// Every re-packing of a step in ONE launch: the job table is fixed once the trainer is finalized (pointers into the flat
// parameter buffer and the packed-weight buffers), and ~70 launches of 2-5 us each, one behind the other, were 0.26 ms of every
// step whatever the batch (3 % of the 10-site step).  Block b serves the job whose block range holds it.
__global__ __launch_bounds__(256) void pack_jobs_kernel(const PackJob* __restrict__ jobs, int n_jobs) {
    int j = 0;
    while (j + 1 < n_jobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;
    const PackJob q = jobs[j];
    const long long idx = (long long)((int)blockIdx.x - q.first_block) * 256 + threadIdx.x;
    switch (q.type) {
    case 0: pack_frag_body(idx, q.dst, q.src, q.i[0], q.i[1], q.i[2], q.i[3], q.i[4], q.l[0], q.l[1], q.l[2], q.i[5], q.omap, q.cmap); break;
    case 1: if (idx < q.i[1]) q.dst[idx] = idx < q.i[0] ? q.src[idx] : 0.f; break;
    case 2: pack_wc_body((int)idx, q.dst, q.src, q.i[0], q.i[1]); break;
    default: pack_wct_body((int)idx, q.dst, q.src, q.i[0], q.i[1]); break;
    }
}
void launch_pack_jobs(const PackJob* jobs, int n_jobs, int n_blocks, hipStream_t s) {
    hipLaunchKernelGGL(pack_jobs_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, jobs, n_jobs);
}
int pack_job_blocks(const PackJob& q) {
    long long n = 0;
    switch (q.type) {
    case 0: n = (long long)q.i[0] * q.i[1] * q.i[2] * 256; break;
    case 1: n = q.i[1]; break;
    case 2: n = (long long)2 * q.i[1] * 2 * 256; break;
    default: n = (long long)q.i[1] * HPAD * HPAD; break;
    }
    return (int)((n + 255) / 256);
}

void launch_pack_wct(float* dst, const float* src, int H, int L, hipStream_t s) {
    hipLaunchKernelGGL(pack_wct_kernel, dim3((L * HPAD * HPAD + 255) / 256), dim3(256), 0, s, dst, src, H, L);
}

// dst[i] = i < n ? src[i] : 0 for i < n_pad   (biases padded to the kernels' channel capacity)
__global__ __launch_bounds__(256) void pad_copy_kernel(float* __restrict__ dst, const float* __restrict__ src, int n, int n_pad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_pad) dst[i] = i < n ? src[i] : 0.f;
}

void launch_pad_copy(float* dst, const float* src, int n, int n_pad, hipStream_t s) {
    hipLaunchKernelGGL(pad_copy_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, s, dst, src, n, n_pad);
}

void launch_pack_wc(float* dst, const float* src, int H, int L, hipStream_t s) {
    hipLaunchKernelGGL(pack_wc_kernel, dim3((2 * L * 2 * 256 + 255) / 256), dim3(256), 0, s, dst, src, H, L);
}

// ------------------------------------------------------------------------------------------------
// final max + mean pool backward (dl4vc/model.py:824-839): the max's gradient goes to the FIRST read attaining it
// (torch's max_pool2d keeps the first index on ties: all-padding rows of a pileup are identical)
// ------------------------------------------------------------------------------------------------
template <int PT>       // positions per workgroup: 32, or 8 when 32 would leave most CUs without a workgroup (10 sites: 70 -> 260)
__global__ __launch_bounds__(256) void final_pool_bwd_kernel(const v4f* __restrict__ y, const float* __restrict__ dfeat, long long fs,
                                                             v4f* __restrict__ g, int R, int L, int C) {
    constexpr int PS = PT + 1;
    __shared__ float dmax[CPAD * PS], dmean[CPAD * PS];
    const int pt = blockIdx.x, site = blockIdx.y, tid = threadIdx.x;
    const float* row = dfeat + (size_t)site * fs;
    for (int idx = tid; idx < CPAD * PT; idx += 256) {
        const int c = idx / PT, pp = idx % PT, p = pt * PT + pp;
        const bool ok = c < C && p < L;
        dmax[c * PS + pp] = ok ? row[(size_t)c * L + p] : 0.f;
        dmean[c * PS + pp] = ok ? row[(size_t)C * L + (size_t)c * L + p] : 0.f;
    }
    __syncthreads();
    const int c4 = tid & 31, pl = tid >> 5;
    const int n4 = L * (CPAD / 4);
    const float inv = 1.f / (float)R;
#pragma unroll
    for (int part = 0; part < PT / 8; ++part) {
        const int pp = part * 8 + pl, p = pt * PT + pp;
        if (p >= L) continue;
        const size_t off = (size_t)p * (CPAD / 4) + c4;
        v4f mx = y[((size_t)site * R) * n4 + off];
        int arg[4] = {0, 0, 0, 0};
        for (int r = 1; r < R; ++r) {
            const v4f v = y[((size_t)site * R + r) * n4 + off];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (v[j] > mx[j]) { mx[j] = v[j]; arg[j] = r; }
        }
        v4f dm, da;
#pragma unroll
        for (int j = 0; j < 4; ++j) { dm[j] = dmax[(c4 * 4 + j) * PS + pp]; da[j] = dmean[(c4 * 4 + j) * PS + pp] * inv; }
        for (int r = 0; r < R; ++r) {
            v4f o = da;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += (arg[j] == r) ? dm[j] : 0.f;
            g[((size_t)site * R + r) * n4 + off] = o;
        }
    }
}

void launch_final_pool_bwd(const float* y, const float* dfeat, long long fs, float* g, int n_sites, int R, int L, int C, hipStream_t s) {
    if (((L + 31) / 32) * n_sites >= 512)
        hipLaunchKernelGGL(final_pool_bwd_kernel<32>, dim3((L + 31) / 32, n_sites), dim3(256), 0, s, (const v4f*)y, dfeat, fs, (v4f*)g, R, L, C);
    else
        hipLaunchKernelGGL(final_pool_bwd_kernel<8>, dim3((L + 7) / 8, n_sites), dim3(256), 0, s, (const v4f*)y, dfeat, fs, (v4f*)g, R, L, C);
}

// ------------------------------------------------------------------------------------------------
// highway compression backward (dl4vc/model.py:776-777,859): per layer  hw[row][o] = sum_{p,c} Wc[o][c][p] h[row][p][c]
// ------------------------------------------------------------------------------------------------
// dhw[row][o] = dfeat_hw * (feat_hw > 0)   (the ReLU sits on the concatenated highways, model.py:859)
__device__ __forceinline__ float dhw_of(const float* dfeat, const float* feat, long long fs, int feat_off, int layer, int H, int R,
                                        int row, int o) {
    const int site = row / R, r = row - site * R;
    const size_t i = (size_t)site * fs + feat_off + (size_t)layer * H * R + (size_t)o * R + r;
    return feat[i] > 0.f ? dfeat[i] : 0.f;
}

// dhw[layer][row][o] (zero for o >= H): the operand of both highway GEMMs
__global__ __launch_bounds__(256) void highway_dhw_kernel(const float* __restrict__ dfeat, const float* __restrict__ feat, long long fs,
                                                          int feat_off, float* __restrict__ dhw, int n_rows, int R, int H) {
    const int layer = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_rows * HPAD) return;
    const int row = idx >> 5, o = idx & 31;
    dhw[(size_t)layer * n_rows * HPAD + idx] = o < H ? dhw_of(dfeat, feat, fs, feat_off, layer, H, R, row, o) : 0.f;
}

void launch_highway_dhw(const float* dfeat, const float* feat, long long fs, int feat_off, float* dhw, int n_sites, int R, int H,
                        int layers, hipStream_t s) {
    const int n_rows = n_sites * R;
    hipLaunchKernelGGL(highway_dhw_kernel, dim3((n_rows * HPAD + 255) / 256, layers), dim3(256), 0, s, dfeat, feat, fs, feat_off, dhw, n_rows, R, H);
}

// gWc[o][c][p] = t[o][p*HPAD + c]  (the wgrad GEMM's output is [o][e = (p, c)]: the torch layout wants (o, c, p))
__global__ __launch_bounds__(256) void highway_wc_transpose_kernel(const float* __restrict__ t, float* __restrict__ g_wc, int L, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * H * L) return;
    const int p = idx % L, c = (idx / L) % H, o = idx / (L * H);
    g_wc[idx] = t[(size_t)o * L * HPAD + p * HPAD + c];
}

void launch_highway_wc_transpose(const float* t, float* g_wc, int L, int H, hipStream_t s) {
    hipLaunchKernelGGL(highway_wc_transpose_kernel, dim3((H * H * L + 255) / 256), dim3(256), 0, s, t, g_wc, L, H);
}

// ------------------------------------------------------------------------------------------------
// highway compression backward on the matrix cores, one launch each for all layers (round 5; they were 7 x 2 launches of the generic
// tiled GEMM -- K = 32 and a 165-MB output at 2.6 TB/s; K = rows with both operands row-slow at 1.8 TB/s, split-K partials, a
// transpose launch per layer).
//   dh[l][row][p][c]  = sum_o dhw[l][row][o] Wc[l][o][c][p]            (write stream: highway_dh_kernel)
//   gWc[l][o][c][p]   = sum_row dhw[l][row][o] h[l][row][p][c]         (read stream of h: highway_gwc_kernel)
// ------------------------------------------------------------------------------------------------
// dh: the forward highway kernel's structure (dan_kernels.hip) with the roles turned: wave = 16 reads x all positions, the reads'
// dhw (16 x 32) sits in the wave's registers for the whole walk as the B operand, the position's WcT[p] (32 c x 32 o = 4 KiB) is the
// A operand, fetched once per workgroup through a two-phase LDS ring; a lane ends up with four consecutive c of one read: one
// 16-byte store per channel tile, a read's 128 bytes of a position complete behind two instructions, positions in order.
__global__ __launch_bounds__(512) void highway_dh_kernel(const float* __restrict__ dhw, const float* __restrict__ wct, float* __restrict__ dh,
                                                         int n_rows, int L) {
    __shared__ __attribute__((aligned(16))) char ring[2][8][4][1024];      // [phase parity][position of the phase][c tile 2 x o group 2][lane * 16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;
    const int layer = blockIdx.y;
    const int row0 = blockIdx.x * 128 + wave * 16;
    const int row = min(row0 + i16, n_rows - 1);
    const float* brow = dhw + ((size_t)layer * n_rows + row) * HPAD + kk * 4;
    const v4f b0 = *(const v4f*)brow, b1 = *(const v4f*)(brow + 16);       // o groups 0..15, 16..31 of this lane's read
    // A fragment (c tile mt, o group g) of position p: lane (c16 = i16, kk) <- WcT[p][16 mt + c16][16 g + 4 kk ..]
    const float* wl = wct + (size_t)layer * L * HPAD * HPAD + (size_t)i16 * HPAD + kk * 4;
    auto wfrag = [&](int p, int j) { return *(const v4f*)(wl + (size_t)p * HPAD * HPAD + (j >> 1) * 16 * HPAD + (j & 1) * 16); };
    float* drow = dh + ((size_t)layer * n_rows + row) * (size_t)L * HPAD + kk * 4;
    const bool live = row0 + i16 < n_rows;
    const int n_ph = (L + 7) >> 3;
    v4f wq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wq[j] = wfrag(min(wave, L - 1), j);
#pragma unroll
    for (int j = 0; j < 4; ++j) *(v4f*)(&ring[0][wave][j][lane * 16]) = wq[j];
    __syncthreads();
    for (int ph = 0; ph < n_ph; ++ph) {
        if (ph + 1 < n_ph) {
            const int pn = min(8 * (ph + 1) + wave, L - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) wq[j] = wfrag(pn, j);
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const int p = 8 * ph + d;
            if (p < L) {
                const char* wr = &ring[ph & 1][d][0][lane * 16];
                const v4f a00 = *(const v4f*)wr, a01 = *(const v4f*)(wr + 1024), a10 = *(const v4f*)(wr + 2048), a11 = *(const v4f*)(wr + 3072);
                v4f c0 = splat(0.f), c1 = splat(0.f);
#pragma unroll
                for (int sI = 0; sI < 4; ++sI) { c0 = mfma16(a00[sI], b0[sI], c0); c1 = mfma16(a10[sI], b0[sI], c1); }
#pragma unroll
                for (int sI = 0; sI < 4; ++sI) { c0 = mfma16(a01[sI], b1[sI], c0); c1 = mfma16(a11[sI], b1[sI], c1); }
                if (live) {
                    *(v4f*)(drow + (size_t)p * HPAD) = c0;               // c = 4 kk .. 4 kk + 3
                    *(v4f*)(drow + (size_t)p * HPAD + 16) = c1;          // c = 16 + 4 kk ..
                }
            }
        }
        if (ph + 1 < n_ph) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *(v4f*)(&ring[(ph + 1) & 1][wave][j][lane * 16]) = wq[j];
        }
        __syncthreads();
    }
}

void launch_highway_dh(const float* dhw, const float* wct, float* dh, int n_rows, int L, int layers, hipStream_t s) {
    hipLaunchKernelGGL(highway_dh_kernel, dim3((n_rows + 127) / 128, layers), dim3(512), 0, s, dhw, wct, dh, n_rows, L);
}

// gWc: one workgroup = (layer, eight positions), wave = ONE position, all reads in order (no split: one accumulator chain per
// element, deterministic).  MFMA k = the read: of a block of 16 reads lane (i16, kk) supplies reads 4 s + kk (s = the MFMA of the
// group) -- A = dhw[read][o] from an LDS copy of the block that all eight waves share, B = h[read][p][c] by one 4-byte load per lane
// and MFMA (a wave instruction moves 4 reads x 64 bytes; 32 of them in flight).  The (o, c) block of the position goes straight
// to the torch layout gWc[o][c][p].
constexpr int GW_ROWS = 64;                                   // reads per staged dhw block
__global__ __launch_bounds__(512) void highway_gwc_kernel(const float* __restrict__ dhw, const float* __restrict__ h, float* __restrict__ g_base,
                                                          const long long* __restrict__ w_off, int n_rows, int L, int H) {
    __shared__ __attribute__((aligned(16))) float sd[2][GW_ROWS * HPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;
    const int layer = blockIdx.y;
    const int p = blockIdx.x * 8 + wave;
    const bool pos_ok = p < L;
    const int pc = min(p, L - 1);
    const float* hl = h + (size_t)layer * n_rows * (size_t)L * HPAD + (size_t)pc * HPAD + i16;
    const float* dl = dhw + (size_t)layer * n_rows * HPAD;
    v4f acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = splat(0.f);
    const int n_blk = (n_rows + GW_ROWS - 1) / GW_ROWS;
    auto stage = [&](int blk, int buf) {                          // 64 reads x 32 floats: one 16-byte load per thread, zero past the end
        const int r = blk * GW_ROWS + (tid >> 3);
        const v4f v = r < n_rows ? *(const v4f*)(dl + (size_t)r * HPAD + (tid & 7) * 4) : splat(0.f);
        *(v4f*)(&sd[buf][(tid >> 3) * HPAD + (tid & 7) * 4]) = v;
    };
    // h of a block: 4 groups of 16 reads x (s 4) x (c tile 2) one-float loads
    float hb[2][4][4][2];
    auto request = [&](int blk, int slot) {
#pragma unroll
        for (int gI = 0; gI < 4; ++gI)
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
                const int r = min(blk * GW_ROWS + gI * 16 + 4 * sI + kk, n_rows - 1);       // (clamped reads multiply a zero of the dhw block)
                const float* src = hl + (size_t)r * L * HPAD;
                hb[slot][gI][sI][0] = src[0];
                hb[slot][gI][sI][1] = src[16];
            }
    };
    stage(0, 0);
    request(0, 0);
    __syncthreads();
    auto block = [&](int blk, auto slot_c) {                      // (the slot is a compile-time index: the loop below walks pairs)
        constexpr int cur = decltype(slot_c)::value;
        if (blk + 1 < n_blk) { stage(blk + 1, cur ^ 1); request(blk + 1, cur ^ 1); }
#pragma unroll
        for (int gI = 0; gI < 4; ++gI)
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
                const float* ar = &sd[cur][(gI * 16 + 4 * sI + kk) * HPAD + i16];
                const float a0 = ar[0], a1 = ar[16];
                const float x0 = hb[cur][gI][sI][0], x1 = hb[cur][gI][sI][1];
                acc[0][0] = mfma16(a0, x0, acc[0][0]);
                acc[0][1] = mfma16(a0, x1, acc[0][1]);
                acc[1][0] = mfma16(a1, x0, acc[1][0]);
                acc[1][1] = mfma16(a1, x1, acc[1][1]);
            }
        __syncthreads();
    };
    for (int blk = 0; blk < n_blk; blk += 2) {
        block(blk, std::integral_constant<int, 0>{});
        if (blk + 1 < n_blk) block(blk + 1, std::integral_constant<int, 1>{});
    }
    if (!pos_ok) return;
    float* g = g_base + w_off[layer];                          // torch layout (o, c, p)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = 16 * x + 4 * kk + j, c = 16 * y + i16;
                if (o < H && c < H) g[((size_t)o * H + c) * L + p] = acc[x][y][j];
            }
}

void launch_highway_gwc(const float* dhw, const float* h, float* g_base, const long long* w_off, int n_rows, int L, int H, int layers, hipStream_t s) {
    hipLaunchKernelGGL(highway_gwc_kernel, dim3((L + 7) / 8, layers), dim3(512), 0, s, dhw, h, g_base, w_off, n_rows, L, H);
}

constexpr int HB_ROWS = 8;
__global__ __launch_bounds__(256) void highway_bwd_kernel(const float* __restrict__ dfeat, const float* __restrict__ feat, long long fs,
                                                          int feat_off, const float* __restrict__ wc, long long wc_layer,
                                                          float* __restrict__ dh, long long dh_layer, int n_rows, int R, int L, int H) {
    __shared__ float d[HB_ROWS][HPAD];
    const int layer = blockIdx.y, row0 = blockIdx.x * HB_ROWS, tid = threadIdx.x;
    {
        const int rr = tid >> 5, o = tid & 31, row = row0 + rr;
        d[rr][o] = (row < n_rows && o < H) ? dhw_of(dfeat, feat, fs, feat_off, layer, H, R, row, o) : 0.f;
    }
    __syncthreads();
    const float* w = wc + (size_t)layer * wc_layer;             // transposed copy WcT[p][c][o] (launch_pack_wct): 128-byte runs per thread
    for (int e = tid; e < L * HPAD; e += 256) {
        float acc[HB_ROWS];
#pragma unroll
        for (int rr = 0; rr < HB_ROWS; ++rr) acc[rr] = 0.f;
        const v4f* wv4 = (const v4f*)(w + (size_t)e * HPAD);
#pragma unroll
        for (int o4 = 0; o4 < HPAD / 4; ++o4) {
            const v4f wv = wv4[o4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rr = 0; rr < HB_ROWS; ++rr) acc[rr] = fmaf(d[rr][o4 * 4 + j], wv[j], acc[rr]);
        }
#pragma unroll
        for (int rr = 0; rr < HB_ROWS; ++rr)
            if (row0 + rr < n_rows) dh[(size_t)layer * dh_layer + (size_t)(row0 + rr) * L * HPAD + e] = acc[rr];
    }
}

void launch_highway_bwd(const float* dfeat, const float* feat, long long fs, int feat_off, const float* wc, long long wc_layer,
                        float* dh, long long dh_layer, int n_sites, int R, int L, int H, int layers, hipStream_t s) {
    const int n_rows = n_sites * R;
    hipLaunchKernelGGL(highway_bwd_kernel, dim3((n_rows + HB_ROWS - 1) / HB_ROWS, layers), dim3(256), 0, s, dfeat, feat, fs, feat_off, wc,
                       wc_layer, dh, dh_layer, n_rows, R, L, H);
}

// gWc[o][c][p] partial over a block of rows: thread = one (p, c) element, 32 outputs in registers
constexpr int HW_SPLITS = 8;
__global__ __launch_bounds__(256) void highway_wgrad_kernel(const float* __restrict__ dfeat, const float* __restrict__ feat, long long fs,
                                                            int feat_off, int layer, const float* __restrict__ h, float* __restrict__ partial,
                                                            int n_rows, int R, int L, int H) {
    __shared__ float d[32][HPAD];
    const int tid = threadIdx.x, split = blockIdx.y;
    const int e = blockIdx.x * 256 + tid, n_e = L * HPAD;
    const int per = (n_rows + HW_SPLITS - 1) / HW_SPLITS, lo = split * per, hi = min(n_rows, lo + per);
    float acc[HPAD];
#pragma unroll
    for (int o = 0; o < HPAD; ++o) acc[o] = 0.f;
    for (int r0 = lo; r0 < hi; r0 += 32) {
        __syncthreads();
        for (int i = tid; i < 32 * HPAD; i += 256) {
            const int rr = i >> 5, o = i & 31, row = r0 + rr;
            d[rr][o] = (row < hi && o < H) ? dhw_of(dfeat, feat, fs, feat_off, layer, H, R, row, o) : 0.f;
        }
        __syncthreads();
        if (e < n_e) {
            const int nr = min(32, hi - r0);
            for (int rr = 0; rr < nr; ++rr) {
                const float hv = h[(size_t)(r0 + rr) * n_e + e];
#pragma unroll
                for (int o = 0; o < HPAD; ++o) acc[o] = fmaf(d[rr][o], hv, acc[o]);
            }
        }
    }
    if (e < n_e) {
#pragma unroll
        for (int o = 0; o < HPAD; ++o) partial[((size_t)split * HPAD + o) * n_e + e] = acc[o];
    }
}

// gbc[o] = sum over rows of dhw[row][o]: one block per split of the rows, fixed-order tree inside, splits summed by the reduce
__global__ __launch_bounds__(256) void highway_bias_partial_kernel(const float* __restrict__ dfeat, const float* __restrict__ feat, long long fs,
                                                                  int feat_off, float* __restrict__ bias_partial, int n_rows, int R, int H) {
    __shared__ float red[8][HPAD];
    const int tid = threadIdx.x, o = tid & 31, lane_row = tid >> 5;
    // blockIdx.y = layer (every layer of a step in ONE launch; feat_off is layer 0's block of the feature row)
    const int layer = blockIdx.y;
    const int per = (n_rows + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = min(n_rows, lo + per);
    float sum = 0.f;
    if (o < H)
        for (int row = lo + lane_row; row < hi; row += 8) sum += dhw_of(dfeat, feat, fs, feat_off, layer, H, R, row, o);
    red[lane_row][o] = sum;
    __syncthreads();
    if (tid < HPAD) {
        float t = 0.f;
        for (int i = 0; i < 8; ++i) t += red[i][tid];
        bias_partial[((size_t)layer * gridDim.x + blockIdx.x) * HPAD + tid] = t;
    }
}

// g_base + g_off[layer] (or g_bc itself when g_off is null and the grid is one layer): the compression biases' gradients are not
// contiguous in the flat buffer
__global__ __launch_bounds__(64) void highway_bias_reduce_kernel(const float* __restrict__ bias_partial, int n_blocks, float* __restrict__ g_bc,
                                                                 const long long* __restrict__ g_off, int H) {
    const int o = threadIdx.x, layer = blockIdx.x;
    if (o >= H) return;
    const float* bp = bias_partial + (size_t)layer * n_blocks * HPAD;
    float* dst = g_off ? g_bc + g_off[layer] : g_bc;
    dst[o] = ordered_sum<float>(n_blocks, [&](int b) { return bp[b * HPAD + o]; });
}

__global__ __launch_bounds__(256) void highway_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ g_wc, int L, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;             // over (o, c, p) of the torch layout (H, H, 1, L)
    if (idx >= H * H * L) return;
    const int p = idx % L, c = (idx / L) % H, o = idx / (L * H);
    const int n_e = L * HPAD, e = p * HPAD + c;
    float sum = 0.f;
    for (int sp = 0; sp < HW_SPLITS; ++sp) sum += partial[((size_t)sp * HPAD + o) * n_e + e];
    g_wc[idx] = sum;
}

// gbc[o] = sum over rows of dhw[row][o] for ONE layer (feat_off points at that layer's block of the feature row)
void launch_highway_bias_grad(const float* dfeat, const float* feat, long long fs, int feat_off, float* partial, float* g_bc,
                              int n_sites, int R, int H, hipStream_t s) {
    constexpr int BIAS_BLOCKS = 64;
    const int n_rows = n_sites * R;
    hipLaunchKernelGGL(highway_bias_partial_kernel, dim3(BIAS_BLOCKS), dim3(256), 0, s, dfeat, feat, fs, feat_off, partial, n_rows, R, H);
    hipLaunchKernelGGL(highway_bias_reduce_kernel, dim3(1), dim3(64), 0, s, partial, BIAS_BLOCKS, g_bc, (const long long*)nullptr, H);
}
// every layer at once: partial [layers][64][HPAD]; the gradient of layer l's bias goes to g_base + g_off[l] (g_off on the device)
void launch_highway_bias_grad_all(const float* dfeat, const float* feat, long long fs, int feat_off, float* partial, float* g_base,
                                  const long long* g_off, int n_sites, int R, int H, int layers, hipStream_t s) {
    constexpr int BIAS_BLOCKS = 64;
    const int n_rows = n_sites * R;
    hipLaunchKernelGGL(highway_bias_partial_kernel, dim3(BIAS_BLOCKS, (unsigned)layers), dim3(256), 0, s, dfeat, feat, fs, feat_off, partial, n_rows, R, H);
    hipLaunchKernelGGL(highway_bias_reduce_kernel, dim3((unsigned)layers), dim3(64), 0, s, partial, BIAS_BLOCKS, g_base, g_off, H);
}

void launch_highway_wgrad(const float* dfeat, const float* feat, long long fs, int feat_off, const float* h, long long h_layer,
                          float* partial, float* g_wc, long long wc_layer, float* g_bc, int n_sites, int R, int L, int H, int layers,
                          int layer_stride_b, hipStream_t s) {
    const int n_rows = n_sites * R, n_e = L * HPAD;
    for (int l = 0; l < layers; ++l) {
        hipLaunchKernelGGL(highway_wgrad_kernel, dim3((n_e + 255) / 256, HW_SPLITS), dim3(256), 0, s, dfeat, feat, fs, feat_off, l,
                           h + (size_t)l * h_layer, partial, n_rows, R, L, H);
        hipLaunchKernelGGL(highway_wgrad_reduce_kernel, dim3((H * H * L + 255) / 256), dim3(256), 0, s, partial, g_wc + (size_t)l * wc_layer, L, H);
    }
    // (one layer per call: the compression layers' gradient tensors are not contiguous in the flat buffer)
    constexpr int BIAS_BLOCKS = 64;
    float* bias_partial = partial + (size_t)HW_SPLITS * HPAD * n_e;
    hipLaunchKernelGGL(highway_bias_partial_kernel, dim3(BIAS_BLOCKS), dim3(256), 0, s, dfeat, feat, fs, feat_off, bias_partial, n_rows, R, H);
    hipLaunchKernelGGL(highway_bias_reduce_kernel, dim3(1), dim3(64), 0, s, bias_partial, BIAS_BLOCKS, g_bc, (const long long*)nullptr, H);
    (void)layer_stride_b;
}

// ------------------------------------------------------------------------------------------------
// Layer 1's backward by bins (round 5).  conv1's input column is a SUM of terms -- token embeddings + positional encoding, q, strand, three
// mask flags -- so everything the step needs from dz_1 = d loss / d conv1 pre-activation is a handful of sums of dz_1:
//     S[p][o]            = sum over rows of dz[row][p][o]                                        (-> bias gradient, positional term)
//     BIN[kind][t][k][o] = sum over (row, p) with token_kind[row][p + t - 1] == k of dz[row][p][o]   (kind: read / reference token)
//     GS[t][j][o]        = sum over (row, p) of dz[row][p][o] * scalar_j[row][p + t - 1]             (q, strand, refmatch, varmatch, lenmask)
// from which  gW1[o][read emb e][t] = sum_k BIN[read][t][k][o] E[k][e] + sum_p S[p][o] pe[p + t - 1][e]  (ref emb: BIN[ref]),
// gW1[o][scalar j][t] = GS[t][j][o],  gb1[o] = sum_p S[p][o],  and the embedding gradient (padding_idx 0, scale_grad_by_freq)
// tot_read[k][e] = sum_{t,o} W1[o][e][t] BIN[read][t][k][o]  (what the conv's data gradient, summed per token, amounts to).
// One pass over dz_1 (two tensors, 1.3 GB at 64 sites) replaces the K = 1.3 M weight-gradient GEMM in encode form, the 3-tap
// data-gradient launch of layer 1 (whose only reader was the embedding gradient) and the embedding-gradient launches.
// Deterministic: a workgroup walks its rows in order, a thread owns its accumulators, partials are added in workgroup order in double.
// ------------------------------------------------------------------------------------------------
constexpr int L0B_S = MPOS * CPAD;                               // floats of a workgroup's partial: S [MPOS][CPAD]
constexpr int L0B_BIN = 2 * 3 * VOCAB * CPAD;                    //   BIN [kind 2][tap 3][token 10][CPAD]
constexpr int L0B_GS = 3 * 5 * CPAD;                             //   GS [tap 3][scalar 5][CPAD]
constexpr int L0B_CNT = 2 * 16;                                  //   token counts [kind 2][16]
constexpr int L0B_FLOATS = L0B_S + L0B_BIN + L0B_GS + L0B_CNT;   // 36 256 <= 3 CPAD CPAD: the weight-gradient partial buffer holds them
static_assert(L0B_FLOATS <= 3 * CPAD * CPAD, "layer-1 bins fit a slice of the weight-gradient partial buffer");
static_assert(L0B_FLOATS == L0_BINS_TOTALS, "dan_train.h sizes the totals");
// The binned sums ARE matrix products: BIN[kind][t] (and GS[t]) = A_kind x dz shifted by the tap, with A_kind [16][column] = ten one-hot
// rows of the column's token (kind: read / reference) and, kind 0 only, the five scalar channels -- sixteen rows: one MFMA tile.  Products
// with 0 / 1 are exact; a wave owns one 16-channel tile of all six (kind, tap) accumulators across the rows it walks.
constexpr int L0B_DZ_S = CPAD + 4;                               // floats per dz row in LDS: 4 rows of a B fragment hit disjoint banks
constexpr int L0B_A = 16;                                        // floats per column of A_kind

__global__ __launch_bounds__(512) void l0_bins_kernel(WgradArgs a, int n_sites) {
    extern __shared__ __attribute__((aligned(16))) float l0b_lds[];
    float* const dz = l0b_lds + L0B_DZ_S;                        // [-1 .. L][L0B_DZ_S]: a zero row either side of the read
    float* const amat = l0b_lds + (MPOS + 2) * L0B_DZ_S;         // [kind 2][MPOS][16]
    const int tid = threadIdx.x, L = a.L, R = a.R, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c4 = tid & 31, pr = tid >> 5;
    for (int i = tid; i < ((MPOS + 2) * L0B_DZ_S + 2 * MPOS * L0B_A) / 4; i += 512) ((v4f*)l0b_lds)[i] = splat(0.f);
    constexpr int NS = (MPOS + 15) / 16;                         // 13 sweeps of 16 columns
    v4f s_acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) s_acc[k] = splat(0.f);
    v4f acc[2][3];                                               // [kind][tap]: rows 4 (lane >> 4) + j of A x channel 16 wave + (lane & 15)
#pragma unroll
    for (int kd = 0; kd < 2; ++kd)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[kd][t] = splat(0.f);
    float cnt_read = 0.f, cnt_ref = 0.f;                         // (threads 0..9: occurrences of token tid)
    const Coef3 ck = load_coef(a.a_coef, c4 * 4);
    __syncthreads();
    for (int row = blockIdx.x; row < a.n_rows; row += gridDim.x) {
        const int site = row / R;
        // ---- stage dz of the row (the transform of the weight-gradient launches), add it to S; the row's A matrices
        {
            const v4f* s1 = (const v4f*)(a.a1 + (size_t)row * L * CPAD);
            const v4f* s2 = (const v4f*)(a.a2 + (size_t)row * L * CPAD);
            v4f r1[NS], r2[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int p = pr + 16 * k;
                r1[k] = p < L ? s1[(size_t)p * (CPAD / 4) + c4] : splat(0.f);
                r2[k] = p < L ? s2[(size_t)p * (CPAD / 4) + c4] : splat(0.f);
            }
            const size_t rbase = (size_t)row * L, sbase = (size_t)site * L;
            int tok = 0, q = 0, st = 0, rf = 0, rm = 0, vm = 0;
            if (tid < L) {
                tok = a.reads[rbase + tid]; q = a.qual[rbase + tid]; st = a.strand[rbase + tid];
                rf = a.ref[sbase + tid]; rm = a.ref_mask[sbase + tid]; vm = a.var_mask[sbase + tid];
            }
            const int agree_ref = __syncthreads_and((rm == 0) || (tok == rm));      // (also: the previous row's products are done with dz and A)
            const int agree_var = __syncthreads_and((vm == 0) || (tok == vm));
            if (tid < L) {
                const int tk = min(tok, VOCAB - 1), rk = min(rf, VOCAB - 1);
                float* a0 = amat + tid * L0B_A;
                float* a1 = amat + (MPOS + tid) * L0B_A;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    v4f v0, v1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v0[j] = (4 * g + j == tk) ? 1.f : 0.f; v1[j] = (4 * g + j == rk) ? 1.f : 0.f; }
                    if (g == 2) { v0[2] = (float)q * 0.01f; v0[3] = (float)st * 0.5f; }
                    if (g == 3) { v0[0] = (rm != 0 && agree_ref) ? 1.f : 0.f; v0[1] = (vm != 0 && agree_var) ? 1.f : 0.f; v0[2] = (rm != 0) ? 1.f : 0.f; v0[3] = 0.f; }
                    *(v4f*)(a0 + 4 * g) = v0; *(v4f*)(a1 + 4 * g) = v1;
                }
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int p = pr + 16 * k;
                if (p < L) {
                    const v4f v = apply_transform(r1[k], r2[k], ck, a.a_coef != nullptr, a.a_mask);
                    *(v4f*)(dz + p * L0B_DZ_S + c4 * 4) = v;
                    s_acc[k] += v;
                }
            }
        }
        __syncthreads();
        // ---- acc[kind][t] += A_kind[.][column q] x dz[q - t + 1][.] over the columns, four per MFMA: A: row lane & 15, column 4 s + (lane >> 4)
        {
            const int m = lane & 15, kq = lane >> 4;
            const float* pa0 = amat + kq * L0B_A + m;
            const float* pa1 = pa0 + MPOS * L0B_A;
            const float* pb = dz + (kq + 1) * L0B_DZ_S + 16 * wave + m;          // tap t reads row q - t + 1
            const int n_steps = (L + 3) >> 2;
            for (int sI = 0; sI < n_steps; ++sI) {
                const float a0 = pa0[sI * 4 * L0B_A], a1 = pa1[sI * 4 * L0B_A];
                float bt[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) bt[t] = pb[(sI * 4 - t) * L0B_DZ_S];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc[0][t] = mfma16(a0, bt[t], acc[0][t]);
                    acc[1][t] = mfma16(a1, bt[t], acc[1][t]);
                }
            }
        }
        if (tid < VOCAB) {
            for (int p = 0; p < L; ++p) {
                cnt_read += amat[p * L0B_A + tid];
                if (row == site * R) cnt_ref += amat[(MPOS + p) * L0B_A + tid];              // the ref lookup: once per site
            }
        }
    }
    __syncthreads();
    // ---- this workgroup's partial
    float* out = a.partial + (size_t)blockIdx.x * (3 * CPAD * CPAD);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int p = pr + 16 * k;
        if (p < MPOS) *(v4f*)(out + p * CPAD + c4 * 4) = s_acc[k];
    }
    {
        const int n = lane & 15, g = lane >> 4, o = 16 * wave + n;
#pragma unroll
        for (int kd = 0; kd < 2; ++kd)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;                     // row of A
                    if (r < VOCAB) out[L0B_S + ((size_t)(kd * 3 + t) * VOCAB + r) * CPAD + o] = acc[kd][t][j];
                    else if (kd == 0 && r < VOCAB + 5) out[L0B_S + L0B_BIN + ((size_t)t * 5 + (r - VOCAB)) * CPAD + o] = acc[kd][t][j];
                }
    }
    if (tid < VOCAB) { out[L0B_S + L0B_BIN + L0B_GS + tid] = cnt_read; out[L0B_S + L0B_BIN + L0B_GS + 16 + tid] = cnt_ref; }
    else if (tid < 16 || (tid >= 16 + VOCAB && tid < 32)) out[L0B_S + L0B_BIN + L0B_GS + tid] = 0.f;     // (the unused count slots)
}

// sums of the workgroups' partials in workgroup order (double)
__global__ __launch_bounds__(256) void l0_bins_reduce_kernel(const float* __restrict__ partial, int wgs, double* __restrict__ tot) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L0B_FLOATS) return;
    tot[i] = ordered_sum<double>(wgs, [&](int w) { return (double)partial[(size_t)w * (3 * CPAD * CPAD) + i]; });
}

// gW1 (reference layout [cout][cin][3]), gb1, g_emb from the totals
__global__ __launch_bounds__(256) void l0_grads_kernel(const double* __restrict__ tot, const float* __restrict__ emb, const float* __restrict__ pe,
                                                       const float* __restrict__ w1, const int* __restrict__ canon, int L, int n_out, int n_in,
                                                       float* __restrict__ g_w, float* __restrict__ g_b, float* __restrict__ g_emb) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const double* S = tot;
    const double* BIN = tot + L0B_S;
    const double* GS = tot + L0B_S + L0B_BIN;
    const double* CNT = tot + L0B_S + L0B_BIN + L0B_GS;
    const int n_w = 3 * n_out * n_in;
    if (idx < n_w) {
        const int i = idx % n_in, o = (idx / n_in) % n_out, t = idx / (n_in * n_out);
        const int c = canon[i];                                  // canonical channel of the reference's input channel i
        double g = 0.0;
        if (c < 2 * EMBED) {
            const int kind = c / EMBED, e = c - kind * EMBED;
            for (int k = 0; k < VOCAB; ++k) g += BIN[((size_t)(kind * 3 + t) * VOCAB + k) * CPAD + o] * (double)emb[k * EMBED + e];
            for (int p = 0; p < L; ++p) {
                const int pq = p + t - 1;
                if (pq >= 0 && pq < L) g += S[(size_t)p * CPAD + o] * (double)pe[(size_t)pq * EMBED + e];
            }
        } else {
            const int j = c - 2 * EMBED;
            g = GS[((size_t)t * 5 + j) * CPAD + o];
        }
        g_w[((size_t)o * n_in + i) * 3 + t] = (float)g;
    } else if (idx < n_w + n_out) {
        const int o = idx - n_w;
        double g = 0.0;
        for (int p = 0; p < L; ++p) g += S[(size_t)p * CPAD + o];
        g_b[o] = (float)g;
    } else if (idx < n_w + n_out + VOCAB * EMBED) {
        const int j = idx - n_w - n_out, k = j / EMBED, e = j % EMBED;
        double g = 0.0;
        if (k != 0) {                                            // padding_idx = base_enum['pad'] = 0
            const double cr = CNT[k], cf = CNT[16 + k];
            double tr = 0.0, tf = 0.0;
            for (int t = 0; t < 3; ++t)
                for (int o = 0; o < n_out; ++o) {
                    tr += (double)w1[((size_t)o * n_in + e) * 3 + t] * BIN[((size_t)(0 * 3 + t) * VOCAB + k) * CPAD + o];
                    tf += (double)w1[((size_t)o * n_in + EMBED + e) * 3 + t] * BIN[((size_t)(1 * 3 + t) * VOCAB + k) * CPAD + o];
                }
            if (cr > 0.0) g += tr / cr;
            if (cf > 0.0) g += tf / cf;
        }
        g_emb[j] = (float)g;
    }
}

int l0_bins_lds_bytes() { return ((MPOS + 2) * L0B_DZ_S + 2 * MPOS * L0B_A) * (int)sizeof(float); }

// Returns 0, or the hipError_t of the attribute call when the device refuses the kernel's dynamic LDS size (nothing is launched then).
// The attribute belongs to (function, device): set on every call -- one driver table lookup -- rather than once per process, which
// would leave a second device of the same process without it.
int launch_l0_backward(const WgradArgs& a, int n_sites, double* tot, const float* w1, const int* canon, int n_out, int n_in, float* g_w,
                       float* g_b, float* g_emb, hipStream_t s) {
    const hipError_t e = hipFuncSetAttribute((const void*)l0_bins_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, l0_bins_lds_bytes());
    if (e != hipSuccess) return (int)e;
    const int wgs = a.n_rows < TRAIN_PARTIAL_WGS ? a.n_rows : TRAIN_PARTIAL_WGS;
    hipLaunchKernelGGL(l0_bins_kernel, dim3(wgs), dim3(512), l0_bins_lds_bytes(), s, a, n_sites);
    hipLaunchKernelGGL(l0_bins_reduce_kernel, dim3((L0B_FLOATS + 255) / 256), dim3(256), 0, s, a.partial, wgs, tot);
    const int total = 3 * n_out * n_in + n_out + VOCAB * EMBED;
    hipLaunchKernelGGL(l0_grads_kernel, dim3((total + 255) / 256), dim3(256), 0, s, tot, a.emb, a.pe, w1, canon, a.L, n_out, n_in, g_w, g_b, g_emb);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Layer 1's forward by table (the inference path's walk, dan_kernels.h L0_*): the weights change every step, so the tables -- token pairs,
// positional term in three variants, scalar-channel rows -- are rebuilt from the current conv1 weights and embeddings by one small launch
// per step; the walk writes a_1 = relu(conv1 + bias) and the row's BatchNorm sums.  Replaces the encode-form 3-tap row launch of layer 1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l0_train_tables_kernel(const float* __restrict__ w1, const int* __restrict__ inv, const float* __restrict__ emb,
                                                              const float* __restrict__ pe, int L, int n_out, int n_in, float* __restrict__ tab) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    auto W = [&](int o, int c, int t) -> double { return (o < n_out && inv[c] >= 0) ? (double)w1[((size_t)o * n_in + inv[c]) * 3 + t] : 0.0; };
    const int n_tj = 3 * L0_NTJ * CPAD, n_sc = 5 * 3 * CPAD, n_pe = 3 * L * CPAD;
    if (idx < n_tj) {
        const int o = idx % CPAD, ent = (idx / CPAD) % L0_NTJ, t = idx / (CPAD * L0_NTJ);
        double v = 0.0;
        if (ent < 100) {
            const int ta = ent / 10, tb = ent % 10;
            for (int e = 0; e < EMBED; ++e) v += W(o, e, t) * (double)emb[ta * EMBED + e] + W(o, EMBED + e, t) * (double)emb[tb * EMBED + e];
        }
        tab[L0_TJ_OFF + idx] = (float)v;
    } else if (idx < n_tj + n_sc) {
        const int j = idx - n_tj, o = j % CPAD, kt = j / CPAD, k = kt / 3, t = kt % 3;
        tab[L0_WSC_OFF + j] = (float)W(o, 2 * EMBED + k, t);
    } else if (idx < n_tj + n_sc + n_pe) {
        const int j = idx - n_tj - n_sc, o = j % CPAD, p = (j / CPAD) % L, var = j / (CPAD * L);
        double v = 0.0;
        for (int t = 0; t < 3; ++t) {
            const int pq = p + t - 1;
            if (pq < 0 || pq >= L || (t == 0 && var == 1) || (t == 2 && var == 2)) continue;
            for (int e = 0; e < EMBED; ++e) v += (W(o, e, t) + W(o, EMBED + e, t)) * (double)pe[(size_t)pq * EMBED + e];
        }
        tab[L0_PE_OFF + j] = (float)v;
    }
}

__global__ __launch_bounds__(512) void l0_train_forward_kernel(EncodeSrc e, const float* __restrict__ tab, const float* __restrict__ bias_p, int R, int L,
                                                               float* __restrict__ a_out, float* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) float tokf[(MPOS + 2) * L0_TOK];
    __shared__ __attribute__((aligned(16))) float red[16][2][CPAD];
    const int row = blockIdx.x, site = row / R, tid = threadIdx.x;
    const size_t rbase = (size_t)row * L, sbase = (size_t)site * L;
    int tok = 0, rm = 0, vm = 0;
    if (tid < L) { tok = e.reads[rbase + tid]; rm = e.ref_mask[sbase + tid]; vm = e.var_mask[sbase + tid]; }
    const int agree_ref = __syncthreads_and((rm == 0) || (tok == rm));      // model.py:592-593
    const int agree_var = __syncthreads_and((vm == 0) || (tok == vm));
    if (tid <= L + 1) {                                          // thread j stages column j - 1; the columns either side of the read: the zero entry
        const int pc = tid - 1;
        const bool inside = pc >= 0 && pc < L;
        int tk = 0, qq = 0, ss = 0, rr = 0, mm = 0, vv = 0;
        if (inside) {
            tk = e.reads[rbase + pc]; qq = e.qual[rbase + pc]; ss = e.strand[rbase + pc];
            rr = e.ref[sbase + pc]; mm = e.ref_mask[sbase + pc]; vv = e.var_mask[sbase + pc];
        }
        const v4f t0 = {__builtin_bit_cast(float, inside ? min(tk, VOCAB - 1) * 10 + min(rr, VOCAB - 1) : 100), (float)qq * 0.01f, (float)ss * 0.5f,
                        (inside && mm != 0) ? 1.f : 0.f};
        const v4f t1 = {(inside && vv != 0 && agree_var) ? 1.f : 0.f, (tid == 0 && agree_ref) ? 1.f : 0.f, 0.f, 0.f};
        *(v4f*)(tokf + tid * L0_TOK) = t0; *(v4f*)(tokf + tid * L0_TOK + 4) = t1;
    }
    __syncthreads();
    const int c4 = tid & 31, pr = tid >> 5;
    const v4f bias = *(const v4f*)(bias_p + c4 * 4);
    v4f wq[3], wst[3], wlen[3], wvar[3];
    {
        const float agree = tokf[5];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            auto wv = [&](int k) { return *(const v4f*)(tab + L0_WSC_OFF + (k * 3 + t) * CPAD + c4 * 4); };
            wq[t] = wv(0); wst[t] = wv(1); wvar[t] = wv(3);
            wlen[t] = wv(4) + agree * wv(2);
        }
    }
    const v4f* tj = (const v4f*)(tab + L0_TJ_OFF) + c4;
    const v4f* pev = (const v4f*)(tab + L0_PE_OFF) + c4;
    v4f s0 = splat(0.f), s1 = splat(0.f);
    v4f* out = (v4f*)(a_out + (size_t)row * L * CPAD) + c4;
    constexpr int NS = (MPOS + 15) / 16;
#pragma unroll 4
    for (int k = 0; k < NS; ++k) {
        const int p = pr + 16 * k;
        if (p < L) {
            const int var = (p == 0) ? 1 : (p == L - 1) ? 2 : 0;
            v4f acc = bias + pev[((size_t)var * L + p) * (CPAD / 4)];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const v4f sa = *(const v4f*)(tokf + (p + t) * L0_TOK);
                const float sv = tokf[(p + t) * L0_TOK + 4];
                acc += tj[(size_t)(t * L0_NTJ + __builtin_bit_cast(int, sa[0])) * (CPAD / 4)];
                acc += sa[1] * wq[t] + sa[2] * wst[t] + sa[3] * wlen[t] + sv * wvar[t];
            }
            v4f v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[j], 0.f);
            out[(size_t)p * (CPAD / 4)] = v;
            s0 += v; s1 += v * v;
        }
    }
    if (stats) {                                                 // the row's sums, the sixteen position classes added in order
        *(v4f*)(&red[pr][0][c4 * 4]) = s0; *(v4f*)(&red[pr][1][c4 * 4]) = s1;
        __syncthreads();
        if (tid < 2 * CPAD) {
            const int st = tid / CPAD, ch = tid % CPAD;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += red[i][st][ch];
            stats[((size_t)row * 2 + st) * CPAD + ch] = sum;
        }
    }
}

void launch_l0_train_tables(const float* w1, const int* inv, const float* emb, const float* pe, int L, int n_out, int n_in, float* tab, hipStream_t s) {
    const int total = 3 * L0_NTJ * CPAD + 5 * 3 * CPAD + 3 * L * CPAD;
    hipLaunchKernelGGL(l0_train_tables_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w1, inv, emb, pe, L, n_out, n_in, tab);
}

int launch_l0_train_forward(const RowArgs& enc, const float* tab, const float* bias, int n_rows, float* a_out, float* stats, hipStream_t s) {
    EncodeSrc e{enc.reads, enc.qual, enc.strand, enc.ref, enc.ref_mask, enc.var_mask, enc.emb, enc.pe};
    hipLaunchKernelGGL(l0_train_forward_kernel, dim3(n_rows), dim3(512), 0, s, e, tab, bias, enc.R, enc.L, a_out, stats);
    return n_rows;                                               // entries of stats written
}

// ------------------------------------------------------------------------------------------------
// FC stack: tiled MFMA GEMM  C[m][n] = sum_k opA(m,k) opB(n,k) (+ bias[n], ReLU)
// 128 x 128 x 32 tiles through LDS as in the inference fc_kernel; an operand that is K-slow in memory (X[k*ld + i]: the
// transposed uses of the backward pass) is staged as [k][i] rows and its fragments are read with four ds_read_b32 (row
// stride 132: the two k-groups of a 32-lane half sit 4 rows = 16 banks apart) instead of one ds_read_b128.
// ------------------------------------------------------------------------------------------------
constexpr int GM = 128, GN = 128, GK = 32, GS_KC = GK + 8, GS_KS = GM + 4;
template <bool A_KSLOW, bool B_KSLOW>
__global__ __launch_bounds__(512) void gemm_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb,
                                                   const float* __restrict__ bias, float* __restrict__ C, long long ldc, int M, int N,
                                                   int K, int relu, int tiles_n, int k_per_split, float* __restrict__ split_out) {
    constexpr int SZA = A_KSLOW ? GK * GS_KS : GM * GS_KC, SZB = B_KSLOW ? GK * GS_KS : GN * GS_KC;
    __shared__ __attribute__((aligned(16))) float sa[2][SZA], sb[2][SZB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int bm = (blockIdx.x / tiles_n) * GM, bn = (blockIdx.x % tiles_n) * GN;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32;
    const int M4 = (M + 3) & ~3, N4 = (N + 3) & ~3;
    v4f ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (A_KSLOW) {                                       // rows k0 + (tid >> 5) + 16 i, columns bm + 4 (tid & 31)
                const int k = k0 + (tid >> 5) + 16 * i, m = bm + 4 * (tid & 31);
                ra[i] = (k < K && m < M4) ? *(const v4f*)(A + (size_t)k * lda + m) : splat(0.f);
            } else {                                             // rows bm + (tid >> 3) + 64 i, k = k0 + 4 (tid & 7)
                const int m = min(bm + (tid >> 3) + 64 * i, M - 1), k = k0 + 4 * (tid & 7);
                ra[i] = (k < K) ? *(const v4f*)(A + (size_t)m * lda + k) : splat(0.f);
            }
            if (B_KSLOW) {
                const int k = k0 + (tid >> 5) + 16 * i, n = bn + 4 * (tid & 31);
                rb[i] = (k < K && n < N4) ? *(const v4f*)(B + (size_t)k * ldb + n) : splat(0.f);
            } else {
                const int n = min(bn + (tid >> 3) + 64 * i, N - 1), k = k0 + 4 * (tid & 7);
                rb[i] = (k < K) ? *(const v4f*)(B + (size_t)n * ldb + k) : splat(0.f);
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (A_KSLOW) *(v4f*)(&sa[buf][((tid >> 5) + 16 * i) * GS_KS + 4 * (tid & 31)]) = ra[i];
            else *(v4f*)(&sa[buf][((tid >> 3) + 64 * i) * GS_KC + 4 * (tid & 7)]) = ra[i];
            if (B_KSLOW) *(v4f*)(&sb[buf][((tid >> 5) + 16 * i) * GS_KS + 4 * (tid & 31)]) = rb[i];
            else *(v4f*)(&sb[buf][((tid >> 3) + 64 * i) * GS_KC + 4 * (tid & 7)]) = rb[i];
        }
    };
    v4f acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = splat(0.f);
    // split-K: blockIdx.y owns k in [k_lo, k_hi); its partial tile goes to split_out[split][M][N] (summed by
    // gemm_split_reduce_kernel in split order, then bias / ReLU)
    const int k_lo = blockIdx.y * k_per_split, k_hi = min(K, k_lo + k_per_split);
    const int KT = (k_hi - k_lo + GK - 1) / GK;
    const int Kfull = K;
    K = k_hi;
    gload(k_lo);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) gload(k_lo + (kt + 1) * GK);
#pragma unroll
        for (int g = 0; g < GK / 16; ++g) {
            v4f a[4], w[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (A_KSLOW) {
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx) a[i][sidx] = sa[cur][(16 * g + 4 * kk + sidx) * GS_KS + wm + 16 * i + r16];
                } else {
                    a[i] = *(const v4f*)(&sa[cur][(wm + 16 * i + r16) * GS_KC + 16 * g + 4 * kk]);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (B_KSLOW) {
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx) w[j][sidx] = sb[cur][(16 * g + 4 * kk + sidx) * GS_KS + wn + 16 * j + r16];
                } else {
                    w[j] = *(const v4f*)(&sb[cur][(wn + 16 * j + r16) * GS_KC + 16 * g + 4 * kk]);
                }
            }
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(w[j][sidx], a[i][sidx], acc[i][j]);
        }
        if (kt + 1 < KT) sstore(cur ^ 1);
        __syncthreads();
    }
    (void)Kfull;
    // The B operand goes FIRST into the MFMA: the result tile is then C^T, a lane holds C[m = .. + r16][n = .. + 4 kk + 0..3] --
    // four consecutive columns, one 16-byte store (with A first a lane held four ROWS of one column: 4-byte stores, and the
    // store-bound shapes -- the FC1 weight gradient's 269 MB, the highway's dh -- ran at 1.5-2.6 TB/s).  Same products, same sums.
    const bool vec = ((N & 3) == 0) && (split_out || ((ldc & 3) == 0));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n0 = bn + wn + j * 16 + kk * 4;
        if (n0 >= N) continue;
        v4f b = splat(0.f);
        if (bias && !split_out) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) b[jj] = (n0 + jj < N) ? bias[n0 + jj] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = bm + wm + i * 16 + r16;
            if (m >= M) continue;
            v4f v = acc[i][j] + b;
            if (!split_out && relu) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = fmaxf(v[jj], 0.f);
            }
            float* dst = split_out ? split_out + ((size_t)blockIdx.y * M + m) * N + n0 : C + (size_t)m * ldc + n0;
            if (vec) *(v4f*)dst = v;
            else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) if (n0 + jj < N) dst[jj] = v[jj];
            }
        }
    }
}

__global__ __launch_bounds__(256) void gemm_split_reduce_kernel(const float* __restrict__ split_out, int splits, const float* __restrict__ bias,
                                                                float* __restrict__ C, long long ldc, int M, int N, int relu) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * N) return;
    const int m = idx / N, n = idx % N;
    float v = 0.f;
    for (int sp = 0; sp < splits; ++sp) v += split_out[((size_t)sp * M + m) * N + n];
    if (bias) v += bias[n];
    if (relu) v = fmaxf(v, 0.f);
    C[(size_t)m * ldc + n] = v;
}

// ------------------------------------------------------------------------------------------------
// FC1 at a FEW sites per GPU (batch 80 over 8 GPUs = 10): three products that each move the 1024 x 73 856 weight matrix (or its
// gradient) once and do 10 multiply-adds per element -- weight streaming, not matrix work.  On the tiled MFMA kernel they take
// 130-210 us each (1.3-2 TB/s, at any batch size: its k-tile is 128 bytes of each of 128 rows); here a wave instruction moves ONE
// KILOBYTE OF ONE ROW and the few-row operand sits in LDS, registers or scalar registers.  M <= 16.
//   forward        out[b][n]  = sum_k x[b][k] W[n][k]        x slab in LDS, a wave per weight row, lane sums reduced per row
//   data gradient  dx[b][k]   = sum_n d[b][n] W[n][k]        M accumulator vectors per lane, d[b][n] wave-uniform
//   weight grad.   gW[n][k]   = sum_b d[b][n] x[b][k]        x slab in registers, rows written as they are formed
// K-slabs of 1024 floats (a 256-thread workgroup = four 1-KB instructions wide); partials in launch_gemm's split layout.
// ------------------------------------------------------------------------------------------------
constexpr int SK_MAXM = 16, SK_SLAB = 1024, SK_THREADS = 256, SK_DROWS = 128;
// sum over the 64 lanes of a wave, every lane gets it: four DPP steps inside the 16-lane rows (one instruction each once the
// compiler folds the move into the add), two cross-row steps through the permute network
__device__ __forceinline__ float wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__global__ __launch_bounds__(SK_THREADS) void fc_skinny_fwd_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ W, long long ldw,
                                                                   float* __restrict__ part, int M, int N, int K, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [M][SK_SLAB]: sized by the launch, so that 3-4 workgroups share a CU at 10 sites
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * SK_SLAB;
    for (int i = tid; i < M * (SK_SLAB / 4); i += SK_THREADS) {
        const int b = i / (SK_SLAB / 4), k = k0 + 4 * (i - b * (SK_SLAB / 4));
        ((v4f*)xs)[i] = (k < K) ? *(const v4f*)(x + (size_t)b * ldx + k) : splat(0.f);
    }
    __syncthreads();
    const int n_lo = blockIdx.y * rows_per_wg, n_hi = min(N, n_lo + rows_per_wg);
    auto load_row = [&](int n, v4f (&w)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + 4 * (j * 64 + lane);
            w[j] = (n < n_hi && k < K) ? *(const v4f*)(W + (size_t)n * ldw + k) : splat(0.f);
        }
    };
    v4f w[4], w1[4], w2[4];
    load_row(n_lo + wave, w);
    load_row(n_lo + wave + 4, w1);
    for (int n = n_lo + wave; n < n_hi; n += 4) {
        load_row(n + 8, w2);                                    // two rows of this wave ahead: they travel under this one's arithmetic
        float mine = 0.f;
#pragma unroll
        for (int b = 0; b < SK_MAXM; ++b) {
            if (b < M) {
                v4f a4 = splat(0.f);
#pragma unroll
                for (int j = 0; j < 4; ++j) a4 += w[j] * *(const v4f*)(xs + b * SK_SLAB + 4 * (j * 64 + lane));
                const float t = wave_sum((a4[0] + a4[1]) + (a4[2] + a4[3]));
                mine = (lane == b) ? t : mine;
            }
        }
        if (lane < M) part[((size_t)blockIdx.x * M + lane) * N + n] = mine;
#pragma unroll
        for (int j = 0; j < 4; ++j) { w[j] = w1[j]; w1[j] = w2[j]; }
    }
}

__global__ __launch_bounds__(SK_THREADS) void fc_skinny_dgrad_kernel(const float* __restrict__ d, long long ldd, const float* __restrict__ W, long long ldw,
                                                                     float* __restrict__ part, int M, int N, int K, int rows_per_wg) {
    // out[b][f] = sum over n in this workgroup's rows of d[b][n] W[n][f];  N = output columns (f), K = weight rows (n)
    // the workgroup's d[b][rows] in LDS (rows_per_wg <= SK_DROWS, a multiple of 4; zero past the range): four rows' factors of a
    // b are ONE broadcast 16-byte read (as forty scalar loads per four rows the loop waited on the scalar cache every iteration)
    __shared__ __attribute__((aligned(16))) float ds[SK_MAXM * SK_DROWS];
    const int tid = threadIdx.x;
    const int f = blockIdx.x * SK_SLAB + 4 * tid;
    const int n_lo = blockIdx.y * rows_per_wg, n_hi = min(K, n_lo + rows_per_wg);
    for (int i = tid; i < M * SK_DROWS; i += SK_THREADS) {
        const int b = i / SK_DROWS, r = i - b * SK_DROWS;
        ds[i] = (n_lo + r < n_hi) ? d[(size_t)b * ldd + n_lo + r] : 0.f;
    }
    __syncthreads();
    v4f acc[SK_MAXM];
#pragma unroll
    for (int b = 0; b < SK_MAXM; ++b) acc[b] = splat(0.f);
    const bool in = f < N;
    auto load4 = [&](int n, v4f (&w)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = (in && n + r < n_hi) ? *(const v4f*)(W + (size_t)(n + r) * ldw + f) : splat(0.f);
    };
    v4f w[4], wn[4];
    load4(n_lo, w);
    for (int n = n_lo; n < n_hi; n += 4) {
        load4(n + 4, wn);
#pragma unroll
        for (int b = 0; b < SK_MAXM; ++b) {
            if (b < M) {
                const v4f dv = *(const v4f*)(ds + b * SK_DROWS + (n - n_lo));
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[b] += splat(dv[r]) * w[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = wn[r];
    }
    if (in) {
#pragma unroll
        for (int b = 0; b < SK_MAXM; ++b)
            if (b < M) *(v4f*)(part + ((size_t)blockIdx.y * M + b) * N + f) = acc[b];
    }
}

__global__ __launch_bounds__(SK_THREADS) void fc_skinny_wgrad_kernel(const float* __restrict__ d, long long ldd, const float* __restrict__ x, long long ldx,
                                                                     float* __restrict__ C, long long ldc, int M, int N, int K, int rows_per_wg) {
    // C[n][f] = sum_b d[b][n] x[b][f];  M = rows n of C, N = columns f, K = the few b
    const int tid = threadIdx.x;
    const int f = blockIdx.x * SK_SLAB + 4 * tid;
    if (f >= N) return;
    v4f xr[SK_MAXM];
#pragma unroll
    for (int b = 0; b < SK_MAXM; ++b) xr[b] = (b < K) ? *(const v4f*)(x + (size_t)b * ldx + f) : splat(0.f);
    const int n_lo = blockIdx.y * rows_per_wg, n_hi = min(M, n_lo + rows_per_wg);
    for (int n = n_lo; n < n_hi; ++n) {
        v4f v = splat(0.f);
#pragma unroll
        for (int b = 0; b < SK_MAXM; ++b)
            if (b < K) v += splat(d[(size_t)b * ldd + n]) * xr[b];
        *(v4f*)(C + (size_t)n * ldc + f) = v;
    }
}

void launch_gemm(const float* A, long long lda, int a_kslow, const float* B, long long ldb, int b_kslow, const float* bias,
                 float* C, long long ldc, int M, int N, int K, int relu, float* split_ws, long long split_ws_floats, hipStream_t s) {
    // FC1 at a few sites per GPU: the three weight-streaming forms (fc_skinny_*_kernel); vector accesses need the 16-byte
    // alignment the trainer's padded strides give
    const bool al = ((lda | ldb | ldc) & 3) == 0;
    if (split_ws && al && !a_kslow && !b_kslow && M <= SK_MAXM && K >= 8192 && (long long)((K + SK_SLAB - 1) / SK_SLAB) * M * N <= split_ws_floats) {
        const int slabs = (K + SK_SLAB - 1) / SK_SLAB, groups = std::max(1, std::min(N / 16, (768 + slabs - 1) / slabs));
        const int rows = (N + groups - 1) / groups;
        hipLaunchKernelGGL(fc_skinny_fwd_kernel, dim3((unsigned)slabs, (unsigned)((N + rows - 1) / rows)), dim3(SK_THREADS), (size_t)M * SK_SLAB * sizeof(float), s, A, lda, B, ldb, split_ws, M, N, K, rows);
        hipLaunchKernelGGL(gemm_split_reduce_kernel, dim3((M * N + 255) / 256), dim3(256), 0, s, split_ws, slabs, bias, C, ldc, M, N, relu);
        return;
    }
    if (split_ws && al && !a_kslow && b_kslow && M <= SK_MAXM && N >= 8192 && (N & 3) == 0 && !bias && !relu) {
        const int slabs = (N + SK_SLAB - 1) / SK_SLAB;
        int groups = std::max(1, std::min(K / 16, (768 + slabs - 1) / slabs));
        while (groups > 1 && (long long)groups * M * N > split_ws_floats) --groups;
        const int rows = ((K + groups - 1) / groups + 3) & ~3;
        groups = (K + rows - 1) / rows;
        if ((long long)groups * M * N <= split_ws_floats && rows <= SK_DROWS) {
            hipLaunchKernelGGL(fc_skinny_dgrad_kernel, dim3((unsigned)slabs, (unsigned)groups), dim3(SK_THREADS), 0, s, A, lda, B, ldb, split_ws, M, N, K, rows);
            hipLaunchKernelGGL(gemm_split_reduce_kernel, dim3((M * N + 255) / 256), dim3(256), 0, s, split_ws, groups, bias, C, ldc, M, N, relu);
            return;
        }
    }
    if (al && a_kslow && b_kslow && K <= SK_MAXM && N >= 8192 && (N & 3) == 0 && !bias && !relu) {
        const int slabs = (N + SK_SLAB - 1) / SK_SLAB, groups = std::max(1, std::min(M / 16, (768 + slabs - 1) / slabs));
        const int rows = (M + groups - 1) / groups;
        hipLaunchKernelGGL(fc_skinny_wgrad_kernel, dim3((unsigned)slabs, (unsigned)((M + rows - 1) / rows)), dim3(SK_THREADS), 0, s, A, lda, B, ldb, C, ldc, M, N, K, rows);
        return;
    }
    const int tiles_m = (M + GM - 1) / GM, tiles_n = (N + GN - 1) / GN, tiles = tiles_m * tiles_n;
    // too few tiles to fill 256 CUs and a long K: split K (deterministic: partials summed in split order)
    // (More splits do not help the skinny products: FC1's forward takes 166 us at 32, 64 and 256 splits alike -- 189 us + a 66-us
    // reduction at 256 -- and the highway weight gradient 91 us at 5 splits, 104 us at 10: a k-tile is 128 bytes of each of 128
    // rows, and that access pattern, not the number of workgroups, sets the rate.)
    int splits = 1;
    if (split_ws && tiles < 64 && K >= 16 * GK) {               // (from K = 512 on: the 10-site step's highway gradients have K = 1000)
        splits = std::min(256 / tiles, K / (8 * GK));
        while (splits > 1 && (long long)splits * M * N > split_ws_floats) --splits;
    }
    int k_per = K;
    if (splits > 1) {
        k_per = ((K + splits - 1) / splits + GK - 1) / GK * GK;
        splits = (K + k_per - 1) / k_per;
    }
    float* so = splits > 1 ? split_ws : nullptr;
    const dim3 grid((unsigned)tiles, (unsigned)splits), blk(512);
    if (!a_kslow && !b_kslow) hipLaunchKernelGGL((gemm_kernel<false, false>), grid, blk, 0, s, A, lda, B, ldb, bias, C, ldc, M, N, K, relu, tiles_n, k_per, so);
    else if (!a_kslow && b_kslow) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, blk, 0, s, A, lda, B, ldb, bias, C, ldc, M, N, K, relu, tiles_n, k_per, so);
    else if (a_kslow && !b_kslow) hipLaunchKernelGGL((gemm_kernel<true, false>), grid, blk, 0, s, A, lda, B, ldb, bias, C, ldc, M, N, K, relu, tiles_n, k_per, so);
    else hipLaunchKernelGGL((gemm_kernel<true, true>), grid, blk, 0, s, A, lda, B, ldb, bias, C, ldc, M, N, K, relu, tiles_n, k_per, so);
    if (splits > 1)
        hipLaunchKernelGGL(gemm_split_reduce_kernel, dim3((M * N + 255) / 256), dim3(256), 0, s, so, splits, bias, C, ldc, M, N, relu);
}

// ------------------------------------------------------------------------------------------------
// elementwise pieces of the FC stack
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask, float scale,
                                                      float* __restrict__ y, int rows, int cols, long long ld) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)rows * cols) return;
    const int r = (int)(idx / cols), c = (int)(idx % cols);
    const float m = mask ? (float)mask[idx] * scale : 1.f;
    y[(size_t)r * ld + c] = x[(size_t)r * ld + c] * m;
}

void launch_dropout(const float* x, const uint8_t* mask, float scale, float* y, int rows, int cols, long long ld, hipStream_t s) {
    const long long n = (long long)rows * cols;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, mask, scale, y, rows, cols, ld);
}

__global__ __launch_bounds__(256) void dropout_relu_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask, float scale,
                                                               const float* __restrict__ act, int relu, float* __restrict__ dx, int rows,
                                                               int cols, long long ld) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)rows * cols) return;
    const int r = (int)(idx / cols), c = (int)(idx % cols);
    const size_t o = (size_t)r * ld + c;
    float v = dy[o] * (mask ? (float)mask[idx] * scale : 1.f);
    if (relu && !(act[o] > 0.f)) v = 0.f;
    dx[o] = v;
}

void launch_dropout_relu_bwd(const float* dy, const uint8_t* mask, float scale, const float* act, int relu, float* dx, int rows,
                             int cols, long long ld, hipStream_t s) {
    const long long n = (long long)rows * cols;
    hipLaunchKernelGGL(dropout_relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dy, mask, scale, act, relu, dx, rows,
                       cols, ld);
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int rows, int cols, long long ld, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float sum = 0.f;
    for (int r = 0; r < rows; ++r) sum += x[(size_t)r * ld + c];
    out[c] = sum;
}

void launch_colsum(const float* x, int rows, int cols, long long ld, float* out, hipStream_t s) {
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, x, rows, cols, ld, out);
}

// counter-based keep mask (splitmix64 of (seed, stream, index)): NOT torch's generator -- the reference's masks are
// irreproducible outside torch; parity tests pass the reference's recorded masks explicitly instead
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ mask, long long n, float p, unsigned long long seed,
                                                           unsigned long long stream_id) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (stream_id + 1) + (unsigned long long)idx * 0xD1342543DE82EF95ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    mask[idx] = u >= p ? 1 : 0;
}

void launch_dropout_mask(uint8_t* mask, long long n, float p, unsigned long long seed, unsigned long long stream_id, hipStream_t s) {
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mask, n, p, seed, stream_id);
}

// ------------------------------------------------------------------------------------------------
// heads + losses + their gradients with respect to the 27 head outputs: ONE workgroup (batch-wide normalisers)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_neg_abs(float x) { return log1pf(expf(-fabsf(x))); }

// SoftBCEWithLogitsFocalLoss (objectives.py:77-112) of one site: n classes, logits x, target index tgt, example weight wgt,
// category weights pw (already divided by their sum).  Returns the site's loss; dx = d(site loss)/dx; *close = the flag.
__device__ float focal_site(const float* x, int n, int tgt, float wgt, const float* pw, float eps, float window, float alpha,
                            float gamma, float* dx, int* close) {
    float mx = x[0];
    for (int j = 1; j < n; ++j) mx = fmaxf(mx, x[j]);
    float p[3], den = 0.f;
    for (int j = 0; j < n; ++j) { p[j] = expf(x[j] - mx); den += p[j]; }
    float loss = 0.f, dist = 0.f, ce[3], fw[3], dfw_dp[3];
    for (int j = 0; j < n; ++j) {
        p[j] = fminf(fmaxf(p[j] / den, 0.f), 1.f);
        const float y = (j == tgt) ? 1.f - eps : eps / (float)(n - 1);
        ce[j] = wgt * (fmaxf(x[j], 0.f) - x[j] * y + softplus_neg_abs(x[j]));          // BCE-with-logits, reduction none
        const float pt = y * p[j] + (1.f - y) * (1.f - p[j]);
        const float om = 1.f - pt;
        fw[j] = powf(om, gamma) * pw[j];
        // d fw / d p_j = pw gamma (1 - pt)^(gamma - 1) * -(2y - 1)
        dfw_dp[j] = (gamma == 0.f) ? 0.f : pw[j] * gamma * powf(om, gamma - 1.f) * -(2.f * y - 1.f);
        loss += alpha * fw[j] * ce[j];
        dist += fabsf(p[j] - y);
    }
    *close = (dist * 0.5f) <= eps * window;
    // dL/dx_k = alpha [ fw_k wgt (sigmoid(x_k) - y_k) + sum_j ce_j dfw_j/dp_j p_j (delta_jk - p_k) ]
    float s = 0.f;
    for (int j = 0; j < n; ++j) s += ce[j] * dfw_dp[j] * p[j];
    for (int k = 0; k < n; ++k) {
        const float y = (k == tgt) ? 1.f - eps : eps / (float)(n - 1);
        const float sig = 1.f / (1.f + expf(-x[k]));
        dx[k] = alpha * (fw[k] * wgt * (sig - y) + ce[k] * dfw_dp[k] * p[k] - p[k] * s);
    }
    return loss;
}

__device__ const float BASE_CW[VOCAB] = {0.001f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.001f, 0.001f, 1.f, 0.001f};      // trainer.py:312-313

// head outputs (model.py:919-921,953-958): one workgroup per site, one wave-slice per head
__global__ __launch_bounds__(64) void heads_logits_kernel(LossArgs a) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (t >= NHEAD) return;
    const float* x = a.hidden + (size_t)b * a.hid_stride;
    const float* w = a.wh + (size_t)t * a.hid;
    float acc = 0.f;
    for (int k = 0; k < a.hid; ++k) acc = fmaf(x[k], w[k], acc);
    a.logits[(size_t)b * NHEAD + t] = acc + a.bh[t];
}

__global__ __launch_bounds__(256) void heads_loss_kernel(LossArgs a, float* site_terms) {
    __shared__ float red[2][256];
    const int tid = threadIdx.x;
    // ---- the cross-entropy normalisers sum_b w[y_b]
    float wsum_b = 0.f, wsum_r = 0.f;
    for (int b = tid; b < a.B; b += 256) {
        wsum_b += BASE_CW[min((int)a.var_base[b], VOCAB - 1)];
        wsum_r += BASE_CW[min((int)a.var_ref[b], VOCAB - 1)];
    }
    red[0][tid] = wsum_b; red[1][tid] = wsum_r;
    __syncthreads();
    if (tid < 2) {
        float sum = 0.f;
        for (int i = 0; i < 256; ++i) sum += red[tid][i];
        red[tid][0] = sum;
    }
    __syncthreads();
    const float wb_tot = a.ce_den[0] > 0.f ? a.ce_den[0] : red[0][0], wr_tot = a.ce_den[1] > 0.f ? a.ce_den[1] : red[1][0];
    const float invB = 1.f / (a.mean_sites > 0.f ? a.mean_sites : (float)a.B);
    const float pw2[2] = {a.fp_weight / (a.fp_weight + 1.f), 1.f / (a.fp_weight + 1.f)};
    const float pw3[3] = {a.fp_weight / (a.fp_weight + 2.f), 1.f / (a.fp_weight + 2.f), 1.f / (a.fp_weight + 2.f)};
    for (int b = tid; b < a.B; b += 256) {
        const float* z = a.logits + (size_t)b * NHEAD;
        float* d = a.dlogits + (size_t)b * NHEAD;
        float* term = site_terms + (size_t)b * 8;
        const float wgt = a.weight[b];
        // Bin: target = label <= 1 (trainer.py:134); VT: target = var_type
        float dx[3];
        int cl;
        const float lb = focal_site(z, 2, a.label[b] <= 1 ? 1 : 0, wgt, pw2, a.label_smoothing, a.close_window, a.focal_alpha,
                                    a.focal_gamma, dx, &cl);
        a.close[b * 2] = (uint8_t)cl;
        d[0] = dx[0] * invB * a.binary_weight; d[1] = dx[1] * invB * a.binary_weight;
        const float lv = focal_site(z + 2, 3, a.var_type[b], wgt, pw3, a.label_smoothing, a.close_window, a.focal_alpha,
                                    a.focal_gamma, dx, &cl);
        a.close[b * 2 + 1] = (uint8_t)cl;
        for (int j = 0; j < 3; ++j) d[2 + j] = dx[j] * invB * a.aux_weight;
        // AF: F.binary_cross_entropy(sigmoid(z), t, weight) (trainer.py:309; log clamped at -100 as torch does)
        const float af = 1.f / (1.f + expf(-z[5])), t_af = a.allele_freq[b];
        const float l_af = -wgt * (t_af * fmaxf(logf(af), -100.f) + (1.f - t_af) * fmaxf(log1pf(-af), -100.f));
        d[5] = wgt * (af - t_af) * invB * a.aux_weight * a.aux_allele_weight;
        // coverage: F.mse_loss(leaky_relu(z), cov / 100) (trainer.py:141,310)
        const float cv = z[6] > 0.f ? z[6] : 0.01f * z[6], t_cv = a.coverage[b] * 0.01f;
        const float l_cv = (cv - t_cv) * (cv - t_cv);
        d[6] = 2.f * (cv - t_cv) * (z[6] > 0.f ? 1.f : 0.01f) * invB * a.aux_weight;
        // bases: F.cross_entropy(., weight) = sum_b w[y] nll / sum_b w[y]  (trainer.py:312-313)
        float l_vb = 0.f, l_vr = 0.f;
        for (int hd = 0; hd < 2; ++hd) {
            const float* v = z + 7 + hd * VOCAB;
            const int y = min((int)(hd ? a.var_ref[b] : a.var_base[b]), VOCAB - 1);
            const float wy = BASE_CW[y], tot = hd ? wr_tot : wb_tot;
            float mx = v[0];
            for (int j = 1; j < VOCAB; ++j) mx = fmaxf(mx, v[j]);
            float den = 0.f;
            for (int j = 0; j < VOCAB; ++j) den += expf(v[j] - mx);
            const float lse = mx + logf(den);
            (hd ? l_vr : l_vb) = wy * (lse - v[y]);
            for (int j = 0; j < VOCAB; ++j)
                d[7 + hd * VOCAB + j] = wy * (expf(v[j] - lse) - (j == y ? 1.f : 0.f)) / tot * a.aux_weight * a.aux_bases_weight;
        }
        term[0] = lb; term[1] = lv; term[2] = l_af; term[3] = l_cv; term[4] = l_vb; term[5] = l_vr;
    }
    __threadfence_block();
    __syncthreads();
    if (tid < 6) {
        float sum = 0.f;
        for (int b = 0; b < a.B; ++b) sum += site_terms[(size_t)b * 8 + tid];
        red[0][tid] = tid < 4 ? sum * invB : sum / (tid == 4 ? wb_tot : wr_tot);
    }
    __syncthreads();
    if (tid == 0) {
        const float bin = red[0][0], vt = red[0][1], af = red[0][2], cov = red[0][3], vb = red[0][4], vr = red[0][5];
        a.losses[1] = bin; a.losses[2] = vt; a.losses[3] = af; a.losses[4] = cov; a.losses[5] = vb; a.losses[6] = vr;
        a.losses[0] = bin * a.binary_weight + (vt + af * a.aux_allele_weight + cov + (vb + vr) * a.aux_bases_weight) * a.aux_weight;   // trainer.py:426-427
    }
}

void launch_heads_loss(const LossArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(heads_logits_kernel, dim3(a.B), dim3(64), 0, s, a);
    hipLaunchKernelGGL(heads_loss_kernel, dim3(1), dim3(256), 0, s, a, a.site_terms);
}

__global__ __launch_bounds__(256) void heads_bwd_hidden_kernel(const float* __restrict__ dl, const float* __restrict__ wh, int hid,
                                                               int hid_stride, float* __restrict__ dh) {
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < hid; k += 256) {
        float acc = 0.f;
        for (int j = 0; j < NHEAD; ++j) acc = fmaf(dl[(size_t)b * NHEAD + j], wh[(size_t)j * hid + k], acc);
        dh[(size_t)b * hid_stride + k] = acc;
    }
}

__global__ __launch_bounds__(256) void heads_bwd_weight_kernel(const float* __restrict__ dl, const float* __restrict__ hidden, int B, int hid,
                                                               int hid_stride, float* __restrict__ gwh, float* __restrict__ gbh) {
    const int j = blockIdx.x;
    for (int k = threadIdx.x; k < hid; k += 256) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc = fmaf(dl[(size_t)b * NHEAD + j], hidden[(size_t)b * hid_stride + k], acc);
        gwh[(size_t)j * hid + k] = acc;
    }
    if (threadIdx.x == 0) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dl[(size_t)b * NHEAD + j];
        gbh[j] = acc;
    }
}

void launch_heads_bwd(const float* dlogits, const float* hidden, const float* wh, int B, int hid, int hid_stride, float* dhidden,
                      float* gwh, float* gbh, hipStream_t s) {
    hipLaunchKernelGGL(heads_bwd_hidden_kernel, dim3(B), dim3(256), 0, s, dlogits, wh, hid, hid_stride, dhidden);
    hipLaunchKernelGGL(heads_bwd_weight_kernel, dim3(NHEAD), dim3(256), 0, s, dlogits, hidden, B, hid, hid_stride, gwh, gbh);
}

// ------------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam (trainer.py:437-439; torch.optim.Adam, no weight decay)
// ------------------------------------------------------------------------------------------------
constexpr int SUMSQ_PER_BLOCK = 256 * 64;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, double* __restrict__ bp) {
    __shared__ double red[256];
    // 16-byte loads, eight in flight (n is a multiple of 4: every tensor of the flat buffer is 16-byte aligned); the squares are
    // added in double in a fixed order
    const long long lo4 = (long long)blockIdx.x * (SUMSQ_PER_BLOCK / 4), n4 = n >> 2;
    const v4f* g4 = (const v4f*)g;
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i += 8) {
        v4f v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long long idx = lo4 + (long long)(i + j) * 256 + threadIdx.x;
            v[j] = idx < n4 ? g4[idx] : splat(0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += (double)v[j][e] * (double)v[j][e];
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) bp[blockIdx.x] = red[0];
}

void launch_sumsq(const float* g, long long n, double* block_partials, int* n_blocks, hipStream_t s) {
    const int nb = (int)((n + SUMSQ_PER_BLOCK - 1) / SUMSQ_PER_BLOCK);
    *n_blocks = nb;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, s, g, n, block_partials);
}

__global__ __launch_bounds__(256) void clip_coef_kernel(const double* __restrict__ bp, int nb, float clip, float* out) {
    __shared__ double red[256];
    double part = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) part += bp[i];            // fixed assignment, fixed tree: deterministic
    red[threadIdx.x] = part;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const double sum = red[0];
    const float norm = (float)sqrt(sum);
    float coef = 1.f;
    if (clip > 0.f) coef = fminf(1.f, clip / (norm + 1e-6f));
    out[0] = norm;
    out[1] = coef;
}

void launch_clip_coef(const double* block_partials, int n_blocks, float clip, float* out, hipStream_t s) {
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, s, block_partials, n_blocks, clip, out);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   long long n, const float* __restrict__ clip_out, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;      // one 16-byte vector of each of the four streams per thread
    if (idx * 4 >= n) return;
    const float clip = clip_out[1];
    const v4f g4 = ((const v4f*)g)[idx], m4 = ((const v4f*)m)[idx], v4 = ((const v4f*)v)[idx];
    v4f p4 = ((const v4f*)p)[idx], mo, vo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gi = g4[e] * clip;
        const float mi = b1 * m4[e] + (1.f - b1) * gi;
        const float vi = b2 * v4[e] + (1.f - b2) * gi * gi;
        mo[e] = mi;
        vo[e] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p4[e] = p4[e] - (lr / bc1) * (mi / denom);
    }
    ((v4f*)m)[idx] = mo;
    ((v4f*)v)[idx] = vo;
    ((v4f*)p)[idx] = p4;
}

void launch_adam(float* p, const float* g, float* m, float* v, long long n, const float* clip_out, float lr, float b1, float b2,
                 float eps, float bc1, float bc2, hipStream_t s) {
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, clip_out, lr, b1, b2, eps, bc1, bc2);
}

}  // namespace dan
