/*
 * dl4vc_dan.h -- C ABI of the MI355X-native DAN inference forward (libdl4vc_dan.so).
 *
 * Drop-in boundary for ONE path of NVIDIA-Genomics-Research/DL4VC: the forward of
 * dl4vc/model.py::Basic2DNet as called from dl4vc/trainer.py:569-572 (inference) plus the score
 * post-processing of dl4vc/trainer.py:609-623.  The reference has no FFI of its own -- the seam is a
 * Python call and a checkpoint format -- so each entry point below names the
 * reference interface it replaces.  The reference-side binding (ctypes) is shown in INTEGRATION.md.
 *
 * Plain C: pointers and sizes only, no torch types.  One handle is bound to one HIP device and is
 * NOT thread-safe; use one handle per process/GPU (sites shard across GPUs with no collective).
 * Every function returns 0 on success or a negative dan_status; dan_last_error() gives the text.
 */
#ifndef DL4VC_DAN_H
#define DL4VC_DAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history: 1 = first release; 2 = dan_config.conv_algo; 3 = dan_config.skip_empty_rows (struct grows at the end);
 * 4 = dan_forward_async / dan_wait; 5 = dan_config.bf16_form (the library reads no environment variable any more);
 * 6 = dan_source_hash, dan_query("chunk_sites_auto"). */
#define DAN_ABI_VERSION 6
#define DAN_MAX_LAYERS 16

typedef enum dan_status {
    DAN_OK = 0,
    DAN_ERR_INVALID_ARG = -1,   /* bad pointer / size / unsupported configuration                */
    DAN_ERR_MISSING_TENSOR = -2,/* finalize(): a weight the configuration needs was never set     */
    DAN_ERR_SHAPE = -3,         /* set_tensor()/finalize(): shape disagrees with the configuration*/
    DAN_ERR_STATE = -4,         /* call order (forward before finalize, set_tensor after, ...)    */
    DAN_ERR_HIP = -5,           /* a HIP runtime call failed (text in dan_last_error)             */
    DAN_ERR_NO_DEVICE = -6      /* no usable gfx950 device                                        */
} dan_status;

/* Structural configuration.  Replaces the keyword arguments of Basic2DNet.__init__
 * (dl4vc/model.py:35-53) as main.py passes them (main.py:99-112). */
typedef struct dan_config {
    int32_t reads;            /* num_single_reads   (R)   model.py:41   */
    int32_t length;           /* single_read_len    (L)   model.py:41   */
    int32_t layers;           /* total_conv_layers        model.py:45   */
    int32_t c_init;           /* init_conv_channels       model.py:36   */
    int32_t c_final;          /* final_conv_channels      model.py:36   */
    int32_t dil_mid;          /* middle_layer_dilation    model.py:52   */
    int32_t dil_final;        /* final_layer_dilation     model.py:52   */
    uint32_t pool_layers_mask;/* bit l set <=> l in conv_1d_pool_layers (1-based)  model.py:49 */
    int32_t residual_start;   /* residual_layer_start (0 = none)        model.py:45 */
    int32_t use_bn;           /* use_batchnorm            model.py:49   */
    int32_t use_q;            /* use_q_scores             model.py:38   */
    int32_t use_strand;       /* use_strands              model.py:38   */
    int32_t use_mask;         /* use_reads_ref_var_mask   model.py:40   */
    int32_t bottleneck;       /* bottleneck_channels == bottleneck_linear_outputs, 0 = no highway */
    int32_t fc_sizes[2];      /* layer_sizes              model.py:35   */
    int32_t precision;        /* 0 = fp32 MFMA (parity path); 1 = bf16x3 split (hi+lo bf16, 3 MFMAs per
                               * product, L <= 304: two units per read above 208 columns, like precision 0); 2 = plain bf16
                               * (L <= 304, BASELINE config 5)                                      */
    int32_t device_id;        /* HIP device ordinal                                               */
    int32_t max_batch;        /* sites per FC macro-batch (0 = 4096)                              */
    int32_t chunk_sites;      /* sites per conv-stack chunk (0 = largest power of two whose y + h fit 48 GB and a third of the
                               * device memory that is free at dan_create) */
    int32_t conv_algo;        /* fp32 path, form of the 3-tap convolutions after the first layer: 1 = direct implicit
                               * GEMM; 2 = Winograd F(2,3) over the dilated positions (needs every such layer to
                               * have dilation 2: 4 exact-fp32 GEMMs per 2 outputs instead of 6); 0 = Winograd
                               * where it applies, else direct                                     */
    int32_t skip_empty_rows;  /* 1 = compute the all-padding rows of a pileup (reads, q-scores and strand bytes all zero:
                               * the rows below the site's coverage) once per site and let every other such row use
                               * that result -- their encoded input is identical, so every output is bit-identical to
                               * computing all rows (which is what 0 does, and what the reference does)              */
    int32_t bf16_form;        /* precision 2, form of the conv-stack kernel: 0 = eight waves, 32x32x16 tiles (default);
                               * 1 = sixteen waves, 16x16x32 tiles -- a second, independently written implementation of the
                               * same arithmetic that the parity tests hold to the same oracle (~7 % slower)            */
} dan_config;

typedef struct dan_handle dan_t;

/* Lifecycle.  Replaces Basic2DNet(**flags) -> load_state_dict() -> eval()
 * (main.py:99-124, trainer.py:476). */
int dan_abi_version(void);
/* Identity of the kernel + C-ABI sources this library was built from (first 16 hex digits of their sha256, set by
 * dl4vc_amd/csrc/Makefile; "unknown" for a build that did not pass it).  Measurement plumbing: bench.py stamps it on its line
 * and reports profiled HBM traffic only when profiles/rNN_traffic.json was captured with the same sources.  No counterpart in
 * the reference. */
const char* dan_source_hash(void);
int dan_create(const dan_config* cfg, dan_t** out);
/* One checkpoint tensor by its state-dict name WITHOUT the "module." prefix (main.py:117,196), fp32,
 * C-contiguous, host memory; copied.  Names: SURVEY.md section 8b "Weights contract".  The two
 * FC Linear layers are addressed as "fc.0.weight"/"fc.0.bias"/"fc.1.*" (their state-dict index depends
 * on the dropout flag, model.py:369-377; the Python loader maps them by structure). */
int dan_set_tensor(dan_t* h, const char* name, const float* data, const int64_t* shape, int32_t ndim);
/* Validate every shape against the configuration, fold BN, repack for MFMA, upload. */
int dan_finalize(dan_t* h);
void dan_destroy(dan_t* h);
const char* dan_last_error(const dan_t* h);   /* h may be NULL: error of the last failed dan_create */

/* Forward on HOST buffers (synchronous).  Replaces  model(reads, ref, q_scores=..., strands=...,
 * ..., ref_masks=..., var_masks=...)  at trainer.py:569-572 plus the softmax at trainer.py:609-623.
 * Inputs are uint8 in the HDF5-native order: reads/qual/strand [B][R][L], ref/ref_mask/var_mask [B][L]
 * (binary_trust_vector, af_scores, ref_bases, var_bases are accepted but never read by the reference's
 * forward in the supported configuration, so they are not part of the ABI).
 * Outputs (caller-owned, any may be NULL): bin_logits [B][2], vt_logits [B][3], vt_prob [B][3] =
 * softmax(vt_logits) = (NV,HV,OV), bp [B] = 1 - softmax(bin_logits)[0]. */
int dan_forward(dan_t* h, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand,
                const uint8_t* ref, const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites,
                float* bin_logits, float* vt_logits, float* vt_prob, float* bp);
/* Same, plus the auxiliary heads of model.py:953-958: aux [B][22] = af(1, sigmoid), cov(1, leaky_relu),
 * vb(10), vr(10). */
int dan_forward_aux(dan_t* h, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand,
                    const uint8_t* ref, const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites,
                    float* bin_logits, float* vt_logits, float* vt_prob, float* bp, float* aux);
/* Forward on DEVICE buffers (all pointers are HIP device pointers on the handle's device), enqueued
 * on `stream` (a hipStream_t, NULL = default stream); asynchronous -- the caller synchronises. */
int dan_forward_device(dan_t* h, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand,
                       const uint8_t* ref, const uint8_t* ref_mask, const uint8_t* var_mask,
                       int64_t n_sites, float* bin_logits, float* vt_logits, float* vt_prob, float* bp,
                       float* aux, void* stream);

/* Asynchronous forward on HOST buffers, double-buffered (SURVEY.md section 8b "Ownership": the reference's loop
 * overlaps nothing -- DataLoader hand-off, H2D scatter, forward, .cpu() and the per-scalar '%.8f' formatting run back to
 * back, trainer.py:518-572,629-630, utils.py:168-178).  dan_forward_async() copies the six input planes into one of TWO
 * pinned staging slots (the caller's input buffers are free again when it returns), enqueues H2D on a copy stream, the
 * forward on the compute stream and the D2H of the scores into the slot's pinned output block on a second copy stream,
 * and returns a ticket.  dan_wait(ticket) blocks until that batch's scores have arrived and copies them to the output
 * pointers given at enqueue (caller-owned, must stay valid until then; any may be NULL).  At most two tickets can be
 * in flight: a third dan_forward_async before the oldest was waited for fails with DAN_ERR_STATE.  n_sites <= max_batch.
 * Results are bit-identical to dan_forward.  Typical loop: enqueue batch k+1, wait for k, format k's VCF records. */
int dan_forward_async(dan_t* h, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand,
                      const uint8_t* ref, const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites,
                      float* bin_logits, float* vt_logits, float* vt_prob, float* bp, float* aux, int64_t* ticket);
int dan_wait(dan_t* h, int64_t ticket);

/* Parity-test taps (debug export).  dan_set_tap(): keep a copy of the activations after conv layer
 * `layer` (1..layers; 0 = the encoded 48-channel input; -1 = off) of the LAST chunk processed.
 * dan_read_buffer(): copy an internal fp32 buffer to the host: "tap" [sites][R][L][128],
 * "feature" [B][feature_stride], "hidden0" [B][fc0], "hidden1" [B][fc1]; returns the number of floats
 * copied (<= capacity) or a negative status.  dan_query(): integer facts by name ("feature_width",
 * "feature_stride", "chunk_sites", "max_batch", "cpad", "tap_sites"). */
/* (With skip_empty_rows = 1 the tap and "y" rows of skipped empty pileup rows are not written: read their source row.) */
int dan_set_tap(dan_t* h, int32_t layer);
int64_t dan_read_buffer(dan_t* h, const char* name, float* dst, int64_t capacity);
int64_t dan_query(const dan_t* h, const char* what);

/* Kernel timing with HIP events on the launch stream (bench.py's roofline leg).  When enabled,
 * every launch of the conv-stack segment kernel is bracketed by events; dan_kernel_stats() returns
 * launches and total milliseconds since the last enable/reset for kernel "conv_segment" (others:
 * "fc", "highway", "pool"). */
int dan_profile_enable(dan_t* h, int32_t on);
int dan_kernel_stats(dan_t* h, const char* kernel, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* DL4VC_DAN_H */
