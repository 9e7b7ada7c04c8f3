/*
 * dl4vc_dan_train.h -- C ABI of the MI355X-native DAN TRAINING STEP (libdl4vc_dan.so).
 *
 * Replaces, for one mini-batch, the body of the reference's training loop dl4vc/trainer.py:109-439:
 *   model(...) in train mode (trainer.py:213-217; BatchNorm on batch statistics, dropout in the FC stack),
 *   the focal soft-BCE on the Bin and VT heads (trainer.py:221-224, dl4vc/objectives.py:49-112), the AF / coverage /
 *   base auxiliary losses (trainer.py:309-313), the mix (trainer.py:425-427), loss.backward() (:435),
 *   clip_grad_norm_ (:437-438) and optimizer.step() of optim.Adam(model.parameters(), lr) (:439, main.py:116).
 * What stays on the host, as in the reference: batch assembly and the targets (dl4vc/dataset.py:583-680), the example
 * weights (trainer.py:151-172), the close-example bookkeeping (trainer.py:258-267), the epoch loop (main.py:151-199).
 *
 * Plain C, host pointers in, host pointers out.  One trainer = one HIP device = one process (data parallelism: one
 * process per GPU, gradients averaged between dan_train_backward and dan_train_apply through dan_train_grad_buffer).
 * Every function returns 0 or a negative dan_status (dl4vc_dan.h); text via dan_train_last_error().
 */
#ifndef DL4VC_DAN_TRAIN_H
#define DL4VC_DAN_TRAIN_H

#include "dl4vc_dan.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The flags of train_variant_caller.sh:101-151 that reach one step (arguments.py defaults in brackets). */
typedef struct dan_train_hyper {
    float lr;                 /* --lr 0.0002                      [0.01]  main.py:116                 */
    float beta1, beta2;       /* torch.optim.Adam defaults 0.9 / 0.999                                */
    float adam_eps;           /* 1e-8                                                                 */
    float grad_clip;          /* --grad-clip 1.0                  [0 = off]  trainer.py:437-438       */
    float label_smoothing;    /* --label-smoothing 0.001          objectives.py:79-81                 */
    float close_match_window; /* --close_match_window 2.0         objectives.py:112                   */
    float focal_alpha;        /* --focal_loss_alpha 1.            objectives.py:105                   */
    float focal_gamma;        /* --focal_loss_gamma 0.2           objectives.py:101                   */
    float fp_train_weight;    /* --fp-train-weight 0.2            trainer.py:84-96 (pos_weight[0])    */
    float binary_weight;      /* --binary-weight                  [1.0]  trainer.py:426               */
    float aux_weight;         /* --auxillary-loss-weight 1.0      trainer.py:427                      */
    float aux_bases_weight;   /* --auxillary-loss-bases-weight 0.01                                   */
    float aux_allele_weight;  /* --auxillary-loss-allele-weight 0.001                                 */
    float dropout;            /* --model-hidden-dropout 0.1       model.py:369-377                    */
} dan_train_hyper;

/* Per-site training targets as the reference's dataset yields them (dataset.py:672-680), host arrays of n_sites. */
typedef struct dan_train_targets {
    const uint8_t* label;          /* {0 TP, 1 FN, 2 FP}; binary target = label <= 1   trainer.py:132-134 */
    const uint8_t* var_type;       /* {0 none, 1 homo.., 2 ..} as parse_vcf gives it   trainer.py:137     */
    const float* allele_freq;      /* [0, 1]                                            trainer.py:139     */
    const float* coverage;         /* raw read count (scaled by 1/100 inside)           trainer.py:141     */
    const uint8_t* var_base_enum;  /* token 0..9                                        trainer.py:143     */
    const uint8_t* var_ref_enum;   /* token 0..9                                        trainer.py:144     */
    const float* weight;           /* example weight total_class_weight                 trainer.py:151-172 */
} dan_train_targets;

typedef struct dan_trainer dan_trainer_t;

/* Lifecycle: Basic2DNet(**flags) + optim.Adam(...) (main.py:99-117).  max_batch = sites per step (--batch-size).
 * Only precision 0 (fp32) trains.  Tensors: the same names as dan_set_tensor (dl4vc_dan.h) -- every parameter and the
 * BatchNorm running statistics must be set before dan_train_finalize. */
int dan_train_create(const dan_config* cfg, const dan_train_hyper* hyper, int32_t max_batch, dan_trainer_t** out);
int dan_train_set_tensor(dan_trainer_t* t, const char* name, const float* data, const int64_t* shape, int32_t ndim);
int dan_train_finalize(dan_trainer_t* t);
void dan_train_destroy(dan_trainer_t* t);
const char* dan_train_last_error(const dan_trainer_t* t);

/* Forward (train mode) + losses + backward of one batch: gradients of every parameter are left on the device.
 * Inputs as dan_forward (uint8 planes, HDF5-native order).  dropout_masks: three host arrays of KEEP flags (1 = keep),
 * [B][feature_width], [B][fc0], [B][fc1] -- the masks of the three nn.Dropout of conv2hidden -- or NULL to draw them
 * on the device from (seed, step) (the reference draws them from torch's global generator, which nothing else can
 * reproduce: parity tests pass the masks the reference drew).  Ignored when hyper.dropout == 0.
 * Outputs (host, any may be NULL): losses[7] = total, bin, vt, af, cov, vb, vr (trainer.py:425-434);
 * close[n_sites][2] = (bin_close, vt_close) flags of objectives.py:112 (the loop feeds vt_close to its sampler). */
int dan_train_backward(dan_trainer_t* t, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand,
                       const uint8_t* ref, const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites,
                       const dan_train_targets* targets, const uint8_t* const* dropout_masks, uint64_t seed,
                       float* losses, uint8_t* close);
/* The same step in two halves, for the data-parallel exchange to overlap the backward pass: _begin stages the inputs and
 * ENQUEUES the whole forward + backward (it returns while the device works); _end waits for it and copies the outputs.
 * In between, dan_train_wait_bucket(t, 0) returns as soon as the gradients of bucket 0 are final -- the FC stack and the
 * heads, 87 % of the 311-MB buffer at the published sizes, which the backward pass produces FIRST -- so that their
 * all-reduce runs under the convolution layers' backward (nn.DataParallel reduces after the whole backward, main.py:117);
 * bucket 1 (embedding, convolution / BatchNorm / residual / bottleneck / compression layers) is final at _end.
 * dan_train_grad_bucket gives each bucket's window of dan_train_grad_buffer; the two windows tile it. */
int dan_train_backward_begin(dan_trainer_t* t, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand,
                             const uint8_t* ref, const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites,
                             const dan_train_targets* targets, const uint8_t* const* dropout_masks, uint64_t seed);
int dan_train_wait_bucket(dan_trainer_t* t, int32_t bucket);
int dan_train_backward_end(dan_trainer_t* t, float* losses, uint8_t* close);
int dan_train_grad_bucket(dan_trainer_t* t, int32_t bucket, int64_t* offset, int64_t* count);

/* Data parallelism, exactness: nn.DataParallel (main.py:117) computes every loss on the gathered FULL batch, so its .mean()
 * terms divide by the full batch size and its two weighted cross-entropies (trainer.py:312-313) by the full batch's sum of
 * class weights.  For the average of the ranks' gradients to equal that gradient also when the shards differ, a rank passes,
 * for the NEXT dan_train_backward[_begin] only, those normalisers divided by the number of ranks: full-batch sites / ranks,
 * and (sum over the full batch of the base-class weight of var_base_enum, resp. var_ref_enum) / ranks.  Zero = the rank's
 * own (the single-GPU behaviour).  The class weights are 0.001 for tokens 0, 6, 7, 9 and 1 otherwise. */
int dan_train_set_global_batch(dan_trainer_t* t, float sites_per_rank, float vb_weight_per_rank, float vr_weight_per_rank);

/* clip_grad_norm_ + Adam on the gradients currently on the device; advances the step counter.  grad_norm (may be NULL)
 * receives the total 2-norm before clipping. */
int dan_train_apply(dan_trainer_t* t, float* grad_norm);
/* = dan_train_backward + dan_train_apply */
int dan_train_step(dan_trainer_t* t, const uint8_t* reads, const uint8_t* qual, const uint8_t* strand,
                   const uint8_t* ref, const uint8_t* ref_mask, const uint8_t* var_mask, int64_t n_sites,
                   const dan_train_targets* targets, const uint8_t* const* dropout_masks, uint64_t seed,
                   float* losses, uint8_t* close, float* grad_norm);

/* Replaces  optimizer.param_groups[0]['lr'] *= args.lr_decay  (main.py:166). */
int dan_train_set_lr(dan_trainer_t* t, float lr);

/* Flat fp32 gradient buffer on the device (all parameters, fixed order) for the data-parallel average: ranks all-reduce
 * it (RCCL) between dan_train_backward and dan_train_apply.  Replaces nn.DataParallel's reduce-add (main.py:117). */
void* dan_train_grad_buffer(dan_trainer_t* t, int64_t* n_floats);

/* Copy a tensor to the host in its state-dict shape: "<name>" = current parameter / BN running statistic
 * (model.state_dict(), main.py:194-199), "grad:<name>", "m:<name>", "v:<name>" (Adam moments, optimizer.state_dict()).
 * Debug buffers of the last step: "act:a<l>", "act:x<l>" [B][R][L][128], "act:h<l>" [B][R][L][32] (l = 1..layers),
 * "feature" [B][feature_stride], "logits" / "dlogits" [B][27].  Returns floats copied or a negative status. */
int64_t dan_train_get_tensor(dan_trainer_t* t, const char* name, float* dst, int64_t capacity);
/* Overwrite a parameter or running statistic after finalize (checkpoint restore). */
int dan_train_put_tensor(dan_trainer_t* t, const char* name, const float* src, int64_t count);
int64_t dan_train_query(const dan_trainer_t* t, const char* what);   /* "step", "max_batch", "num_param_floats", "feature_stride" */

#ifdef __cplusplus
}
#endif
#endif /* DL4VC_DAN_TRAIN_H */
