"""CPU oracle for ONE TRAINING STEP of the DL4VC "DAN" network  --  TEST INFRASTRUCTURE ONLY.

Checker for the training path (BASELINE config 4); only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  The shipped path
(``dl4vc_amd``) never routes through it.

From-scratch restatement (plain torch on CPU; autograd does the differentiation here -- the product computes
every gradient with hand-written HIP kernels) of what one iteration of the reference's training loop does:

  train-mode forward ......... dl4vc/model.py:434-961 with ``model.train()`` (trainer.py:69): BatchNorm2d uses the
                               BATCH statistics over (B, R, L) per channel and updates running_mean/var with momentum
                               0.1 and the unbiased variance (model.py:217,223,229,749-751); the three nn.Dropout of
                               ``conv2hidden`` (before FC1, after each FC's ReLU, model.py:369-377) draw masks
  losses ..................... dl4vc/trainer.py:82-96 (criteria), :134-172 (targets and example weights),
                               :221-224 (focal soft-BCE on Bin and VT), :309-313 (AF / coverage / base aux losses),
                               :425-427 (the mix); dl4vc/objectives.py:49-112 (SoftBCEWithLogitsFocalLoss)
  backward, clip, Adam ....... dl4vc/trainer.py:435-439 (``clip_grad_norm_``, ``optimizer.step()``), main.py:116
                               (``optim.Adam(model.parameters(), lr=args.lr)``: betas 0.9/0.999, eps 1e-8)
  embedding gradient ......... model.py:143-145: ``padding_idx = 0`` and ``scale_grad_by_freq=True`` (each of the two
                               lookups -- reads, ref -- divides a row's gradient by that row's count in the lookup)

The dropout masks are an explicit input (like the read subsets of deep pileups): the reference draws them from torch's
global RNG, which no other implementation can reproduce; ``oracle/gen_golden.py`` records the masks the reference drew.

Pinned by ``tests/golden/train_*.npz``: losses, every gradient tensor, the Adam-updated state and the BN running
statistics after one call of the reference's own ``trainer.train`` on one batch (tests/test_train_oracle.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from oracle.dan_oracle import OracleSpec, spec_from, _strip, fc_keys, encode, BN_EPS, VOCAB

COVERAGE_SCALE_FACTOR = 1.0 / 100.0                      # trainer.py:61
BASE_CLASS_WEIGHT = (0.001, 1., 1., 1., 1., 1., 0.001, 0.001, 1., 0.001)     # trainer.py:312-313
BN_MOMENTUM = 0.1                                        # nn.BatchNorm2d default


@dataclass
class TrainHyper:
    """The flags of train_variant_caller.sh:101-151 that reach one training step (defaults = the published script)."""
    lr: float = 0.0002
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8
    grad_clip: float = 1.0                 # --grad-clip; 0 = off (trainer.py:437)
    label_smoothing: float = 0.001
    close_match_window: float = 2.0
    focal_alpha: float = 1.0
    focal_gamma: float = 0.2
    fp_train_weight: float = 0.2           # pos_weight[0] of both criteria (trainer.py:84-96)
    non_snp_train_weight: float = 2.0      # example weight of non-SNP sites (trainer.py:169-171)
    binary_weight: float = 1.0             # arguments.py:47
    aux_weight: float = 1.0                # --auxillary-loss-weight
    aux_bases_weight: float = 0.01         # --auxillary-loss-bases-weight
    aux_allele_weight: float = 0.001       # --auxillary-loss-allele-weight
    dropout: float = 0.1                   # --model-hidden-dropout


def smoothed_one_hot(n_classes: int, target: torch.Tensor, eps: float, dtype) -> torch.Tensor:
    """objectives.py:79-81: eps/(n-1) everywhere, 1-eps at the target index."""
    oh = torch.full((target.shape[0], n_classes), eps / (n_classes - 1), dtype=dtype)
    oh.scatter_(1, target.reshape(-1, 1), 1.0 - eps)
    return oh


def focal_soft_bce(logits: torch.Tensor, target: torch.Tensor, weight: torch.Tensor, pos_weight: torch.Tensor,
                   hp: TrainHyper):
    """SoftBCEWithLogitsFocalLoss.forward with logits=True (objectives.py:77-112).  Returns (loss, close flags)."""
    n = logits.shape[1]
    oh = smoothed_one_hot(n, target, hp.label_smoothing, logits.dtype)
    ce = F.binary_cross_entropy_with_logits(logits, oh, weight, reduction="none")        # :84-86
    p = F.softmax(logits, dim=1).clamp(0.0, 1.0)                                           # :93-95
    pt = oh * p + (1 - oh) * (1 - p)                                                       # :100
    w = (1 - pt) ** hp.focal_gamma                                                         # :101
    w = w * pos_weight / pos_weight.sum()                                                  # :103
    loss = (hp.focal_alpha * w * ce).sum(dim=1).mean()                                     # :105-109
    dist = (p - oh).abs().sum(dim=1) / 2.0                                                 # :112
    close = dist <= hp.label_smoothing * hp.close_match_window
    return loss, close


def train_forward(sd: Dict[str, torch.Tensor], spec: OracleSpec, x: torch.Tensor, dropout_masks: Optional[Sequence],
                  hp: TrainHyper, taps: Optional[dict] = None, forced: Optional[dict] = None):
    """Train-mode forward from the encoded input ``x`` (B,Cin,R,L).  Returns (outputs dict, {layer: (mean, biased var)}).

    ``forced`` (tests only): the DISCRETE decisions of another evaluation of the same step, imposed on this one -- "relu<l>" /
    "hrelu<l>": bool (B,C,R,L) masks used in place of the conv / bottleneck ReLU's own sign test (x * mask: the same function and
    the same derivative wherever the two agree), "argmax": int (B,C,L), the read that takes the final max.  A ReLU input or a
    top-1 / top-2 gap within an fp32 rounding error of its edge is decided either way by correct fp32 evaluations, and one such
    decision moves gradient tensors by up to 2e-2 of their max; with the device's decisions imposed, the float64 result is the exact
    gradient of the network the device differentiated and the comparison needs no loose bar."""
    B, _, R, L = x.shape
    forced = forced or {}

    def relu(t, key):
        if key in forced:
            return t * torch.as_tensor(np.asarray(forced[key])).to(t.dtype)
        return F.relu(t)

    pool = None
    hws = []
    stats = {}
    for l in range(1, spec.layers + 1):
        _, _, dil = spec.layer_dims(l)
        residual = x                                                 # model.py:732
        if (l - 1) in spec.pool_layers:
            x = x + pool                                             # model.py:742
        x = F.conv2d(x, sd["conv1D_layers.%d.weight" % (l - 1)], sd["conv1D_layers.%d.bias" % (l - 1)],
                     padding=(0, dil), dilation=(1, dil))
        if taps is not None:
            taps["pre%d" % l] = x.detach().numpy().copy()           # (the ReLU's input: which mask decisions sit on a rounding error)
        x = relu(x, "relu%d" % l)                                    # model.py:749
        if spec.use_bn:                                              # training-mode BN after the ReLU, model.py:750-751
            p = "bn1D_layers.%d." % (l - 1)
            mu = x.mean(dim=(0, 2, 3))
            var = x.var(dim=(0, 2, 3), unbiased=False)
            stats[l] = (mu.detach(), var.detach(), B * R * L)
            x = (x - mu[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
            x = x * sd[p + "weight"][None, :, None, None] + sd[p + "bias"][None, :, None, None]
        if spec.is_residual(l):
            i = l - spec.residual_start                              # model.py:760
            x = F.conv2d(x, sd["residual_conv_layers.%d.weight" % i], sd["residual_conv_layers.%d.bias" % i]) + residual
        if taps is not None:
            taps["conv%d" % l] = x.detach().numpy().copy()
        if l in spec.pool_layers:
            pool = x.mean(dim=2, keepdim=True)                       # model.py:772
        if spec.bottleneck > 0:
            h = F.conv2d(x, sd["conv1D_bottleneck_layers.%d.weight" % (l - 1)], sd["conv1D_bottleneck_layers.%d.bias" % (l - 1)])
            if taps is not None:
                taps["hpre%d" % l] = h.detach().numpy().copy()
            h = relu(h, "hrelu%d" % l)
            hw = F.conv2d(h, sd["conv1D_compression_layers.%d.weight" % (l - 1)],
                          sd["conv1D_compression_layers.%d.bias" % (l - 1)])
            hws.append(hw.squeeze(3).reshape(B, -1))
    if "argmax" in forced:
        mx = torch.gather(x, 2, torch.as_tensor(np.asarray(forced["argmax"])).long()[:, :, None, :])
    else:
        mx = x.max(dim=2, keepdim=True).values
    av = x.mean(dim=2, keepdim=True)
    feat = torch.cat((mx, av), dim=1).reshape(B, -1)
    if spec.bottleneck > 0:
        if taps is not None:
            taps["hwpre"] = torch.cat(hws, dim=1).detach().numpy().copy()     # (B, layers*H*R), layer-major, channel-major, read-minor
        feat = torch.cat((feat, F.relu(torch.cat(hws, dim=1))), dim=1)
    if taps is not None:
        taps["feature"] = feat.detach().numpy().copy()

    def drop(t, i):                                                  # nn.Dropout(p) in training: mask / (1 - p)
        if hp.dropout <= 0.0:
            return t
        m = torch.as_tensor(np.asarray(dropout_masks[i])).to(t.dtype)
        return t * m / (1.0 - hp.dropout)

    hidden = drop(feat, 0)                                           # conv2hidden.0            model.py:371-372
    for i, k in enumerate(fc_keys(sd)):                              # Linear, ReLU, Dropout    model.py:374
        hidden = F.linear(hidden, sd[k + ".weight"], sd[k + ".bias"])
        if taps is not None:
            taps["fcpre%d" % i] = hidden.detach().numpy().copy()
        hidden = drop(F.relu(hidden), i + 1)
    if taps is not None:
        taps["hidden"] = hidden.detach().numpy().copy()

    def head(name):
        return F.linear(hidden, sd[name + ".weight"], sd[name + ".bias"])

    if taps is not None:
        taps["covpre"] = head("fcHidden2Coverage").detach().numpy().copy()
    out = {"bin_logits": head("fcHidden2BinTarget"), "vt_logits": head("fcHidden2VT"),
           "af": torch.sigmoid(head("fcHidden2AF")), "cov": F.leaky_relu(head("fcHidden2Coverage")),
           "vb": head("fcHidden2VB"), "vr": head("fcHidden2VR")}
    return out, stats


def example_weights(is_snp: np.ndarray, hp: TrainHyper, trust_weight: Optional[np.ndarray] = None) -> np.ndarray:
    """trainer.py:151,169-172: (is_snp + (1 - is_snp) * non_snp_weight) * binary_trust_weight  -> (B,)."""
    s = np.asarray(is_snp, np.float32)
    w = s + (1.0 - s) * np.float32(hp.non_snp_train_weight)
    if trust_weight is not None:
        w = w * np.asarray(trust_weight, np.float32)
    return w.astype(np.float32)


def losses(out: Dict[str, torch.Tensor], targets: Dict[str, np.ndarray], hp: TrainHyper):
    """trainer.py:132-144 (targets), :221-224, :309-313, :425-427.  ``targets``: label (B,) {0 TP, 1 FN, 2 FP},
    var_type (B,) {0 none, 1 homo?, 2 ...} as the dataset yields it, allele_freq (B,) float, coverage (B,) RAW read
    count, var_base_enum / var_ref_enum (B,) tokens, weight (B,) example weight (``example_weights``)."""
    dt = out["bin_logits"].dtype
    label = torch.as_tensor(np.asarray(targets["label"])).long()
    t_bin = (label <= 1).long()                                                            # trainer.py:134
    t_vt = torch.as_tensor(np.asarray(targets["var_type"])).long()
    t_af = torch.as_tensor(np.asarray(targets["allele_freq"], np.float32)).to(dt).reshape(-1, 1)
    t_cov = torch.as_tensor(np.asarray(targets["coverage"], np.float32)).to(dt).reshape(-1, 1) * COVERAGE_SCALE_FACTOR
    t_vb = torch.as_tensor(np.asarray(targets["var_base_enum"])).long()
    t_vr = torch.as_tensor(np.asarray(targets["var_ref_enum"])).long()
    w = torch.as_tensor(np.asarray(targets["weight"], np.float32)).to(dt).reshape(-1, 1)   # total_class_weight (B,1)
    pw2 = torch.tensor([hp.fp_train_weight, 1.0], dtype=dt)
    pw3 = torch.tensor([hp.fp_train_weight, 1.0, 1.0], dtype=dt)
    bin_loss, bin_close = focal_soft_bce(out["bin_logits"], t_bin, w, pw2, hp)
    vt_loss, vt_close = focal_soft_bce(out["vt_logits"], t_vt, w, pw3, hp)
    af_loss = F.binary_cross_entropy(out["af"], t_af, weight=w)                            # trainer.py:309
    cov_loss = F.mse_loss(out["cov"], t_cov)                                               # trainer.py:310
    cw = torch.tensor(BASE_CLASS_WEIGHT, dtype=dt)
    vb_loss = F.cross_entropy(out["vb"], t_vb, weight=cw)                                  # trainer.py:312
    vr_loss = F.cross_entropy(out["vr"], t_vr, weight=cw)                                  # trainer.py:313
    loss = bin_loss * hp.binary_weight                                                     # trainer.py:426-427
    loss = loss + (vt_loss + af_loss * hp.aux_allele_weight + cov_loss + (vb_loss + vr_loss) * hp.aux_bases_weight) * hp.aux_weight
    return {"loss": loss, "bin": bin_loss, "vt": vt_loss, "af": af_loss, "cov": cov_loss, "vb": vb_loss, "vr": vr_loss,
            "bin_close": bin_close, "vt_close": vt_close}


TRAINABLE_SKIP = ("pe", "running_mean", "running_var", "num_batches_tracked", "bin_output_weights", "vt_output_weights")


def trainable(name: str) -> bool:
    """Parameters that receive a gradient: everything but the ``pe`` buffer, the BN running statistics and the two
    early-loss mixing scalars (model.py:429-431), which no loss term of the supported configuration touches (their
    ``.grad`` stays None, so clip_grad_norm_ and Adam skip them)."""
    return not name.endswith(TRAINABLE_SKIP)


def train_step_oracle(state_dict, cfg, planes, targets, hp: TrainHyper = TrainHyper(), dropout_masks=None,
                      adam_state: Optional[dict] = None, step: int = 1, dtype=torch.float32, taps: bool = False,
                      forced: Optional[dict] = None):
    """One optimisation step.  Returns a dict of numpy arrays:
      loss terms ('loss','bin','vt','af','cov','vb','vr'), 'bin_close','vt_close' (B,) bool, outputs ('out:<name>'),
      'grad:<param>' (before clipping), 'grad_norm', 'clip_coef', 'new:<tensor>' (parameters after Adam, BN running stats
      after the momentum update), 'm:<param>' / 'v:<param>' (Adam moments)."""
    spec = spec_from(cfg)
    sd = _strip(state_dict, dtype)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if trainable(k) and v.is_floating_point()}
    live = dict(sd)
    live.update(params)
    reads, qual, strand, ref, ref_mask, var_mask = planes
    # encode with a differentiable embedding lookup (dan_oracle.encode is no_grad): same channel order
    rd = torch.as_tensor(np.asarray(reads)).long()
    rf = torch.as_tensor(np.asarray(ref)).long()
    B, R, L = rd.shape
    with torch.no_grad():
        x_const = encode(spec, sd, reads, qual, strand, ref, ref_mask, var_mask, dtype)      # (B,Cin,R,L), for the non-embedding channels
    E = live["embeddings.weight"]
    pe = sd["pe"].reshape(-1, spec.embed_dim)[:L]
    # nn.Embedding(padding_idx=0, scale_grad_by_freq=True) (model.py:143-145): one lookup per index tensor
    r_emb = F.embedding(rd, E, padding_idx=0, scale_grad_by_freq=True) + pe
    f_emb = (F.embedding(rf, E, padding_idx=0, scale_grad_by_freq=True) + pe).unsqueeze(1).expand(B, R, L, spec.embed_dim)
    emb = torch.cat((r_emb, f_emb), dim=3).permute(0, 3, 1, 2)
    x = torch.cat((emb, x_const[:, 2 * spec.embed_dim:]), dim=1)
    tp = {} if taps else None
    out, stats = train_forward(live, spec, x, dropout_masks, hp, tp, forced)      # (``forced``: see train_forward; tests only)
    ls = losses(out, targets, hp)
    ls["loss"].backward()
    res: Dict[str, np.ndarray] = {}
    for k in ("loss", "bin", "vt", "af", "cov", "vb", "vr"):
        res[k] = np.asarray(ls[k].detach().numpy())
    res["bin_close"] = ls["bin_close"].numpy()
    res["vt_close"] = ls["vt_close"].numpy()
    for k, v in out.items():
        res["out:" + k] = v.detach().numpy()
    if tp:
        for k, v in tp.items():
            res["tap:" + k] = v
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
    for k, g in grads.items():
        res["grad:" + k] = g.numpy().copy()
    # clip_grad_norm_(parameters, max_norm) (trainer.py:437-438): total 2-norm, coef = max_norm / (norm + 1e-6) clamped to 1
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(dtype)
    coef = torch.ones((), dtype=dtype)
    if hp.grad_clip > 0:
        coef = torch.clamp(hp.grad_clip / (total + 1e-6), max=1.0)
    res["grad_norm"] = np.asarray(total.numpy())
    res["clip_coef"] = np.asarray(coef.numpy())
    # Adam (torch.optim.Adam, no weight decay, no amsgrad)
    b1, b2 = hp.beta1, hp.beta2
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    for k, p in params.items():
        g = grads[k] * coef
        m = torch.as_tensor(adam_state["m:" + k]).to(dtype) if adam_state else torch.zeros_like(p)
        v = torch.as_tensor(adam_state["v:" + k]).to(dtype) if adam_state else torch.zeros_like(p)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        denom = v.sqrt() / (bc2 ** 0.5) + hp.adam_eps
        new = p.detach() - (hp.lr / bc1) * m / denom
        res["new:" + k] = new.numpy()
        res["m:" + k] = m.numpy()
        res["v:" + k] = v.numpy()
    # BN running statistics: momentum 0.1, UNBIASED batch variance
    for l, (mu, var, n) in stats.items():
        p = "bn1D_layers.%d." % (l - 1)
        res["new:" + p + "running_mean"] = ((1 - BN_MOMENTUM) * sd[p + "running_mean"] + BN_MOMENTUM * mu).numpy()
        res["new:" + p + "running_var"] = ((1 - BN_MOMENTUM) * sd[p + "running_var"] + BN_MOMENTUM * var * (n / (n - 1.0))).numpy()
    return res


def state_errors(new, ref_new, ref_grad, lr):
    """How far an Adam-updated tensor is from the reference's, split by how well-conditioned the update is.

    Adam's first steps move every element by ~lr * g / (|g| + eps): where |g| is at the level of fp32 summation noise
    (or of eps = 1e-8) the SIGN of g, and with it the whole +-lr step, depends on the summation order -- two correct
    implementations legitimately differ by up to 2 * lr there.  Returns (worst |delta| / max|ref_new| over the elements whose
    reference gradient is significant (|g| >= 1e-3 * max|g|), worst |delta| / lr over the other elements)."""
    new, ref_new = np.asarray(new, np.float64), np.asarray(ref_new, np.float64)
    d = np.abs(new - ref_new)
    if ref_grad is None:
        return float(d.max(initial=0.0)) / max(1e-12, float(np.abs(ref_new).max(initial=0.0))), 0.0
    g = np.abs(np.asarray(ref_grad, np.float64))
    sig = g >= 1e-3 * g.max(initial=0.0)
    a = float(d[sig].max(initial=0.0)) / max(1e-12, float(np.abs(ref_new).max(initial=0.0)))
    b = float(d[~sig].max(initial=0.0)) / lr
    return a, b
