"""CPU oracle for the DL4VC "DAN" inference forward  --  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The shipped path (``dl4vc_amd``) never routes through it and fails loudly
when the HIP extension is missing.

It is a from-scratch functional restatement (plain ``torch.nn.functional`` on
CPU tensors, fp32 or fp64) of the reference network
``/root/reference/dl4vc/model.py::Basic2DNet.forward`` for the configuration
family the published scripts use.  It is
pinned two ways (see ``oracle/gen_golden.py`` and ``tests/test_oracle_golden.py``):

* against golden vectors produced by importing the reference itself in the
  build container (``tests/golden/dan_*.npz``), and
* live against the reference at full production shape when ``/root/reference``
  is present (``tests/test_vs_live_reference.py``, which also fuzzes random structural configurations).

Data layout differs from the reference on purpose: all per-read tensors are
``[site][read][pos]`` uint8 (the HDF5-native order,
``tools/convert_bam_single_reads.py:694-698``), not the ``(B, pos, read)``
transpose the reference builds in ``dl4vc/dataset.py:521``.

Reference lines followed, by block:
  encode .................. dl4vc/model.py:450-451, 463-470, 501-517, 534-561, 576-627, 719
  conv stack .............. dl4vc/model.py:211-262 (construction), 728-778 (loop)
  read pooling / feature .. dl4vc/model.py:302-304, 824-839, 848-859, 911-912
  FC + heads .............. dl4vc/model.py:362-377, 406-415, 917-921, 953-958
  score post-processing ... dl4vc/trainer.py:609-623
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

# dl4vc/model.py:16,24 -- scale factors applied to the strand enum and the q-score
STRAND_SCALE = 0.5
QUAL_SCALE = 1.0 / 100.0
BN_EPS = 1e-5          # nn.BatchNorm2d default, dl4vc/model.py:217,223,229
VOCAB = 10             # len(enum_base), dl4vc/base_enum.py:13


@dataclass
class OracleSpec:
    """Structural configuration (the subset of Basic2DNet flags that is in scope)."""
    reads: int = 100                 # num_single_reads  (model.py:41)
    length: int = 201                # single_read_len
    layers: int = 7                  # total_conv_layers
    c_init: int = 128                # init_conv_channels
    c_final: int = 128               # final_conv_channels
    dil_mid: int = 2                 # middle_layer_dilation
    dil_final: int = 2               # final_layer_dilation
    pool_layers: Sequence[int] = (2,)   # conv_1d_pool_layers (1-based, pool AFTER layer)
    residual_start: int = 5          # residual_layer_start (0 = none)
    use_bn: bool = True
    use_q: bool = True
    use_strand: bool = True
    use_mask: bool = True
    bottleneck: int = 32             # bottleneck_channels == bottleneck_linear_outputs
    fc_sizes: Sequence[int] = (1024, 256)
    embed_dim: int = 20

    @property
    def in_channels(self) -> int:
        return 2 * self.embed_dim + int(self.use_q) + int(self.use_strand) + (3 if self.use_mask else 0)

    def layer_dims(self, l: int):
        """(c_in, c_out, dilation) of 1-based conv layer l -- model.py:211-229."""
        if l == 1:
            return self.in_channels, self.c_init, 1
        if l < self.layers:
            return self.c_init, self.c_init, self.dil_mid
        return self.c_init, self.c_final, self.dil_final

    def is_residual(self, l: int) -> bool:
        """model.py:246."""
        return (self.residual_start > 0 and l >= self.residual_start
                and not (l == self.layers and self.c_init != self.c_final))

    @property
    def feature_width(self) -> int:
        return 2 * self.c_final * self.length + self.layers * self.bottleneck * self.reads


def spec_from(cfg) -> OracleSpec:
    """Build an OracleSpec from any object/dict carrying the same field names."""
    if isinstance(cfg, OracleSpec):
        return cfg
    get = (lambda k, d: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d: getattr(cfg, k, d))
    base = OracleSpec()
    kw = {f: get(f, getattr(base, f)) for f in base.__dataclass_fields__}
    kw["pool_layers"] = tuple(kw["pool_layers"])
    kw["fc_sizes"] = tuple(kw["fc_sizes"])
    return OracleSpec(**kw)


def _strip(sd: Dict[str, "np.ndarray | torch.Tensor"], dtype) -> Dict[str, torch.Tensor]:
    """Accept reference-style state dicts: optional ``module.`` prefix (main.py:117,196)."""
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        t = torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v
        if t.is_floating_point():
            t = t.to(dtype)
        out[k] = t
    return out


def fc_keys(sd: Dict[str, torch.Tensor]) -> List[str]:
    """``conv2hidden.N`` Linear indices in order: .1/.4 with dropout, .0/.3 without (model.py:369-377)."""
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("conv2hidden.") and k.endswith(".weight")})
    return ["conv2hidden.%d" % i for i in idx]


@torch.no_grad()
def encode(spec: OracleSpec, sd, reads, qual, strand, ref, ref_mask, var_mask, dtype=torch.float32):
    """Rows A5: uint8 [B][R][L] (+[B][L]) -> float (B, Cin, R, L).   model.py:450-627,719."""
    reads = torch.as_tensor(np.asarray(reads)).long()
    ref = torch.as_tensor(np.asarray(ref)).long()
    B, R, L = reads.shape
    E = sd["embeddings.weight"]                       # (10, 20)
    pe = sd["pe"].reshape(-1, spec.embed_dim)[:L]     # (L, 20)   model.py:466
    r_emb = E[reads] + pe                             # (B,R,L,20)  model.py:450,506
    f_emb = (E[ref] + pe).unsqueeze(1).expand(B, R, L, spec.embed_dim)   # model.py:451,502-507
    chans = [r_emb, f_emb]                            # model.py:517
    if spec.use_q:
        q = torch.as_tensor(np.asarray(qual)).to(dtype) * QUAL_SCALE     # model.py:536
        chans.append(q.unsqueeze(-1))
    if spec.use_strand:
        s = torch.as_tensor(np.asarray(strand)).to(dtype) * STRAND_SCALE  # model.py:551
        chans.append(s.unsqueeze(-1))
    if spec.use_mask:
        rm = torch.as_tensor(np.asarray(ref_mask)).long()               # (B,L)
        vm = torch.as_tensor(np.asarray(var_mask)).long()

        def match(mask):
            on = (mask != 0)                                            # model.py:579,606
            # read agrees iff reads*on == mask at EVERY position          model.py:592-593
            agree = ((reads * on.unsqueeze(1).long()) == mask.unsqueeze(1)).all(dim=2)   # (B,R)
            return (on.unsqueeze(1) & agree.unsqueeze(2)).to(dtype)     # (B,R,L)   model.py:599

        refmatch = match(rm)
        varmatch = match(vm)
        lenmask = (rm != 0).to(dtype).unsqueeze(1).expand(B, R, L)      # model.py:578-584
        chans += [refmatch.unsqueeze(-1), varmatch.unsqueeze(-1), lenmask.unsqueeze(-1)]   # model.py:625
    x = torch.cat(chans, dim=3)                       # (B,R,L,Cin)
    return x.permute(0, 3, 1, 2).contiguous()         # (B,Cin,R,L)  == transpose(1,3) of (B,L,R,C)


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    """fp32 -> nearest bf16 (ties to even, what ``v_cvt_pk_bf16_f32`` does) -> fp32."""
    return t.to(torch.bfloat16).to(t.dtype)


BF16_MODES = (None, "operands", "storage")


@torch.no_grad()
def conv_layer(spec: OracleSpec, sd, l: int, x, pool=None, bf16: Optional[str] = None):
    """One pass of the layer loop, model.py:728-778, for the 1-based layer ``l``: input ``x`` (B,C,R,L) = the previous layer's
    output (or the encoded input), ``pool`` = the read-mean to add first when layer l-1 is a pool layer.
    Returns (y_l, h_l or None): the layer's output and its ReLU'd bottleneck (B,H,R,L) -- model.py:774.

    ``bf16`` (BASELINE config 5; the reference has no such mode -- it is what "run the GEMMs on the bf16 matrix cores" means):
      "operands": the two operands of every conv / residual 1x1 / bottleneck GEMM are rounded to bf16, sums are fp32,
                  everything else (bias, ReLU, BatchNorm, residual add, pooling, compression, FC) is the fp32 arithmetic
                  above.  Pinned against the live reference run with bf16-rounded weights and forward-pre-hooks that round
                  the inputs of those modules (tests/test_vs_live_reference.py).
      "storage":  "operands" plus the roundings of the HIP kernel's bf16 activation STORAGE (dan_kernels_bf16p.hip): a
                  layer's output y_l, the BatchNorm output feeding the residual 1x1 and the bottleneck h_l are held as bf16 --
                  so the residual branch adds the bf16 value of x and the read-mean averages bf16 values; the read-mean itself
                  enters the next layer unrounded, as its own fp32 convolution (see below).  Each is one more rounding of a value "operands" rounds anyway at its next
                  use as an operand; the GPU parity test holds the kernel to THIS mode layer by layer."""
    assert bf16 in BF16_MODES
    rb = bf16_round if bf16 else (lambda t: t)
    st = bf16_round if bf16 == "storage" else (lambda t: t)
    _, _, dil = spec.layer_dims(l)
    residual = x                                                 # model.py:732 (before the pool add)
    W = sd["conv1D_layers.%d.weight" % (l - 1)]
    b = sd["conv1D_layers.%d.bias" % (l - 1)]
    if pool is not None and bf16 == "storage":
        # the kernel convolves the stored bf16 y alone and seeds its accumulators with conv(pool) of the site, computed once per
        # site in fp32 (launch_conv_pool): conv(y + pool) = conv(y) + conv(pool) with the SUM never rounded to bf16
        x = F.conv2d(rb(x), rb(W), b, padding=(0, dil), dilation=(1, dil)) + \
            F.conv2d(pool, rb(W), None, padding=(0, dil), dilation=(1, dil))
    else:
        if pool is not None:
            x = x + pool                                         # model.py:742
        x = F.conv2d(rb(x), rb(W), b, padding=(0, dil), dilation=(1, dil))
    x = F.relu(x)                                                # model.py:749
    if spec.use_bn:                                              # eval-mode BN AFTER the ReLU, model.py:750-751
        p = "bn1D_layers.%d." % (l - 1)
        x = F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"],
                         sd[p + "weight"], sd[p + "bias"], training=False, eps=BN_EPS)
    x = st(x)
    if spec.is_residual(l):
        i = l - spec.residual_start                              # model.py:760
        x = F.conv2d(rb(x), rb(sd["residual_conv_layers.%d.weight" % i]), sd["residual_conv_layers.%d.bias" % i])
        x = st(x + residual)                                     # model.py:761
    h = None
    if spec.bottleneck > 0:
        h = st(F.relu(F.conv2d(rb(x), rb(sd["conv1D_bottleneck_layers.%d.weight" % (l - 1)]),
                               sd["conv1D_bottleneck_layers.%d.bias" % (l - 1)])))       # model.py:774
    return x, h


@torch.no_grad()
def dan_forward_oracle(state_dict, cfg, reads, qual, strand, ref, ref_mask, var_mask,
                       taps: bool = False, dtype=torch.float32, bf16: Optional[str] = None) -> Dict[str, np.ndarray]:
    """Full forward.  Returns numpy arrays:
    bin_logits (B,2) vt_logits (B,3) af (B,1) cov (B,1) vb (B,10) vr (B,10)  -- model.py:919-958
    vt_prob (B,3) bp (B,)                                                  -- trainer.py:609-623
    with ``taps``: conv{l} (B,C,R,L), hw{l} (B,H*R), pool{l}, feature (B,F), hidden (B,fc[-1]).
    ``bf16``: None (the reference's fp32 arithmetic) | "operands" | "storage" -- see ``conv_layer``.
    """
    spec = spec_from(cfg)
    sd = _strip(state_dict, dtype)
    out: Dict[str, np.ndarray] = {}
    x = encode(spec, sd, reads, qual, strand, ref, ref_mask, var_mask, dtype)
    if bf16 == "storage":
        x = bf16_round(x)
    B, _, R, L = x.shape
    if taps:
        out["encoded"] = x.numpy().copy()
    pool = None
    hws = []
    for l in range(1, spec.layers + 1):
        x, h = conv_layer(spec, sd, l, x, pool if (l - 1) in spec.pool_layers else None, bf16)
        if taps:
            out["conv%d" % l] = x.numpy().copy()
        if l in spec.pool_layers:
            # AvgPool2d((MAX_READS,1), ceil_mode) over all R rows, model.py:194,772: the same pooling op as the reference -- rows
            # summed IN ORDER, then divided by the rows present -- and not x.mean(dim=2), whose pairwise sum differs in the last bit:
            # harmless in fp32, but in the bf16 modes a last-bit difference of y + pool flips bf16 roundings (325 of 1.8 M elements
            # at 128 reads x 301 columns, each a 1e-3-of-max change downstream: found by the R > 100 fixtures of round 6)
            pool = F.avg_pool2d(x, (R, 1))
            if taps:
                out["pool%d" % l] = pool.numpy().copy()
        if spec.bottleneck > 0:
            if taps:
                out["h%d" % l] = h.numpy().copy()
            hw = F.conv2d(h, sd["conv1D_compression_layers.%d.weight" % (l - 1)],
                          sd["conv1D_compression_layers.%d.bias" % (l - 1)])            # (B,H,R,1) model.py:776
            hw = hw.squeeze(3).reshape(B, -1)                    # channel-major, read-minor  model.py:777
            hws.append(hw)
            if taps:
                out["hw%d" % l] = hw.numpy().copy()
    mx = x.max(dim=2, keepdim=True).values                       # model.py:825
    av = F.avg_pool2d(x, (R, 1))                                 # model.py:304,826 (in-order sum, as above)
    feat = torch.cat((mx, av), dim=1).reshape(B, -1)             # model.py:833,839
    if spec.bottleneck > 0:
        feat = torch.cat((feat, F.relu(torch.cat(hws, dim=1))), dim=1)   # model.py:854,859,912
    if taps:
        out["feature"] = feat.numpy().copy()
    hidden = feat
    for k in fc_keys(sd):                                        # Dropout is identity in eval, model.py:917
        hidden = F.relu(F.linear(hidden, sd[k + ".weight"], sd[k + ".bias"]))
    if taps:
        out["hidden"] = hidden.numpy().copy()

    def head(name):
        return F.linear(hidden, sd[name + ".weight"], sd[name + ".bias"])

    xbin = head("fcHidden2BinTarget")                            # model.py:919
    xvt = head("fcHidden2VT")                                    # model.py:921
    out["bin_logits"] = xbin.numpy()
    out["vt_logits"] = xvt.numpy()
    out["af"] = torch.sigmoid(head("fcHidden2AF")).numpy()       # model.py:953-954
    out["cov"] = F.leaky_relu(head("fcHidden2Coverage")).numpy()  # model.py:955-956
    out["vb"] = head("fcHidden2VB").numpy()
    out["vr"] = head("fcHidden2VR").numpy()
    out["bp"] = (1.0 - F.softmax(xbin, dim=1)[:, 0]).numpy()     # trainer.py:620-621
    out["vt_prob"] = F.softmax(xvt, dim=1).numpy()               # trainer.py:623
    return out


# ----------------------------------------------------------------------------------------------
# Seeded weights of the reference's shapes (used by tests, smoke and bench; no checkpoint offline)
# ----------------------------------------------------------------------------------------------
# Seeded weights and the positional-encoding buffer are synthetic-data helpers of the package (dl4vc_amd/synth.py): the
# benchmark needs them without touching the oracle.  Re-exported here for the tests and the golden generator.
from dl4vc_amd.synth import sinusoid_pe, random_state_dict   # noqa: E402,F401
