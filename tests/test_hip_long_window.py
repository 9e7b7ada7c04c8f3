"""Parity paths (fp32 and, since the same round, bf16x3) for windows of 209..304 columns (VERDICT r4 "missing" 5; SURVEY.md section 8a: R and L are configuration;
reference: dl4vc/model.py:41 single_read_len, :214-231).  A read above 208 columns does not fit the LDS image of the fp32 segment
kernel, so it is computed as TWO overlapping units (csrc/dan_kernels.h plan_units, segment_kernel<., ., SPLIT = true>): each unit
is an ordinary <= 208-column read of the kernel, the overlap is the segment's receptive-field radius, a unit stores only its own
columns, y crosses segments out of place.  Held to the reference's own fp32 outputs (tests/golden/long_*.npz) and to the oracle
at the fp32 path's bars.  The bf16x3 kernel (csrc/dan_kernels_bf16x.hip, segmentx_kernel<SPLIT = true>) takes the same unit plan:
the same fixtures and structures at the bars of tests/test_hip_bf16.py.  GPU only."""
import dataclasses

import numpy as np
import pytest

from golden_util import load_case, long_cases, input_tuple
from dl4vc_amd.config import DanConfig, PRECISION_BF16X3, PRECISION_BF16
from dl4vc_amd.model import DanNet
from dl4vc_amd import synth
from oracle.dan_oracle import dan_forward_oracle, random_state_dict
from test_hip_parity import cfg_from, close, SCORE_ATOL, TAP_RTOL, ALGOS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("case", long_cases())
def test_long_window_golden_outputs_and_taps(case, algo):
    """The reference's fp32 forward at 301 columns (the inputs and weights of bf16_operands_l301) and at 304 columns with two pool
    layers (three segments): scores 1e-4 absolute; logits, heads, the feature row, the hidden layer and the conv2 / conv7 taps
    1e-4 of the tensor's magnitude -- columns either side of the cut between the units included."""
    spec, w, inp, out = load_case(case)
    cfg = cfg_from(spec, conv_algo=algo)
    assert cfg.length > 208
    net = DanNet(cfg).load_state_dict(w)
    got = net.forward_u8(*input_tuple(inp), aux=True)
    for k in ("vt_prob", "bp"):
        close(got[k], out[k], SCORE_ATOL, "%s:%s" % (case, k))
    for k in ("bin_logits", "vt_logits", "af", "cov", "vb", "vr"):
        close(got[k], out[k], TAP_RTOL, "%s:%s" % (case, k))
    F, Fs = net.handle.query("feature_width"), net.handle.query("feature_stride")
    B, R, L = inp["reads"].shape
    feat = net.handle.read_buffer("feature", B * Fs).reshape(B, Fs)[:, :F]
    close(feat, out["feature"], TAP_RTOL, case + ":feature")
    hid = net.handle.read_buffer("hidden1", B * cfg.fc_sizes[1]).reshape(B, -1)
    close(hid, out["hidden"], TAP_RTOL, case + ":hidden")
    cpad = net.handle.query("cpad")
    for layer in (2, 7):
        net.handle.set_tap(layer)
        net.forward_u8(*input_tuple(inp))
        tap = net.handle.read_buffer("tap", B * R * L * cpad).reshape(B, R, L, cpad)
        ref = out["conv%d" % layer]
        g = np.transpose(tap[:ref.shape[0], :, :, :ref.shape[1]], (0, 3, 1, 2))
        close(g, ref, TAP_RTOL, "%s:conv%d" % (case, layer))
        # ... column by column: the worst column must not be one next to the cut (a leak of the inner zero padding would sit there)
        err_col = np.abs(g - ref).max(axis=(0, 1, 2))
        mid = (L + 1) // 2
        assert err_col[mid - 12:mid + 12].max() <= TAP_RTOL * max(1.0, float(np.abs(ref).max())), (case, layer, int(err_col.argmax()))
        assert np.all(tap[..., ref.shape[1]:] == 0), "pad channels must stay zero"
    net.close()


@pytest.mark.parametrize("case", long_cases())
def test_long_window_golden_outputs_and_taps_bf16x3(case):
    """The same two reference runs on the split-bf16 kernel (precision 1): scores 1e-4 absolute, the six heads, the feature row, the
    hidden layer and the conv2 / conv7 taps within 1e-4 of the tensor's magnitude -- the bars the 201-column fixtures are held to."""
    spec, w, inp, out = load_case(case)
    cfg = cfg_from(spec, precision=PRECISION_BF16X3)
    net = DanNet(cfg).load_state_dict(w)
    assert net.handle.query("bf16x3_split_kernel") == 1
    got = net.forward_u8(*input_tuple(inp), aux=True)
    errs = {}
    for k in ("vt_prob", "bp"):
        errs[k] = float(np.abs(got[k] - out[k]).max())
        close(got[k], out[k], SCORE_ATOL, "%s:%s" % (case, k))
    for k in ("bin_logits", "vt_logits", "af", "cov", "vb", "vr"):
        errs[k] = float(np.abs(got[k] - out[k]).max()) / max(1.0, float(np.abs(out[k]).max()))
        close(got[k], out[k], TAP_RTOL, "%s:%s" % (case, k))
    F, Fs = net.handle.query("feature_width"), net.handle.query("feature_stride")
    B, R, L = inp["reads"].shape
    feat = net.handle.read_buffer("feature", B * Fs).reshape(B, Fs)[:, :F]
    close(feat, out["feature"], TAP_RTOL, case + ":feature")
    hid = net.handle.read_buffer("hidden1", B * cfg.fc_sizes[1]).reshape(B, -1)
    close(hid, out["hidden"], TAP_RTOL, case + ":hidden")
    cpad = net.handle.query("cpad")
    for layer in (2, 7):
        net.handle.set_tap(layer)
        net.forward_u8(*input_tuple(inp))
        tap = net.handle.read_buffer("tap", B * R * L * cpad).reshape(B, R, L, cpad)
        ref = out["conv%d" % layer]
        g = np.transpose(tap[:ref.shape[0], :, :, :ref.shape[1]], (0, 3, 1, 2))
        close(g, ref, TAP_RTOL, "%s:conv%d" % (case, layer))
        assert np.all(tap[..., ref.shape[1]:] == 0), "pad channels must stay zero"
    print("bf16x3 %s: " % case + " ".join("%s %.2g" % kv for kv in errs.items()))
    net.close()


LONG_STRUCTURES = {
    # (reads, length, layers, pools, residual_start, c_init, c_final, bottleneck, extra)
    "l240_pools_1_3": dict(reads=9, length=240, layers=5, pool_layers=(1, 3), residual_start=4, c_init=48, c_final=128, bottleneck=8),
    "l304_production_structure": dict(reads=6, length=304, layers=7, pool_layers=(2,), residual_start=5, c_init=128, c_final=128, bottleneck=32),
    "l209_one_past_the_image": dict(reads=5, length=209, layers=4, pool_layers=(), residual_start=2, c_init=16, c_final=16, bottleneck=0, use_bn=False),
    "l257_dilation_3_no_mask": dict(reads=11, length=257, layers=6, pool_layers=(4,), residual_start=0, c_init=128, c_final=48, bottleneck=32,
                                    dil_mid=3, dil_final=1, use_mask=False, use_q=False),
    "l288_one_layer_segments": dict(reads=3, length=288, layers=4, pool_layers=(1, 2, 3), residual_start=3, c_init=48, c_final=48, bottleneck=8),
}


@pytest.mark.parametrize("name", sorted(LONG_STRUCTURES))
def test_long_window_structures_all_forms_agree(name):
    """Structures at 209 / 240 / 257 / 288 / 304 columns against the fp32 oracle: the Winograd form (where the dilations allow it),
    the direct form, empty rows computed once per site (persistent workgroups walking (row, unit) pairs) and 2-site chunks --
    the last two bit-identical to the first."""
    kw = dict(LONG_STRUCTURES[name], fc_sizes=(32, 16))
    _all_forms_agree(name, DanConfig(**kw), 900 + sorted(LONG_STRUCTURES).index(name), 5)


def _random_long_structure(seed):
    """tests/test_hip_parity.py::random_structure at 209..304 columns, with the dilations drawn as well (dilation 2 everywhere: the
    Winograd form, six tiles per lane on the split kernel; anything else: the direct form on the 13-tile image; dilation 3-4 and many
    layers: bf16x3 units above 192 columns, its seven-tile split form)."""
    from test_hip_parity import random_structure
    kw, n = random_structure(seed, (209, 305))
    rng = np.random.default_rng(7000 + seed)
    if rng.random() < 0.5:
        kw["dil_mid"], kw["dil_final"] = int(rng.integers(1, 5)), int(rng.choice([1, 2, 4]))
    return kw, max(2, n)


@pytest.mark.parametrize("seed", list(range(100, 110)))
def test_long_window_random_structures_all_forms_agree(seed):
    """Round 6: the split kernels were re-tiled for the window (six Winograd tiles per lane / six column tiles per wave for units of
    up to 190 / 192 columns); the five named structures above are joined by seeded random ones -- reads, window 209..304, layers,
    pools, residual start, widths, bottleneck, input planes, dilations -- through the same checks.  A structure whose segment reaches
    further sideways than a unit has room for is refused by dan_create (and skipped here)."""
    kw, n = _random_long_structure(seed)
    try:
        DanNet(DanConfig(**kw)).close()
    except RuntimeError as e:
        assert "sideways" in str(e), e
        pytest.skip("refused by dan_create: %s" % str(e)[:80])
    _all_forms_agree("seed %d %s" % (seed, kw), DanConfig(**kw), 4000 + seed, n)


def _all_forms_agree(name, cfg, seed, n_sites):
    sd = random_state_dict(cfg, seed=seed)
    batch = synth.make_sites(n_sites, reads=cfg.reads, length=cfg.length, seed=seed + 50)
    want = dan_forward_oracle(sd, cfg, *batch.arrays())
    outs = {}
    for tag, c, kwn in (("auto", cfg, {}), ("direct", dataclasses.replace(cfg, conv_algo=1), {}),
                        ("skip", dataclasses.replace(cfg, skip_empty_rows=True), {}), ("chunks", cfg, dict(chunk_sites=2, max_batch=4))):
        net = DanNet(c, **kwn).load_state_dict(sd)
        outs[tag] = got = net.forward_u8(*batch.arrays(), aux=True)
        net.close()
        for k in ("vt_prob", "bp"):
            close(got[k], want[k], SCORE_ATOL, "%s %s %s" % (name, tag, k))
        for k in ("vt_logits", "bin_logits", "af", "cov", "vb", "vr"):
            close(got[k], want[k], TAP_RTOL, "%s %s %s" % (name, tag, k))
    for k in outs["auto"]:
        assert np.array_equal(outs["auto"][k], outs["skip"][k]), (name, "skip", k)
        assert np.array_equal(outs["auto"][k], outs["chunks"][k]), (name, "chunks", k)
    # ... and on the bf16x3 kernel: the oracle at the same bars (the score bar grows with the logits' magnitude above 16, the stated
    # limit of a two-piece operand: tests/test_hip_parity.py::test_random_structures_bf16x3), skip / chunks bit-identical
    cx = dataclasses.replace(cfg, precision=PRECISION_BF16X3)
    mag = max(float(np.abs(want["vt_logits"]).max()), float(np.abs(want["bin_logits"]).max()))
    bar = SCORE_ATOL * max(1.0, mag / 16.0)
    xo = {}
    for tag, c, kwn in (("x3", cx, {}), ("x3 skip", dataclasses.replace(cx, skip_empty_rows=True), {}), ("x3 chunks", cx, dict(chunk_sites=2, max_batch=4))):
        net = DanNet(c, **kwn).load_state_dict(sd)
        xo[tag] = got = net.forward_u8(*batch.arrays(), aux=True)
        net.close()
        for k in ("vt_prob", "bp"):
            close(got[k], want[k], bar, "%s %s %s" % (name, tag, k))
        close(got["vt_logits"], want["vt_logits"], TAP_RTOL, "%s %s vt_logits" % (name, tag))
    for k in xo["x3"]:
        assert np.array_equal(xo["x3"][k], xo["x3 skip"][k]), (name, "x3 skip", k)
        assert np.array_equal(xo["x3"][k], xo["x3 chunks"][k]), (name, "x3 chunks", k)


def test_config5_shape_in_fp32():
    """BASELINE config 5's shape (128 reads x 301 columns) at precision 0, production width: scores within 1e-4 of the oracle; the
    allele masks of one site moved next to the cut between the units (column 151) and of another to the window's last column,
    so that the agreement predicates -- which a unit takes over the WHOLE window -- decide channels the other unit encodes."""
    cfg = DanConfig(reads=128, length=301)
    assert cfg.feature_width == 105728
    sd = random_state_dict(cfg, seed=3)
    batch = synth.make_sites(3, reads=128, length=301, seed=4)
    arrs = [a.copy() for a in batch.arrays()]
    for site, col in ((1, 151), (2, 300)):
        src = int(np.flatnonzero(arrs[4][site])[0])
        arrs[4][site, col], arrs[5][site, col] = arrs[4][site, src], arrs[5][site, src]
        arrs[4][site, src] = arrs[5][site, src] = 0
    net = DanNet(cfg).load_state_dict(sd)
    got = net.forward_u8(*arrs, aux=True)
    net.close()
    want = dan_forward_oracle(sd, cfg, *arrs)
    for k in ("vt_prob", "bp"):
        close(got[k], want[k], SCORE_ATOL, k)
    close(got["vt_logits"], want["vt_logits"], TAP_RTOL, "vt_logits")
    print("128 x 301 fp32: max |vt_prob - oracle| %.3g" % float(np.abs(got["vt_prob"] - want["vt_prob"]).max()))
    net = DanNet(dataclasses.replace(cfg, precision=PRECISION_BF16X3)).load_state_dict(sd)
    gx = net.forward_u8(*arrs, aux=True)
    net.close()
    for k in ("vt_prob", "bp"):
        close(gx[k], want[k], SCORE_ATOL, "bf16x3 " + k)
    close(gx["vt_logits"], want["vt_logits"], TAP_RTOL, "bf16x3 vt_logits")
    print("128 x 301 bf16x3: max |vt_prob - oracle| %.3g" % float(np.abs(gx["vt_prob"] - want["vt_prob"]).max()))


def test_window_limits_per_precision():
    """fp32 and bf16x3 take windows up to 304 columns (two units per read), plain bf16 as well (its two images hold 304); 305 is
    refused everywhere, and so is -- at precisions 0 and 1 -- a long window whose segment reaches further sideways than half an
    image has room for (sixteen unpooled layers at dilation 4: 61 columns + 152 > 208), with the reason in the message."""
    DanNet(DanConfig(reads=8, length=301)).close()
    DanNet(DanConfig(reads=8, length=304, precision=PRECISION_BF16)).close()
    DanNet(DanConfig(reads=8, length=301, precision=PRECISION_BF16X3)).close()
    for prec in (0, PRECISION_BF16X3, PRECISION_BF16):
        with pytest.raises(RuntimeError, match="length"):
            DanNet(DanConfig(reads=8, length=305, precision=prec))
    deep = dict(reads=4, layers=16, pool_layers=(), residual_start=0, dil_mid=4, dil_final=4, c_init=16, c_final=16, bottleneck=4, fc_sizes=(16, 8))
    for prec in (0, PRECISION_BF16X3):
        with pytest.raises(RuntimeError, match="sideways"):
            DanNet(DanConfig(length=304, precision=prec, **deep))
    DanNet(DanConfig(length=208, **deep)).close()              # the same network on one unit is fine
    DanNet(DanConfig(length=280, **deep)).close()              # 140 + 61 <= 208
