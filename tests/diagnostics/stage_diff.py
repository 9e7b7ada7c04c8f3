#!/usr/bin/env python3
"""GPU debugging aid: per-stage max error of the HIP path against the oracle (encoded input, every conv
layer, feature blocks, hidden, scores) for one configuration.  Usage: python tests/diagnostics/stage_diff.py [small|prod]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dl4vc_amd.config import DanConfig          # noqa: E402
from dl4vc_amd.model import DanNet              # noqa: E402
from dl4vc_amd import synth                     # noqa: E402
from oracle.dan_oracle import dan_forward_oracle, random_state_dict   # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    prec = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if which == "small":
        cfg = DanConfig(reads=8, c_init=16, c_final=16, bottleneck=4, fc_sizes=(16, 8), precision=prec)
    elif which == "mid":
        cfg = DanConfig(reads=16, c_init=128, c_final=128, bottleneck=32, fc_sizes=(64, 32), precision=prec)
    elif which == "stress":
        cfg = DanConfig(reads=128, length=301, precision=prec)
    else:
        cfg = DanConfig(reads=64, precision=prec)
    print("config", which, "precision", prec)
    sd = random_state_dict(cfg, seed=1)
    batch = synth.make_sites(3 if which != "stress" else 2, reads=cfg.reads, length=cfg.length, seed=2)
    want = dan_forward_oracle(sd, cfg, *batch.arrays(), taps=True)
    net = DanNet(cfg).load_state_dict(sd)
    B, R, L = batch.reads.shape
    cpad = net.handle.query("cpad")

    def report(name, got, ref):
        err = np.abs(got.astype(np.float64) - ref)
        print("%-10s max|err| %.3e   max|ref| %.3e   argmax %s" % (name, err.max(), np.abs(ref).max(),
                                                                  np.unravel_index(err.argmax(), err.shape)))

    for layer in range(0, cfg.layers + 1):
        net.handle.set_tap(layer)
        net.forward_u8(*batch.arrays())
        tap = net.handle.read_buffer("tap", B * R * L * cpad).reshape(B, R, L, cpad)
        ref = want["encoded"] if layer == 0 else want["conv%d" % layer]          # (B,C,R,L)
        C = ref.shape[1]
        if layer == 0:
            # canonical 48-channel order == reference order when q, strand and mask are all on
            got = np.transpose(tap[..., :C], (0, 3, 1, 2))
        else:
            got = np.transpose(tap[..., :C], (0, 3, 1, 2))
        report("encoded" if layer == 0 else "conv%d" % layer, got, ref)
    net.handle.set_tap(-1)
    got = net.forward_u8(*batch.arrays(), aux=True)
    F, Fs = net.handle.query("feature_width"), net.handle.query("feature_stride")
    feat = net.handle.read_buffer("feature", B * Fs).reshape(B, Fs)[:, :F]
    CL = cfg.c_final * L
    report("feat.max", feat[:, :CL], want["feature"][:, :CL])
    report("feat.mean", feat[:, CL:2 * CL], want["feature"][:, CL:2 * CL])
    if cfg.bottleneck:
        report("feat.hw", feat[:, 2 * CL:], want["feature"][:, 2 * CL:])
    hid = net.handle.read_buffer("hidden1", B * cfg.fc_sizes[1]).reshape(B, -1)
    report("hidden", hid, want["hidden"])
    for k in ("bin_logits", "vt_logits", "vt_prob", "bp", "af", "cov", "vb", "vr"):
        report(k, got[k], want[k])


if __name__ == "__main__":
    main()
