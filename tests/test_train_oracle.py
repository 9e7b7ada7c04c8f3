"""The training oracle (oracle/dan_train_oracle.py) against the fixtures the reference's own training loop produced
(tests/golden/train_*.npz, oracle/gen_golden_train.py): losses, every gradient, clip norm, Adam state, BN running stats."""
import numpy as np
import pytest

from golden_util import load_train_case, train_cases
from oracle.dan_train_oracle import TrainHyper, train_step_oracle, state_errors, example_weights, trainable

GRAD_RTOL = 1e-4          # of the tensor's max magnitude


def test_fixture_set_is_complete():
    assert set(train_cases()) >= {"train_small", "train_var_nobn", "train_var_pool24", "train_var_l5res2", "train_var_nohw",
                                  "train_var_cfinal"}


@pytest.mark.parametrize("case", train_cases())
def test_train_oracle_matches_reference_training_loop(case):
    spec, hyper, w, steps, final, adam, close = load_train_case(case)
    hp = TrainHyper(**hyper)
    state, moments = dict(w), None
    for s, st in enumerate(steps):
        # example weights are host logic: (is_snp + (1 - is_snp) * non_snp_weight), trainer.py:169-172
        np.testing.assert_allclose(example_weights(st["targets"]["is_snp"], hp), st["targets"]["weight"])
        r = train_step_oracle(state, spec, st["planes"], st["targets"], hp, dropout_masks=st["masks"], adam_state=moments,
                              step=s + 1)
        for k in ("loss", "bin", "vt"):
            assert abs(float(r[k]) - float(st[k])) < 1e-5, (case, s, k)
        for k, ref in st["out"].items():
            assert np.abs(r["out:" + k] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (case, s, k)
        assert np.array_equal(r["bin_close"], st["bin_close"]) and np.array_equal(r["vt_close"], st["vt_close"])
        assert abs(float(r["grad_norm"]) - float(st["grad_norm"])) <= 1e-5 * float(st["grad_norm"])
        for k, g in st["grad"].items():
            err = np.abs(r["grad:" + k] - g).max()
            assert err <= GRAD_RTOL * np.abs(g).max() + 1e-12, (case, s, k, err, np.abs(g).max())
        # parameters the reference leaves without a gradient: ours must be exactly zero there
        for k in r:
            if k.startswith("grad:") and k[5:] not in st["grad"]:
                assert not np.any(r[k]), k
        state = dict(state)
        for k, v in r.items():
            if k.startswith("new:"):
                state[k[4:]] = v
        moments = {k: v for k, v in r.items() if k.startswith(("m:", "v:"))}
    last = steps[-1]["grad"]
    for k, ref in final.items():
        a, b = state_errors(state[k], ref, last.get(k), hp.lr)
        assert a < 1e-4 and b < 2.1 * len(steps), (case, k, a, b)
    for k, ref in adam.items():
        mine = moments[("m:" if k.startswith("adam_m:") else "v:") + k.split(":", 1)[1]]
        assert np.abs(mine - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-30), (case, k)
    # the loop writes the VT close flags of each batch back into the dataset (trainer.py:263-264)
    flags = np.concatenate([st["vt_close"] for st in steps])
    assert np.array_equal(close[:len(flags)], flags)


def test_dropout_masks_matter_and_bn_uses_batch_statistics():
    """Guards against an oracle that silently ignores its train-mode inputs."""
    spec, hyper, w, steps, *_ = load_train_case("train_small")
    hp, st = TrainHyper(**hyper), steps[0]
    base = train_step_oracle(w, spec, st["planes"], st["targets"], hp, dropout_masks=st["masks"])
    ones = [np.ones_like(m) for m in st["masks"]]
    other = train_step_oracle(w, spec, st["planes"], st["targets"], hp, dropout_masks=ones)
    assert abs(float(base["loss"]) - float(other["loss"])) > 1e-6
    half = tuple(p[:3] for p in st["planes"])
    tg = {k: v[:3] for k, v in st["targets"].items()}
    sub = train_step_oracle(w, spec, half, tg, hp, dropout_masks=[m[:3] for m in st["masks"]])
    assert np.abs(sub["out:vt_logits"] - base["out:vt_logits"][:3]).max() > 1e-6      # batch statistics couple the sites


def test_forced_decisions_mode_reproduces_the_step_it_took_them_from_and_follows_a_flipped_one():
    """oracle.dan_train_oracle.train_forward(forced=...) -- the GPU tests' edge branch imposes the device's ReLU / max decisions on
    the float64 oracle.  With the oracle's OWN decisions imposed every output and gradient is reproduced (x * mask == relu(x)
    wherever the mask is the sign test); with one conv-ReLU decision flipped on an operand near zero, the result is the gradient of
    that neighbouring network: it differs, and is reproduced by flipping the same decision again."""
    import torch
    spec, hyper, w, steps, *_ = load_train_case("train_small")
    hp, st = TrainHyper(**hyper), steps[0]
    base = train_step_oracle(w, spec, st["planes"], st["targets"], hp, dropout_masks=st["masks"], dtype=torch.float64, taps=True)
    forced = {}
    for l in range(1, spec["layers"] + 1):
        forced["relu%d" % l] = base["tap:pre%d" % l] > 0
        forced["hrelu%d" % l] = base["tap:hpre%d" % l] > 0
    forced["argmax"] = base["tap:conv%d" % spec["layers"]].argmax(axis=2)
    same = train_step_oracle(w, spec, st["planes"], st["targets"], hp, dropout_masks=st["masks"], dtype=torch.float64, forced=forced)
    for k, v in base.items():
        if k.startswith(("grad:", "out:")) or k in ("loss", "grad_norm"):
            assert np.abs(same[k] - v).max() <= 1e-12 * max(1.0, float(np.abs(v).max())), k
    pre = base["tap:pre3"]
    i = np.unravel_index(np.abs(pre).argmin(), pre.shape)              # the decision closest to its edge
    flipped = dict(forced)
    flipped["relu3"] = forced["relu3"].copy()
    flipped["relu3"][i] = ~flipped["relu3"][i]
    other = train_step_oracle(w, spec, st["planes"], st["targets"], hp, dropout_masks=st["masks"], dtype=torch.float64, forced=flipped)
    moved = max(float(np.abs(other[k] - base[k]).max()) / max(float(np.abs(base[k]).max()), 1e-30) for k in base if k.startswith("grad:"))
    assert moved > 1e-9, "flipping a decision changed nothing"
    again = train_step_oracle(w, spec, st["planes"], st["targets"], hp, dropout_masks=st["masks"], dtype=torch.float64, forced=flipped)
    assert all(np.array_equal(again[k], other[k]) for k in other if k.startswith("grad:"))
