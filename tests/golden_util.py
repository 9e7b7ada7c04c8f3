"""Loading of the committed golden fixtures (written by oracle/gen_golden.py)."""
import glob
import json
import os

import numpy as np

from conftest import GOLDEN


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    spec = json.loads(bytes(z["spec_json"]).decode())
    weights = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    inputs = {k[3:]: z[k] for k in z.files if k.startswith("in:")}
    outputs = {k[4:]: z[k] for k in z.files if k.startswith("out:")}
    return spec, weights, inputs, outputs


def model_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "dan_*.npz")))


def long_cases():
    """Windows of 209..304 columns (oracle/gen_golden.py::gen_long_window_fixtures: the reference's fp32 forward)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "long_*.npz")))


def input_tuple(inputs):
    return tuple(inputs[k] for k in ("reads", "qual", "strand", "ref", "ref_mask", "var_mask"))


def train_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "train_*.npz")))


def load_train_case(name):
    """tests/golden/train_*.npz (oracle/gen_golden_train.py: the reference's own trainer.train on one or two batches) ->
    (spec dict, hyper dict, weights, [step dict], final state, adam moments, close_examples)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    spec = json.loads(bytes(z["spec_json"]).decode())
    hyper = json.loads(bytes(z["hyper_json"]).decode())
    weights = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    steps = []
    for s in range(int(z["n_steps"])):
        tag = "s%d:" % s
        st = {"planes": tuple(z[tag + "in:" + k] for k in ("reads", "qual", "strand", "ref", "ref_mask", "var_mask")),
              "targets": {k[len(tag) + 3:]: z[k] for k in z.files if k.startswith(tag + "tg:")},
              "masks": [], "out": {k[len(tag) + 4:]: z[k] for k in z.files if k.startswith(tag + "out:")},
              "grad": {k[len(tag) + 5:]: z[k] for k in z.files if k.startswith(tag + "grad:")},
              "vcfrec": bytes(z[tag + "vcfrec"]).decode().split("\n")}
        i = 0
        while tag + "mask%d" % i in z.files:
            shape = tuple(int(v) for v in z[tag + "mask%d_shape" % i])
            st["masks"].append(np.unpackbits(z[tag + "mask%d" % i])[:int(np.prod(shape))].reshape(shape))
            i += 1
        for k in ("loss", "bin", "vt", "grad_norm", "bin_close", "vt_close"):
            st[k] = z[tag + k]
        steps.append(st)
    final = {k[6:]: z[k] for k in z.files if k.startswith("final:")}
    adam = {k: z[k] for k in z.files if k.startswith(("adam_m:", "adam_v:"))}
    return spec, hyper, weights, steps, final, adam, z["close_examples"]
