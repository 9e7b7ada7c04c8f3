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


def input_tuple(inputs):
    return tuple(inputs[k] for k in ("reads", "qual", "strand", "ref", "ref_mask", "var_mask"))
