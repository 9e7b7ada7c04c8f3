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


def many_reads_cases():
    """More than 100 reads per site (oracle/gen_golden.py::gen_many_reads_fixtures: the reference's fp32 forward with its module
    constant MAX_READS raised to the fixture's read count): 128 x 301 (BASELINE config 5's shape) and 101 x 201."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "reads_*.npz")))


def check_against_bf16_fixture(got, ref, what, atol=2e-5):
    """A CPU evaluation in a bf16 mode against a fixture the reference wrote in that mode ON ANOTHER MACHINE.  On the machine class
    the fixture was generated on (the build container) the oracle runs the same torch kernels in the same order as the reference did
    and the two agree to fp32 roundoff (`atol` of the tensor's magnitude; bit for bit since round 6).  A host with another vector
    ISA sums its convolutions in another order; a last-bit difference then flips the bf16 rounding of an operand here and there
    (measured by forcing ATEN_CPU_CAPABILITY / ONEDNN_MAX_CPU_ISA to AVX2: 0.25 % of a 128 x 301 conv7 tap beyond 2e-5, worst
    3.7e-3 of the maximum; sums over a read -- highways, the feature row, heads -- 1.4e-3).  Such a run is accepted on the
    flip-consistent bar: an activation tap >= 99 % within `atol` and nowhere beyond 2e-2, anything else within 5e-3 -- a bar the
    fp32 evaluation of the same site misses by far (every element of a tap moves; the callers assert that).  Returns which bar held."""
    scale = max(1.0, float(np.abs(ref).max())) if ref.size else 1.0
    d = np.abs(np.asarray(got, np.float64) - ref) / scale
    worst = float(d.max()) if d.size else 0.0
    if worst <= atol:
        return "roundoff"
    if what.split(":")[-1].startswith("conv"):
        inside = float((d <= atol).mean())
        assert inside >= 0.99 and worst <= 2e-2, "%s: %.4f of the elements within %.0e, worst %.3g of max" % (what, inside, atol, worst)
    else:
        assert worst <= 5e-3, "%s: %.3g of max" % (what, worst)
    return "flip-consistent"


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
