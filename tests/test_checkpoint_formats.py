"""The checkpoint seam (SURVEY.md section 8b "Weights contract"): what `--modelload` must open.

The published `checkpoint.pth.tar` was written by a torch-1.2-era `torch.save` (docs/Step-by-step.md:14): the LEGACY
serialisation (a pickle stream with the tensor storages appended -- not the zip container torch has written since 1.6), a dict
`{'epoch', 'state_dict', 'best_loss', 'optimizer'}` (main.py:194-199) whose state-dict keys carry DataParallel's `module.`
prefix (main.py:117).  Every checkpoint the other tests write is the new zip format; this file writes the old one and loads it
through `dl4vc_amd.model.load_checkpoint` (main.py:121-124's `torch.load(map_location='cpu')['state_dict']`).  CPU only; the
GPU twin (`main.py --modelload <legacy file>` end to end) is tests/test_cli_gpu.py::test_main_py_loads_a_legacy_format_checkpoint."""
import zipfile

import numpy as np
import pytest
import torch

from dl4vc_amd.config import DanConfig
from dl4vc_amd.model import load_checkpoint, normalise_state_dict
from oracle.dan_oracle import random_state_dict

SMALL = DanConfig(reads=8, c_init=16, c_final=16, bottleneck=4, fc_sizes=(32, 16))


def reference_style_checkpoint(cfg, seed, dropout_indices=True):
    """A dict shaped like the reference's save_checkpoint argument: module.-prefixed tensors incl. num_batches_tracked counters, the
    FC layers under their nn.Sequential indices (1 / 4 behind Dropout modules, model.py:369-377), an Adam state dict over the
    parameters in order (main.py:116,198)."""
    sd = random_state_dict(cfg, seed=seed)
    out = {}
    for k, v in sd.items():
        if k.startswith("fc."):                                     # our generator may name them by structure
            i, part = k.split(".")[1:]
            k = "conv2hidden.%d.%s" % ((1, 4)[int(i)] if dropout_indices else (0, 3)[int(i)], part)
        elif k.startswith("conv2hidden.") and not dropout_indices:
            i, part = k.split(".")[1:]
            k = "conv2hidden.%d.%s" % ({1: 0, 4: 3}.get(int(i), int(i)), part)
        out["module." + k] = torch.from_numpy(np.array(v))
    for layer in range(cfg.layers):
        out["module.bn1D_layers.%d.num_batches_tracked" % layer] = torch.tensor(1234 + layer, dtype=torch.int64)
    params = [torch.nn.Parameter(t.clone()) for k, t in out.items()
              if t.dtype == torch.float32 and "running_" not in k and not k.endswith(".pe")]
    opt = torch.optim.Adam(params, lr=1e-3)
    for p in params:
        p.grad = torch.full_like(p, 1e-3)
    opt.step()
    return {"epoch": 7, "state_dict": out, "best_loss": 0.123, "optimizer": opt.state_dict()}, sd


@pytest.mark.parametrize("legacy", [True, False])
def test_load_checkpoint_opens_the_legacy_and_the_zip_format(tmp_path, legacy):
    ck, sd = reference_style_checkpoint(SMALL, seed=3)
    path = str(tmp_path / "checkpoint.pth.tar")
    torch.save(ck, path, _use_new_zipfile_serialization=not legacy)
    assert zipfile.is_zipfile(path) == (not legacy), "the fixture must be in the format the test names"
    got = load_checkpoint(path)
    want = normalise_state_dict(sd)
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    for k in want:
        assert got[k].dtype == np.float32 and got[k].flags["C_CONTIGUOUS"]
        assert np.array_equal(got[k], want[k]), k
    assert not any("num_batches_tracked" in k or k.startswith("module.") for k in got)
    assert "fc.0.weight" in got and "fc.1.bias" in got and not any(k.startswith("conv2hidden.") for k in got)


def test_fc_layers_are_mapped_by_structure_not_by_index(tmp_path):
    """A model built without Dropout modules numbers its Linear layers 0 / 3, one with them 1 / 4 (model.py:369-377)."""
    ck, sd = reference_style_checkpoint(SMALL, seed=4, dropout_indices=False)
    assert any(k.startswith("module.conv2hidden.0.") for k in ck["state_dict"])
    path = str(tmp_path / "c.pth.tar")
    torch.save(ck, path, _use_new_zipfile_serialization=False)
    got = load_checkpoint(path)
    want = normalise_state_dict(sd)
    for k in ("fc.0.weight", "fc.0.bias", "fc.1.weight", "fc.1.bias"):
        assert np.array_equal(got[k], want[k]), k


def test_a_bare_state_dict_file_loads_too(tmp_path):
    """`torch.save(model.state_dict())` without the wrapping dict (what a user exporting weights by hand writes)."""
    ck, sd = reference_style_checkpoint(SMALL, seed=5)
    path = str(tmp_path / "weights.pt")
    torch.save(ck["state_dict"], path, _use_new_zipfile_serialization=False)
    got = load_checkpoint(path)
    assert np.array_equal(got["embeddings.weight"], sd["embeddings.weight"])


def test_legacy_checkpoint_with_cuda_storages_maps_to_cpu(tmp_path):
    """The published file was saved from GPU tensors; `map_location='cpu'` (main.py:123) is what lets a box without that device
    open it.  A legacy stream tags each storage with its location; rewrite the tags of a CPU-saved file to 'cuda:0' and load."""
    ck, sd = reference_style_checkpoint(SMALL, seed=6)
    path = str(tmp_path / "gpu_saved.pth.tar")
    torch.save(ck, path, _use_new_zipfile_serialization=False)
    raw = open(path, "rb").read()
    # the pickled persistent ids are tuples ('storage', <type>, <key>, <location>, <numel>, ...); location is the short string 'cpu'
    tagged = raw.replace(b"X\x03\x00\x00\x00cpu", b"X\x06\x00\x00\x00cuda:0")
    assert tagged != raw
    open(path, "wb").write(tagged)
    got = load_checkpoint(path)
    assert np.array_equal(got["embeddings.weight"], sd["embeddings.weight"])
