"""The C-ABI library loads and exports every symbol include/dl4vc_dan.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from dl4vc_amd import capi
from dl4vc_amd.config import DanConfig, UnsupportedModelOption, production_config

HEADER = os.path.join(ROOT, "include", "dl4vc_dan.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.isfile(capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return capi.load_library()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dan_[a-z_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = declared_functions()
    assert set(names) == set(capi.SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_loader_header_symbols_exported():
    from dl4vc_amd import loader
    if not loader.available():
        import __graft_entry__ as g
        g.build()
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dl4vc_loader.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b((?:dl|pe)_[a-z_]+)\s*\(", text)))
    assert set(names) == set(loader.SYMBOLS)
    nl = loader.load_library()
    for n in names:
        assert hasattr(nl, n), n


def test_training_header_symbols_exported(lib):
    """include/dl4vc_dan_train.h: every declared entry point is exported and bound."""
    from dl4vc_amd import train
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dl4vc_dan_train.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(dan_train_[a-z_]+)\s*\(", text)))
    assert set(names) == set(train.TRAIN_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n
    body = re.search(r"typedef struct dan_train_hyper \{(.*?)\} dan_train_hyper;", text, flags=re.S).group(1)
    fields = []
    for decl in re.findall(r"float\s+([a-z0-9_, ]+);", body):
        fields += [f.strip() for f in decl.split(",")]
    assert fields == [f[0] for f in train._CHyper._fields_]
    body = re.search(r"typedef struct dan_train_targets \{(.*?)\} dan_train_targets;", text, flags=re.S).group(1)
    assert re.findall(r"\*\s*([a-z_]+)\s*;", body) == [f[0] for f in train._CTargets._fields_]


def test_abi_version(lib):
    assert lib.dan_abi_version() == capi.ABI_VERSION


def test_config_struct_layout():
    # 23 x 4-byte fields, no padding: must match struct dan_config in the header
    assert ctypes.sizeof(capi.DanCConfig) == 4 * 23
    cc = capi.c_config(production_config(reads=64), device_id=3)
    assert (cc.reads, cc.length, cc.layers, cc.device_id) == (64, 201, 7, 3)
    assert cc.pool_layers_mask == 1 << 2 and list(cc.fc_sizes) == [1024, 256]


def test_header_struct_fields_match_binding():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct dan_config \{(.*?)\} dan_config;", text, flags=re.S).group(1)
    names = re.findall(r"u?int32_t\s+([a-z0-9_]+)(?:\[\d+\])?\s*;", body)
    assert names == [f[0] for f in capi.DanCConfig._fields_]


def test_create_rejects_bad_config_without_touching_the_gpu(lib):
    h = ctypes.c_void_p()
    cc = capi.c_config(DanConfig(reads=8), 0)
    cc.length = 400                      # beyond the LDS-resident window of the fp32 path
    assert lib.dan_create(ctypes.byref(cc), ctypes.byref(h)) == -1
    assert b"length" in lib.dan_last_error(None)
    assert not h.value
    cc = capi.c_config(DanConfig(reads=8, dil_mid=3, conv_algo=2), 0)      # Winograd form needs dilation 2
    assert lib.dan_create(ctypes.byref(cc), ctypes.byref(h)) == -1
    assert b"dilation 2" in lib.dan_last_error(None)


def test_the_library_reads_no_environment_variable():
    """kernel forms are chosen through dan_config (bf16_form), not through the process environment."""
    src = os.path.join(ROOT, "dl4vc_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".cpp", ".h")) and not name.startswith(("dan_loader", "dan_pileup")):
            assert "getenv" not in open(os.path.join(src, name)).read(), name
    h = ctypes.c_void_p()
    cc = capi.c_config(DanConfig(reads=8), 0)
    cc.bf16_form = 1                       # a form of the precision-2 kernel only
    lib_ = capi.load_library()
    assert lib_.dan_create(ctypes.byref(cc), ctypes.byref(h)) == -1 and b"bf16_form" in lib_.dan_last_error(None)


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path|no HIP device"):
        capi.DanHandle(DanConfig(reads=8))


def test_unsupported_flags_are_rejected():
    import types
    base = dict(model_conv_layers=7, model_highway_single_reads=True, model_concat_hw_reads=True,
                model_pool_combine_dimension=0)
    DanConfig.from_args(types.SimpleNamespace(**base))
    for bad in (dict(early_loss_layers=[3]), dict(use_transformer=True), dict(model_pool_combine_dimension=2048),
                dict(model_skip_final_maxpool=True), dict(model_concat_hw_reads=False), dict(model_use_AF=True)):
        with pytest.raises(UnsupportedModelOption):
            DanConfig.from_args(types.SimpleNamespace(**{**base, **bad}))
    with pytest.raises(UnsupportedModelOption):
        DanConfig(residual_start=1)


def test_flop_model_matches_survey():
    # 397 184 MAC per read-position; 16.12 / 10.35 GFLOP per site
    assert production_config().macs_per_position() == 397184
    assert abs(production_config(100).flops_per_site() / 1e9 - 16.12) < 0.01
    assert abs(production_config(64).flops_per_site() / 1e9 - 10.35) < 0.01
    assert production_config(64).input_bytes_per_site() == 39195


def test_pe_options_struct_matches_the_ctypes_mirror():
    from dl4vc_amd import loader
    text = open(os.path.join(ROOT, "include", "dl4vc_loader.h")).read()
    body = re.search(r"typedef struct pe_options \{(.*?)\} pe_options;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"int32_t\s+([a-z_]+)\s*;", body)
    assert names == [f[0] for f in loader.PileupOptions._fields_]
