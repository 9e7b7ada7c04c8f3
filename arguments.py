"""Command-line surface of ``main.py`` -- a drop-in for the reference's flag set.

``call_variants.sh`` passes training-only flags to the inference run as well (reference:
call_variants.sh:101-147), so every flag of the reference parser (arguments.py:5-135) is accepted with
the same name, type and default; the ones the inference hot path acts on are marked ``*``.  The table
form (rather than a transcription of the reference's ``add_argument`` calls) is deliberate: one row
per flag = (names, kind, default).  Kinds: ``flag`` = store_true, ``int``/``float``/``str`` scalars,
``ints``/``strs`` = one-or-more values.
"""
from __future__ import annotations

import argparse

_FLAGS = [
    # ---- files / run control
    ("--train_file", "str", None), ("--test_file*", "req-str", None), ("--debug", "flag", False),
    ("--loss-debug-freq", "int", 0), ("--max-train-batches", "int", 0), ("--max-test-batches*", "int", 0),
    ("--batch-size", "int", 1000), ("--test-batch-size*", "int", 1000), ("--epochs", "int", 20),
    ("--epochs_skip_eval", "int", 0), ("--lr", "float", 0.01), ("--lr-decay", "float", 1.0),
    ("--grad-clip", "float", 0.0),
    # ---- loss shaping (training only)
    ("--label-smoothing", "float", 0.0), ("--close_match_window", "float", 2.0), ("--focal_loss_gamma", "float", 0.0),
    ("--focal_loss_alpha", "float", 1.0), ("--close_examples_sample_rate", "float", 1.0),
    ("--save_hard_example_records", "flag", False),
    ("--use-var-type-threshold*", "flag", False), ("--binary-weight", "float", 1.0), ("--no-cuda", "flag", False),
    ("--seed*", "int", 1), ("--log-interval", "int", 10),
    ("--save_vcf_records*", "flag", False), ("--save_vcf_records_file*", "str", ""), ("--sample_vcf*", "str", None),
    ("--gpus*", "int", 1), ("--num-data-workers", "int", 5), ("--modelsave", "str", "checkpoint.pth.tar"),
    ("--modelload*", "str", None),
    ("--train_holdout_chromosomes", "strs", []), ("--test_holdout_chromosomes", "strs", []),
    ("--shuffle_test", "flag", False), ("--gatk-table", "str", ""), ("--giab-table", "str", ""),
    ("--test-trust-region-table", "str", ""), ("--train-trust-region-table", "str", ""),
    ("--non-trust-train-weight", "float", 0.01), ("--fp-train-weight", "float", 1.0), ("--trust-snp-only", "flag", False),
    ("--non-snp-train-weight", "float", 1.0), ("--auxillary-loss-weight", "float", 0.0),
    ("--auxillary-loss-bases-weight", "float", 0.1), ("--auxillary-loss-allele-weight", "float", 1.0),
    ("--aux-keep-candidate-af", "flag", False), ("--early_loss_layers*", "ints", []),
    ("--early_loss_weight", "float", 0.1), ("--learn_early_loss_weight", "flag", False),
    ("--layer_loss_weight", "float", 0.01),
    # ---- augmentation (training only)
    ("--delay_augmentation_epochs", "int", 0), ("--rm_var_reads_rate", "float", 0.0),
    ("--rm_non_var_reads_rate", "float", 0.0), ("--training_use_directional_augmentation", "flag", False),
    ("--augmented_example_weight", "float", 0.2), ("--delta_loss_weight", "float", 10.0),
    ("--augment-single-reads", "flag", False), ("--augment-reference", "flag", False),
    ("--reads-dynamic-downsample-rate", "float", 0.0), ("--reads-dynamic-downsample-prob", "float", 0.0),
    # ---- model structure
    ("--model-conv-layers*", "int", 5), ("--model-ave-pool-layers*", "ints", [2]),
    ("--model-residual-layer-start*", "int", 0), ("--model-init-conv-channels*", "int", 128),
    ("--model-final-conv-channels*", "int", 128), ("--model_final_layer_dilation*", "int", 1),
    ("--model_middle_layer_dilation*", "int", 1), ("--model-hidden-dropout*", "float", 0.0),
    ("--model-batchnorm*", "flag", False), ("--model-use-q-scores*", "flag", False),
    ("--model-use-strands*", "flag", False), ("--model-highway-single-reads*", "flag", False),
    ("--model-bottleneck-size*", "int", 32), ("--model_concat_hw_reads*", "flag", False),
    ("--model-use-naive-var-vector*", "flag", False), ("--model-use-reads-ref-var-mask*", "flag", False),
    ("--model-use-AF*", "flag", False), ("--model_skip_final_maxpool*", "flag", False),
    ("--model_pool_combine_dimension*", "int", 2048),
    # ---- transformer variant (rejected by the hot path, parsed for compatibility)
    ("--use_transformer*", "flag", False), ("--transformer_encoder_heads", "int", 4),
    ("--num_transformer_layers", "int", 4), ("--transformer_feedforward_dim", "int", 64),
    ("--final_transformer_dims", "int", 64), ("--transformer_residual", "flag", False),
    ("--transformer_encoder_dropout", "float", 0.1),
]

# additions of this implementation (not in the reference); all optional
_EXTRA = [
    ("--reads-seed", "int", 0, "pins the random read subset of pileups deeper than 100 reads (the reference draws "
                               "it from an unseeded RNG, dl4vc/dataset.py:274-281)"),
    ("--sites-per-launch", "int", 4096, "candidate sites per device launch (FC macro-batch)"),
    ("--shard", "str", "", "i/n: process only the i-th of n contiguous site shards (multi-GPU launch sets this)"),
    ("--precision", "str", "fp32", "conv-stack arithmetic: fp32 (exact fp32 MFMA, default), bf16x3 (split bf16, scores "
                                   "within 1e-4, ~2.3x faster) or bf16"),
    ("--compute-empty-rows", "flag", False, "compute every all-padding pileup row separately, as the reference does (default: once "
                                            "per site -- their inputs are identical, the outputs bit-identical)"),
    ("--conv-algo", "str", "auto", "fp32 conv form: auto (Winograd F(2,3) where every layer after the first has "
                                   "dilation 2), direct, or winograd"),
]

_TYPES = {"int": int, "float": float, "str": str}


def create_arg_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="DL4VC DAN variant caller -- MI355X-native inference path")
    for name, kind, default in _FLAGS:
        name = name.rstrip("*")
        if kind == "flag":
            p.add_argument(name, action="store_true", default=default)
        elif kind == "req-str":
            p.add_argument(name, type=str, required=True)
        elif kind in ("ints", "strs"):
            p.add_argument(name, type=int if kind == "ints" else str, nargs="+", default=list(default))
        else:
            p.add_argument(name, type=_TYPES[kind], default=default)
    for name, kind, default, text in _EXTRA:
        if kind == "flag":
            p.add_argument(name, action="store_true", default=default, help=text)
        else:
            p.add_argument(name, type=_TYPES[kind], default=default, help=text)
    return p
