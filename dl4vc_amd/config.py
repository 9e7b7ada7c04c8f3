"""Structural configuration of the DAN forward.

Mirrors the subset of ``Basic2DNet.__init__`` keyword arguments (reference:
dl4vc/model.py:35-53) that the published scripts exercise (call_variants.sh:101-147), and the
translation from ``main.py`` command-line flags to them (reference: main.py:99-112).  Variants that
are reachable from flags but out of scope are rejected with a clear error instead of being
silently ignored ("Unsupported-by-design").
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Sequence, Tuple

# dl4vc/model.py:25-28, dl4vc/dataset.py:398
SINGLE_READ_LENGTH = 201
NUM_SINGLE_READS = 100
MAX_READS = 100
MIN_RESIDUAL_LAYER = 2
MAX_LAYERS = 16       # capacity of the C-ABI config struct (include/dl4vc_dan.h)
L0_MAX_LAYERS = 12    # csrc/dan_kernels.h: layer 1 is summed from tables when the first segment has at most this many layers

PRECISION_F32 = 0     # v_mfma_f32_16x16x4_f32: exact fp32 FMA chains (parity path, the default)
PRECISION_BF16X3 = 1  # split-bf16: hi+lo operands, three bf16 MFMAs per product, fp32 accumulate (L <= 304)
PRECISION_BF16 = 2    # plain bf16 operands, fp32 accumulate (BASELINE config 5: 128 reads x 301 bp; L <= 304)


class UnsupportedModelOption(ValueError):
    """A Basic2DNet option outside the hot path's scope was requested."""


@dataclass(frozen=True)
class DanConfig:
    reads: int = NUM_SINGLE_READS            # num_single_reads
    length: int = SINGLE_READ_LENGTH         # single_read_len
    layers: int = 7                          # total_conv_layers          (--model-conv-layers)
    c_init: int = 128                        # init_conv_channels         (--model-init-conv-channels)
    c_final: int = 128                       # final_conv_channels        (--model-final-conv-channels)
    dil_mid: int = 2                         # middle_layer_dilation      (--model_middle_layer_dilation)
    dil_final: int = 2                       # final_layer_dilation       (--model_final_layer_dilation)
    pool_layers: Tuple[int, ...] = (2,)      # conv_1d_pool_layers        (--model-ave-pool-layers)
    residual_start: int = 5                  # residual_layer_start       (--model-residual-layer-start)
    use_bn: bool = True                      # use_batchnorm              (--model-batchnorm)
    use_q: bool = True                       # use_q_scores               (--model-use-q-scores)
    use_strand: bool = True                  # use_strands                (--model-use-strands)
    use_mask: bool = True                    # use_reads_ref_var_mask     (--model-use-reads-ref-var-mask)
    bottleneck: int = 32                     # bottleneck_channels/_linear_outputs (--model-bottleneck-size)
    fc_sizes: Tuple[int, ...] = (1024, 256)  # layer_sizes (constructor default, model.py:35)
    embed_dim: int = 20                      # embed_dim   (constructor default)
    precision: int = PRECISION_F32
    conv_algo: int = 0                       # fp32 path: 0 auto, 1 direct 3-tap GEMM, 2 Winograd F(2,3) (include/dl4vc_dan.h)
    skip_empty_rows: bool = False            # all-padding pileup rows computed once per site (bit-identical outputs)
    bf16_form: int = 0                       # precision 2: 0 = eight-wave kernel form (default), 1 = sixteen-wave form (include/dl4vc_dan.h)

    def __post_init__(self):
        object.__setattr__(self, "pool_layers", tuple(int(p) for p in self.pool_layers))
        object.__setattr__(self, "fc_sizes", tuple(int(s) for s in self.fc_sizes))
        self.validate()

    # ---- derived quantities -------------------------------------------------------------
    @property
    def in_channels(self) -> int:
        """model.py:169-178: 2*embed (+q) (+strand) (+3 mask channels)."""
        return 2 * self.embed_dim + int(self.use_q) + int(self.use_strand) + (3 if self.use_mask else 0)

    def layer_dims(self, l: int) -> Tuple[int, int, int]:
        """(c_in, c_out, dilation) of the 1-based conv layer ``l`` (model.py:211-229)."""
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
    def pooled_width(self) -> int:
        return 2 * self.c_final * self.length

    @property
    def highway_width(self) -> int:
        return self.layers * self.bottleneck * self.reads

    @property
    def feature_width(self) -> int:
        """model.py:296,327,336-338 -- 73 856 at the production shape."""
        return self.pooled_width + self.highway_width

    def macs_per_position(self) -> int:
        """Conv-stack multiply-accumulates per (read, position) -- SURVEY.md section 8d."""
        total = 0
        for l in range(1, self.layers + 1):
            cin, cout, _ = self.layer_dims(l)
            total += 3 * cin * cout
            if self.is_residual(l):
                total += cout * cout
            total += cout * self.bottleneck + self.bottleneck * self.bottleneck
        return total

    def flops_per_site(self) -> float:
        """Algorithmic FLOPs of one candidate site (2 x MAC), conv stack + FC + heads."""
        fc = 0
        sizes = (self.feature_width,) + tuple(self.fc_sizes)
        for a, b in zip(sizes[:-1], sizes[1:]):
            fc += a * b
        fc += sizes[-1] * 27
        return 2.0 * (self.reads * self.length * self.macs_per_position() + fc)

    def winograd_applies(self) -> bool:
        """The fp32 path runs the 3-tap convolutions after the first layer in Winograd F(2,3) form when all of them have
        dilation 2 (the production network) and ``conv_algo`` does not force the direct form."""
        dil_ok = (self.layers < 3 or self.dil_mid == 2) and (self.layers < 2 or self.dil_final == 2)
        return self.precision == PRECISION_F32 and dil_ok and self.conv_algo != 1 and self.layers > 1

    def executed_macs_per_position(self) -> float:
        """MFMA multiply-accumulates per (read, position) actually issued: ``macs_per_position`` with the Winograd layers'
        3-tap term replaced by 4 GEMMs per 2 outputs (2 * cin * cout).  Tile padding is not counted."""
        total = float(self.macs_per_position())
        if self.winograd_applies():
            for l in range(2, self.layers + 1):
                cin, cout, _ = self.layer_dims(l)
                total -= cin * cout
        first_segment = min(self.pool_layers) if self.pool_layers else self.layers
        if self.precision == PRECISION_F32 and first_segment <= L0_MAX_LAYERS:
            # the fp32 path computes layer 1 from tables on the vector ALUs (csrc/dan_kernels.h L0_*): its 3 * cin * cout
            # multiply-accumulates per position are algorithmic work but issue no MFMA.  (The kernel keeps the GEMM form when its
            # first segment has more than L0_MAX_LAYERS layers -- csrc/dan_kernels.hip `l0_lookup` -- and, for that one forward,
            # when a debug tap sits on the encoded input; neither occurs in a measured run.)
            cin, cout, _ = self.layer_dims(1)
            total -= 3 * cin * cout
        return total

    def input_bytes_per_site(self) -> int:
        return 3 * self.reads * self.length + 3 * self.length

    def to_dict(self):
        return asdict(self)

    # ---- validation ---------------------------------------------------------------------
    def validate(self):
        if not (1 <= self.layers <= MAX_LAYERS):
            raise UnsupportedModelOption("layers must be in 1..%d" % MAX_LAYERS)
        if self.reads < 1 or self.length < 8:
            raise UnsupportedModelOption("reads >= 1 and length >= 8 required")
        if self.residual_start > 0 and self.residual_start < MIN_RESIDUAL_LAYER:
            # model.py:209
            raise UnsupportedModelOption("Do not allow residuals starting at conv layer %d" % self.residual_start)
        for p in self.pool_layers:
            if not (1 <= p < self.layers):
                raise UnsupportedModelOption("pool layer %d must lie in 1..layers-1" % p)
        if self.layers == 1 and self.c_init != self.c_final:
            # model.py:214 vs :257,275: the only layer has init_conv_channels outputs, everything after it is sized by
            # final_conv_channels -- the reference builds such a model and fails in its first forward
            raise UnsupportedModelOption("a single conv layer needs init_conv_channels == final_conv_channels")
        if self.bottleneck < 0 or len(self.fc_sizes) != 2:
            raise UnsupportedModelOption("bottleneck >= 0 and exactly two FC layers required")
        if self.conv_algo not in (0, 1, 2):
            raise UnsupportedModelOption("conv_algo must be 0 (auto), 1 (direct) or 2 (winograd)")
        if self.bf16_form not in (0, 1) or (self.bf16_form and self.precision != PRECISION_BF16):
            raise UnsupportedModelOption("bf16_form is 0 or 1 and selects a form of the precision-2 (bf16) kernel")
        if self.embed_dim != 20:
            raise UnsupportedModelOption("embed_dim is fixed at 20 in the reference's scripts")

    # ---- main.py flag namespace -> config  (reference: main.py:99-112) --------------------
    @classmethod
    def from_args(cls, args, reads: int = NUM_SINGLE_READS, length: int = SINGLE_READ_LENGTH) -> "DanConfig":
        def flag(name, default=None):
            return getattr(args, name, default)

        if flag("early_loss_layers"):
            raise UnsupportedModelOption("--early_loss_layers is not supported (model.py:864-900)")
        if flag("use_transformer"):
            raise UnsupportedModelOption("--use_transformer is not supported (model.py:279-294)")
        if flag("model_pool_combine_dimension", 0) > 0:
            raise UnsupportedModelOption("--model_pool_combine_dimension > 0 is not supported (model.py:308-310); "
                                         "the published scripts pass 0")
        if flag("model_skip_final_maxpool"):
            raise UnsupportedModelOption("--model_skip_final_maxpool is not supported (model.py:824-835)")
        if flag("model_use_naive_var_vector") or flag("model_use_AF"):
            raise UnsupportedModelOption("naive variant encoding / AF input are deprecated in the reference "
                                         "(model.py:355-360)")
        highway = bool(flag("model_highway_single_reads", False))
        if highway and not flag("model_concat_hw_reads", False):
            raise UnsupportedModelOption("averaged highways (no --model_concat_hw_reads) are not supported "
                                         "(model.py:856-857)")
        return cls(reads=reads, length=length,
                   layers=int(flag("model_conv_layers", 5)),
                   c_init=int(flag("model_init_conv_channels", 128)),
                   c_final=int(flag("model_final_conv_channels", 128)),
                   dil_mid=int(flag("model_middle_layer_dilation", 1)),
                   dil_final=int(flag("model_final_layer_dilation", 1)),
                   pool_layers=tuple(flag("model_ave_pool_layers", [2])),
                   residual_start=int(flag("model_residual_layer_start", 0)),
                   use_bn=bool(flag("model_batchnorm", False)),
                   use_q=bool(flag("model_use_q_scores", False)),
                   use_strand=bool(flag("model_use_strands", False)),
                   use_mask=bool(flag("model_use_reads_ref_var_mask", False)),
                   bottleneck=int(flag("model_bottleneck_size", 32)) if highway else 0)


def production_config(reads: int = NUM_SINGLE_READS, length: int = SINGLE_READ_LENGTH) -> DanConfig:
    """The only published configuration (call_variants.sh:101-147)."""
    return DanConfig(reads=reads, length=length)
