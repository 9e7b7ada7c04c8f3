"""Seeded synthetic candidate sites of the reference's input shape.

There is no network for HG002 data or the published checkpoint, so benchmarks and parity tests
run on pileups drawn by this generator.  The convolution cost is data independent; the generator
is realistic only as far as needed to exercise the allele-match predicate, empty rows, read
start/end tokens, gap columns and inserts:

* ``ref`` tokens uniform in {A,T,G,C} with 1 % gap/N (token 5);
* coverage ``n ~ clip(N(0.78 R, 0.19 R), 1, R)`` non-empty rows (50 +- 12 at R = 64), the rest all-pad;
* a non-empty row covers a contiguous span of ~0.75 L columns placed uniformly, token 6 / 7 at the
  span ends, 0 outside; tokens follow ``ref`` with 1 % substitutions and 0.5 % gaps, ``noinsert``
  (8) where ``ref`` is a gap;
* one allele per site -- 80 % SNP, 10 % insert (<= 5 bases), 10 % delete (<= 5 bases) -- carried by a
  fraction AF in {0.1, 0.5, 1.0} of the rows that cover the centre;
* ``q ~ U{2..41}`` on covered columns, strand in {1, 2} per row;
* allele masks built by ``dl4vc_amd.alleles`` (row A3).

Layout is the HDF5-native one: ``reads/qual/strand[site][read][pos]`` uint8, ``ref/ref_mask/var_mask
[site][pos]`` uint8.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List

import numpy as np

from . import vocab as V
from .alleles import allele_mask_vectors, CENTER

_BASES = "ATGC"        # token t (1..4) -> character _BASES[t-1]


@dataclass
class SiteBatch:
    reads: np.ndarray       # (B,R,L) u8
    qual: np.ndarray        # (B,R,L) u8
    strand: np.ndarray      # (B,R,L) u8
    ref: np.ndarray         # (B,L) u8
    ref_mask: np.ndarray    # (B,L) u8
    var_mask: np.ndarray    # (B,L) u8
    vcfrec: List[str]
    num_reads: np.ndarray   # (B,) i32  rows that carry a read

    def __len__(self):
        return self.reads.shape[0]

    def slice(self, lo, hi) -> "SiteBatch":
        return SiteBatch(self.reads[lo:hi], self.qual[lo:hi], self.strand[lo:hi], self.ref[lo:hi],
                         self.ref_mask[lo:hi], self.var_mask[lo:hi], self.vcfrec[lo:hi], self.num_reads[lo:hi])

    def arrays(self):
        return self.reads, self.qual, self.strand, self.ref, self.ref_mask, self.var_mask


def make_sites(n_sites: int, reads: int = 64, length: int = 201, seed: int = 0,
               chrom: str = "chr20", first_pos: int = 100000) -> SiteBatch:
    if length != 201:
        return _make_sites_generic(n_sites, reads, length, seed, chrom, first_pos)
    rng = np.random.default_rng(seed)
    B, R, L = n_sites, reads, length
    rd = np.zeros((B, R, L), np.uint8)
    ql = np.zeros((B, R, L), np.uint8)
    st = np.zeros((B, R, L), np.uint8)
    rf = np.zeros((B, L), np.uint8)
    rmask = np.zeros((B, L), np.uint8)
    vmask = np.zeros((B, L), np.uint8)
    recs: List[str] = []
    nreads = np.zeros(B, np.int32)
    span = int(round(0.75 * L))
    for b in range(B):
        ref = rng.integers(1, 5, L).astype(np.uint8)
        ref[rng.random(L) < 0.01] = V.GAP
        ref[CENTER - 1:CENTER + 8] = np.where(ref[CENTER - 1:CENTER + 8] == V.GAP,
                                              rng.integers(1, 5, 9), ref[CENTER - 1:CENTER + 8])
        kind = rng.random()
        af = (0.1, 0.5, 1.0)[rng.integers(0, 3)]
        n = int(np.clip(round(rng.normal(0.78 * R, 0.19 * R)), 1, R))
        nreads[b] = n
        ref_base = int(ref[CENTER])
        if kind < 0.8:                                   # SNP
            alt = int(rng.choice([t for t in (1, 2, 3, 4) if t != ref_base]))
            ref_s, alt_s = _BASES[ref_base - 1], _BASES[alt - 1]
            ins_len = del_len = 0
        elif kind < 0.9:                                 # insert: open gap columns after the centre
            ins_len = int(rng.integers(1, 6))
            ins = rng.integers(1, 5, ins_len)
            ref[CENTER + 1:CENTER + 1 + ins_len] = V.GAP
            ref_s = _BASES[ref_base - 1]
            alt_s = ref_s + "".join(_BASES[t - 1] for t in ins)
            del_len = 0
        else:                                            # delete
            del_len = int(rng.integers(1, 6))
            ref_s = "".join(_BASES[t - 1] for t in ref[CENTER:CENTER + 1 + del_len])
            alt_s = _BASES[ref_base - 1]
            ins_len = 0
        for r in range(n):
            lo = int(rng.integers(0, L - span + 1))
            hi = lo + span                               # covered columns [lo, hi)
            row = ref[lo:hi].copy()
            sub = rng.random(span) < 0.01
            row[sub] = rng.integers(1, 5, int(sub.sum()))
            row[rng.random(span) < 0.005] = V.GAP
            row[ref[lo:hi] == V.GAP] = V.NOINSERT
            covers = lo < CENTER - 1 and hi > CENTER + 8
            carries = covers and (rng.random() < af)
            if covers:
                c = CENTER - lo
                row[c] = ref_base                          # clean centre unless the allele is carried
                if ins_len:
                    row[c + 1:c + 1 + ins_len] = V.NOINSERT
                if carries:
                    if kind < 0.8:
                        row[c] = alt
                    elif ins_len:
                        row[c + 1:c + 1 + ins_len] = ins
                    else:
                        row[c + 1:c + 1 + del_len] = V.GAP
            row[0], row[-1] = V.START, V.END
            rd[b, r, lo:hi] = row
            ql[b, r, lo:hi] = rng.integers(2, 42, span)
            st[b, r, lo:hi] = rng.integers(1, 3)
        rf[b] = ref
        rec = "\t".join((chrom, str(first_pos + 7 * b), ".", ref_s, alt_s, "50", ".",
                         "DP=%d;AF=%.4f" % (n, af), "GT:GQ", "1:50"))
        recs.append(rec)
        rmask[b], vmask[b] = allele_mask_vectors(rec, ref)
    return SiteBatch(rd, ql, st, rf, rmask, vmask, recs, nreads)


def _make_sites_generic(n_sites, reads, length, seed, chrom, first_pos) -> SiteBatch:
    """Windows other than 201 columns (stress shape 128 x 301): SNP-only alleles at column 100 (the middle column of a window
    too short to have one), masks written directly (the reference's mask builder is hard-wired to 201 columns, dataset.py:114)."""
    rng = np.random.default_rng(seed)
    B, R, L = n_sites, reads, length
    rf = rng.integers(1, 5, (B, L)).astype(np.uint8)
    rd = np.zeros((B, R, L), np.uint8)
    ql = np.zeros((B, R, L), np.uint8)
    st = np.zeros((B, R, L), np.uint8)
    rmask = np.zeros((B, L), np.uint8)
    vmask = np.zeros((B, L), np.uint8)
    recs, nreads = [], np.zeros(B, np.int32)
    span = max(2, int(round(0.75 * L)))
    centre = CENTER if L > CENTER else L // 2
    for b in range(B):
        n = int(np.clip(round(rng.normal(0.78 * R, 0.19 * R)), 1, R))
        nreads[b] = n
        ref_base = int(rf[b, centre])
        alt = int(rng.choice([t for t in (1, 2, 3, 4) if t != ref_base]))
        af = (0.1, 0.5, 1.0)[rng.integers(0, 3)]
        for r in range(n):
            lo = int(rng.integers(0, L - span + 1))
            hi = lo + span
            row = rf[b, lo:hi].copy()
            sub = rng.random(span) < 0.01
            row[sub] = rng.integers(1, 5, int(sub.sum()))
            if lo < centre < hi - 1 and lo != centre:
                row[centre - lo] = alt if rng.random() < af else ref_base
            row[0], row[-1] = V.START, V.END
            rd[b, r, lo:hi] = row
            ql[b, r, lo:hi] = rng.integers(2, 42, span)
            st[b, r, lo:hi] = rng.integers(1, 3)
        rmask[b, centre] = ref_base
        vmask[b, centre] = alt
        recs.append("\t".join((chrom, str(first_pos + 7 * b), ".", _BASES[ref_base - 1], _BASES[alt - 1],
                               "50", ".", "DP=%d;AF=%.4f" % (n, af), "GT:GQ", "1:50")))
    return SiteBatch(rd, ql, st, rf, rmask, vmask, recs, nreads)


def tile_sites(batch: SiteBatch, n_sites: int) -> SiteBatch:
    """Repeat a generated batch up to ``n_sites`` (bench sizes: generating 65 536 distinct sites in
    Python would take minutes; the forward's cost is data independent)."""
    reps = -(-n_sites // len(batch))

    def t(a):
        return np.concatenate([a] * reps, axis=0)[:n_sites]

    return SiteBatch(t(batch.reads), t(batch.qual), t(batch.strand), t(batch.ref), t(batch.ref_mask),
                     t(batch.var_mask), (batch.vcfrec * reps)[:n_sites], t(batch.num_reads))


# ------------------------------------------------------------------------------------------
# seeded weights of the reference's checkpoint shapes (there is no network for the published checkpoint)
# ------------------------------------------------------------------------------------------
def sinusoid_pe(length: int, dim: int) -> np.ndarray:
    """The registered buffer ``pe`` -- model.py:154-162."""
    pos = np.arange(0.0, length, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0.0, dim, 2, dtype=np.float32) * np.float32(-(np.log(10000.0) / dim))).astype(np.float32)
    pe = np.zeros((length, dim), dtype=np.float32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe[None]


def random_state_dict(cfg, seed: int = 0, dropout_keys: bool = True) -> Dict[str, np.ndarray]:
    """Seeded N(0, 1/fan_in) weights with randomised BN statistics, reference key names and shapes
    (SURVEY.md section 8b 'Weights contract').  Biases are N(0, 0.1).  ``cfg``: a DanConfig or any object with the same
    structural fields and helpers (layers, layer_dims, is_residual, bottleneck, length, embed_dim, feature_width, ...)."""
    spec = cfg
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}

    def w(*shape, fan_in=None):
        fan_in = fan_in or int(np.prod(shape[1:]))
        return (rng.standard_normal(shape) * np.sqrt(1.0 / fan_in)).astype(np.float32)

    def bias(n):
        return (rng.standard_normal(n) * 0.1).astype(np.float32)

    sd["embeddings.weight"] = (rng.standard_normal((V.VOCAB_SIZE, spec.embed_dim)) * 0.5).astype(np.float32)
    sd["pe"] = sinusoid_pe(spec.length, spec.embed_dim)
    for l in range(1, spec.layers + 1):
        cin, cout, _ = spec.layer_dims(l)
        # He-style gain (x2) keeps activations O(1) through the ReLU stack
        sd["conv1D_layers.%d.weight" % (l - 1)] = w(cout, cin, 1, 3) * np.float32(np.sqrt(2.0))
        sd["conv1D_layers.%d.bias" % (l - 1)] = bias(cout)
        p = "bn1D_layers.%d." % (l - 1)
        sd[p + "weight"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        sd[p + "bias"] = bias(cout)
        sd[p + "running_mean"] = rng.uniform(0.0, 0.5, cout).astype(np.float32)
        sd[p + "running_var"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        if spec.is_residual(l):
            i = l - spec.residual_start
            sd["residual_conv_layers.%d.weight" % i] = w(cout, cout, 1, 1)
            sd["residual_conv_layers.%d.bias" % i] = bias(cout)
        if spec.bottleneck > 0:
            H = spec.bottleneck
            sd["conv1D_bottleneck_layers.%d.weight" % (l - 1)] = w(H, cout, 1, 1) * np.float32(np.sqrt(2.0))
            sd["conv1D_bottleneck_layers.%d.bias" % (l - 1)] = bias(H)
            sd["conv1D_compression_layers.%d.weight" % (l - 1)] = w(H, H, 1, spec.length)
            sd["conv1D_compression_layers.%d.bias" % (l - 1)] = bias(H)
    sizes = [spec.feature_width] + list(spec.fc_sizes)
    for i in range(len(sizes) - 1):
        k = "conv2hidden.%d" % ((1 + 3 * i) if dropout_keys else 3 * i)
        sd[k + ".weight"] = w(sizes[i + 1], sizes[i]) * np.float32(np.sqrt(2.0))
        sd[k + ".bias"] = bias(sizes[i + 1])
    hid = sizes[-1]
    for name, n in (("fcHidden2BinTarget", 2), ("fcHidden2VT", 3), ("fcHidden2AF", 1),
                    ("fcHidden2Coverage", 1), ("fcHidden2VB", V.VOCAB_SIZE), ("fcHidden2VR", V.VOCAB_SIZE)):
        sd[name + ".weight"] = w(n, hid) * np.float32(2.0)
        sd[name + ".bias"] = bias(n)
    sd["bin_output_weights"] = np.full((1,), 0.1, np.float32)
    sd["vt_output_weights"] = np.full((1,), 0.1, np.float32)
    return sd


def torch_default_init(cfg, seed: int = 1, dropout_keys: bool = True) -> Dict[str, np.ndarray]:
    """A fresh model's state in the reference's key names, drawn from the distributions torch's constructors use for the
    reference's modules (nn.Conv2d / nn.Linear: kaiming_uniform(a=sqrt 5) = U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight
    and bias; nn.Embedding: N(0, 1) with the padding row zeroed; BatchNorm: weight 1, bias 0, running mean 0 / var 1;
    dl4vc/model.py:143-145,211-262,362-377,406-415).  The VALUES differ from a reference run with the same --seed (torch's
    generator and construction order are not reproduced): training from scratch starts from an equivalent, not identical,
    point.  Resume from a checkpoint (--modelload) for an identical one."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}

    def uni(shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return rng.uniform(-b, b, shape).astype(np.float32)

    emb = rng.standard_normal((V.VOCAB_SIZE, cfg.embed_dim)).astype(np.float32)
    emb[0] = 0.0
    sd["embeddings.weight"] = emb
    sd["pe"] = sinusoid_pe(cfg.length, cfg.embed_dim)
    for l in range(1, cfg.layers + 1):
        cin, cout, _ = cfg.layer_dims(l)
        sd["conv1D_layers.%d.weight" % (l - 1)] = uni((cout, cin, 1, 3), cin * 3)
        sd["conv1D_layers.%d.bias" % (l - 1)] = uni((cout,), cin * 3)
        p = "bn1D_layers.%d." % (l - 1)
        sd[p + "weight"] = np.ones(cout, np.float32)
        sd[p + "bias"] = np.zeros(cout, np.float32)
        sd[p + "running_mean"] = np.zeros(cout, np.float32)
        sd[p + "running_var"] = np.ones(cout, np.float32)
        if cfg.is_residual(l):
            i = l - cfg.residual_start
            sd["residual_conv_layers.%d.weight" % i] = uni((cout, cout, 1, 1), cout)
            sd["residual_conv_layers.%d.bias" % i] = uni((cout,), cout)
        if cfg.bottleneck > 0:
            H = cfg.bottleneck
            sd["conv1D_bottleneck_layers.%d.weight" % (l - 1)] = uni((H, cout, 1, 1), cout)
            sd["conv1D_bottleneck_layers.%d.bias" % (l - 1)] = uni((H,), cout)
            sd["conv1D_compression_layers.%d.weight" % (l - 1)] = uni((H, H, 1, cfg.length), H * cfg.length)
            sd["conv1D_compression_layers.%d.bias" % (l - 1)] = uni((H,), H * cfg.length)
    sizes = [cfg.feature_width] + list(cfg.fc_sizes)
    for i in range(len(sizes) - 1):
        k = "conv2hidden.%d" % ((1 + 3 * i) if dropout_keys else 3 * i)
        sd[k + ".weight"] = uni((sizes[i + 1], sizes[i]), sizes[i])
        sd[k + ".bias"] = uni((sizes[i + 1],), sizes[i])
    hid = sizes[-1]
    for name, n in (("fcHidden2BinTarget", 2), ("fcHidden2VT", 3), ("fcHidden2AF", 1), ("fcHidden2Coverage", 1),
                    ("fcHidden2VB", V.VOCAB_SIZE), ("fcHidden2VR", V.VOCAB_SIZE)):
        sd[name + ".weight"] = uni((n, hid), hid)
        sd[name + ".bias"] = uni((n,), hid)
    sd["bin_output_weights"] = np.full((1,), 0.1, np.float32)
    sd["vt_output_weights"] = np.full((1,), 0.1, np.float32)
    return sd


# ---- labelled candidate records for the training path (main.py --train_file): synthetic sites + truth columns ----
GT_COLUMN = ("GT:0/1", "GT:1/1", "GT:0/0", "GT:1|0", "GT:0/1", "GT:1/1", "GT:./.", "GT:0|1")
LABELS = (0, 0, 2, 1, 0, 1, 2, 0)          # {0: TP, 1: FN, 2: FP}  trainer.py:133


def make_labelled_records(n_sites: int, reads: int, seed: int):
    """Candidate records in the converter's schema with a truth column (vcfrec column 11, utils.py:59-70)."""
    sites = make_sites(n_sites, reads=reads, seed=seed)
    from .hdf5_schema import record_dtype
    recs = np.zeros(n_sites, dtype=record_dtype(200, 201))
    for i in range(n_sites):
        recs[i]["name"] = ("chr20:%d" % (1000 + 7 * i)).encode()
        recs[i]["single_reads"][:reads] = sites.reads[i]
        recs[i]["q-scores"][:reads] = sites.qual[i]
        recs[i]["strand"][:reads] = sites.strand[i]
        recs[i]["ref_bases"] = sites.ref[i]
        recs[i]["num_reads"] = int(sites.num_reads[i])
        recs[i]["label"] = LABELS[i % len(LABELS)]
        rec = sites.vcfrec[i] + "\t" + GT_COLUMN[i % len(GT_COLUMN)]
        assert len(rec) < 128
        recs[i]["vcfrec"] = rec.encode()
    return recs
