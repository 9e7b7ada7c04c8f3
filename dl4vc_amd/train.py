"""Host-side mirror of the reference's TRAINING step (BASELINE config 4).

``DanTrainer`` stands where ``model`` + ``optimizer`` stand inside one iteration of ``dl4vc/trainer.py::train``
(trainer.py:109-439): it is built from the same structural flags and hyper-parameters, takes a reference-format
``state_dict`` (main.py:121-124), and one call of ``train_step`` does what the loop body does between
``optimizer.zero_grad()`` (:201) and ``optimizer.step()`` (:439) -- train-mode forward, the loss mix, backward, gradient
clipping, Adam -- in hand-written HIP kernels behind the C ABI of ``include/dl4vc_dan_train.h``.  The host keeps what the
reference keeps on the host: batch assembly and targets (``dl4vc_amd/train_data.py``), example weights, the close-example
bookkeeping.  There is no PyTorch/CPU fallback: construction fails loudly without the extension or a GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Mapping, Sequence

import numpy as np

from .capi import load_library, c_config, DanCConfig
from .config import DanConfig
from .model import normalise_state_dict

TRAIN_SYMBOLS = ("dan_train_create", "dan_train_set_tensor", "dan_train_finalize", "dan_train_destroy", "dan_train_last_error",
                 "dan_train_backward", "dan_train_apply", "dan_train_step", "dan_train_set_lr", "dan_train_grad_buffer", "dan_train_get_tensor",
                 "dan_train_put_tensor", "dan_train_query", "dan_train_backward_begin", "dan_train_backward_end", "dan_train_wait_bucket",
                 "dan_train_grad_bucket", "dan_train_set_global_batch")

LOSS_NAMES = ("loss", "bin", "vt", "af", "cov", "vb", "vr")


@dataclass(frozen=True)
class TrainHyper:
    """``struct dan_train_hyper``: the flags of train_variant_caller.sh:101-151 that reach one step."""
    lr: float = 0.0002
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8
    grad_clip: float = 1.0
    label_smoothing: float = 0.001
    close_match_window: float = 2.0
    focal_alpha: float = 1.0
    focal_gamma: float = 0.2
    fp_train_weight: float = 0.2
    binary_weight: float = 1.0
    aux_weight: float = 1.0
    aux_bases_weight: float = 0.01
    aux_allele_weight: float = 0.001
    dropout: float = 0.1
    non_snp_train_weight: float = 2.0       # host side only: example weights (trainer.py:169-172)

    @classmethod
    def from_args(cls, args) -> "TrainHyper":
        """main.py / arguments.py flag namespace -> hyper-parameters (reference: trainer.py:82-96,425-438, main.py:116)."""
        return cls(lr=args.lr, grad_clip=args.grad_clip, label_smoothing=args.label_smoothing,
                   close_match_window=args.close_match_window, focal_alpha=args.focal_loss_alpha,
                   focal_gamma=args.focal_loss_gamma, fp_train_weight=args.fp_train_weight,
                   binary_weight=args.binary_weight, aux_weight=args.auxillary_loss_weight,
                   aux_bases_weight=args.auxillary_loss_bases_weight, aux_allele_weight=args.auxillary_loss_allele_weight,
                   dropout=args.model_hidden_dropout, non_snp_train_weight=args.non_snp_train_weight)


class _CHyper(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("lr", "beta1", "beta2", "adam_eps", "grad_clip", "label_smoothing", "close_match_window",
                                         "focal_alpha", "focal_gamma", "fp_train_weight", "binary_weight", "aux_weight",
                                         "aux_bases_weight", "aux_allele_weight", "dropout")]


class _CTargets(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("label", "var_type", "allele_freq", "coverage", "var_base_enum", "var_ref_enum", "weight")]


def _bind(lib):
    vp = C.c_void_p
    if getattr(lib, "_train_bound", False):
        return lib
    lib.dan_train_create.argtypes = [C.POINTER(DanCConfig), C.POINTER(_CHyper), C.c_int32, C.POINTER(vp)]
    lib.dan_train_set_tensor.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_int32]
    lib.dan_train_finalize.argtypes = [vp]
    lib.dan_train_destroy.argtypes = [vp]
    lib.dan_train_destroy.restype = None
    lib.dan_train_last_error.argtypes = [vp]
    lib.dan_train_last_error.restype = C.c_char_p
    planes = [vp] * 6
    lib.dan_train_backward.argtypes = [vp] + planes + [C.c_int64, C.POINTER(_CTargets), C.POINTER(vp), C.c_uint64, vp, vp]
    lib.dan_train_backward_begin.argtypes = [vp] + planes + [C.c_int64, C.POINTER(_CTargets), C.POINTER(vp), C.c_uint64]
    lib.dan_train_backward_end.argtypes = [vp, vp, vp]
    lib.dan_train_wait_bucket.argtypes = [vp, C.c_int32]
    lib.dan_train_grad_bucket.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.dan_train_set_global_batch.argtypes = [vp, C.c_float, C.c_float, C.c_float]
    lib.dan_train_apply.argtypes = [vp, C.POINTER(C.c_float)]
    lib.dan_train_step.argtypes = [vp] + planes + [C.c_int64, C.POINTER(_CTargets), C.POINTER(vp), C.c_uint64, vp, vp, C.POINTER(C.c_float)]
    lib.dan_train_set_lr.argtypes = [vp, C.c_float]
    lib.dan_train_grad_buffer.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.dan_train_grad_buffer.restype = vp
    lib.dan_train_get_tensor.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.c_int64]
    lib.dan_train_get_tensor.restype = C.c_int64
    lib.dan_train_put_tensor.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.c_int64]
    lib.dan_train_query.argtypes = [vp, C.c_char_p]
    lib.dan_train_query.restype = C.c_int64
    lib._train_bound = True
    return lib


def example_weights(is_snp, hyper: TrainHyper, trust_weight=None) -> np.ndarray:
    """trainer.py:151,169-172: (is_snp + (1 - is_snp) * non_snp_train_weight) * binary_trust_weight."""
    s = np.asarray(is_snp, np.float32)
    w = s + (1.0 - s) * np.float32(hyper.non_snp_train_weight)
    if trust_weight is not None:
        w = w * np.asarray(trust_weight, np.float32)
    return w.astype(np.float32)


BASE_CLASS_WEIGHT = np.array([0.001, 1., 1., 1., 1., 1., 0.001, 0.001, 1., 0.001])      # trainer.py:312-313


def base_class_weight_sums(targets: Mapping) -> np.ndarray:
    """[sites, sum of class weights of var_base_enum, of var_ref_enum] of this rank's share of a batch: summed over the ranks
    and divided by their number these are the arguments of ``DanTrainer.set_global_batch``."""
    vb = np.asarray(targets["var_base_enum"]).astype(np.int64).reshape(-1)
    vr = np.asarray(targets["var_ref_enum"]).astype(np.int64).reshape(-1)
    return np.array([len(vb), BASE_CLASS_WEIGHT[np.minimum(vb, 9)].sum(), BASE_CLASS_WEIGHT[np.minimum(vr, 9)].sum()], np.float64)


def average_gradients(grad, world_size: int, all_reduce) -> None:
    """Data-parallel gradient average between ``backward`` and ``apply``: ``all_reduce(grad)`` must sum ``grad`` (a flat
    fp32 tensor / array view of the device gradient buffer) over the ranks in place; the mean is what a full-batch
    ``loss.backward()`` of nn.DataParallel (main.py:117) yields for equally sized per-rank batches.  One collective of the
    whole flat buffer per step (RCCL over xGMI when the buffer is a GPU tensor)."""
    if world_size <= 1:
        return
    all_reduce(grad)
    grad *= 1.0 / world_size


class GradientExchange:
    """The data-parallel gradient average of one step, bucket by bucket (replaces nn.DataParallel's
    reduce-add, main.py:117).  ``start(bucket)`` begins averaging one window of the flat gradient buffer whose contents are
    final; ``finish()`` returns when every started bucket holds the mean.  On the GPU the collectives and the shard sums run
    on a side stream, so a bucket started in the middle of the backward pass (``DanTrainer.wait_bucket``) is exchanged
    under the rest of it.

    ``direct`` (the default on RCCL): reduce-scatter and all-gather in their DIRECT form -- all-to-all of the 1/world
    chunks (every rank sends chunk j straight to rank j: on MI355X all 7 xGMI links of a GPU carry 1/8 of the bucket at
    once, where a ring moves 2 x 7/8 of it over one link), the shard summed in rank order (the same bits on every rank),
    then an all-gather of the shards.  Otherwise (gloo rehearsals, world sizes that do not divide a tail) a plain
    all-reduce."""

    def __init__(self, dist, world_size: int, direct=None):
        self.dist, self.world = dist, int(world_size)
        self.direct = (dist is not None and self.world > 1 and dist.get_backend() == "nccl") if direct is None else bool(direct)
        self._side = None
        self._recv: Dict[tuple, object] = {}

    def start(self, bucket) -> None:
        if self.world <= 1 or bucket.numel() == 0:
            return
        if bucket.is_cuda:
            import torch
            if self._side is None:
                self._side = torch.cuda.Stream(device=bucket.device)
            with torch.cuda.stream(self._side):
                self._mean(bucket)
        else:
            self._mean(bucket)

    def _mean(self, b) -> None:
        import torch
        w, n = self.world, b.numel()
        m = n // w * w if self.direct else 0
        if m:
            key = (b.data_ptr(), m)
            if key not in self._recv:
                self._recv[key] = torch.empty(m, dtype=b.dtype, device=b.device)
            recv = self._recv[key]
            self.dist.all_to_all_single(recv, b[:m])
            shard = recv.view(w, m // w).sum(0)
            shard *= 1.0 / w
            self.dist.all_gather_into_tensor(b[:m], shard)
        if n > m:
            tail = b[m:]
            self.dist.all_reduce(tail)
            tail *= 1.0 / w

    def finish(self) -> None:
        if self._side is not None:
            self._side.synchronize()


class _DevBuf:
    """``__cuda_array_interface__`` view of a raw device pointer (so torch can wrap the gradient buffer zero-copy)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class DanTrainer:
    def __init__(self, config: DanConfig, hyper: TrainHyper = TrainHyper(), max_batch: int = 64, device_id: int = 0):
        self._h = None
        self.lib = _bind(load_library())
        self.config, self.hyper, self.max_batch = config, hyper, max_batch
        cc = c_config(config, device_id)
        ch = _CHyper(**{n: getattr(hyper, n) for n, _ in _CHyper._fields_})
        h = C.c_void_p()
        rc = self.lib.dan_train_create(C.byref(cc), C.byref(ch), max_batch, C.byref(h))
        if rc != 0:
            raise RuntimeError("dan_train_create failed (%d): %s" % (rc, self.lib.dan_train_last_error(None).decode()))
        self._h = h
        self._loaded = False
        # state-dict indices of the two FC Linear layers: 1 / 4 when the model carries dropout modules, 0 / 3 when it does not
        # (model.py:369-377); a loaded checkpoint's own indices win
        self._fc_keys = ("conv2hidden.1", "conv2hidden.4") if hyper.dropout > 0 else ("conv2hidden.0", "conv2hidden.3")
        self._extra: Dict[str, np.ndarray] = {}

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.dan_train_last_error(self._h).decode()))

    def close(self):
        if self._h is not None:
            self.lib.dan_train_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass

    # ---- lifecycle ----------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Mapping[str, object]) -> "DanTrainer":
        """Reference checkpoint keys (``module.`` prefix accepted; ``conv2hidden.{1,4}`` with dropout, ``{0,3}`` without,
        model.py:369-377) -> initial parameters + BN running statistics."""
        if self._loaded:
            raise RuntimeError("state already loaded; create a new DanTrainer")
        fc = sorted({int(k.split("conv2hidden.")[1].split(".")[0]) for k in state_dict if "conv2hidden." in k and k.endswith("weight")})
        if len(fc) == 2:
            self._fc_keys = tuple("conv2hidden.%d" % i for i in fc)
        sd = normalise_state_dict(state_dict)
        for k in ("bin_output_weights", "vt_output_weights"):   # early-loss mixing scalars: no loss term reaches them
            if k in sd:
                self._extra[k] = sd.pop(k)
        for k, v in sd.items():
            a = np.ascontiguousarray(v, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(self.lib.dan_train_set_tensor(self._h, k.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), shape, a.ndim),
                        "dan_train_set_tensor(%s)" % k)
        self._shapes = {k: tuple(np.shape(v)) for k, v in sd.items()}
        self._check(self.lib.dan_train_finalize(self._h), "dan_train_finalize")
        self._loaded = True
        return self

    # ---- one step -----------------------------------------------------------------------------------------------
    def _marshal(self, planes, targets, dropout_masks):
        reads = np.ascontiguousarray(planes[0], np.uint8)
        B = reads.shape[0]
        R, L = self.config.reads, self.config.length
        shapes = [(B, R, L)] * 3 + [(B, L)] * 3
        ins = []
        for a, shp, nm in zip(planes, shapes, ("reads", "qual", "strand", "ref", "ref_mask", "var_mask")):
            a = np.ascontiguousarray(a, np.uint8)
            if a.shape != shp:
                raise ValueError("%s has shape %s, expected %s" % (nm, a.shape, shp))
            ins.append(a)
        u8 = lambda k: np.ascontiguousarray(np.asarray(targets[k]).reshape(-1), np.uint8)        # noqa: E731
        f32 = lambda k: np.ascontiguousarray(np.asarray(targets[k]).reshape(-1), np.float32)     # noqa: E731
        tg = {"label": u8("label"), "var_type": u8("var_type"), "allele_freq": f32("allele_freq"), "coverage": f32("coverage"),
              "var_base_enum": u8("var_base_enum"), "var_ref_enum": u8("var_ref_enum"), "weight": f32("weight")}
        for k, v in tg.items():
            if v.shape != (B,):
                raise ValueError("target %s has shape %s, expected (%d,)" % (k, v.shape, B))
        ct = _CTargets(**{k: v.ctypes.data for k, v in tg.items()})
        mp = None
        keep = []
        if dropout_masks is not None and self.hyper.dropout > 0:
            widths = (self.query("feature_width"), self.config.fc_sizes[0], self.config.fc_sizes[1])
            arr = (C.c_void_p * 3)()
            for i, (m, w) in enumerate(zip(dropout_masks, widths)):
                m = np.ascontiguousarray(np.asarray(m).reshape(B, -1), np.uint8)
                if m.shape != (B, w):
                    raise ValueError("dropout mask %d has shape %s, expected (%d, %d)" % (i, m.shape, B, w))
                keep.append(m)
                arr[i] = m.ctypes.data
            mp = arr
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        return B, [p(a) for a in ins] + [B, C.byref(ct), mp], (ins, tg, ct, keep)

    @staticmethod
    def _outputs(losses, close, norm=None):
        out = {k: float(v) for k, v in zip(LOSS_NAMES, losses)}
        out["bin_close"] = close[:, 0].astype(bool)
        out["vt_close"] = close[:, 1].astype(bool)
        if norm is not None:
            out["grad_norm"] = float(norm.value)
        return out

    def _call(self, fn, planes, targets, dropout_masks, seed, with_norm):
        B, head, _alive = self._marshal(planes, targets, dropout_masks)
        losses = np.zeros(7, np.float32)
        close = np.zeros((B, 2), np.uint8)
        args = [self._h] + head + [C.c_uint64(seed), losses.ctypes.data_as(C.c_void_p), close.ctypes.data_as(C.c_void_p)]
        norm = C.c_float(0.0) if with_norm else None
        if with_norm:
            args.append(C.byref(norm))
        self._check(fn(*args), fn.__name__)
        return self._outputs(losses, close, norm)

    # ---- the same step in two halves (``GradientExchange`` runs between them) --------------------------------------
    def backward_begin(self, planes: Sequence, targets: Mapping, dropout_masks=None, seed: int = 0) -> None:
        """Stage the inputs and enqueue forward + backward; returns while the device works."""
        B, head, _alive = self._marshal(planes, targets, dropout_masks)      # (the inputs are copied before the call returns)
        self._check(self.lib.dan_train_backward_begin(*([self._h] + head + [C.c_uint64(seed)])), "dan_train_backward_begin")
        self._pending_B = B

    def wait_bucket(self, bucket: int) -> None:
        """Block until the gradients of ``bucket`` are final (0: FC stack + heads, produced first; 1: everything else)."""
        self._check(self.lib.dan_train_wait_bucket(self._h, int(bucket)), "dan_train_wait_bucket")

    def backward_end(self) -> Dict[str, object]:
        losses = np.zeros(7, np.float32)
        close = np.zeros((self._pending_B, 2), np.uint8)
        self._check(self.lib.dan_train_backward_end(self._h, losses.ctypes.data_as(C.c_void_p), close.ctypes.data_as(C.c_void_p)),
                    "dan_train_backward_end")
        return self._outputs(losses, close)

    def set_global_batch(self, sites_per_rank: float, vb_weight_per_rank: float, vr_weight_per_rank: float) -> None:
        """Full-batch normalisers / ranks for the next backward (``base_class_weight_sums`` gives the local sums to all-reduce)."""
        self._check(self.lib.dan_train_set_global_batch(self._h, float(sites_per_rank), float(vb_weight_per_rank), float(vr_weight_per_rank)),
                    "dan_train_set_global_batch")

    def grad_buckets(self):
        """[(offset, count)] of bucket 0 and bucket 1 in the flat gradient buffer (``grad_tensor``): together they tile it."""
        out = []
        for b in (0, 1):
            off, n = C.c_int64(0), C.c_int64(0)
            self._check(self.lib.dan_train_grad_bucket(self._h, b, C.byref(off), C.byref(n)), "dan_train_grad_bucket")
            out.append((int(off.value), int(n.value)))
        return out

    def backward(self, planes: Sequence, targets: Mapping, dropout_masks=None, seed: int = 0) -> Dict[str, object]:
        """Train-mode forward + losses + backward; gradients stay on the device (``grad``, ``grad_tensor``)."""
        return self._call(self.lib.dan_train_backward, planes, targets, dropout_masks, seed, False)

    def apply(self) -> float:
        """clip_grad_norm_ + Adam (trainer.py:437-439); returns the gradient norm before clipping."""
        norm = C.c_float(0.0)
        self._check(self.lib.dan_train_apply(self._h, C.byref(norm)), "dan_train_apply")
        return float(norm.value)

    def train_step(self, planes: Sequence, targets: Mapping, dropout_masks=None, seed: int = 0) -> Dict[str, object]:
        return self._call(self.lib.dan_train_step, planes, targets, dropout_masks, seed, True)

    def set_lr(self, lr: float) -> None:
        """``optimizer.param_groups[0]['lr'] *= args.lr_decay`` (main.py:166)."""
        import dataclasses
        self._check(self.lib.dan_train_set_lr(self._h, float(lr)), "dan_train_set_lr")
        self.hyper = dataclasses.replace(self.hyper, lr=float(lr))

    # ---- state --------------------------------------------------------------------------------------------------
    def query(self, what: str) -> int:
        v = self.lib.dan_train_query(self._h, what.encode())
        if v < 0:
            self._check(int(v), "dan_train_query(%s)" % what)
        return int(v)

    def tensor(self, name: str, shape=None) -> np.ndarray:
        """A parameter / running statistic (``name``), its gradient (``grad:name``) or Adam moment (``m:``/``v:``)."""
        base = name.split(":", 1)[1] if name.split(":", 1)[0] in ("grad", "m", "v") else name
        shape = shape or self._shapes.get(base)
        if shape is None:
            raise KeyError(name)
        out = np.empty(int(np.prod(shape)), np.float32)
        n = self.lib.dan_train_get_tensor(self._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), out.size)
        if n < 0:
            self._check(int(n), "dan_train_get_tensor(%s)" % name)
        return out[:n].reshape(shape)

    def debug_buffer(self, name: str, n_floats: int) -> np.ndarray:
        out = np.empty(int(n_floats), np.float32)
        n = self.lib.dan_train_get_tensor(self._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), out.size)
        if n < 0:
            self._check(int(n), "dan_train_get_tensor(%s)" % name)
        return out[:n]

    def put_tensor(self, name: str, array) -> None:
        a = np.ascontiguousarray(array, np.float32).reshape(-1)
        self._check(self.lib.dan_train_put_tensor(self._h, name.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size),
                    "dan_train_put_tensor(%s)" % name)

    def state_dict(self, prefix: str = "") -> Dict[str, np.ndarray]:
        """``model.state_dict()`` in the reference's key names (main.py:194-199; ``prefix='module.'`` for the DataParallel
        form the reference saves)."""
        out = {}
        for k in self._shapes:
            v = self.tensor(k)
            if k.startswith("fc."):
                i, part = int(k.split(".")[1]), k.split(".")[2]
                k = "%s.%s" % (self._fc_keys[i], part)
            out[prefix + k] = v
        for k, v in self._extra.items():
            out[prefix + k] = v
        return out

    def grad_tensor(self):
        """The flat device gradient buffer as a torch CUDA tensor (zero-copy) for the data-parallel all-reduce."""
        import torch
        n = C.c_int64(0)
        ptr = self.lib.dan_train_grad_buffer(self._h, C.byref(n))
        if not ptr:
            raise RuntimeError("no gradient buffer (finalize first)")
        return torch.as_tensor(_DevBuf(int(ptr), int(n.value)), device="cuda")
