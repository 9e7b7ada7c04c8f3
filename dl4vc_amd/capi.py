"""ctypes binding of ``libdl4vc_dan.so`` (C ABI: include/dl4vc_dan.h).

There is no CPU fallback: if the shared library is missing or no HIP device is usable the
constructor raises (SURVEY.md section 8b "Errors": every entry point returns an int status,
this wrapper turns non-zero into ``RuntimeError`` with ``dan_last_error()``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from .config import DanConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdl4vc_dan.so")
ABI_VERSION = 6

# every symbol include/dl4vc_dan.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = ("dan_abi_version", "dan_source_hash", "dan_create", "dan_set_tensor", "dan_finalize", "dan_destroy", "dan_last_error",
           "dan_forward", "dan_forward_aux", "dan_forward_device", "dan_forward_async", "dan_wait", "dan_set_tap", "dan_read_buffer", "dan_query",
           "dan_profile_enable", "dan_kernel_stats")


class DanCConfig(C.Structure):
    """``struct dan_config`` -- field order and types must match the header."""
    _fields_ = [("reads", C.c_int32), ("length", C.c_int32), ("layers", C.c_int32), ("c_init", C.c_int32),
                ("c_final", C.c_int32), ("dil_mid", C.c_int32), ("dil_final", C.c_int32),
                ("pool_layers_mask", C.c_uint32), ("residual_start", C.c_int32), ("use_bn", C.c_int32),
                ("use_q", C.c_int32), ("use_strand", C.c_int32), ("use_mask", C.c_int32), ("bottleneck", C.c_int32),
                ("fc_sizes", C.c_int32 * 2), ("precision", C.c_int32), ("device_id", C.c_int32),
                ("max_batch", C.c_int32), ("chunk_sites", C.c_int32), ("conv_algo", C.c_int32), ("skip_empty_rows", C.c_int32),
                ("bf16_form", C.c_int32)]


_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("DL4VC_DAN_LIB", LIB_PATH)
    if not os.path.isfile(p):
        raise RuntimeError("HIP extension %s not found: build it with `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (or `make -C dl4vc_amd/csrc`).  There is no CPU fallback." % p)
    lib = C.CDLL(p)
    u8p, f32p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.c_void_p
    lib.dan_abi_version.restype = C.c_int
    lib.dan_source_hash.restype = C.c_char_p
    lib.dan_create.argtypes = [C.POINTER(DanCConfig), C.POINTER(vp)]
    lib.dan_set_tensor.argtypes = [vp, C.c_char_p, f32p, C.POINTER(C.c_int64), C.c_int32]
    lib.dan_finalize.argtypes = [vp]
    lib.dan_destroy.argtypes = [vp]
    lib.dan_destroy.restype = None
    lib.dan_last_error.argtypes = [vp]
    lib.dan_last_error.restype = C.c_char_p
    planes = [vp] * 6
    lib.dan_forward.argtypes = [vp] + planes + [C.c_int64] + [vp] * 4
    lib.dan_forward_aux.argtypes = [vp] + planes + [C.c_int64] + [vp] * 5
    lib.dan_forward_device.argtypes = [vp] + planes + [C.c_int64] + [vp] * 5 + [vp]
    lib.dan_forward_async.argtypes = [vp] + planes + [C.c_int64] + [vp] * 5 + [C.POINTER(C.c_int64)]
    lib.dan_wait.argtypes = [vp, C.c_int64]
    lib.dan_set_tap.argtypes = [vp, C.c_int32]
    lib.dan_read_buffer.argtypes = [vp, C.c_char_p, f32p, C.c_int64]
    lib.dan_read_buffer.restype = C.c_int64
    lib.dan_query.argtypes = [vp, C.c_char_p]
    lib.dan_query.restype = C.c_int64
    lib.dan_profile_enable.argtypes = [vp, C.c_int32]
    lib.dan_kernel_stats.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    if lib.dan_abi_version() != ABI_VERSION:
        raise RuntimeError("libdl4vc_dan ABI %d != binding %d" % (lib.dan_abi_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def source_hash() -> str:
    """``dan_source_hash()`` of the loaded library: which kernel + C-ABI sources it was built from."""
    return load_library().dan_source_hash().decode()


HASHED_SOURCES = ("dan_*.hip", "dan_*.h", "dan_capi.cpp", "dan_train_capi.cpp")


def tree_source_hash(csrc_dir: Optional[str] = None) -> str:
    """The same digest computed from the sources in the tree (csrc/Makefile: sha256 over the sorted files, 16 hex digits).  Equal to
    ``source_hash()`` exactly when the library is a build of the tree as it stands."""
    import glob
    import hashlib
    d = csrc_dir or os.path.join(_HERE, "csrc")
    files = sorted({f for pat in HASHED_SOURCES for f in glob.glob(os.path.join(d, pat))}, key=os.path.basename)
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def c_config(cfg: DanConfig, device_id: int = 0, max_batch: int = 0, chunk_sites: int = 0) -> DanCConfig:
    mask = 0
    for p in cfg.pool_layers:
        mask |= 1 << p
    return DanCConfig(reads=cfg.reads, length=cfg.length, layers=cfg.layers, c_init=cfg.c_init, c_final=cfg.c_final,
                      dil_mid=cfg.dil_mid, dil_final=cfg.dil_final, pool_layers_mask=mask,
                      residual_start=cfg.residual_start, use_bn=int(cfg.use_bn), use_q=int(cfg.use_q),
                      use_strand=int(cfg.use_strand), use_mask=int(cfg.use_mask), bottleneck=cfg.bottleneck,
                      fc_sizes=(C.c_int32 * 2)(*cfg.fc_sizes), precision=cfg.precision, device_id=device_id,
                      max_batch=max_batch, chunk_sites=chunk_sites, conv_algo=cfg.conv_algo, skip_empty_rows=int(cfg.skip_empty_rows),
                      bf16_form=int(cfg.bf16_form))


def _u8(a, shape, name):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.shape != shape:
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, shape))
    return a


class DanHandle:
    """Owns one ``dan_t``.  Not thread-safe; one per process / GPU."""

    def __init__(self, cfg: DanConfig, device_id: int = 0, max_batch: int = 0, chunk_sites: int = 0):
        self._h = None
        self.lib = load_library()
        self.cfg = cfg
        h = C.c_void_p()
        cc = c_config(cfg, device_id, max_batch, chunk_sites)
        rc = self.lib.dan_create(C.byref(cc), C.byref(h))
        if rc != 0:
            raise RuntimeError("dan_create failed (%d): %s" % (rc, self.lib.dan_last_error(None).decode()))
        self._h = h

    # ---- plumbing ---------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.dan_last_error(self._h).decode()))

    def close(self):
        if self._h is not None:
            self.lib.dan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass

    # ---- lifecycle --------------------------------------------------------------------------
    def set_tensor(self, name: str, array) -> None:
        a = np.ascontiguousarray(array, dtype=np.float32)
        shape = (C.c_int64 * a.ndim)(*a.shape)
        self._check(self.lib.dan_set_tensor(self._h, name.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), shape, a.ndim),
                    "dan_set_tensor(%s)" % name)

    def finalize(self) -> None:
        self._check(self.lib.dan_finalize(self._h), "dan_finalize")

    def query(self, what: str) -> int:
        v = self.lib.dan_query(self._h, what.encode())
        if v < 0:
            self._check(int(v), "dan_query(%s)" % what)
        return int(v)

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, reads, qual, strand, ref, ref_mask, var_mask, aux: bool = False) -> Dict[str, np.ndarray]:
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        B = reads.shape[0]
        R, L = self.cfg.reads, self.cfg.length
        reads = _u8(reads, (B, R, L), "reads")
        qual = _u8(qual, (B, R, L), "qual")
        strand = _u8(strand, (B, R, L), "strand")
        ref = _u8(ref, (B, L), "ref")
        ref_mask = _u8(ref_mask, (B, L), "ref_mask")
        var_mask = _u8(var_mask, (B, L), "var_mask")
        out = {"bin_logits": np.empty((B, 2), np.float32), "vt_logits": np.empty((B, 3), np.float32),
               "vt_prob": np.empty((B, 3), np.float32), "bp": np.empty((B,), np.float32)}
        if aux:
            out["aux"] = np.empty((B, 22), np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        rc = self.lib.dan_forward_aux(self._h, p(reads), p(qual), p(strand), p(ref), p(ref_mask), p(var_mask), B,
                                      p(out["bin_logits"]), p(out["vt_logits"]), p(out["vt_prob"]), p(out["bp"]),
                                      p(out["aux"]) if aux else None)
        self._check(rc, "dan_forward")
        if aux:
            a = out.pop("aux")
            out.update(af=a[:, 0:1].copy(), cov=a[:, 1:2].copy(), vb=a[:, 2:12].copy(), vr=a[:, 12:22].copy())
        return out

    def forward_async(self, reads, qual, strand, ref, ref_mask, var_mask, aux: bool = False):
        """Enqueue one batch (<= max_batch sites) on the double-buffered asynchronous path; returns a token for
        ``wait``.  The input arrays are free again on return; at most two batches may be in flight."""
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        B = reads.shape[0]
        R, L = self.cfg.reads, self.cfg.length
        ins = [_u8(reads, (B, R, L), "reads"), _u8(qual, (B, R, L), "qual"), _u8(strand, (B, R, L), "strand"),
               _u8(ref, (B, L), "ref"), _u8(ref_mask, (B, L), "ref_mask"), _u8(var_mask, (B, L), "var_mask")]
        out = {"bin_logits": np.empty((B, 2), np.float32), "vt_logits": np.empty((B, 3), np.float32),
               "vt_prob": np.empty((B, 3), np.float32), "bp": np.empty((B,), np.float32)}
        if aux:
            out["aux"] = np.empty((B, 22), np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        t = C.c_int64(-1)
        rc = self.lib.dan_forward_async(self._h, *[p(a) for a in ins], B, p(out["bin_logits"]), p(out["vt_logits"]),
                                        p(out["vt_prob"]), p(out["bp"]), p(out["aux"]) if aux else None, C.byref(t))
        self._check(rc, "dan_forward_async")
        return (int(t.value), out)                 # `out` keeps the destination arrays alive until wait()

    def wait(self, token) -> Dict[str, np.ndarray]:
        ticket, out = token
        self._check(self.lib.dan_wait(self._h, ticket), "dan_wait")
        if "aux" in out:
            a = out.pop("aux")
            out.update(af=a[:, 0:1].copy(), cov=a[:, 1:2].copy(), vb=a[:, 2:12].copy(), vr=a[:, 12:22].copy())
        return out

    def forward_device(self, ptrs, n_sites: int, out_ptrs, stream: int = 0) -> None:
        """Raw device pointers (ints): ptrs = (reads, qual, strand, ref, ref_mask, var_mask),
        out_ptrs = (bin_logits, vt_logits, vt_prob, bp, aux) with 0 for unwanted outputs."""
        v = lambda x: C.c_void_p(int(x)) if x else None   # noqa: E731
        rc = self.lib.dan_forward_device(self._h, *[v(x) for x in ptrs], int(n_sites), *[v(x) for x in out_ptrs], v(stream))
        self._check(rc, "dan_forward_device")

    # ---- debug taps / profiling ---------------------------------------------------------------
    def set_tap(self, layer: int) -> None:
        self._check(self.lib.dan_set_tap(self._h, layer), "dan_set_tap")

    def read_buffer(self, name: str, n_floats: int) -> np.ndarray:
        dst = np.empty(int(n_floats), np.float32)
        n = self.lib.dan_read_buffer(self._h, name.encode(), dst.ctypes.data_as(C.POINTER(C.c_float)), dst.size)
        if n < 0:
            self._check(int(n), "dan_read_buffer(%s)" % name)
        return dst[:n]

    def profile(self, on: bool) -> None:
        self._check(self.lib.dan_profile_enable(self._h, int(on)), "dan_profile_enable")

    def kernel_stats(self, kernel: str):
        n, ms = C.c_int64(0), C.c_double(0.0)
        self._check(self.lib.dan_kernel_stats(self._h, kernel.encode(), C.byref(n), C.byref(ms)), "dan_kernel_stats")
        return int(n.value), float(ms.value)
