"""ctypes binding of ``libdl4vc_loader.so`` -- the native batched candidate loader (row N1).

Stands where the reference's ``DataLoader(ContextDatasetFromNumpy(...), num_workers=5)`` stands
(main.py:86-94, dl4vc/dataset.py:494-680): it yields batches of the six uint8 planes in record order.
Same results as ``dl4vc_amd.dataset.assemble_batch`` (byte for byte, including the seeded read subsets of
deep pileups), an order of magnitude faster: chunk inflate + assembly run in C++ worker threads with a
bounded prefetch ring while the GPU works on the previous batch.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterator, Optional

import numpy as np

from .synth import SiteBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DL4VC_LOADER_LIB: another build of the same library -- the CPU AddressSanitizer pass of tools/asan_pileup.sh)
LIB_PATH = os.environ.get("DL4VC_LOADER_LIB") or os.path.join(_HERE, "csrc", "libdl4vc_loader.so")
SYMBOLS = ("dl_open", "dl_num_records", "dl_num_sites", "dl_window", "dl_next", "dl_close", "dl_last_error",
           "dl_select_rows", "dl_allele_masks", "pe_open", "pe_encode", "pe_close", "pe_last_error")
_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError("%s not built (make -C dl4vc_amd/csrc)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        lib.dl_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_uint64, C.c_int32,
                                C.c_int32, C.c_int32, C.POINTER(vp)]
        lib.dl_num_records.argtypes = [vp]; lib.dl_num_records.restype = C.c_int64
        lib.dl_num_sites.argtypes = [vp]; lib.dl_num_sites.restype = C.c_int64
        lib.dl_window.argtypes = [vp]; lib.dl_window.restype = C.c_int32
        lib.dl_next.argtypes = [vp] * 10; lib.dl_next.restype = C.c_int64
        lib.dl_close.argtypes = [vp]; lib.dl_close.restype = None
        lib.dl_last_error.argtypes = [vp]; lib.dl_last_error.restype = C.c_char_p
        lib.dl_select_rows.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        lib.dl_allele_masks.argtypes = [C.c_char_p, vp, vp, vp]
        lib.pe_open.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, vp, C.POINTER(vp)]
        lib.pe_encode.argtypes = [vp, C.POINTER(C.c_char_p), vp, C.c_int64, vp, vp, vp, vp, vp, vp, C.c_int32]
        lib.pe_close.argtypes = [vp]; lib.pe_close.restype = None
        lib.pe_last_error.argtypes = [vp]; lib.pe_last_error.restype = C.c_char_p
        _lib = lib
    return _lib


def available() -> bool:
    return os.path.isfile(LIB_PATH)


class NativeLoader:
    """Iterates ``SiteBatch``es of at most ``batch_sites`` records of ``[lo, hi)`` in order."""

    def __init__(self, path: str, reads: int, batch_sites: int = 4096, lo: int = 0, hi: Optional[int] = None,
                 seed: Optional[int] = None, threads: int = 8, prefetch: int = 3):
        self.lib = load_library()
        self._h = C.c_void_p()
        self.reads, self.batch_sites = reads, batch_sites
        h5 = os.environ.get("DL4VC_LIBHDF5", "")
        rc = self.lib.dl_open(path.encode(), h5.encode(), reads, lo, -1 if hi is None else hi, batch_sites,
                              0 if seed is None else int(seed) & 0xFFFFFFFF, int(seed is not None), threads, prefetch,
                              C.byref(self._h))
        if rc != 0:
            self._h = None
            raise OSError("dl_open(%s): %s" % (path, self.lib.dl_last_error(None).decode()))
        self.window = self.lib.dl_window(self._h)

    def __len__(self):
        return int(self.lib.dl_num_sites(self._h))

    @property
    def num_records(self) -> int:
        return int(self.lib.dl_num_records(self._h))

    def __iter__(self) -> Iterator[SiteBatch]:
        B, R, L = self.batch_sites, self.reads, self.window
        while True:
            rd = np.empty((B, R, L), np.uint8); ql = np.empty_like(rd); st = np.empty_like(rd)
            rf = np.empty((B, L), np.uint8); rm = np.empty_like(rf); vm = np.empty_like(rf)
            vcf = np.zeros((B, 129), np.uint8); nr = np.empty(B, np.int32); bl = np.empty(B, np.uint8)
            p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
            n = self.lib.dl_next(self._h, p(rd), p(ql), p(st), p(rf), p(rm), p(vm), p(vcf), p(nr), p(bl))
            if n == 0:
                return
            if n < 0:
                raise ValueError("native loader: %s" % self.lib.dl_last_error(self._h).decode())
            recs = [bytes(vcf[i]).split(b"\x00", 1)[0].decode() for i in range(n)]
            batch = SiteBatch(rd[:n], ql[:n], st[:n], rf[:n], rm[:n], vm[:n], recs, nr[:n])
            batch.blacklist = bl[:n].astype(bool)
            yield batch

    def close(self):
        if self._h is not None:
            self.lib.dl_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass


class PileupOptions(C.Structure):
    """``pe_options`` of include/dl4vc_loader.h."""
    _fields_ = [(n, C.c_int32) for n in ("window_size", "max_reads", "max_insert_length", "max_insert_length_variant", "min_base_quality")]


class NativePileupEncoder:
    """ctypes face of the native pileup encoder (``pe_*``, row N4): image planes of a run of locations, in order."""

    def __init__(self, bam_path: str, fasta_path: str, window_size: int, max_reads: int, max_insert_length: int,
                 max_insert_length_variant: int, min_base_quality: int = 0, bai_path: Optional[str] = None):
        self.lib = load_library()
        self._h = C.c_void_p()
        self.window, self.max_reads = 2 * window_size + 1, max_reads
        opt = PileupOptions(window_size, max_reads, max_insert_length, max_insert_length_variant, min_base_quality)
        rc = self.lib.pe_open(bam_path.encode(), bai_path.encode() if bai_path else None, fasta_path.encode(), C.byref(opt), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise RuntimeError("pe_open failed: %s" % self.lib.pe_last_error(None).decode())

    def encode(self, contigs, positions, threads: int = 1):
        """-> (reads, qual, strand [n][max_reads][W] u8, ref [n][W] u8, num_reads [n] i32, status [n] i8)."""
        n = len(positions)
        names = (C.c_char_p * max(n, 1))(*[c.encode() for c in contigs])
        pos = np.ascontiguousarray(positions, np.int32)
        R, W = self.max_reads, self.window
        reads, qual, strand = (np.zeros((n, R, W), np.uint8) for _ in range(3))
        ref = np.zeros((n, W), np.uint8)
        num = np.zeros(n, np.int32)
        status = np.zeros(n, np.int8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        rc = self.lib.pe_encode(self._h, names, p(pos), n, p(reads), p(qual), p(strand), p(ref), p(num), p(status), int(threads))
        if rc != 0:
            raise RuntimeError("pe_encode failed: %s" % self.lib.pe_last_error(self._h).decode())
        return reads, qual, strand, ref, num, status

    def close(self):
        if self._h is not None:
            self.lib.pe_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
