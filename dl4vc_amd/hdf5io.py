"""Reading (and, for fixtures, writing) the candidate HDF5 file without h5py.

The reference reads ``hdfile['data'][idx]`` through h5py (dl4vc/dataset.py:500-512), one gzip-chunked
compound record at a time.  h5py is not installable here, so this module offers two back-ends behind
one class: h5py when it is importable, otherwise ``libhdf5`` (1.10) through ctypes.  Records are read
in RANGES (one H5Dread per batch) in the packed on-disk layout of ``hdf5_schema.record_dtype`` -- no
type conversion, no per-record Python work.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
from typing import Optional

import numpy as np

from .hdf5_schema import DATASET_NAME, record_dtype

hid_t = C.c_int64
hsize_t = C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0x0000, 0x0001, 0x0002
H5S_SELECT_SET = 0
H5T_COMPOUND = 6
H5S_UNLIMITED = 0xFFFFFFFFFFFFFFFF

_LIB = None


def _find_libhdf5() -> Optional[str]:
    cands = [os.environ.get("DL4VC_LIBHDF5", ""), "/opt/conda/lib/libhdf5.so", ctypes.util.find_library("hdf5") or ""]
    for c in cands:
        if c and (os.path.isabs(c) and os.path.isfile(c) or not os.path.isabs(c)):
            return c
    return None


def libhdf5():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _find_libhdf5()
    if path is None:
        raise RuntimeError("neither h5py nor libhdf5 is available: cannot read candidate HDF5 files "
                           "(set DL4VC_LIBHDF5=/path/to/libhdf5.so)")
    lib = C.CDLL(path)
    lib.H5open()
    sig = {
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]), "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Fclose": (C.c_int, [hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Dclose": (C.c_int, [hid_t]),
        "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sselect_hyperslab": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Sclose": (C.c_int, [hid_t]),
        "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dset_extent": (C.c_int, [hid_t, C.POINTER(hsize_t)]),
        "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tclose": (C.c_int, [hid_t]), "H5Tget_class": (C.c_int, [hid_t]),
        "H5Tcreate": (hid_t, [C.c_int, C.c_size_t]), "H5Tinsert": (C.c_int, [hid_t, C.c_char_p, C.c_size_t, hid_t]),
        "H5Tarray_create2": (hid_t, [hid_t, C.c_uint, C.POINTER(hsize_t)]), "H5Tcopy": (hid_t, [hid_t]),
        "H5Tset_size": (C.c_int, [hid_t, C.c_size_t]),
        "H5Tget_nmembers": (C.c_int, [hid_t]), "H5Tget_member_offset": (C.c_size_t, [hid_t, C.c_uint]),
        "H5Tget_member_name": (C.c_void_p, [hid_t, C.c_uint]), "H5free_memory": (C.c_int, [C.c_void_p]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pset_chunk": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]),
        "H5Pset_deflate": (C.c_int, [hid_t, C.c_uint]), "H5Pclose": (C.c_int, [hid_t]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    lib._g = lambda sym: hid_t.in_dll(lib, sym).value       # noqa: E731
    _LIB = lib
    return lib


def _h5py():
    try:
        import h5py       # noqa: F401
        return h5py
    except Exception:     # noqa: BLE001
        return None


class CandidateFile:
    """Read-only view of the ``data`` dataset: ``len()``, ``read(lo, hi) -> structured array``."""

    def __init__(self, path: str):
        if not os.path.isfile(path):
            raise FileNotFoundError(path)
        self.path = path
        self._h5 = _h5py()
        self._dtype = None
        if self._h5 is not None:
            self._f = self._h5.File(path, "r")
            self._d = self._f[DATASET_NAME]
            self._n = len(self._d)
            return
        lib = self._lib = libhdf5()
        self._fid = lib.H5Fopen(path.encode(), H5F_ACC_RDONLY, 0)
        if self._fid < 0:
            raise OSError("cannot open %s as HDF5" % path)
        self._did = lib.H5Dopen2(self._fid, DATASET_NAME.encode(), 0)
        if self._did < 0:
            lib.H5Fclose(self._fid)
            raise KeyError("%s has no dataset '%s'" % (path, DATASET_NAME))
        self._tid = lib.H5Dget_type(self._did)
        sid = lib.H5Dget_space(self._did)
        dims = (hsize_t * 1)()
        if lib.H5Sget_simple_extent_ndims(sid) != 1:
            raise ValueError("dataset '%s' must be one-dimensional" % DATASET_NAME)
        lib.H5Sget_simple_extent_dims(sid, dims, None)
        lib.H5Sclose(sid)
        self._n = int(dims[0])
        self._itemsize = int(lib.H5Tget_size(self._tid))
        self._dtype = self._infer_dtype()

    def _infer_dtype(self) -> np.dtype:
        """Match the file's compound type against the schema (by item size and member offsets)."""
        lib = self._lib
        if lib.H5Tget_class(self._tid) != H5T_COMPOUND:
            raise ValueError("dataset '%s' is not a compound type" % DATASET_NAME)
        offs = {}
        for i in range(lib.H5Tget_nmembers(self._tid)):
            p = lib.H5Tget_member_name(self._tid, i)
            offs[C.string_at(p).decode()] = int(lib.H5Tget_member_offset(self._tid, i))
            lib.H5free_memory(p)
        # window fixed at 201 (dl4vc/dataset.py:114); the stored read count follows from the item size
        for store in (200, 100, 50, 300, 400, 1000):
            dt = record_dtype(store, 201)
            if dt.itemsize == self._itemsize and all(dt.fields[k][1] == offs.get(k, -1) for k in dt.names):
                return dt
        raise ValueError("unrecognised record layout (item size %d, members %s)" % (self._itemsize, sorted(offs)))

    def __len__(self):
        return self._n

    @property
    def dtype(self) -> np.dtype:
        return self._dtype if self._dtype is not None else self._d.dtype

    def read(self, lo: int, hi: int) -> np.ndarray:
        lo, hi = max(0, int(lo)), min(int(hi), self._n)
        n = max(0, hi - lo)
        if self._h5 is not None:
            return self._d[lo:hi]
        out = np.empty(n, dtype=self._dtype)
        if n == 0:
            return out
        lib = self._lib
        fs = lib.H5Dget_space(self._did)
        start, count = (hsize_t * 1)(lo), (hsize_t * 1)(n)
        lib.H5Sselect_hyperslab(fs, H5S_SELECT_SET, start, None, count, None)
        ms = lib.H5Screate_simple(1, count, None)
        rc = lib.H5Dread(self._did, self._tid, ms, fs, 0, out.ctypes.data_as(C.c_void_p))
        lib.H5Sclose(ms)
        lib.H5Sclose(fs)
        if rc < 0:
            raise OSError("H5Dread failed on %s[%d:%d]" % (self.path, lo, hi))
        return out

    def read_field(self, lo: int, hi: int, name: str) -> np.ndarray:
        """One member of records ``[lo, hi)`` (e.g. ``vcfrec``) without materialising the 124-KB records: the memory type
        handed to H5Dread is a compound holding only that member, libhdf5 extracts it while it inflates the chunks."""
        lo, hi = max(0, int(lo)), min(int(hi), self._n)
        n = max(0, hi - lo)
        if self._h5 is not None:
            return self._d.fields(name)[lo:hi]
        ft = self._dtype.fields[name][0]
        sub = np.dtype([(name, ft)])
        out = np.empty(n, dtype=sub)
        if n == 0:
            return out[name]
        lib = self._lib
        mt = _h5_compound_type(lib, sub)
        fs = lib.H5Dget_space(self._did)
        start, count = (hsize_t * 1)(lo), (hsize_t * 1)(n)
        lib.H5Sselect_hyperslab(fs, H5S_SELECT_SET, start, None, count, None)
        ms = lib.H5Screate_simple(1, count, None)
        rc = lib.H5Dread(self._did, mt, ms, fs, 0, out.ctypes.data_as(C.c_void_p))
        lib.H5Sclose(ms)
        lib.H5Sclose(fs)
        lib.H5Tclose(mt)
        if rc < 0:
            raise OSError("H5Dread(%s) failed on %s[%d:%d]" % (name, self.path, lo, hi))
        return out[name]

    def close(self):
        if self._h5 is not None:
            self._f.close()
            return
        if getattr(self, "_did", -1) >= 0:
            self._lib.H5Tclose(self._tid)
            self._lib.H5Dclose(self._did)
            self._lib.H5Fclose(self._fid)
            self._did = -1

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _h5_compound_type(lib, dt: np.dtype) -> int:
    """numpy record dtype -> packed HDF5 compound (the converter's layout, convert_bam_single_reads.py:694-698)."""
    tid = lib.H5Tcreate(H5T_COMPOUND, dt.itemsize)
    base = {np.dtype(np.uint8): lib._g("H5T_STD_U8LE_g"), np.dtype(np.uint16): lib._g("H5T_STD_U16LE_g"),
            np.dtype(np.int32): lib._g("H5T_STD_I32LE_g")}
    made = []
    for name in dt.names:
        ft, off = dt.fields[name][0], dt.fields[name][1]
        if ft.kind == "S":
            m = lib.H5Tcopy(lib._g("H5T_C_S1_g"))
            lib.H5Tset_size(m, ft.itemsize)
        elif ft.subdtype is not None:
            sub, shape = ft.subdtype
            dims = (hsize_t * len(shape))(*shape)
            m = lib.H5Tarray_create2(base[sub], len(shape), dims)
        else:
            m = lib.H5Tcopy(base[ft])
        made.append(m)
        if lib.H5Tinsert(tid, name.encode(), off, m) < 0:
            raise OSError("H5Tinsert(%s) failed" % name)
    for m in made:
        lib.H5Tclose(m)
    return tid


def write_candidates(path: str, records: np.ndarray, gzip: int = 4, chunk: int = 8) -> None:
    """Create ``path`` with the resizable, gzip-chunked 1-D compound dataset ``data`` the converter writes
    (convert_bam_single_reads.py:659).  Used for fixtures and the synthetic 1k-site plumbing case."""
    h5 = _h5py()
    if h5 is not None:
        with h5.File(path, "w") as f:
            f.create_dataset(DATASET_NAME, maxshape=(None,), data=records, compression="gzip")
        return
    lib = libhdf5()
    fid = lib.H5Fcreate(path.encode(), H5F_ACC_TRUNC, 0, 0)
    if fid < 0:
        raise OSError("cannot create %s" % path)
    tid = _h5_compound_type(lib, records.dtype)
    n = len(records)
    dims, maxd = (hsize_t * 1)(n), (hsize_t * 1)(H5S_UNLIMITED)
    sid = lib.H5Screate_simple(1, dims, maxd)
    pl = lib.H5Pcreate(lib._g("H5P_CLS_DATASET_CREATE_ID_g"))
    lib.H5Pset_chunk(pl, 1, (hsize_t * 1)(max(1, min(chunk, n))))
    if gzip:
        lib.H5Pset_deflate(pl, gzip)
    did = lib.H5Dcreate2(fid, DATASET_NAME.encode(), tid, sid, 0, pl, 0)
    if did < 0:
        raise OSError("H5Dcreate2 failed")
    buf = np.ascontiguousarray(records)
    rc = lib.H5Dwrite(did, tid, 0, 0, 0, buf.ctypes.data_as(C.c_void_p))
    for closer, h in ((lib.H5Dclose, did), (lib.H5Pclose, pl), (lib.H5Sclose, sid), (lib.H5Tclose, tid), (lib.H5Fclose, fid)):
        closer(h)
    if rc < 0:
        raise OSError("H5Dwrite failed")


def records_from_sites(batch, store_reads: int = 200, label: int = 2) -> np.ndarray:
    """Pack a ``synth.SiteBatch`` into converter-format records (label 2 = candidate / 'FP' as the inference
    converter writes, convert_bam_single_reads.py:573-574)."""
    B, R, L = batch.reads.shape
    recs = np.zeros(B, dtype=record_dtype(store_reads, L))
    for i in range(B):
        fields = batch.vcfrec[i].split("\t")
        recs[i]["name"] = ("%s:%s" % (fields[0], fields[1])).encode()[:16]
        recs[i]["single_reads"][:R] = batch.reads[i]
        recs[i]["q-scores"][:R] = batch.qual[i]
        recs[i]["strand"][:R] = batch.strand[i]
        recs[i]["ref_bases"] = batch.ref[i]
        recs[i]["num_reads"] = int(batch.num_reads[i])
        recs[i]["label"] = label
        recs[i]["vcfrec"] = batch.vcfrec[i].encode()[:128]
    return recs


def append_candidates(path: str, records: np.ndarray) -> int:
    """Extend the ``data`` dataset of an existing file by ``records`` (the converter's ``df.resize`` + slice assignment,
    convert_bam_single_reads.py:660-671); returns the new length."""
    h5 = _h5py()
    if h5 is not None:
        with h5.File(path, "a") as f:
            d = f[DATASET_NAME]
            n0 = d.shape[0]
            d.resize((n0 + len(records),))
            d[n0:] = records
            return n0 + len(records)
    lib = libhdf5()
    fid = lib.H5Fopen(path.encode(), H5F_ACC_RDWR, 0)
    if fid < 0:
        raise OSError("cannot open %s for appending" % path)
    did = lib.H5Dopen2(fid, DATASET_NAME.encode(), 0)
    if did < 0:
        lib.H5Fclose(fid)
        raise KeyError("%s has no dataset '%s'" % (path, DATASET_NAME))
    sid = lib.H5Dget_space(did)
    dims = (hsize_t * 1)()
    lib.H5Sget_simple_extent_dims(sid, dims, None)
    lib.H5Sclose(sid)
    n0, n = int(dims[0]), len(records)
    rc = 0
    if n:
        rc = lib.H5Dset_extent(did, (hsize_t * 1)(n0 + n))
        tid = _h5_compound_type(lib, records.dtype)
        fs = lib.H5Dget_space(did)
        count = (hsize_t * 1)(n)
        lib.H5Sselect_hyperslab(fs, H5S_SELECT_SET, (hsize_t * 1)(n0), None, count, None)
        ms = lib.H5Screate_simple(1, count, None)
        buf = np.ascontiguousarray(records)
        if rc >= 0:
            rc = lib.H5Dwrite(did, tid, ms, fs, 0, buf.ctypes.data_as(C.c_void_p))
        for closer, h in ((lib.H5Sclose, ms), (lib.H5Sclose, fs), (lib.H5Tclose, tid)):
            closer(h)
    lib.H5Dclose(did)
    lib.H5Fclose(fid)
    if rc < 0:
        raise OSError("appending %d records to %s failed" % (n, path))
    return n0 + n
