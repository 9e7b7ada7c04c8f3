"""On-disk record schema of the candidate HDF5 file.

Reference: tools/convert_bam_single_reads.py:694-698 (dtype), :659 (dataset "data", 1-D, resizable,
gzip).  The compound type is PACKED (numpy default, no alignment): 123 965 bytes per record at
``--max-reads 200`` and a 201-column window.  ``num_reads`` and ``label`` are scalars on disk
(numpy-1.15 semantics of a ``(name, type, 1)`` field); files written by newer numpy would hold them
as ``(1,)`` sub-arrays with the same bytes, so readers go by offset.
"""
from __future__ import annotations

import numpy as np

DATASET_NAME = "data"
STORE_MAX_READS = 200          # dl4vc/dataset.py:399, call_variants.sh:92 (--max-reads 200)
WINDOW = 201
VCFREC_BYTES = 128             # long indel records are silently truncated by the converter (S128)
NAME_BYTES = 16


def record_dtype(store_reads: int = STORE_MAX_READS, window: int = WINDOW) -> np.dtype:
    fields = [
        ("name", "S%d" % NAME_BYTES),
        ("ref", np.uint8, (5, window)),
        ("reads", np.uint16, (5, window)),
        ("single_reads", np.uint8, (store_reads, window)),
        ("ref_bases", np.uint8, (window,)),
        ("num_reads", np.int32),
        ("label", np.uint8),
        ("vcfrec", "S%d" % VCFREC_BYTES),
        ("q-scores", np.uint8, (store_reads, window)),
        ("strand", np.uint8, (store_reads, window)),
    ]
    return np.dtype(fields)          # packed: offsets are cumulative sizes


def field_offsets(store_reads: int = STORE_MAX_READS, window: int = WINDOW):
    dt = record_dtype(store_reads, window)
    return {name: dt.fields[name][1] for name in dt.names}, dt.itemsize


assert record_dtype().itemsize == 123965
assert field_offsets()[0]["single_reads"] == 3031 and field_offsets()[0]["strand"] == 83765
