"""In-process tail of the pipeline: what call_variants.sh:162-168 of the reference does with
external tools after ``format_vcf`` --

    bcftools norm -m +any  IN > JOIN ; sed -i 's/0\\/2/1\\/2/' JOIN ; sed -i 's/2\\/2/1\\/2/' JOIN ; bgzip -c JOIN > OUT.gz ; tabix -p vcf OUT.gz

PARITY STATUS: **unpinned**.  bcftools / bgzip / tabix (htslib) are third-party tools the reference shells out to; none of
them is present in the build image, so nothing here could be compared with their output.  The restatement follows their
published behaviour for exactly the records this pipeline produces (one sample, FORMAT ``GT:GQ``, INFO ``DP`` Number=1 and
``AF`` Number=A -- tools/candidate_generator.py:198-216 of the reference), and the two ``sed`` lines of the reference script
are themselves evidence for the one non-obvious rule (two heterozygous records joined by bcftools come out as ``0/2``):

  * join (``bcftools norm -m +any``, vcfnorm.c ``merge_biallelics_to_multiallelic``): consecutive records with the same
    CHROM and POS become one record; REF = the longest REF, every ALT is extended by the REF suffix it lacks, duplicate
    alleles share an index; ID = the distinct non-``.`` IDs joined by ``;``; QUAL = the maximum; FILTER = union (``PASS``/``.``
    dropped when something else is present); INFO/FORMAT fields declared ``Number=A`` are concatenated per ALT (``.`` for
    an allele a record does not carry), ``Number=R`` likewise with the REF value first, all other fields keep the FIRST
    record's value; GT: start from the first record, then for every later record each non-reference allele replaces
    the value at the same ploidy slot (remapped to its new index), reference alleles and ``.`` leave the slot alone.
    Unlike bcftools no ``##bcftools_norm*`` header lines are added.
  * the ``sed`` rewrites are literal: FIRST occurrence per line of ``0/2`` -> ``1/2``, then first ``2/2`` -> ``1/2``, on every
    line (header lines included).
  * BGZF (``bgzip -c``): gzip members with the ``BC`` extra field, at most 0xff00 input bytes each, plus the 28-byte EOF
    block (SAM spec 4.1).  Valid BGZF; not byte-identical to htslib's file (deflate engine / level may differ).
  * tabix (``tabix -p vcf``): .tbi per the tabix spec -- UCSC binning (min shift 14, depth 5), 16-kb linear index, chunk
    begin/end as BGZF virtual offsets, adjacent chunks of a bin merged when they share a BGZF block; the optional
    statistics pseudo-bin (37450) is written like htslib does.  ``tabix_query`` reads the index back (used by the tests).
"""
from __future__ import annotations

import gzip
import io
import re
import struct
import zlib
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

# ------------------------------------------------------------------------------------------------------
# header knowledge: Number= of INFO / FORMAT ids
# ------------------------------------------------------------------------------------------------------
_DECL = re.compile(r"##(INFO|FORMAT)=<ID=([^,>]+),Number=([^,>]+)")


def field_numbers(header_lines: Iterable[str]) -> Dict[Tuple[str, str], str]:
    out = {}
    for line in header_lines:
        m = _DECL.match(line)
        if m:
            out[(m.group(1), m.group(2))] = m.group(3)
    return out


# ------------------------------------------------------------------------------------------------------
# bcftools norm -m +any
# ------------------------------------------------------------------------------------------------------
def _merge_per_allele(values: List[Optional[List[str]]], maps: List[List[int]], n_alt: int, with_ref: bool) -> str:
    """values[i] = the comma-split field of record i (None if absent); maps[i][a] = new index of record i's allele a."""
    out = ["."] * (n_alt + (1 if with_ref else 0))
    for vals, amap in zip(values, maps):
        if vals is None:
            continue
        for a, v in enumerate(vals):
            old = a if with_ref else a + 1            # allele index in the record this value belongs to
            if old >= len(amap):
                continue
            new = amap[old]
            slot = new if with_ref else new - 1
            if out[slot] == "." and v != ".":
                out[slot] = v
    return ",".join(out)


def _merge_gt(gts: List[Optional[str]], maps: List[List[int]]) -> Optional[str]:
    alleles: Optional[List[str]] = None
    seps: List[str] = []
    for g, amap in zip(gts, maps):
        if g is None:
            continue
        parts = re.split(r"[/|]", g)
        if alleles is None:                                   # the first record's genotype, re-indexed
            seps = re.findall(r"[/|]", g)
            alleles = [a if a == "." else str(amap[int(a)]) for a in parts]
            continue
        for k, a in enumerate(parts):                         # later records: non-reference alleles overwrite their slot
            if k < len(alleles) and a != "." and int(a) != 0:
                alleles[k] = str(amap[int(a)])
    if alleles is None:
        return None
    out = alleles[0]
    for k in range(1, len(alleles)):
        out += (seps[k - 1] if k - 1 < len(seps) else "/") + alleles[k]
    return out


def _join_group(recs: List[List[str]], numbers: Dict[Tuple[str, str], str]) -> List[str]:
    if len(recs) == 1:
        return recs[0]
    ref = max((r[3] for r in recs), key=len)
    alts: List[str] = []
    maps: List[List[int]] = []
    for r in recs:
        assert ref.startswith(r[3]), "records at one position must share a REF prefix: %s vs %s" % (r[3], ref)
        suffix = ref[len(r[3]):]
        amap = [0]
        for alt in r[4].split(","):
            ext = alt if alt.startswith("<") or alt == "*" or alt == "." else alt + suffix
            if ext not in alts:
                alts.append(ext)
            amap.append(alts.index(ext) + 1)
        maps.append(amap)
    ids: List[str] = []
    for r in recs:
        if r[2] != "." and r[2] not in ids:
            ids.append(r[2])
    quals = [float(r[5]) for r in recs if r[5] != "."]
    qual = "."
    if quals:
        best = max(quals)
        qual = next(r[5] for r in recs if r[5] != "." and float(r[5]) == best)
    filters: List[str] = []
    for r in recs:
        for f in r[6].split(";"):
            if f not in filters:
                filters.append(f)
    real = [f for f in filters if f not in (".", "PASS")]
    filt = ";".join(real) if real else ("PASS" if "PASS" in filters else ".")
    # INFO
    keys: List[str] = []
    parsed = []
    for r in recs:
        d = {}
        if r[7] != ".":
            for item in r[7].split(";"):
                k, _, v = item.partition("=")
                d[k] = v if _ else None
                if k not in keys:
                    keys.append(k)
        parsed.append(d)
    info_items = []
    for k in keys:
        num = numbers.get(("INFO", k), "1")
        if num in ("A", "R"):
            vals = [d[k].split(",") if d.get(k) is not None else None for d in parsed]
            info_items.append("%s=%s" % (k, _merge_per_allele(vals, maps, len(alts), num == "R")))
        else:
            first = next(d for d in parsed if k in d)
            info_items.append(k if first[k] is None else "%s=%s" % (k, first[k]))
    out = [recs[0][0], recs[0][1], ";".join(ids) if ids else ".", ref, ",".join(alts), qual, filt,
           ";".join(info_items) if info_items else "."]
    # FORMAT + samples
    if len(recs[0]) > 8:
        fkeys: List[str] = []
        for r in recs:
            for k in r[8].split(":"):
                if k not in fkeys:
                    fkeys.append(k)
        n_samples = len(recs[0]) - 9
        samples = []
        for s in range(n_samples):
            per_rec = []
            for r in recs:
                ks = r[8].split(":")
                vs = r[9 + s].split(":")
                per_rec.append({k: (vs[i] if i < len(vs) else ".") for i, k in enumerate(ks)})
            vals_out = []
            for k in fkeys:
                if k == "GT":
                    vals_out.append(_merge_gt([d.get("GT") for d in per_rec], maps) or ".")
                    continue
                num = numbers.get(("FORMAT", k), "1")
                if num in ("A", "R"):
                    vals = [d[k].split(",") if k in d else None for d in per_rec]
                    vals_out.append(_merge_per_allele(vals, maps, len(alts), num == "R"))
                else:
                    vals_out.append(next((d[k] for d in per_rec if k in d), "."))
            samples.append(":".join(vals_out))
        out += [":".join(fkeys)] + samples
    return out


def join_multiallelic_lines(lines: Sequence[str]) -> List[str]:
    """Position-sorted VCF lines (with newline) -> lines with same-position records joined (see module docstring)."""
    header = [l for l in lines if l.startswith("#")]
    numbers = field_numbers(header)
    out: List[str] = []
    group: List[List[str]] = []

    def flush():
        if group:
            out.append("\t".join(_join_group(group, numbers)) + "\n")
            group.clear()

    for line in lines:
        if line.startswith("#"):
            flush()
            out.append(line)
            continue
        cols = line.rstrip("\n").split("\t")
        if group and (group[0][0] != cols[0] or group[0][1] != cols[1]):
            flush()
        group.append(cols)
    flush()
    return out


def genotype_rewrites(lines: Sequence[str]) -> List[str]:
    """``sed 's/0\\/2/1\\/2/'`` then ``sed 's/2\\/2/1\\/2/'``: first occurrence per line (call_variants.sh:163-164)."""
    return [l.replace("0/2", "1/2", 1).replace("2/2", "1/2", 1) for l in lines]


# ------------------------------------------------------------------------------------------------------
# BGZF
# ------------------------------------------------------------------------------------------------------
BGZF_BLOCK = 0xff00
BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(data: bytes, level: int) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = c.compress(data) + c.flush()
    bsize = len(body) + 25                    # whole block length - 1
    assert bsize < 65536
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + body +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


class BgzfWriter:
    """Writes BGZF and reports the virtual offset (block start << 16 | offset inside the block) of whatever is written next."""

    def __init__(self, fileobj, level: int = 6):
        self.f, self.level = fileobj, level
        self.buf = bytearray()
        self.block_start = 0

    def tell(self) -> int:
        return (self.block_start << 16) | len(self.buf)

    def write(self, data: bytes) -> None:
        self.buf += data
        while len(self.buf) >= BGZF_BLOCK:
            self._flush(BGZF_BLOCK)

    def _flush(self, n: int) -> None:
        blk = _bgzf_block(bytes(self.buf[:n]), self.level)
        self.f.write(blk)
        self.block_start += len(blk)
        del self.buf[:n]

    def close(self) -> None:
        if self.buf:
            self._flush(len(self.buf))
        self.f.write(BGZF_EOF)


def bgzf_compress(data: bytes, level: int = 6) -> bytes:
    out = io.BytesIO()
    w = BgzfWriter(out, level)
    w.write(data)
    w.close()
    return out.getvalue()


def bgzf_blocks(raw: bytes) -> Iterator[Tuple[int, bytes]]:
    """(file offset of the block, its inflated bytes) for every BGZF block."""
    off = 0
    while off < len(raw):
        assert raw[off:off + 4] == b"\x1f\x8b\x08\x04", "not a BGZF block at %d" % off
        xlen = struct.unpack_from("<H", raw, off + 10)[0]
        extra = raw[off + 12:off + 12 + xlen]
        bsize = None
        i = 0
        while i < len(extra):
            si, slen = extra[i:i + 2], struct.unpack_from("<H", extra, i + 2)[0]
            if si == b"BC":
                bsize = struct.unpack_from("<H", extra, i + 4)[0]
            i += 4 + slen
        assert bsize is not None
        body = raw[off + 12 + xlen:off + bsize + 1 - 8]
        yield off, zlib.decompress(body, -15)
        off += bsize + 1


# ------------------------------------------------------------------------------------------------------
# tabix (.tbi) for VCF
# ------------------------------------------------------------------------------------------------------
def reg2bin(beg: int, end: int) -> int:
    """UCSC binning, 0-based half-open [beg, end) (tabix spec section 'C source code for computing bin number')."""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def reg2bins(beg: int, end: int) -> List[int]:
    end -= 1
    bins = [0]
    for shift, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        bins.extend(range(base + (beg >> shift), base + (end >> shift) + 1))
    return bins


_END = re.compile(r"(?:^|;)END=(\d+)")


def _vcf_interval(cols: List[str]) -> Tuple[int, int]:
    beg = int(cols[1]) - 1
    end = beg + len(cols[3])
    m = _END.search(cols[7]) if len(cols) > 7 else None
    if m:
        end = max(end, int(m.group(1)))
    return beg, max(end, beg + 1)


def write_vcf_gz_with_index(lines: Sequence[str], gz_path: str, level: int = 6) -> None:
    """``bgzip -c`` + ``tabix -p vcf``: writes gz_path and gz_path + '.tbi'.  Records must be sorted by CHROM block, POS."""
    names: List[str] = []
    bins: List[Dict[int, List[List[int]]]] = []
    linear: List[List[int]] = []
    stats: List[List[int]] = []                                # per ref: [first voff, last voff, n records]
    with open(gz_path, "wb") as f:
        w = BgzfWriter(f, level)
        for line in lines:
            data = line.encode()
            if line.startswith("#"):
                w.write(data)
                continue
            cols = line.rstrip("\n").split("\t")
            if not names or names[-1] != cols[0]:
                assert cols[0] not in names, "records of %s are not contiguous" % cols[0]
                names.append(cols[0]); bins.append({}); linear.append([]); stats.append([0, 0, 0])
            tid = len(names) - 1
            beg, end = _vcf_interval(cols)
            v0 = w.tell()
            w.write(data)
            v1 = w.tell()
            chunks = bins[tid].setdefault(reg2bin(beg, end), [])
            if chunks and (chunks[-1][1] >> 16) == (v0 >> 16):
                chunks[-1][1] = v1                              # same BGZF block: extend the previous chunk
            else:
                chunks.append([v0, v1])
            lin = linear[tid]
            for win in range(beg >> 14, ((end - 1) >> 14) + 1):
                while len(lin) <= win:
                    lin.append(0)
                if lin[win] == 0:
                    lin[win] = v0
            st = stats[tid]
            if st[2] == 0:
                st[0] = v0
            st[1] = v1
            st[2] += 1
        w.close()
    for lin in linear:                                          # empty windows point at the next record (htslib fills backwards)
        nxt = 0
        for i in range(len(lin) - 1, -1, -1):
            if lin[i] == 0:
                lin[i] = nxt
            else:
                nxt = lin[i]
    blob = bytearray()
    name_blob = b"".join(n.encode() + b"\0" for n in names)
    blob += b"TBI\1" + struct.pack("<8i", len(names), 2, 1, 2, 0, ord("#"), 0, len(name_blob)) + name_blob
    for tid in range(len(names)):
        b = bins[tid]
        blob += struct.pack("<i", len(b) + 1)
        for bin_id in sorted(b):
            blob += struct.pack("<Ii", bin_id, len(b[bin_id]))
            for c0, c1 in b[bin_id]:
                blob += struct.pack("<QQ", c0, c1)
        blob += struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", stats[tid][0], stats[tid][1], stats[tid][2], 0)
        blob += struct.pack("<i", len(linear[tid])) + b"".join(struct.pack("<Q", v) for v in linear[tid])
    blob += struct.pack("<Q", 0)                                # n_no_coor
    with open(gz_path + ".tbi", "wb") as f:
        f.write(bgzf_compress(bytes(blob), level))


def read_tbi(path: str) -> dict:
    raw = gzip.open(path, "rb").read()
    assert raw[:4] == b"TBI\1"
    n_ref, fmt, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from("<8i", raw, 4)
    off = 36
    names = raw[off:off + l_nm].split(b"\0")[:-1]
    off += l_nm
    refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", raw, off)[0]; off += 4
        b = {}
        for _ in range(n_bin):
            bin_id, n_chunk = struct.unpack_from("<Ii", raw, off); off += 8
            b[bin_id] = [struct.unpack_from("<QQ", raw, off + 16 * i) for i in range(n_chunk)]
            off += 16 * n_chunk
        n_intv = struct.unpack_from("<i", raw, off)[0]; off += 4
        lin = list(struct.unpack_from("<%dQ" % n_intv, raw, off)); off += 8 * n_intv
        refs.append({"bins": b, "linear": lin})
    return {"names": [n.decode() for n in names], "format": fmt, "cols": (col_seq, col_beg, col_end), "meta": chr(meta),
            "skip": skip, "refs": refs}


def tabix_query(gz_path: str, chrom: str, beg1: int, end1: int) -> List[str]:
    """Records of ``chrom`` overlapping the 1-based inclusive region [beg1, end1], found THROUGH the index."""
    idx = read_tbi(gz_path + ".tbi")
    if chrom not in idx["names"]:
        return []
    ref = idx["refs"][idx["names"].index(chrom)]
    beg, end = beg1 - 1, end1
    lin = ref["linear"]
    min_off = lin[min(beg >> 14, len(lin) - 1)] if lin else 0
    chunks = sorted(c for b in reg2bins(beg, end) if b in ref["bins"] and b != 37450 for c in ref["bins"][b] if c[1] > min_off)
    with open(gz_path, "rb") as f:
        raw = f.read()
    ustart, text = {}, bytearray()
    for off, data in bgzf_blocks(raw):                         # virtual offset -> position in the inflated stream
        ustart[off] = len(text)
        text += data
    upos = lambda v: ustart[v >> 16] + (v & 0xffff)            # noqa: E731
    hits: List[str] = []
    for c0, c1 in chunks:
        for line in bytes(text[upos(c0):upos(c1)]).decode().splitlines():
            cols = line.split("\t")
            b, e = _vcf_interval(cols)
            if cols[0] == chrom and b < end and e > beg and line not in hits:
                hits.append(line)
    return hits


# ------------------------------------------------------------------------------------------------------
# the whole tail
# ------------------------------------------------------------------------------------------------------
def finish_calls(thresholded_vcf: str, joined_vcf: str, gz_path: str) -> None:
    """call_variants.sh:162-168 without bcftools / sed / bgzip / tabix."""
    with open(thresholded_vcf) as f:
        lines = f.readlines()
    joined = genotype_rewrites(join_multiallelic_lines(lines))
    with open(joined_vcf, "w") as f:
        f.writelines(joined)
    write_vcf_gz_with_index(joined, gz_path)
