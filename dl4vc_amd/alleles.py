"""Variant-proposal encoding: VCF record -> allele mask vectors.

Host-side logic, restated from the reference's behaviour (not its text):
  * ``allele_token_vectors``  <- dl4vc/dataset.py:86-109  (simple_variant_encoding_vectors)
  * ``allele_mask_vectors``   <- dl4vc/dataset.py:112-250 (get_read_mask_vectors)
  * ``parse_candidate``       <- dl4vc/utils.py:19-72     (parse_vcf)
  * ``count_center_support``  <- dl4vc/dataset.py:340-361 (count_variants_from_single_reads)

The two mask vectors are what the device consumes: length-L uint8 token vectors that are zero
except over the allele span anchored at the window centre (column 100).  A read "matches" an
allele iff it equals the mask at every non-zero column (dl4vc/model.py:576-627).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from . import vocab as V

VAR_ENCODE_LEN = 51          # dl4vc/dataset.py:85
CENTER = 100                 # window centre column, dl4vc/dataset.py:129 (READ_MIDPOINT, model.py:19)


class AlleleMaskError(AssertionError):
    """Raised where the reference's ``assert``s fire (dataset.py:114-243).  The dataset layer turns it
    into all-zero masks + a blacklist flag exactly like dataset.py:644-663 does for AssertionError."""


def _fields(vcf_record: str):
    return vcf_record.strip().split("\t")


def allele_token_vectors(vcf_record: str, insert_limit: int = VAR_ENCODE_LEN,
                         delete_limit: int = VAR_ENCODE_LEN, keep_pad: bool = True
                         ) -> Tuple[np.ndarray, np.ndarray]:
    """REF / ALT strings as uint8 token vectors, padded (or clipped) the way the reference does.

    A limit of 0 means "do not truncate, do not pad" for that allele (dataset.py:91-98)."""
    rec = _fields(vcf_record)
    ref_s, alt_s = rec[3], rec[4]
    if delete_limit > 0:
        ref_s = ref_s[:delete_limit]
    if insert_limit > 0:
        alt_s = alt_s[:insert_limit]
    ref_v = np.zeros(max(delete_limit, len(ref_s)), dtype=np.uint8)
    alt_v = np.zeros(max(insert_limit, len(alt_s)), dtype=np.uint8)
    ref_v[:len(ref_s)] = [V.token_of(c) for c in ref_s]
    alt_v[:len(alt_s)] = [V.token_of(c) for c in alt_s]
    if not keep_pad:                                   # clip at the first pad token
        for name, vec in (("r", ref_v), ("a", alt_v)):
            pads = np.flatnonzero(vec == V.PAD)
            if len(pads):
                if name == "r":
                    ref_v = vec[:pads[0]]
                else:
                    alt_v = vec[:pads[0]]
    return ref_v, alt_v


def _anchor(reference: np.ndarray) -> int:
    """Centre column, rewound past gap columns another allele's insert opened (dataset.py:129-132)."""
    off = CENTER
    while reference[off] == V.GAP:
        off -= 1
    return off


def allele_mask_vectors(vcf_record: str, reference: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(ref_mask, var_mask): uint8[L], zero outside the allele span.

    SNP ``A->G``       : one token each at the anchor.
    Delete ``ATG->A``  : ref ``[A,T,G]``, var ``[A,-,-]``; gap columns of the window that fall inside the
                         span become 0 ("match anything") in both vectors (dataset.py:194-233).
    Insert ``A->ATT``  : ref ``[A,noinsert,noinsert]``, var ``[A,T,T]`` (dataset.py:239-241).
    """
    reference = np.asarray(reference)
    L = len(reference)
    if L != 201:
        raise AlleleMaskError("allele masks assume a 201-column window")            # dataset.py:114
    rec = _fields(vcf_record)
    ref_s, alt_s = rec[3], rec[4]
    ref_v, alt_v = allele_token_vectors(vcf_record, delete_limit=0, keep_pad=False)
    ref_v = ref_v.astype(np.int64)
    alt_v = alt_v.astype(np.int64)

    is_snp = (len(ref_s) == 1 and len(alt_s) == 1
              and ref_s in V.SNP_BASE_CHARS and alt_s in V.SNP_BASE_CHARS)
    if is_snp:
        off = _anchor(reference)
    elif len(ref_s) > len(alt_s):
        off = _anchor(reference)
        if len(alt_s) != 1:
            raise AlleleMaskError("For deletes, expect exactly one base in variant. [%s -> %s]" % (ref_s, alt_s))
    elif len(alt_s) > len(ref_s):
        if len(ref_s) != 1:
            raise AlleleMaskError("For inserts, expect exactly one base in reference. [%s -> %s]" % (ref_s, alt_s))
        off = _anchor(reference)
    else:
        # equal-length non-SNP pairs (MNPs, lower-case 'g' SNPs): the reference falls through with the
        # offsets unbound and dies with UnboundLocalError (dataset.py:177) -- not an AssertionError, so
        # it is not blacklisted there either.
        raise UnboundLocalError("allele pair %s -> %s is neither SNP, insert nor delete" % (ref_s, alt_s))
    if reference[off] != ref_v[0]:
        raise AlleleMaskError("Did not find (first) ref base in reference! [%s -> %s] %d: %s"
                              % (ref_s, alt_s, off, reference[off]))                # dataset.py:173,183

    if len(ref_v) > 1:                                   # delete: spell the deletion out in the variant
        alt_v = np.concatenate((alt_v, np.full(len(ref_v) - len(alt_v), V.GAP, dtype=np.int64)))
        if not np.array_equal(ref_v, reference[off:off + len(ref_v)]):
            # The window has gap columns inside the deleted span: walk the window, absorbing either the
            # next deleted base or a gap column (dataset.py:202-221).
            new_ref, new_alt, k = [], [], 0
            for col in range(off, L):
                if k >= len(ref_v):
                    break
                if reference[col] == ref_v[k]:
                    new_ref.append(int(reference[col]))
                    new_alt.append(int(alt_v[k]))
                    k += 1
                elif reference[col] == V.GAP:
                    new_ref.append(V.GAP)
                    new_alt.append(V.NOINSERT)
                else:
                    raise AlleleMaskError("Mis-match inserting pad delete into reference. [%s -> %s]" % (ref_s, alt_s))
            if k < len(ref_v):
                raise AlleleMaskError("Finished padding, did not reach end of pad insert")
            ref_v = np.array(new_ref, dtype=np.int64)
            alt_v = np.array(new_alt, dtype=np.int64)
            # gap columns inside the span match anything, in both vectors (dataset.py:228-233)
            ref_v[ref_v == V.GAP] = V.PAD
            alt_v[alt_v == V.NOINSERT] = V.PAD
    if len(ref_v) == 1 and len(alt_v) > 1:                # insert: reads without it show 'noinsert'
        ref_v = np.concatenate((ref_v, np.full(len(alt_v) - 1, V.NOINSERT, dtype=np.int64)))
    if len(ref_v) != len(alt_v):
        raise AlleleMaskError("Need to adjust ref, var vectors for same length!")

    ref_mask = np.zeros(L, dtype=np.uint8)
    var_mask = np.zeros(L, dtype=np.uint8)
    ref_mask[off:off + len(ref_v)] = ref_v                # ValueError if the span leaves the window, as numpy does there
    var_mask[off:off + len(alt_v)] = alt_v
    return ref_mask, var_mask


def safe_allele_mask_vectors(vcf_record: str, reference: np.ndarray):
    """Dataset-level wrapper: (ref_mask, var_mask, blacklisted).  dataset.py:644-663."""
    try:
        r, v = allele_mask_vectors(vcf_record, reference)
        return r, v, False
    except AssertionError:
        L = len(reference)
        return np.zeros(L, np.uint8), np.zeros(L, np.uint8), True


_VAR_TYPE = {"homo": 2, "hetero": 1, "none": 0}


def parse_candidate(vcf_record: str) -> Dict[str, object]:
    """Facts the harness needs from one candidate VCF line (reference: utils.py:19-72)."""
    rec = _fields(vcf_record)
    ref_s, alt_s = rec[3], rec[4]
    res: Dict[str, object] = {}
    if len(ref_s) == 1 and len(alt_s) == 1 and ref_s in V.SNP_BASE_CHARS and alt_s in V.SNP_BASE_CHARS:
        res.update(is_snp=True, var_mode=V.MUTATION_SNP, ref_base=V.token_of(ref_s), var_base=V.token_of(alt_s))
    elif len(ref_s) > len(alt_s):
        res.update(is_snp=False, var_mode=V.MUTATION_DELETE, ref_base=V.token_of(ref_s[0]), var_base=V.GAP)
    elif len(ref_s) < len(alt_s):
        res.update(is_snp=False, var_mode=V.MUTATION_INSERT, ref_base=V.token_of(ref_s[0]), var_base=V.NOINSERT)
    else:
        print("Unknown mutation detected!!! %s" % str(rec))
        res["is_snp"] = False
    stats = dict(kv.split("=") for kv in rec[7].split(";"))
    res["allele_freq"] = float(stats["AF"])
    res["coverage"] = int(stats["DP"])
    res["var_type"] = _VAR_TYPE["none"]
    if len(rec) > 10:                                      # an appended truth column such as "GT:0/1"
        gt, var = rec[10].split(":")
        if gt == "GT" and len(var) == 3 and var[1] in "/|":
            if var[0] == "1" and var[2] == "1":
                res["var_type"] = _VAR_TYPE["homo"]
            elif (var[0], var[2]) in (("0", "1"), ("1", "0")):
                res["var_type"] = _VAR_TYPE["hetero"]
    return res


_SUPPORT_TOKENS = frozenset((V.A, V.T, V.C, V.G, V.GAP, V.UNK, V.NOINSERT))   # dataset.py real_base_keys_set


def count_center_support(reads_pos_major: np.ndarray, reference: np.ndarray, var_mode: int):
    """(covered, agree, disagree) at the centre column -- dataset.py:340-361.
    ``reads_pos_major`` is (L, n_reads) as the reference holds it."""
    if var_mode == V.MUTATION_SNP:
        ref_base, col = int(reference[CENTER]), reads_pos_major[CENTER]
    elif var_mode == V.MUTATION_DELETE:
        ref_base, col = int(reference[CENTER + 1]), reads_pos_major[CENTER + 1]
    elif var_mode == V.MUTATION_INSERT:
        ref_base, col = V.NOINSERT, reads_pos_major[CENTER + 1]
    else:
        raise UnboundLocalError("unknown mutation type")
    vals, cnts = np.unique(col, return_counts=True)
    hist = dict(zip(vals.tolist(), cnts.tolist()))
    agree = hist.get(ref_base, 0)
    disagree = sum(hist.get(t, 0) for t in _SUPPORT_TOKENS - {ref_base})
    return agree + disagree, agree, disagree
