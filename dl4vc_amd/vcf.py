"""Scored-VCF output and genotype post-processing.

  * ``start_scored_vcf`` / ``append_scored_records``  <- dl4vc/utils.py:146-158, 162-178
  * ``format_vcf``                                     <- tools/format_vcf.py:51-221

``format_vcf`` defines what "identical genotype calls" means, so its quirks are kept on purpose:
the pending position group is only flushed by a LATER record that passes its threshold (a trailing
group followed only by sub-threshold records is lost, format_vcf.py:203-213), the last group uses a
simpler pruning rule than the others, and a 2-base delete with ``--indel_threshold 0`` is a NameError
there (format_vcf.py:62-84,117).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Iterable, List, Sequence

SCORE_BUCKETS = 50           # tools/format_vcf.py:43
SCORE_FIELD = "BP=%.8f;NV=%.8f;HV=%.8f;OV=%.8f"     # dl4vc/utils.py:175


# ------------------------------------------------------------------------------------------
# A16
# ------------------------------------------------------------------------------------------
def scored_vcf_path(filename: str, epoch: int = 1) -> str:
    """``<dir>/epoch<E>_<basename>`` -- dl4vc/utils.py:150-153 (call_variants.sh:151 reads epoch1_*)."""
    return os.path.join(os.path.dirname(filename), ("epoch%s_" % epoch) + os.path.basename(filename))


def start_scored_vcf(sample_vcf: str, filename: str, epoch: int = 1) -> str:
    """Create the output file holding only the header of ``sample_vcf``.

    The reference copies the header through pysam (utils.py:149-155); pysam is not a dependency here,
    so the '#' lines are copied as text (pysam would additionally normalise them, e.g. inject a
    ``##FILTER=<ID=PASS,...>`` line -- irrelevant to every downstream consumer in call_variants.sh)."""
    out = scored_vcf_path(filename, epoch)
    with open(sample_vcf, "r") as fin, open(out, "w") as fout:
        for line in fin:
            if not line.startswith("#"):
                break
            fout.write(line)
    return out


def score_field(bp: float, vt: Sequence[float]) -> str:
    return SCORE_FIELD % (float(bp), float(vt[0]), float(vt[1]), float(vt[2]))


def scored_record(vcf_record: str, bp: float, vt: Sequence[float]) -> str:
    items = vcf_record.strip().split("\t")
    assert items[2] == ".", "DANGER -- would replace non-empty INFO -- check the hack"      # utils.py:172
    items[2] = score_field(bp, vt)
    return "\t".join(items)


def append_scored_records(vcf_file: str, bp: Iterable[float], vt: Iterable[Sequence[float]],
                          vcf_records: Sequence[str]) -> None:
    """Append one line per site with the scores spliced into the ID column (utils.py:162-178).
    Scores are fp32 values printed with ``%.8f`` (i.e. the exact double of the float32)."""
    bp = list(bp)
    vt = list(vt)
    assert len(bp) == len(vcf_records), "mis-match between results and VCF to save"
    assert os.path.isfile(vcf_file), "VCF file for append does not exist -- need to initialize it"
    with open(vcf_file, "a") as f:
        f.write("".join(scored_record(r, b, v) + "\n" for r, b, v in zip(vcf_records, bp, vt)))


def sort_scored_vcf_lines(lines: Sequence[str]) -> List[str]:
    """In-process twin of the pipeline's sort stage (call_variants.sh:151):
    ``awk '$1 ~ /^#/ {print; next} {print | "sort -k1,1 -k2,2n"}'`` -- header lines first in their original order,
    records by CHROM (byte order, i.e. ``LC_ALL=C``), then POS numerically, ties by the whole line (GNU sort's
    last-resort comparison)."""
    header = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if not l.startswith("#")]

    def key(line):
        cols = line.rstrip("\n").split("\t")
        chrom = cols[0].encode()
        try:
            pos = float(cols[1].strip()) if len(cols) > 1 else 0.0
        except ValueError:
            pos = 0.0                                       # `sort -n` reads a non-number as 0
        return (chrom, pos, line.encode())

    return header + sorted(body, key=key)


# ------------------------------------------------------------------------------------------
# A17
# ------------------------------------------------------------------------------------------
@dataclass
class FormatOptions:
    """Flags of tools/format_vcf.py:227-240; defaults are that script's, call_variants.sh:154-160 passes
    snp 0.1 / indel 0.2 / snp_zygo 0.75 / indel_zygo 0.8."""
    snp_threshold: float = 0.3
    indel_threshold: float = 0.0
    long_indel_threshold: float = 0.0
    delete_threshold: float = 0.0
    snp_zygo_threshold: float = 0.5
    indel_zygo_threshold: float = 0.5
    long_indel_zygo_threshold: float = 0.5
    delete_zygo_threshold: float = 0.5
    multiallele_second_threshold: float = 0.7
    multiallele_homozygous_second_threshold: float = 0.9


PIPELINE_OPTIONS = dict(snp_threshold=0.1, indel_threshold=0.2, snp_zygo_threshold=0.75, indel_zygo_threshold=0.8)


class _Thresholds:
    def __init__(self, o: FormatOptions):
        self.snp, self.snp_hz = o.snp_threshold, o.snp_zygo_threshold
        self.delete = self.delete_hz = None              # stays undefined on the reference's else-branch
        if o.indel_threshold > 0.0:
            self.indel, self.indel_hz = o.indel_threshold, o.indel_zygo_threshold
            if o.long_indel_threshold > 0.0:
                self.long, self.long_hz = o.long_indel_threshold, o.long_indel_zygo_threshold
            else:
                self.long, self.long_hz = self.indel, self.indel_hz
            if o.delete_threshold > 0.0:
                self.delete, self.delete_hz = o.delete_threshold, o.delete_zygo_threshold
            else:
                self.delete, self.delete_hz = self.indel, self.indel_hz
        else:
            self.indel, self.indel_hz = self.snp, self.snp_hz
            self.long, self.long_hz = self.indel, self.indel_hz

    def pick(self, ref_s: str, alt_s: str):
        """(call threshold, homozygous threshold) -- format_vcf.py:112-117,126."""
        if len(ref_s) == 1 and len(alt_s) == 1:
            return self.snp, self.snp_hz
        if len(ref_s) >= 3 or len(alt_s) >= 3:
            return self.long, self.long_hz
        if len(ref_s) > 1 and len(alt_s) == 1:
            if self.delete is None:
                raise NameError("name 'delete_threshold' is not defined")
            return self.delete, self.delete_hz
        return self.indel, self.indel_hz


class _Group:
    """Called alleles sharing one (chrom, pos)."""
    def __init__(self, chrom, pos):
        self.chrom, self.pos = chrom, pos
        self.lines: List[str] = []
        self.scores: List[float] = []
        self.gts: List[str] = []

    def add(self, line, score, gt):
        self.lines.append(line)
        self.scores.append(score)
        self.gts.append(gt)

    def _ranked(self):
        order = sorted(zip(self.scores, self.lines), reverse=True)
        top2 = [self.lines.index(l) for _, l in order][:2]
        assert top2[0] != top2[1]
        return top2

    def resolve(self, o: FormatOptions) -> List[str]:
        """Pruning applied when a later call flushes this group -- format_vcf.py:158-196."""
        lines = self.lines
        if "1/1" in self.gts:
            best = self.gts.index("1/1")
            if len(lines) > 1:
                top2 = self._ranked()
                second_strong = self.scores[top2[1]] >= o.multiallele_homozygous_second_threshold
                first_strong_het = (self.scores[top2[0]] >= o.multiallele_homozygous_second_threshold
                                    and self.gts[top2[0]] != "1/1")
                if not (second_strong or first_strong_het):
                    lines = [lines[best]]
        if len(lines) > 2:
            top2 = self._ranked()
            if self.scores[top2[1]] <= o.multiallele_second_threshold:
                top2 = top2[:1]
            lines = [lines[i] for i in top2]
        return lines

    def resolve_last(self) -> List[str]:
        """Simpler rule used for the file's final group -- format_vcf.py:204-213."""
        lines = self.lines
        if "1/1" in self.gts:
            lines = [lines[self.gts.index("1/1")]]
        if len(lines) > 2:
            top2 = [self.scores.index(s) for s in sorted(self.scores)[-2:]]
            lines = [lines[i] for i in top2]
        return lines


def format_vcf_lines(lines: Sequence[str], options: FormatOptions = None) -> List[str]:
    """Scored, position-sorted VCF lines (with '\\n') -> genotyped lines.  Same behaviour as
    ``tools/format_vcf.py::filter_format_vcf`` minus its prints."""
    o = options or FormatOptions()
    th = _Thresholds(o)
    out: List[str] = []
    group = None
    n = len(lines)
    for lineno, line in enumerate(lines, 1):
        if line[0] == "#":
            out.append(line)
            continue
        items = line.strip("\n").split("\t")
        assert len(items) in (10, 11), 'Line should have 10-11 items (11th is appended "GT:1/1")\n%s' % line
        scores = {k: float(v) for k, v in (s.split("=") for s in items[2].split(";"))}
        call_score = 1.0 - scores["NV"]
        thr, hz_thr = th.pick(items[3], items[4])
        margin = call_score - thr
        if margin < 0.0:
            continue
        gt = "1/1" if scores["OV"] >= hz_thr else "0/1"
        q = int(margin / (1.0 - thr) * SCORE_BUCKETS)
        new_line = "\t".join(items[0:9] + ["%s:%s" % (gt, q)])
        if group is None:
            group = _Group(items[0], items[1])
        elif not (group.chrom == items[0] and group.pos == items[1]):
            out.extend(l + "\n" for l in group.resolve(o))
            group = _Group(items[0], items[1])
        group.add(new_line, call_score, gt)
        if lineno == n:
            out.extend(l + "\n" for l in group.resolve_last())
    return out


def threshold_distance(vcfrecs: Sequence[str], vt_prob, options: FormatOptions = None) -> "np.ndarray":
    """Per site, the distance of its scores to the nearest decision threshold ``format_vcf`` would hold them against: the call
    threshold of the record's allele class on ``1 - NV``, that class's homozygous threshold on ``OV`` (format_vcf.py:107-126)
    and, where the site passes, the two multi-allele thresholds on the call score (:158-196).  A score tolerance cannot by
    itself guarantee identical genotype calls: a site closer to a threshold than the tolerance is where two correct
    evaluations may disagree (hard part 4) -- this is the count the pipeline logs."""
    import numpy as np
    o = options or FormatOptions()
    th = _Thresholds(o)
    vt = np.asarray(vt_prob, np.float64).reshape(-1, 3)
    if len(vcfrecs) == 0:
        return np.empty(0, np.float64)
    # one pass over the records for their allele class (REF / ALT columns only), the distances as array arithmetic: this runs on
    # the scoring path (main.py, every delivered batch), not in a post-processing tool
    picks = np.array([th.pick(*rec.split("\t", 5)[3:5]) for rec in vcfrecs], np.float64).reshape(-1, 2)
    call = 1.0 - vt[:, 0]
    d = np.minimum(np.abs(call - picks[:, 0]), np.abs(vt[:, 2] - picks[:, 1]))
    multi = np.minimum(np.abs(call - o.multiallele_second_threshold), np.abs(call - o.multiallele_homozygous_second_threshold))
    return np.where(call >= picks[:, 0], np.minimum(d, multi), d)


def format_vcf(input_file: str, output_file: str, options: FormatOptions = None) -> None:
    with open(input_file, "r") as f:
        lines = f.readlines()
    with open(output_file, "w") as f:
        f.writelines(format_vcf_lines(lines, options))
