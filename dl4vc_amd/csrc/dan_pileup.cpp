// Native pileup encoder: BAM + FASTA + candidate locations -> the image planes of the
// candidate records main.py scores.  Part of libdl4vc_loader.so (C ABI: include/dl4vc_loader.h, pe_*).
//
// Restates, for the common case, what dl4vc_amd/pileup_encoder.py::process_tracks + finish_record compute -- themselves the
// read-by-read form of the reference's tools/convert_bam_single_reads.py::process_location (:846-1118) and the crop / centre
// / pad step of process_locations_chunk (:720-838) -- together with the layers under them: BGZF blocks (zlib raw inflate +
// CRC), BAM header / records (SAM specification section 4.2), the BAI linear index (section 5.2), a forward-moving window
// over the alignments (each record inflated and parsed once per run of locations), per-read CIGAR resolution (htslib's
// pileup rules: qpos / is_del / is_refskip / merged indel lengths) and FASTA + .fai access.  A location the read-by-read form
// declines (two reads sharing a name:sequence key, a reference skip, a base outside the token table, > 1000 columns, > 8000
// reads) is reported back as status 2 and takes the Python column-by-column path: results are byte-identical to the Python
// module by construction (tests/test_pileup_native.py).  Worker threads take contiguous runs of locations, each with its
// own file handles and window.
#include "../../include/dl4vc_loader.h"

#include <zlib.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

namespace {

std::string g_pe_error;

// ---- token tables (tools/convert_bam_single_reads.py:38-56; dl4vc_amd/pileup_encoder.py::BASE_ENUM) ---------------------------
constexpr uint8_t PAD = 0, START = 6, END = 7, NOINSERT = 8, STRAND_PAD = 200, STRAND_LOWER = 1, STRAND_UPPER = 2, TOK_GAP = 5;
struct Tables {
    uint8_t token[256];
    bool known[256];
    Tables() {
        memset(token, 0, sizeof token);
        memset(known, 0, sizeof known);
        auto set = [&](const char* cs, uint8_t v) { for (; *cs; ++cs) { token[(uint8_t)*cs] = v; known[(uint8_t)*cs] = true; } };
        set("Aa", 1); set("TtUu", 2); set("Gg", 3); set("Cc", 4); set("-*NnXx.,", 5); set("e", 7);
        set("?MmKkRrYySsWwBbVvHhDd", 9);
    }
};
const Tables T;
const char SEQ_CODES[] = "=ACMGRSVTWYHKDBN";
enum { CMATCH, CINS, CDEL, CREF_SKIP, CSOFT_CLIP, CHARD_CLIP, CPADOP, CEQUAL, CDIFF };
inline bool is_aligned(int op) { return op == CMATCH || op == CEQUAL || op == CDIFF; }
inline bool is_refop(int op) { return is_aligned(op) || op == CDEL || op == CREF_SKIP; }
constexpr int FLAG_MASK = 0x4 | 0x100 | 0x200 | 0x400;          // unmapped, secondary, QC fail, duplicate (BAM_DEF_MASK)
constexpr int FREVERSE = 0x10;

// ---- BGZF -------------------------------------------------------------------------------------------------------------------------
struct Bgzf {
    FILE* f = nullptr;                                           // owned: closed with the reader (also when an exception unwinds past it)
    Bgzf() = default;
    Bgzf(const Bgzf&) = delete;
    Bgzf& operator=(const Bgzf&) = delete;
    ~Bgzf() { if (f) fclose(f); }
    int64_t block_start = 0, next_block = 0;
    std::vector<uint8_t> data, raw;
    size_t off = 0;
    std::string err;

    bool load(int64_t file_off) {
        if (fseeko(f, file_off, SEEK_SET) != 0) { err = "seek failed"; return false; }
        uint8_t head[18];
        const size_t got = fread(head, 1, 18, f);
        if (got == 0) { block_start = next_block = file_off; data.clear(); off = 0; return false; }
        if (got < 18 || head[0] != 0x1f || head[1] != 0x8b || head[2] != 8 || head[3] != 4) { err = "not a BGZF block"; return false; }
        const int xlen = head[10] | (head[11] << 8);
        std::vector<uint8_t> extra(xlen);
        memcpy(extra.data(), head + 12, std::min(6, xlen));
        if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, f) != (size_t)(xlen - 6)) { err = "truncated BGZF header"; return false; }
        int bsize = -1;
        for (int i = 0; i + 4 <= xlen;) {
            const int slen = extra[i + 2] | (extra[i + 3] << 8);
            if (extra[i] == 'B' && extra[i + 1] == 'C' && i + 6 <= xlen) bsize = extra[i + 4] | (extra[i + 5] << 8);
            i += 4 + slen;
        }
        if (bsize < 0) { err = "BGZF block without a BC field"; return false; }
        const int body = bsize + 1 - 12 - xlen;
        if (body < 8) { err = "truncated BGZF block"; return false; }
        raw.resize(body);
        if (fread(raw.data(), 1, body, f) != (size_t)body) { err = "truncated BGZF block"; return false; }
        uint32_t crc, isize;
        memcpy(&crc, raw.data() + body - 8, 4);
        memcpy(&isize, raw.data() + body - 4, 4);
        if (isize > 65536) { err = "BGZF block claims more than 64 KiB of data"; return false; }   // (the format's limit; not a size to trust)
        data.resize(isize);
        uint8_t scratch[8];
        z_stream zs{};
        if (inflateInit2(&zs, -15) != Z_OK) { err = "inflateInit2 failed"; return false; }
        zs.next_in = raw.data(); zs.avail_in = body - 8;
        zs.next_out = isize ? data.data() : scratch; zs.avail_out = isize ? isize : (unsigned)sizeof scratch;   // (the empty end-of-file block)
        const int rc = inflate(&zs, Z_FINISH);
        const bool ok = rc == Z_STREAM_END && zs.total_out == isize;
        inflateEnd(&zs);
        if (!ok || (uint32_t)crc32(0L, isize ? data.data() : scratch, isize) != crc) { err = "BGZF block fails its CRC / size check"; return false; }
        block_start = file_off; next_block = file_off + bsize + 1; off = 0;
        return true;
    }
    int64_t tell() const { return (block_start << 16) | (int64_t)off; }
    bool seek(int64_t voff) {
        const int64_t blk = voff >> 16;
        if (blk != block_start || data.empty()) { err.clear(); load(blk); if (!err.empty()) return false; }
        off = (size_t)(voff & 0xffff);
        return true;
    }
    // reads up to n bytes; returns the count (short at end of file); err set on a corrupt block
    size_t read(void* dst, size_t n) {
        size_t done = 0;
        while (n > 0) {
            if (off >= data.size()) {
                err.clear();
                if (!load(next_block)) { if (!err.empty()) return done; break; }
                continue;
            }
            const size_t take = std::min(n, data.size() - off);
            memcpy((uint8_t*)dst + done, data.data() + off, take);
            off += take; done += take; n -= take;
        }
        return done;
    }
};

struct Rec {
    int32_t tid = -1, pos = 0, ref_end = 0;
    uint16_t flag = 0;
    std::string name, seq;
    std::vector<uint32_t> cigar;                                 // (len << 4) | op
    std::vector<uint8_t> qual;
    // resolved against the reference (pileup.py::ReadTrack), filled on first use
    bool resolved = false, has_ref = false;
    int32_t start = 0, end = 0;
    std::vector<int32_t> qpos, indel;
    std::vector<uint8_t> is_del, is_skip;

    void resolve() {
        if (resolved) return;
        resolved = true;
        int n = 0;
        for (uint32_t c : cigar) if (is_refop(c & 0xf)) n += (int)(c >> 4);
        has_ref = false;
        for (uint32_t c : cigar) if (is_refop(c & 0xf)) has_ref = true;
        start = pos; end = pos + n;
        qpos.assign(n, 0); indel.assign(n, 0); is_del.assign(n, 0); is_skip.assign(n, 0);
        int x = 0, y = 0;
        const int nc = (int)cigar.size();
        for (int k = 0; k < nc; ++k) {
            const int op = cigar[k] & 0xf, l = (int)(cigar[k] >> 4);
            if (is_aligned(op)) { for (int i = 0; i < l; ++i) qpos[x + i] = y + i; x += l; y += l; }
            else if (op == CDEL || op == CREF_SKIP) {
                for (int i = 0; i < l; ++i) { qpos[x + i] = y; is_del[x + i] = 1; is_skip[x + i] = op == CREF_SKIP; }
                x += l;
            } else if (op == CINS || op == CSOFT_CLIP) y += l;
            if (is_refop(op) && x > 0 && k + 1 < nc) {
                const int op2 = cigar[k + 1] & 0xf, l2 = (int)(cigar[k + 1] >> 4);
                int v = 0;
                if (op2 == CDEL && op != CDEL) {
                    v = -l2;
                    for (int j = k + 2; j < nc; ++j) { if ((int)(cigar[j] & 0xf) != CDEL) break; v -= (int)(cigar[j] >> 4); }
                } else if (op2 == CINS) {
                    v = l2;
                    for (int j = k + 2; j < nc; ++j) {
                        const int op3 = cigar[j] & 0xf;
                        if (op3 == CINS) v += (int)(cigar[j] >> 4);
                        else if (op3 != CPADOP) break;
                    }
                } else if (op2 == CPADOP && k + 2 < nc) {
                    for (int j = k + 2; j < nc; ++j) {
                        const int op3 = cigar[j] & 0xf;
                        if (op3 == CINS) v += (int)(cigar[j] >> 4);
                        else if (is_refop(op3)) break;
                    }
                }
                indel[x - 1] = v;
            }
        }
    }
};

struct Fasta {
    FILE* f = nullptr;
    struct Entry { int64_t length, offset, lb, lw; };
    std::map<std::string, Entry> index;

    bool open(const std::string& path, std::string& err) {
        f = fopen(path.c_str(), "rb");
        if (!f) { err = "cannot open " + path; return false; }
        FILE* fai = fopen((path + ".fai").c_str(), "r");
        if (fai) {
            char line[4096];
            while (fgets(line, sizeof line, fai)) {
                char name[2048];
                long long a, b, c, d;
                // tab separated: name, length, offset, line bases, line width
                char* tab = strchr(line, '\t');
                if (!tab) continue;
                const size_t nl = (size_t)(tab - line);
                if (nl >= sizeof name) continue;
                memcpy(name, line, nl); name[nl] = 0;
                if (sscanf(tab + 1, "%lld\t%lld\t%lld\t%lld", &a, &b, &c, &d) == 4) index[name] = Entry{a, b, c, d};
            }
            fclose(fai);
        } else {
            // scan (dl4vc_amd/bamio.py::FastaFile._scan)
            std::string name;
            bool have = false;
            Entry e{0, 0, 0, 0};
            int64_t pos = 0;
            std::vector<char> buf(1 << 20);
            std::string line;
            int ch;
            line.reserve(256);
            auto flush_line = [&](const std::string& ln) {
                if (!ln.empty() && ln[0] == '>') {
                    if (have) index[name] = e;
                    size_t a = 1, b = 1;
                    while (b < ln.size() && !isspace((unsigned char)ln[b])) ++b;
                    name = ln.substr(a, b - a);
                    have = true;
                    e = Entry{0, pos + (int64_t)ln.size(), 0, 0};
                } else if (have) {
                    size_t bases = ln.size();
                    while (bases > 0 && (ln[bases - 1] == '\n' || ln[bases - 1] == '\r')) --bases;
                    if (e.lb == 0) { e.lb = (int64_t)bases; e.lw = (int64_t)ln.size(); }
                    e.length += (int64_t)bases;
                }
                pos += (int64_t)ln.size();
            };
            while ((ch = fgetc(f)) != EOF) {
                line.push_back((char)ch);
                if (ch == '\n') { flush_line(line); line.clear(); }
            }
            if (!line.empty()) flush_line(line);
            if (have) index[name] = e;
        }
        return true;
    }
    const Entry* entry(const std::string& ref) const {
        auto it = index.find(ref);
        if (it != index.end()) return &it->second;
        const std::string alt = ref.rfind("chr", 0) == 0 ? ref.substr(3) : "chr" + ref;
        it = index.find(alt);
        return it == index.end() ? nullptr : &it->second;
    }
    // bases of [start, end) as stored; false when the sequence is absent
    bool fetch(const std::string& ref, int64_t start, int64_t end, std::string& out) {
        out.clear();
        const Entry* e = entry(ref);
        if (!e) return false;
        start = std::max<int64_t>(0, start); end = std::min(end, e->length);
        if (end <= start || e->lb <= 0) return true;
        const int64_t first = e->offset + (start / e->lb) * e->lw + start % e->lb;
        const int64_t last = e->offset + ((end - 1) / e->lb) * e->lw + (end - 1) % e->lb;
        std::vector<char> raw((size_t)(last - first + 1));
        fseeko(f, first, SEEK_SET);
        const size_t got = fread(raw.data(), 1, raw.size(), f);
        for (size_t i = 0; i < got; ++i) if (raw[i] != '\n' && raw[i] != '\r') out.push_back(raw[i]);
        return true;
    }
    ~Fasta() { if (f) fclose(f); }
};

struct Bam {
    Bgzf r;
    std::vector<std::string> refs;
    std::map<std::string, int> tid_of;
    int64_t first_record = 0;
    std::string err;

    bool open(const std::string& path) {
        r.f = fopen(path.c_str(), "rb");
        if (!r.f) { err = "cannot open " + path; return false; }
        char magic[4];
        if (r.read(magic, 4) != 4 || memcmp(magic, "BAM\1", 4) != 0) { err = path + " is not a BAM file"; return false; }
        int32_t l_text = 0, n_ref = 0;
        if (r.read(&l_text, 4) != 4) { err = "truncated BAM header"; return false; }
        if (l_text < 0 || l_text > (1 << 30)) { err = "corrupt BAM header (text length)"; return false; }
        std::vector<char> text((size_t)l_text);
        if (r.read(text.data(), text.size()) != text.size() || r.read(&n_ref, 4) != 4) { err = "truncated BAM header"; return false; }
        if (n_ref < 0) { err = "corrupt BAM header (reference count)"; return false; }
        for (int i = 0; i < n_ref; ++i) {
            int32_t ln = 0, len = 0;
            if (r.read(&ln, 4) != 4) { err = "truncated BAM header"; return false; }
            if (ln < 1 || ln > 65536) { err = "corrupt BAM header (reference name length)"; return false; }
            std::vector<char> nm((size_t)ln);
            if (r.read(nm.data(), nm.size()) != nm.size() || r.read(&len, 4) != 4) { err = "truncated BAM header"; return false; }
            refs.emplace_back(nm.data(), ln > 0 ? (size_t)ln - 1 : 0);
            tid_of[refs.back()] = i;
        }
        first_record = r.tell();
        return true;
    }
    int get_tid(const std::string& name) const {
        auto it = tid_of.find(name);
        if (it != tid_of.end()) return it->second;
        const std::string alt = name.rfind("chr", 0) == 0 ? name.substr(3) : "chr" + name;
        it = tid_of.find(alt);
        return it == tid_of.end() ? -1 : it->second;
    }
    // next record; 0 = end of file, 1 = ok, -1 = error
    int next(std::shared_ptr<Rec>& out) {
        int32_t size = 0;
        const size_t g = r.read(&size, 4);
        if (g < 4) { if (!r.err.empty()) err = r.err; return r.err.empty() ? 0 : -1; }
        // every length below comes from the file: checked against the record before anything is sized or indexed by it
        if (size < 32 || size > (1 << 28)) { err = "corrupt BAM record (block_size)"; return -1; }
        std::vector<uint8_t> b((size_t)size);
        if (r.read(b.data(), b.size()) != b.size()) { err = r.err.empty() ? "truncated BAM record" : r.err; return -1; }
        auto rec = std::make_shared<Rec>();
        int32_t tid, pos, l_seq;
        uint8_t l_name;
        uint16_t n_cig, flag;
        memcpy(&tid, &b[0], 4); memcpy(&pos, &b[4], 4);
        l_name = b[8];
        memcpy(&n_cig, &b[12], 2); memcpy(&flag, &b[14], 2); memcpy(&l_seq, &b[16], 4);
        if (l_seq < 0 || (uint64_t)32 + l_name + 4 * (uint64_t)n_cig + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq > (uint64_t)size) {
            err = "corrupt BAM record (name / CIGAR / sequence lengths exceed the record)";
            return -1;
        }
        rec->tid = tid; rec->pos = pos; rec->flag = flag;
        size_t o = 32;
        rec->name.assign((const char*)&b[o], l_name > 0 ? (size_t)l_name - 1 : 0);
        o += l_name;
        rec->cigar.resize(n_cig);
        int nref = 0;
        for (int i = 0; i < n_cig; ++i) {
            uint32_t v;
            memcpy(&v, &b[o + 4 * i], 4);
            rec->cigar[i] = v;
            const int op = v & 0xf;
            if (op == CMATCH || op == CDEL || op == CREF_SKIP || op == CEQUAL || op == CDIFF) nref += (int)(v >> 4);
        }
        rec->ref_end = pos + (nref > 0 ? nref : 1);
        o += 4 * (size_t)n_cig;
        rec->seq.resize((size_t)l_seq);
        for (int i = 0; i < l_seq; ++i) {
            const uint8_t p = b[o + i / 2];
            rec->seq[i] = SEQ_CODES[(i & 1) ? (p & 0xf) : (p >> 4)];
        }
        o += ((size_t)l_seq + 1) / 2;
        rec->qual.assign(b.begin() + o, b.begin() + o + l_seq);
        out = rec;
        return 1;
    }
};

struct Bai {
    std::vector<std::vector<uint64_t>> linear;
    bool load(const std::string& path) {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) return false;
        std::vector<uint8_t> raw;
        uint8_t buf[65536];
        size_t g;
        while ((g = fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + g);
        fclose(f);
        if (raw.size() < 8 || memcmp(raw.data(), "BAI\1", 4) != 0) return false;
        size_t o = 4;
        int32_t n_ref;
        memcpy(&n_ref, &raw[o], 4); o += 4;
        for (int r = 0; r < n_ref; ++r) {
            int32_t n_bin;
            memcpy(&n_bin, &raw[o], 4); o += 4;
            for (int b = 0; b < n_bin; ++b) {
                int32_t n_chunk;
                memcpy(&n_chunk, &raw[o + 4], 4);
                o += 8 + 16 * (size_t)n_chunk;
            }
            int32_t n_intv;
            memcpy(&n_intv, &raw[o], 4); o += 4;
            std::vector<uint64_t> lin((size_t)n_intv);
            if (n_intv > 0) memcpy(lin.data(), &raw[o], 8 * (size_t)n_intv);
            o += 8 * (size_t)n_intv;
            linear.push_back(std::move(lin));
        }
        return true;
    }
    // 0 = "no alignment at or after the window" (BaiIndex.linear_offset returning None)
    uint64_t linear_offset(int tid, int64_t start) const {
        if (tid < 0 || tid >= (int)linear.size()) return 0;
        const auto& lin = linear[tid];
        for (size_t w = (size_t)(std::max<int64_t>(start, 0) >> 14); w < lin.size(); ++w) if (lin[w]) return lin[w];
        return 0;
    }
};

// dl4vc_amd/bamio.py::WindowReader
struct Window {
    Bam* bam = nullptr;
    const Bai* bai = nullptr;
    int64_t max_gap = 1 << 16;
    int tid = -2;
    int64_t last_start = -1, scanned_to = -1, at = 0;
    std::vector<std::shared_ptr<Rec>> kept;
    std::shared_ptr<Rec> pending;
    bool eof = true;

    bool reads(int t, int64_t start, int64_t stop, std::vector<std::shared_ptr<Rec>>& out) {
        out.clear();
        if (t < 0) return true;
        if (t != tid || start < last_start || start > scanned_to + max_gap) {
            kept.clear(); pending.reset(); tid = t; scanned_to = -1; eof = false;
            at = bam->first_record;
            if (bai) {
                const uint64_t off = bai->linear_offset(t, start);
                if (off == 0) eof = true; else at = (int64_t)off;
            }
        } else {
            size_t w = 0;
            for (auto& r : kept) if (r->ref_end > start) kept[w++] = r;
            kept.resize(w);
        }
        last_start = start;
        if (stop > scanned_to && !eof) {
            std::shared_ptr<Rec> rec = pending;
            pending.reset();
            if (!rec) { if (!bam->r.seek(at)) return false; }
            for (;;) {
                if (!rec) {
                    const int g = bam->next(rec);
                    if (g < 0) return false;
                    if (g == 0) { eof = true; break; }
                }
                if (rec->tid != t) {
                    if (rec->tid > t || rec->tid < 0) { eof = true; break; }
                } else if (rec->pos >= stop) { pending = rec; break; }
                else if (rec->ref_end > start) kept.push_back(rec);
                rec.reset();
            }
            at = bam->r.tell();
            scanned_to = stop;
        }
        for (auto& r : kept) if (r->pos < stop && r->ref_end > start) out.push_back(r);
        return true;
    }
};

}  // namespace

struct pe_encoder {
    std::string bam_path, bai_path, fasta_path, err;
    pe_options opt{};
    Bai bai;
    bool have_bai = false;
};

namespace {

int pe_fail(pe_encoder* e, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_pe_error = buf;
    return -1;
}

// one location: status 1 = planes written, 0 = no record (the reference counts an error), 2 = needs the Python column path
int encode_one(const pe_options& opt, Window& win, Bam& bam, Fasta& fasta, const std::string& contig, int32_t pos1,
               uint8_t* reads_out, uint8_t* qual_out, uint8_t* strand_out, uint8_t* ref_out, int32_t* num_reads, std::string& err) {
    const int w = opt.window_size, W = 2 * w + 1, MR = opt.max_reads;
    const int window = w + 2;
    const int64_t start = (int64_t)pos1 - window, stop = (int64_t)pos1 + window + 1;
    const int tid = bam.get_tid(contig);
    if (tid < 0) return 0;
    const int64_t s0 = std::max<int64_t>(start, 0);
    std::string ref;
    if (!fasta.fetch(contig, s0, stop + 64, ref)) return 2;        // (the Python path raises the reference's KeyError)
    std::vector<std::shared_ptr<Rec>> recs;
    if (!win.reads(tid, s0, stop, recs)) { err = bam.err.empty() ? bam.r.err : bam.err; return -1; }
    if (opt.min_base_quality > 0) return 2;
    // resolve_reads + the window filter of process_tracks
    std::vector<Rec*> tracks;
    size_t n_resolved = 0;
    for (auto& rp : recs) {
        Rec& r = *rp;
        if ((r.flag & FLAG_MASK) || r.tid < 0) continue;
        r.resolve();
        if (!r.has_ref) continue;
        ++n_resolved;
        if (r.end > s0 && r.start < stop) tracks.push_back(&r);
    }
    if (n_resolved > 8000) return 2;
    {
        std::unordered_set<std::string> keys;
        for (Rec* t : tracks) if (!keys.insert(t->name + ":" + t->seq).second) return 2;
    }
    const int n_pos = (int)(stop - s0);
    std::vector<int> cover(n_pos + 1, 0), longest(n_pos, 0), cap(n_pos, opt.max_insert_length);
    const int ci = pos1 - 1 - (int)s0;
    if (ci >= 0 && ci < n_pos) cap[ci] = std::max(opt.max_insert_length_variant, opt.max_insert_length);
    std::vector<std::pair<int64_t, int64_t>> spans(tracks.size());
    for (size_t i = 0; i < tracks.size(); ++i) {
        Rec* t = tracks[i];
        const int64_t lo = std::max<int64_t>(t->start, s0), hi = std::min<int64_t>(t->end, stop);
        for (int64_t p = lo; p < hi; ++p) if (t->is_skip[p - t->start]) return 2;
        cover[lo - s0] += 1; cover[hi - s0] -= 1;
        for (int64_t p = lo; p < hi; ++p) {
            const int ins = t->indel[p - t->start];
            if (ins > 0) longest[p - s0] = std::max(longest[p - s0], std::min(ins, cap[p - s0]));
        }
        spans[i] = {lo, hi};
    }
    std::vector<int> pos_idx;
    {
        int run = 0;
        for (int p = 0; p < n_pos; ++p) { run += cover[p]; if (run > 0) pos_idx.push_back(p); }
    }
    auto covered = [&](int p) { return std::binary_search(pos_idx.begin(), pos_idx.end(), p); };
    if (pos_idx.empty() || !(ci >= 0 && ci < n_pos && covered(ci))) return 0;
    if (pos_idx.size() > 1001) return 2;
    std::vector<int64_t> col_of(n_pos, 0), prev_col(n_pos, 0);
    int64_t col = 1, prev = 0;
    for (int p : pos_idx) { col_of[p] = col; prev_col[p] = prev; prev = col; col += 1 + longest[p]; }
    const int64_t end_col = col;                                    // the column the loop would give the next position
    std::vector<size_t> order(tracks.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return spans[a].first < spans[b].first; });
    const int n_rows = (int)tracks.size();
    const int n_cols = (int)end_col + 1;
    std::vector<uint8_t> img((size_t)n_rows * n_cols, 0), qimg((size_t)n_rows * n_cols, 0), simg((size_t)n_rows * n_cols, 0);
    for (int row = 0; row < n_rows; ++row) {
        Rec* t = tracks[order[row]];
        const int64_t lo = spans[order[row]].first, hi = spans[order[row]].second;
        const int a = (int)(lo - t->start), nb = (int)(hi - lo);
        uint8_t* I = &img[(size_t)row * n_cols];
        uint8_t* Q = &qimg[(size_t)row * n_cols];
        uint8_t* S = &simg[(size_t)row * n_cols];
        const uint8_t strand = (t->flag & FREVERSE) ? STRAND_LOWER : STRAND_UPPER;
        const int ls = (int)t->seq.size(), lq = (int)t->qual.size();
        std::vector<uint8_t> quals(nb);
        for (int k = 0; k < nb; ++k) {
            const int qp = t->qpos[a + k];
            const bool del = t->is_del[a + k] != 0;
            const uint8_t ch = qp < ls ? (uint8_t)t->seq[qp] : (uint8_t)'N';
            if (!T.known[ch]) return 2;
            const int64_t c = col_of[lo - s0 + k];
            I[c] = del ? TOK_GAP : T.token[ch];
            quals[k] = qp < lq ? t->qual[qp] : 0;
            Q[c] = quals[k];
            S[c] = del ? STRAND_PAD : strand;
        }
        if (t->start >= s0) {                                       // head column inside the window
            const int64_t pc = prev_col[lo - s0];
            I[pc] = START; Q[pc] = quals[0]; S[pc] = t->is_del[a] ? STRAND_PAD : strand;
        }
        for (int k = 0; k < nb; ++k) {
            const int lg = longest[lo - s0 + k];
            if (lg <= 0) continue;
            const int64_t c0 = col_of[lo - s0 + k] + 1;
            const uint8_t st = t->is_del[a + k] ? STRAND_PAD : strand;
            const int ins = t->indel[a + k];
            if (ins > 0) {
                const int n_ins = std::min(ins, cap[lo - s0 + k]);
                const int q0 = t->qpos[a + k];
                for (int j = 1; j <= n_ins; ++j) {
                    const uint8_t ch = q0 + j < ls ? (uint8_t)t->seq[q0 + j] : (uint8_t)'N';
                    if (!T.known[ch]) return 2;
                    I[c0 + j - 1] = T.token[ch];
                    Q[c0 + j - 1] = quals[k];
                    S[c0 + j - 1] = st;
                }
            }
            for (int j = 0; j < lg; ++j) if (I[c0 + j] == PAD) I[c0 + j] = NOINSERT;
        }
        if (t->end <= stop) {                                       // tail column inside the window
            const int k = nb - 1;
            const int64_t e = col_of[lo - s0 + k] + longest[lo - s0 + k] + 1;
            I[e] = END; Q[e] = quals[k]; S[e] = t->is_del[a + k] ? STRAND_PAD : strand;
        }
        // deletions carry no strand: the row's own strand, forward when it has none (:1063-1078)
        uint8_t best = 0;
        bool any_pad = false;
        for (int c = 0; c < n_cols; ++c) { if (S[c] == STRAND_PAD) any_pad = true; else best = std::max(best, S[c]); }
        if (any_pad) for (int c = 0; c < n_cols; ++c) if (S[c] == STRAND_PAD) S[c] = best ? best : STRAND_UPPER;
    }
    // ---- finish_record (process_locations_chunk :720-838)
    const int64_t center = col_of[ci];
    std::vector<uint8_t> ref_line(n_cols, TOK_GAP);
    for (int p : pos_idx) {
        const int64_t o = p;                                        // s0 + p - ref_start with ref_start = s0
        uint8_t tok = TOK_GAP;                                      // BASE_ENUM[""]
        if (o < (int64_t)ref.size()) {
            const uint8_t ch = (uint8_t)ref[o];
            if (!T.known[ch]) return 2;                              // (the Python path raises the reference's KeyError)
            tok = T.token[ch];
        }
        ref_line[col_of[p]] = tok;
    }
    const int64_t lo = std::max<int64_t>(0, center - w), hi = std::min<int64_t>(center + w + 1, n_cols);
    const int cw = (int)(hi - lo);
    auto first_nonzero_row = [&](const std::vector<uint8_t>& im) {
        for (int r = 0; r < n_rows; ++r) {
            const uint8_t* p = &im[(size_t)r * n_cols + lo];
            for (int c = 0; c < cw; ++c) if (p[c]) return r;
        }
        return 0;                                                    // all rows zero: nothing is trimmed
    };
    const int fb = first_nonzero_row(img), fq = first_nonzero_row(qimg), fs = first_nonzero_row(simg);
    const int nbr = n_rows - fb, nq = n_rows - fq, ns = n_rows - fs;
    const int first = std::max(0, (nbr - MR) / 2);                  // int((n - max_reads) / 2) for n >= max_reads, else 0
    const int last = std::min(first + MR, nbr);
    auto count = [&](int n) { return std::max(0, std::min(last, n) - std::min(first, n)); };
    const int kb = count(nbr), kq = count(nq), ks = count(ns);
    if (kq != kb || ks != kb || kb == 0) return 0;
    const int off = w - (int)(center - lo);
    const int k = std::min(MR, kb);
    memset(reads_out, 0, (size_t)MR * W); memset(qual_out, 0, (size_t)MR * W); memset(strand_out, 0, (size_t)MR * W);
    memset(ref_out, 0, (size_t)W);
    for (int r = 0; r < k; ++r) {
        memcpy(reads_out + (size_t)r * W + off, &img[(size_t)(fb + first + r) * n_cols + lo], cw);
        memcpy(qual_out + (size_t)r * W + off, &qimg[(size_t)(fq + first + r) * n_cols + lo], cw);
        memcpy(strand_out + (size_t)r * W + off, &simg[(size_t)(fs + first + r) * n_cols + lo], cw);
    }
    memcpy(ref_out + off, &ref_line[lo], cw);
    *num_reads = k;
    return 1;
}

}  // namespace

extern "C" {

const char* pe_last_error(const pe_encoder_t* e) { return e ? e->err.c_str() : g_pe_error.c_str(); }

int pe_open(const char* bam_path, const char* bai_path, const char* fasta_path, const pe_options* opt, pe_encoder_t** out) {
    if (!bam_path || !fasta_path || !opt || !out) return pe_fail(nullptr, "pe_open: null argument");
    if (opt->window_size < 1 || opt->max_reads < 1) return pe_fail(nullptr, "pe_open: window_size and max_reads must be positive");
    auto e = std::make_unique<pe_encoder>();
    e->bam_path = bam_path; e->fasta_path = fasta_path; e->opt = *opt;
    {   // both files must open (per-thread handles are opened in pe_encode)
        Bam b;
        if (!b.open(bam_path)) return pe_fail(nullptr, "%s", b.err.c_str());
        Fasta f;
        std::string err;
        if (!f.open(fasta_path, err)) return pe_fail(nullptr, "%s", err.c_str());
    }
    std::vector<std::string> cands;
    if (bai_path && *bai_path) cands.push_back(bai_path);
    else {
        cands.push_back(std::string(bam_path) + ".bai");
        const std::string p(bam_path);
        const size_t dot = p.find_last_of('.'), slash = p.find_last_of('/');
        if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) cands.push_back(p.substr(0, dot) + ".bai");
    }
    for (auto& c : cands) if (e->bai.load(c)) { e->have_bai = true; e->bai_path = c; break; }
    *out = e.release();
    return 0;
}

int pe_encode(pe_encoder_t* e, const char* const* contigs, const int32_t* positions, int64_t n, uint8_t* reads_out, uint8_t* qual_out,
              uint8_t* strand_out, uint8_t* ref_out, int32_t* num_reads_out, int8_t* status_out, int32_t threads) {
    if (!e || (n > 0 && (!contigs || !positions || !reads_out || !qual_out || !strand_out || !ref_out || !num_reads_out || !status_out)))
        return pe_fail(e, "pe_encode: null argument");
    const int W = 2 * e->opt.window_size + 1, MR = e->opt.max_reads;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads > 0 ? threads : 1, n));
    std::vector<std::string> errors((size_t)nt);
    auto work = [&](int ti) {
      try {                                                  // (an exception leaving a std::thread terminates the process)
        const int64_t lo = n * ti / nt, hi = n * (ti + 1) / nt;
        Bam bam;
        Fasta fasta;
        std::string err;
        if (!bam.open(e->bam_path)) { errors[ti] = bam.err; return; }
        if (!fasta.open(e->fasta_path, err)) { errors[ti] = err; return; }
        Window win;
        win.bam = &bam; win.bai = e->have_bai ? &e->bai : nullptr;
        for (int64_t i = lo; i < hi; ++i) {
            const int st = encode_one(e->opt, win, bam, fasta, contigs[i], positions[i], reads_out + (size_t)i * MR * W,
                                      qual_out + (size_t)i * MR * W, strand_out + (size_t)i * MR * W, ref_out + (size_t)i * W,
                                      num_reads_out + i, err);
            if (st < 0) { errors[ti] = err.empty() ? "read error" : err; break; }
            status_out[i] = (int8_t)st;
        }
      } catch (const std::exception& ex) {
        errors[ti] = std::string("pileup encoder: ") + ex.what();
      } catch (...) {
        errors[ti] = "pileup encoder: unknown exception";
      }
    };
    std::vector<std::thread> pool;
    for (int ti = 1; ti < nt; ++ti) pool.emplace_back(work, ti);
    work(0);
    for (auto& t : pool) t.join();
    for (auto& s : errors) if (!s.empty()) return pe_fail(e, "%s", s.c_str());
    return 0;
}

void pe_close(pe_encoder_t* e) { delete e; }

}  // extern "C"
