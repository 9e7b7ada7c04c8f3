// libdl4vc_loader.so -- native batched candidate loader.
//
// Replaces the reference's per-item Python path (dl4vc/dataset.py:494-680: lazily opened h5py generator,
// one gzip-chunked 124 KB compound record per __getitem__, 5 DataLoader worker processes) with a C++
// reader that delivers whole batches of the six uint8 planes the device consumes:
//   * raw chunk reads (H5Dread_chunk, serialised: libhdf5 is not thread-safe) + zlib inflate and site
//     assembly in parallel worker threads, with a bounded in-order prefetch ring;
//   * row selection of sample_single_reads (dataset.py:256-287): first R rows, or for pileups deeper than R
//     a sorted random subset drawn with numpy's legacy RandomState algorithms (MT19937, random_sample,
//     permutation) seeded with (seed + absolute record index) -- bit-compatible with dl4vc_amd/dataset.py;
//   * allele mask vectors of get_read_mask_vectors (dataset.py:112-250) incl. the blacklist fallback.
// libhdf5 is dlopen'ed at run time (no link-time dependency); plain C ABI below.
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dl4vc_loader.h"

namespace {

typedef int64_t hid_t;
typedef uint64_t hsize_t;
typedef int herr_t;

struct H5 {
    void* lib = nullptr;
    herr_t (*open)() = nullptr;
    hid_t (*Fopen)(const char*, unsigned, hid_t) = nullptr;
    herr_t (*Fclose)(hid_t) = nullptr;
    hid_t (*Dopen2)(hid_t, const char*, hid_t) = nullptr;
    herr_t (*Dclose)(hid_t) = nullptr;
    hid_t (*Dget_space)(hid_t) = nullptr;
    hid_t (*Dget_type)(hid_t) = nullptr;
    hid_t (*Dget_create_plist)(hid_t) = nullptr;
    int (*Sget_simple_extent_dims)(hid_t, hsize_t*, hsize_t*) = nullptr;
    herr_t (*Sselect_hyperslab)(hid_t, int, const hsize_t*, const hsize_t*, const hsize_t*, const hsize_t*) = nullptr;
    hid_t (*Screate_simple)(int, const hsize_t*, const hsize_t*) = nullptr;
    herr_t (*Sclose)(hid_t) = nullptr;
    herr_t (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*) = nullptr;
    herr_t (*Dread_chunk)(hid_t, hid_t, const hsize_t*, uint32_t*, void*) = nullptr;
    herr_t (*Dget_chunk_storage_size)(hid_t, const hsize_t*, hsize_t*) = nullptr;
    size_t (*Tget_size)(hid_t) = nullptr;
    herr_t (*Tclose)(hid_t) = nullptr;
    int (*Tget_member_index)(hid_t, const char*) = nullptr;
    size_t (*Tget_member_offset)(hid_t, unsigned) = nullptr;
    int (*Pget_layout)(hid_t) = nullptr;
    int (*Pget_chunk)(hid_t, int, hsize_t*) = nullptr;
    int (*Pget_nfilters)(hid_t) = nullptr;
    int (*Pget_filter2)(hid_t, unsigned, unsigned*, size_t*, unsigned*, size_t, char*, unsigned*) = nullptr;
    herr_t (*Pclose)(hid_t) = nullptr;
};

std::string g_open_error;

template <typename F>
bool sym(void* lib, const char* name, F& f) {
    f = reinterpret_cast<F>(dlsym(lib, name));
    return f != nullptr;
}

bool load_h5(H5& h, const char* path, std::string& err) {
    const char* cands[] = {path && *path ? path : nullptr, "/opt/conda/lib/libhdf5.so", "libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103"};
    for (const char* c : cands) {
        if (!c) continue;
        h.lib = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (h.lib) break;
    }
    if (!h.lib) { err = "cannot dlopen libhdf5 (set DL4VC_LIBHDF5)"; return false; }
    bool ok = sym(h.lib, "H5open", h.open) && sym(h.lib, "H5Fopen", h.Fopen) && sym(h.lib, "H5Fclose", h.Fclose) &&
              sym(h.lib, "H5Dopen2", h.Dopen2) && sym(h.lib, "H5Dclose", h.Dclose) && sym(h.lib, "H5Dget_space", h.Dget_space) &&
              sym(h.lib, "H5Dget_type", h.Dget_type) && sym(h.lib, "H5Dget_create_plist", h.Dget_create_plist) &&
              sym(h.lib, "H5Sget_simple_extent_dims", h.Sget_simple_extent_dims) &&
              sym(h.lib, "H5Sselect_hyperslab", h.Sselect_hyperslab) && sym(h.lib, "H5Screate_simple", h.Screate_simple) &&
              sym(h.lib, "H5Sclose", h.Sclose) && sym(h.lib, "H5Dread", h.Dread) && sym(h.lib, "H5Tget_size", h.Tget_size) &&
              sym(h.lib, "H5Tclose", h.Tclose) && sym(h.lib, "H5Tget_member_index", h.Tget_member_index) &&
              sym(h.lib, "H5Tget_member_offset", h.Tget_member_offset) && sym(h.lib, "H5Pget_layout", h.Pget_layout) &&
              sym(h.lib, "H5Pget_chunk", h.Pget_chunk) && sym(h.lib, "H5Pget_nfilters", h.Pget_nfilters) &&
              sym(h.lib, "H5Pget_filter2", h.Pget_filter2) && sym(h.lib, "H5Pclose", h.Pclose);
    if (!ok) { err = "libhdf5 lacks a required symbol"; return false; }
    sym(h.lib, "H5Dread_chunk", h.Dread_chunk);                           // optional (>= 1.10.3)
    sym(h.lib, "H5Dget_chunk_storage_size", h.Dget_chunk_storage_size);
    h.open();
    return true;
}

// ---------------------------------------------------------------------------------------------
// numpy legacy RandomState (MT19937) -- only what sample_single_reads needs
// ---------------------------------------------------------------------------------------------
struct MT {
    uint32_t mt[624];
    int idx = 624;
    explicit MT(uint32_t seed) {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    }
    uint32_t next32() {
        if (idx >= 624) {
            for (int k = 0; k < 624; ++k) {
                uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    double random_sample() { uint32_t a = next32() >> 5, b = next32() >> 6; return (a * 67108864.0 + b) / 9007199254740992.0; }
    uint32_t interval(uint32_t max) {          // numpy random_interval for max <= 0xffffffff
        if (max == 0) return 0;
        uint32_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        while ((v = (next32() & mask)) > max) {}
        return v;
    }
};

// rows of sample_single_reads(random=True), dataset.py:256-287.  Returns count written, -1 if a seed is needed.
int select_rows(bool have_seed, uint32_t seed, int num_reads, int stored_rows, int max_reads, int32_t* rows) {
    max_reads = std::min(max_reads, stored_rows);
    if (max_reads >= num_reads) {
        for (int i = 0; i < max_reads; ++i) rows[i] = i;
        return max_reads;
    }
    if (!have_seed) return -1;
    MT rng(seed);
    rng.random_sample();                                       // the disabled dynamic-down-sampling coin, dataset.py:531
    const int pool = std::min(stored_rows, num_reads);
    std::vector<int32_t> perm(pool);
    for (int i = 0; i < pool; ++i) perm[i] = i;
    for (int i = pool - 1; i >= 1; --i) {                      // legacy shuffle
        const uint32_t j = rng.interval((uint32_t)i);
        std::swap(perm[i], perm[j]);
    }
    const int k = std::min(max_reads, num_reads);
    std::sort(perm.begin(), perm.begin() + k);
    for (int i = 0; i < k; ++i) rows[i] = perm[i];
    return k;
}

// ---------------------------------------------------------------------------------------------
// allele masks (dl4vc_amd/alleles.py, reference dl4vc/dataset.py:86-250)
// ---------------------------------------------------------------------------------------------
enum { PAD = 0, TOK_A = 1, TOK_T = 2, TOK_G = 3, TOK_C = 4, GAP = 5, END_T = 7, NOINSERT = 8, UNK = 9 };
enum { MASK_OK = 0, MASK_BLACKLIST = 1, MASK_FATAL = 2 };

int token_of(char c) {
    switch (c) {
        case 'A': case 'a': return TOK_A;
        case 'T': case 't': case 'U': case 'u': return TOK_T;
        case 'G': case 'g': return TOK_G;
        case 'C': case 'c': return TOK_C;
        case '-': case '*': case 'N': case 'n': case 'X': case 'x': case '.': return GAP;
        case 'e': return END_T;
        default: break;
    }
    if (strchr("?MmKkRrYySsWwBbVvHhDd", c) && c) return UNK;
    return -1;                                                  // KeyError in the reference
}

bool snp_char(const std::string& s) { return s.size() == 1 && strchr("AaTtCcG", s[0]) != nullptr; }

std::vector<std::string> split_tabs(const char* rec) {
    std::string s(rec);
    while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' ')) s.pop_back();
    size_t b = 0;
    while (b < s.size() && (s[b] == ' ' || s[b] == '\n')) ++b;
    std::vector<std::string> out;
    size_t p = b;
    for (;;) {
        size_t q = s.find('\t', p);
        out.push_back(s.substr(p, q == std::string::npos ? std::string::npos : q - p));
        if (q == std::string::npos) break;
        p = q + 1;
    }
    return out;
}

int allele_masks(const char* vcfrec, const uint8_t* window, int L, uint8_t* ref_mask, uint8_t* var_mask, std::string& why) {
    memset(ref_mask, 0, L);
    memset(var_mask, 0, L);
    if (L != 201) { why = "allele masks assume a 201-column window"; return MASK_BLACKLIST; }
    std::vector<std::string> f = split_tabs(vcfrec);
    if (f.size() < 8) { why = "VCF record has fewer than 8 columns"; return MASK_FATAL; }
    const std::string &ref_s = f[3], alt_full = f[4];
    const std::string alt_s = alt_full.substr(0, 51);          // insert_limit = VAR_ENCODE_LEN, dataset.py:85-93
    std::vector<int> ref_v, alt_v;
    for (char c : ref_s) { int t = token_of(c); if (t < 0) { why = std::string("unknown allele character '") + c + "'"; return MASK_FATAL; } ref_v.push_back(t); }
    for (char c : alt_s) { int t = token_of(c); if (t < 0) { why = std::string("unknown allele character '") + c + "'"; return MASK_FATAL; } alt_v.push_back(t); }
    // parse_vcf needs AF and DP in INFO (utils.py:52-56)
    if (f[7].find("AF=") == std::string::npos || f[7].find("DP=") == std::string::npos) { why = "INFO lacks AF/DP"; return MASK_FATAL; }
    const bool is_snp = ref_s.size() == 1 && alt_full.size() == 1 && snp_char(ref_s) && snp_char(alt_full);
    int off = 100;
    while (off > 0 && window[off] == GAP) --off;
    if (is_snp) {
    } else if (ref_s.size() > alt_full.size()) {
        if (alt_full.size() != 1) { why = "For deletes, expect exactly one base in variant"; return MASK_BLACKLIST; }
    } else if (alt_full.size() > ref_s.size()) {
        if (ref_s.size() != 1) { why = "For inserts, expect exactly one base in reference"; return MASK_BLACKLIST; }
    } else {
        why = "allele pair " + ref_s + " -> " + alt_full + " is neither SNP, insert nor delete";
        return MASK_FATAL;                                      // UnboundLocalError in the reference
    }
    if (ref_v.empty() || window[off] != ref_v[0]) { why = "Did not find (first) ref base in reference"; return MASK_BLACKLIST; }
    if (ref_v.size() > 1) {
        while (alt_v.size() < ref_v.size()) alt_v.push_back(GAP);
        bool same = off + (int)ref_v.size() <= L;
        for (size_t i = 0; same && i < ref_v.size(); ++i) same = window[off + i] == ref_v[i];
        if (!same) {
            std::vector<int> nr, na;
            size_t k = 0;
            for (int col = off; col < L; ++col) {
                if (k >= ref_v.size()) break;
                if (window[col] == ref_v[k]) { nr.push_back(window[col]); na.push_back(alt_v[k]); ++k; }
                else if (window[col] == GAP) { nr.push_back(GAP); na.push_back(NOINSERT); }
                else { why = "Mis-match inserting pad delete into reference"; return MASK_BLACKLIST; }
            }
            if (k < ref_v.size()) { why = "Finished padding, did not reach end of pad insert"; return MASK_BLACKLIST; }
            for (auto& t : nr) if (t == GAP) t = PAD;
            for (auto& t : na) if (t == NOINSERT) t = PAD;
            ref_v.swap(nr); alt_v.swap(na);
        }
    }
    if (ref_v.size() == 1 && alt_v.size() > 1) ref_v.resize(alt_v.size(), NOINSERT);
    if (ref_v.size() != alt_v.size()) { why = "Need to adjust ref, var vectors for same length"; return MASK_BLACKLIST; }
    if (off + (int)ref_v.size() > L) { why = "allele span leaves the window"; return MASK_FATAL; }   // numpy broadcast ValueError
    for (size_t i = 0; i < ref_v.size(); ++i) { ref_mask[off + i] = (uint8_t)ref_v[i]; var_mask[off + i] = (uint8_t)alt_v[i]; }
    return MASK_OK;
}

// ---------------------------------------------------------------------------------------------
struct Batch {
    int64_t first = 0;
    int n = 0;
    std::vector<uint8_t> reads, qual, strand, ref, rmask, vmask, black;
    std::vector<char> vcf;
    std::vector<int32_t> nreads;
    std::string error;
};

}  // namespace

struct dl_loader {
    H5 h5;
    hid_t fid = -1, did = -1, tid = -1;
    int64_t n_records = 0, lo = 0, hi = 0;
    size_t itemsize = 0;
    size_t off_reads = 0, off_ref = 0, off_num = 0, off_vcf = 0, off_q = 0, off_strand = 0;
    int store_rows = 0, L = 201, R = 100;
    int batch = 0;                  // sites per dl_next call
    int piece = 0;                  // sites per worker task (divides batch): parallelism is per piece, not per batch
    uint64_t seed = 0;
    bool use_seed = false;
    int64_t chunk_records = 0;      // 0 = not chunked (or unsupported filter): hyperslab reads
    bool deflate = false;
    mutable std::string err;
    // pipeline
    std::mutex h5_mutex, q_mutex;
    std::condition_variable cv_done, cv_space;
    std::vector<std::thread> workers;
    std::map<int64_t, std::unique_ptr<Batch>> ready;
    int64_t next_issue = 0, next_deliver = 0, n_batches = 0;
    int prefetch = 4;
    bool stop = false;
};

namespace {

bool read_records(dl_loader* l, int64_t lo, int64_t hi, std::vector<uint8_t>& out, std::string& err) {
    const int64_t n = hi - lo;
    out.resize((size_t)n * l->itemsize);
    if (n <= 0) return true;
    H5& h = l->h5;
    if (l->chunk_records > 0 && h.Dread_chunk && h.Dget_chunk_storage_size) {
        const int64_t c = l->chunk_records;
        std::vector<uint8_t> raw, plain((size_t)c * l->itemsize);
        for (int64_t ch = lo / c; ch * c < hi; ++ch) {
            hsize_t offset[1] = {(hsize_t)(ch * c)}, nbytes = 0;
            uint32_t mask = 0;
            {
                std::lock_guard<std::mutex> g(l->h5_mutex);
                if (h.Dget_chunk_storage_size(l->did, offset, &nbytes) < 0) { err = "H5Dget_chunk_storage_size failed"; return false; }
                raw.resize(nbytes);
                if (h.Dread_chunk(l->did, 0, offset, &mask, raw.data()) < 0) { err = "H5Dread_chunk failed"; return false; }
            }
            const uint8_t* src = raw.data();
            if (l->deflate && !(mask & 1u)) {                   // inflate outside the lock: this is the parallel part
                uLongf dlen = plain.size();
                if (uncompress(plain.data(), &dlen, raw.data(), (uLong)nbytes) != Z_OK) { err = "zlib inflate failed"; return false; }
                src = plain.data();
            }
            const int64_t c_lo = ch * c, a = std::max(lo, c_lo), b = std::min(hi, c_lo + c);
            memcpy(out.data() + (size_t)(a - lo) * l->itemsize, src + (size_t)(a - c_lo) * l->itemsize, (size_t)(b - a) * l->itemsize);
        }
        return true;
    }
    std::lock_guard<std::mutex> g(l->h5_mutex);
    hid_t fs = h.Dget_space(l->did);
    hsize_t start[1] = {(hsize_t)lo}, count[1] = {(hsize_t)n};
    h.Sselect_hyperslab(fs, 0, start, nullptr, count, nullptr);
    hid_t ms = h.Screate_simple(1, count, nullptr);
    herr_t rc = h.Dread(l->did, l->tid, ms, fs, 0, out.data());
    h.Sclose(ms);
    h.Sclose(fs);
    if (rc < 0) { err = "H5Dread failed"; return false; }
    return true;
}

std::unique_ptr<Batch> build_batch(dl_loader* l, int64_t b) {
    std::unique_ptr<Batch> out(new Batch());
    const int64_t first = l->lo + b * l->piece, last = std::min(l->hi, first + l->piece);
    const int n = (int)(last - first), R = l->R, L = l->L;
    out->first = first; out->n = n;
    std::vector<uint8_t> recs;
    if (!read_records(l, first, last, recs, out->error)) return out;
    const size_t rl = (size_t)R * L;
    out->reads.assign(n * rl, 0); out->qual.assign(n * rl, 0); out->strand.assign(n * rl, 0);
    out->ref.resize((size_t)n * L); out->rmask.resize((size_t)n * L); out->vmask.resize((size_t)n * L);
    out->black.assign(n, 0); out->vcf.assign((size_t)n * 129, 0); out->nreads.assign(n, 0);
    std::vector<int32_t> rows(std::max(R, l->store_rows));
    for (int i = 0; i < n; ++i) {
        const uint8_t* rec = recs.data() + (size_t)i * l->itemsize;
        int32_t num_reads;
        memcpy(&num_reads, rec + l->off_num, 4);
        out->nreads[i] = num_reads;
        const int store = l->store_rows;
        const int mid = std::max(num_reads, store) / 2;
        const int start = std::max(0, mid - store / 2);                  // dataset.py:517-518
        const int avail = std::max(0, store - start);
        const int k = select_rows(l->use_seed, (uint32_t)(l->seed + (uint64_t)(first + i)), num_reads, avail, R, rows.data());
        if (k < 0) {
            char buf[160];
            snprintf(buf, sizeof buf, "record %lld stores %d reads > %d: a seed is required to pin the read subset",
                     (long long)(first + i), num_reads, R);
            out->error = buf;
            return out;
        }
        if (k < R && num_reads > k) {
            // a pileup so deep that the stored window [start, start + store) runs past the stored rows (num_reads > 2 x
            // store): the reference yields a (L, k) item here and its DataLoader cannot collate it (dataset.py:517-521,
            // :270-281).  Same behaviour in the Python twin (dl4vc_amd/dataset.py::assemble_site): refuse, name the record.
            char buf[200];
            snprintf(buf, sizeof buf, "record %lld: num_reads %d leaves %d stored rows in the sampling window (< %d reads): "
                     "the reference cannot batch this site either", (long long)(first + i), num_reads, k, R);
            out->error = buf;
            return out;
        }
        for (int r = 0; r < k; ++r) {
            const size_t src = (size_t)(start + rows[r]) * L, dst = (size_t)i * rl + (size_t)r * L;
            memcpy(&out->reads[dst], rec + l->off_reads + src, L);
            memcpy(&out->qual[dst], rec + l->off_q + src, L);
            memcpy(&out->strand[dst], rec + l->off_strand + src, L);
        }
        memcpy(&out->ref[(size_t)i * L], rec + l->off_ref, L);
        char* v = &out->vcf[(size_t)i * 129];
        memcpy(v, rec + l->off_vcf, 128);
        v[128] = 0;
        std::string why;
        const int st = allele_masks(v, &out->ref[(size_t)i * L], L, &out->rmask[(size_t)i * L], &out->vmask[(size_t)i * L], why);
        if (st == MASK_FATAL) {
            out->error = "record " + std::to_string(first + i) + ": " + why;
            return out;
        }
        out->black[i] = st == MASK_BLACKLIST;
    }
    return out;
}

void worker_main(dl_loader* l) {
    for (;;) {
        int64_t b;
        {
            std::unique_lock<std::mutex> g(l->q_mutex);
            l->cv_space.wait(g, [&] { return l->stop || (l->next_issue < l->n_batches && l->next_issue < l->next_deliver + l->prefetch); });
            if (l->stop || l->next_issue >= l->n_batches) return;
            b = l->next_issue++;
        }
        std::unique_ptr<Batch> res = build_batch(l, b);
        {
            std::lock_guard<std::mutex> g(l->q_mutex);
            l->ready[b] = std::move(res);
        }
        l->cv_done.notify_all();
    }
}

int fail_open(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_open_error = buf;
    return -1;
}

}  // namespace

extern "C" {

const char* dl_last_error(const dl_loader_t* l) { return l ? l->err.c_str() : g_open_error.c_str(); }

int dl_select_rows(uint32_t seed, int32_t num_reads, int32_t stored_rows, int32_t max_reads, int32_t* rows_out) {
    return select_rows(true, seed, num_reads, stored_rows, max_reads, rows_out);
}

int dl_allele_masks(const char* vcfrec, const uint8_t* window, uint8_t* ref_mask, uint8_t* var_mask) {
    std::string why;
    return allele_masks(vcfrec, window, 201, ref_mask, var_mask, why);
}

int dl_open(const char* path, const char* libhdf5_path, int32_t reads, int64_t lo, int64_t hi, int32_t batch_sites,
            uint64_t seed, int32_t use_seed, int32_t threads, int32_t prefetch, dl_loader_t** out) {
    if (!path || !out || reads < 1 || batch_sites < 1) return fail_open("dl_open: bad argument");
    *out = nullptr;
    std::unique_ptr<dl_loader> l(new dl_loader());
    std::string err;
    if (!load_h5(l->h5, libhdf5_path, err)) return fail_open("%s", err.c_str());
    H5& h = l->h5;
    l->fid = h.Fopen(path, 0, 0);
    if (l->fid < 0) return fail_open("cannot open %s as HDF5", path);
    l->did = h.Dopen2(l->fid, "data", 0);
    if (l->did < 0) { h.Fclose(l->fid); return fail_open("%s has no dataset 'data'", path); }
    l->tid = h.Dget_type(l->did);
    // every error return below closes the type / dataset / file handles (the dlopen'ed libhdf5 is reference-counted by
    // the dynamic loader and deliberately never dlclose'd: it registers atexit handlers)
    auto close_all = [&]() { if (l->tid >= 0) h.Tclose(l->tid); h.Dclose(l->did); h.Fclose(l->fid); l->tid = l->did = l->fid = -1; };
    l->itemsize = h.Tget_size(l->tid);
    hid_t sp = h.Dget_space(l->did);
    hsize_t dims[1] = {0};
    h.Sget_simple_extent_dims(sp, dims, nullptr);
    h.Sclose(sp);
    l->n_records = (int64_t)dims[0];
    auto off = [&](const char* name, size_t& dst) -> bool {
        int i = h.Tget_member_index(l->tid, name);
        if (i < 0) return false;
        dst = h.Tget_member_offset(l->tid, (unsigned)i);
        return true;
    };
    if (!(off("single_reads", l->off_reads) && off("ref_bases", l->off_ref) && off("num_reads", l->off_num) &&
          off("vcfrec", l->off_vcf) && off("q-scores", l->off_q) && off("strand", l->off_strand))) {
        close_all();
        return fail_open("%s: record type lacks a field of the converter schema", path);
    }
    l->L = 201;
    // packed layout: ref_bases follows single_reads (tools/convert_bam_single_reads.py:694-698)
    l->store_rows = (int)((l->off_ref - l->off_reads) / l->L);
    if (l->store_rows < 1 || (l->off_ref - l->off_reads) % l->L || l->off_strand - l->off_q != (size_t)l->store_rows * l->L) {
        close_all();
        return fail_open("%s: unexpected record layout", path);
    }
    hid_t pl = h.Dget_create_plist(l->did);
    if (pl >= 0) {
        if (h.Pget_layout(pl) == 2 /*H5D_CHUNKED*/) {
            hsize_t cd[1] = {0};
            h.Pget_chunk(pl, 1, cd);
            const int nf = h.Pget_nfilters(pl);
            bool supported = true;
            for (int i = 0; i < nf; ++i) {
                unsigned flags = 0, cfg = 0;
                size_t ne = 0;
                int id = h.Pget_filter2(pl, (unsigned)i, &flags, &ne, nullptr, 0, nullptr, &cfg);
                if (id == 1) l->deflate = true; else supported = false;
            }
            if (supported && nf <= 1 && cd[0] > 0) l->chunk_records = (int64_t)cd[0];
        }
        h.Pclose(pl);
    }
    l->R = reads;
    l->lo = std::max<int64_t>(0, lo);
    l->hi = hi < 0 ? l->n_records : std::min(hi, l->n_records);
    if (l->hi < l->lo) l->hi = l->lo;
    l->batch = batch_sites;
    l->piece = 1;
    for (int d = 1; d <= 128 && d <= batch_sites; ++d) if (batch_sites % d == 0) l->piece = d;
    l->seed = seed; l->use_seed = use_seed != 0;
    l->n_batches = (l->hi - l->lo + l->piece - 1) / l->piece;          // counted in pieces
    const int nt = std::max(1, std::min(threads, 64));
    l->prefetch = std::max(std::max(1, prefetch) * (batch_sites / l->piece), 2 * nt);
    dl_loader* raw = l.release();
    for (int i = 0; i < nt; ++i) raw->workers.emplace_back(worker_main, raw);
    *out = raw;
    return 0;
}

int64_t dl_num_records(const dl_loader_t* l) { return l ? l->n_records : -1; }
int64_t dl_num_sites(const dl_loader_t* l) { return l ? l->hi - l->lo : -1; }
int32_t dl_window(const dl_loader_t* l) { return l ? l->L : -1; }

int64_t dl_next(dl_loader_t* l, uint8_t* reads, uint8_t* qual, uint8_t* strand, uint8_t* ref, uint8_t* ref_mask,
                uint8_t* var_mask, char* vcfrec, int32_t* num_reads, uint8_t* blacklist) {
    if (!l) return -1;
    const size_t rl = (size_t)l->R * l->L;
    size_t filled = 0;
    while (filled < (size_t)l->batch && l->next_deliver < l->n_batches) {
        std::unique_ptr<Batch> b;
        {
            std::unique_lock<std::mutex> g(l->q_mutex);
            l->cv_done.wait(g, [&] { return l->ready.count(l->next_deliver) > 0; });
            b = std::move(l->ready[l->next_deliver]);
            l->ready.erase(l->next_deliver);
            l->next_deliver++;
        }
        l->cv_space.notify_all();
        if (!b->error.empty()) { l->err = b->error; return -2; }
        const size_t n = (size_t)b->n;
        if (reads) memcpy(reads + filled * rl, b->reads.data(), n * rl);
        if (qual) memcpy(qual + filled * rl, b->qual.data(), n * rl);
        if (strand) memcpy(strand + filled * rl, b->strand.data(), n * rl);
        if (ref) memcpy(ref + filled * l->L, b->ref.data(), n * l->L);
        if (ref_mask) memcpy(ref_mask + filled * l->L, b->rmask.data(), n * l->L);
        if (var_mask) memcpy(var_mask + filled * l->L, b->vmask.data(), n * l->L);
        if (vcfrec) memcpy(vcfrec + filled * 129, b->vcf.data(), n * 129);
        if (num_reads) memcpy(num_reads + filled, b->nreads.data(), n * sizeof(int32_t));
        if (blacklist) memcpy(blacklist + filled, b->black.data(), n);
        filled += n;
    }
    return (int64_t)filled;
}

void dl_close(dl_loader_t* l) {
    if (!l) return;
    {
        std::lock_guard<std::mutex> g(l->q_mutex);
        l->stop = true;
    }
    l->cv_space.notify_all();
    for (auto& t : l->workers) t.join();
    if (l->tid >= 0) l->h5.Tclose(l->tid);
    if (l->did >= 0) l->h5.Dclose(l->did);
    if (l->fid >= 0) l->h5.Fclose(l->fid);
    delete l;
}

}  // extern "C"
