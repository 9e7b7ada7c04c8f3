/*
 * dl4vc_loader.h -- C ABI of the native batched candidate loader (libdl4vc_loader.so).
 *
 * Replaces, for the inference path, the reference's per-item data pipeline:
 *   DataLoader(ContextDatasetFromNumpy(test_file), batch_size, shuffle=False, num_workers=5)   main.py:86-94
 *   ContextDatasetFromNumpy.__getitem__ / _get_generator                                      dl4vc/dataset.py:494-680
 *   sample_single_reads (row subset)                                                           dl4vc/dataset.py:256-287
 *   get_read_mask_vectors (allele masks) + the blacklist fallback                              dl4vc/dataset.py:112-250, 644-663
 * It reads the HDF5 file the converter writes (dataset "data", packed compound records,
 * tools/convert_bam_single_reads.py:659,694-698) and delivers batches of the six uint8 planes that
 * dan_forward() consumes, in record order.  Host-only C++; libhdf5 is dlopen'ed at run time.
 * Every function returns 0 / a count on success and a negative value on error (text: dl_last_error).
 */
#ifndef DL4VC_LOADER_H
#define DL4VC_LOADER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dl_loader dl_loader_t;

/* Open `hdf_path` for records [lo, hi) (hi < 0 = to the end).  `reads` = R rows per site (max_reads of the
 * dataset class, dl4vc/dataset.py:409); `batch_sites` = sites per dl_next() call.  Pileups that store more than R
 * reads get a sorted random subset drawn like numpy's legacy RandomState seeded with (seed + absolute record
 * index) -- exactly np.random.seed(s); np.random.random(); np.random.choice(...) of the reference (dataset.py:274-281,
 * 531); with use_seed = 0 such a record is an error instead of an unpinned draw.  `threads` worker threads inflate
 * and assemble 128-site pieces; at most `prefetch` batches are prepared ahead.  `libhdf5_path` may be NULL/"". */
int dl_open(const char* hdf_path, const char* libhdf5_path, int32_t reads, int64_t lo, int64_t hi, int32_t batch_sites,
            uint64_t seed, int32_t use_seed, int32_t threads, int32_t prefetch, dl_loader_t** out);
int64_t dl_num_records(const dl_loader_t* l);          /* len(dataset), dl4vc/dataset.py:489-491 */
int64_t dl_num_sites(const dl_loader_t* l);            /* hi - lo */
int32_t dl_window(const dl_loader_t* l);               /* columns per read (201) */
/* Next batch in record order into caller buffers (any may be NULL): reads/qual/strand [n][R][L], ref/ref_mask/
 * var_mask [n][L], vcfrec [n][129] NUL-terminated, num_reads [n], blacklist [n].  Returns n (0 at the end). */
int64_t dl_next(dl_loader_t* l, uint8_t* reads, uint8_t* qual, uint8_t* strand, uint8_t* ref, uint8_t* ref_mask,
                uint8_t* var_mask, char* vcfrec, int32_t* num_reads, uint8_t* blacklist);
void dl_close(dl_loader_t* l);
const char* dl_last_error(const dl_loader_t* l);       /* l may be NULL: error of the last failed dl_open */

/* Exposed for parity tests: the subset draw and the allele masks on their own.
 * dl_select_rows: rows_out gets the chosen stored rows, returns their count.
 * dl_allele_masks: 0 = ok, 1 = blacklisted (the reference's AssertionError cases: zero masks), 2 = fatal (the
 * reference dies with a non-assert exception). */
int dl_select_rows(uint32_t seed, int32_t num_reads, int32_t stored_rows, int32_t max_reads, int32_t* rows_out);
int dl_allele_masks(const char* vcfrec, const uint8_t* window /*[201]*/, uint8_t* ref_mask, uint8_t* var_mask);

/* ---- native pileup encoder -------------------------------------------------------------
 * Replaces the per-location work of the reference's converter, which it spreads over 80 pysam processes:
 *   process_location                      tools/convert_bam_single_reads.py:846-1118   (pileup columns -> three image planes)
 *   process_locations_chunk, image part   tools/convert_bam_single_reads.py:720-838    (crop / trim / centre / pad -> record)
 *   samfile.pileup(...) / fetch           tools/convert_bam_single_reads.py:873-874    (BGZF, BAM records, BAI, CIGAR walk)
 * pe_encode fills, for each location (contig name, 1-based VCF POS), the image fields of the converter's record
 * (:694-698): single_reads / q-scores / strand [max_reads][2 w + 1], ref_bases [2 w + 1], num_reads; name / label / vcfrec are
 * text the caller already has.  status: 1 = planes written, 0 = the location yields no record (no read over the candidate's
 * column; the reference counts an error), 2 = the location needs the column-by-column form (reads sharing a name:sequence
 * key, a reference skip, a base outside the token table, > 1000 columns, > 8000 reads, min_base_quality > 0): the Python
 * module encodes those, so results are identical to dl4vc_amd/pileup_encoder.py for every location.  `threads` workers take
 * contiguous runs of locations (locations sorted by position keep every alignment parsed once per run). */
typedef struct pe_encoder pe_encoder_t;
typedef struct pe_options {
    int32_t window_size;                 /* --window-size (100) */
    int32_t max_reads;                   /* --max-reads (call_variants.sh: 200) */
    int32_t max_insert_length;           /* --max-insert-length (10) */
    int32_t max_insert_length_variant;   /* --max-insert-length-variant (50) */
    int32_t min_base_quality;            /* --min-base-quality (0) */
} pe_options;
int pe_open(const char* bam_path, const char* bai_path /* NULL: <bam>.bai, <stem>.bai, else a linear scan */, const char* fasta_path,
            const pe_options* opt, pe_encoder_t** out);
int pe_encode(pe_encoder_t* e, const char* const* contigs, const int32_t* positions, int64_t n, uint8_t* reads_out, uint8_t* qual_out,
              uint8_t* strand_out, uint8_t* ref_out, int32_t* num_reads_out, int8_t* status_out, int32_t threads);
void pe_close(pe_encoder_t* e);
const char* pe_last_error(const pe_encoder_t* e);      /* e may be NULL: error of the last failed pe_open */

#ifdef __cplusplus
}
#endif
#endif /* DL4VC_LOADER_H */
