/*
 * paffy_hip.h -- C-ABI of the MI355X (gfx950) PAF hot path.
 *
 * Boundary: the reference has no FFI; its hot path is the per-command loop
 *     while ((paf = paf_read_with_buffer(..)) != NULL) { <transform>; paf_check; paf_write_with_buffer; }
 * in impl/paf_{shatter,invert,trim,add_mismatches}.c and the whole-file body of
 * impl/paf_tile.c. Each entry point below replaces that loop for a whole batch of PAF text
 * at once; the host drivers (host/paffy_*.c, same `paffy <cmd>` CLI and flags) call nothing
 * else. Signatures are plain pointers and sizes: no HIP or torch types.
 *
 * Data stays text at the boundary (the PAF line grammar of impl/paf.c:317-389 is the wire
 * format between paffy processes), so a batch is: device pointer to '\n'-separated lines in,
 * device pointer to the output lines out, byte-exact to what the reference command(s) write.
 *
 * Two-phase protocol, mirroring paf_estimate_buffer_size -> realloc -> paf_write_to_buffer
 * (impl/paf.c:391-406): paffy_hip_plan() parses and transforms the batch and reports the exact
 * output size and the first failing record; paffy_hip_emit() writes the bytes.
 */
#ifndef PAFFY_HIP_H_
#define PAFFY_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct paffy_hip_ctx paffy_hip_ctx;

/* One stage = one reference command in a `paffy a | paffy b | ...` chain. */
enum {
    PAFFY_INVERT = 1,            /* impl/paf_invert.c:84-89: paf_invert, paf_check, write            */
    PAFFY_TRIM_IDENTITY = 2,     /* impl/paf_trim.c:117-119: paf_trim_unreliable_tails(p0=-r, p1=-t)  */
    PAFFY_TRIM_FIXED = 3,        /* impl/paf_trim.c:120-122: paf_trim_end_fraction(p1=-t)             */
    PAFFY_SHATTER = 4,           /* impl/paf_shatter.c:88-95: paf_shatter, write each block (last stage only) */
    PAFFY_ADD_MISMATCHES = 5,    /* impl/paf_add_mismatches.c:113-131: paf_encode_mismatches          */
    PAFFY_REMOVE_MISMATCHES = 6, /* impl/paf_add_mismatches.c:110-112: paf_remove_mismatches          */
    PAFFY_PASS = 7,              /* paf_read -> paf_write only (normalises tags, impl/paf.c:317-389)  */
    PAFFY_FILTER = 8,            /* impl/paf_filter.c:120-156: records failing the thresholds of paffy_hip_set_filter vanish */
    PAFFY_STATS = 10,            /* paf_stats_calc(.., zero_counts = 0) of every record into the plan's running sums (the aggregate of
                                    `paffy view -s`, impl/paf_view.c:163-168); the record passes on unchanged. See paffy_hip_plan_stats() */
    PAFFY_TRIM_ENDS = 9,         /* paf_trim_ends(paf, n), impl/paf.c:575-598: n aligned bases off each end; n = the 64 bits of (p0, p1),
                                    see paffy_stage_trim_ends(); then paf_check like the other trims */
    PAFFY_CHECK = 11             /* paf_check alone (impl/paf.c:427-461): the record passes on unchanged or fails */
};
/* OR-ed into a transform's kind: without the paf_check that the command loops run after it (impl/paf_invert.c:84-89) -- what the
 * library functions of inc/paf.h do (paf_invert, paf_trim_ends ... check nothing, impl/paf.c:463-598) */
#define PAFFY_NO_CHECK 0x100
#define PAFFY_MAX_STAGES 8

typedef struct {
    int32_t kind;
    float p0; /* trim: trim_by_identity_fraction, a float as in impl/paf_trim.c:16 (default 0.05) */
    float p1; /* trim: trim_end_fraction, a float as in impl/paf_trim.c:14 (default 1.0)          */
} paffy_stage;

/* PAFFY_TRIM_ENDS carries its int64 argument in the two float slots, bit for bit. */
static inline paffy_stage paffy_stage_trim_ends(int64_t end_bases) {
    paffy_stage s;
    union { int64_t i; float f[2]; } u;
    u.i = end_bases;
    s.kind = PAFFY_TRIM_ENDS;
    s.p0 = u.f[0];
    s.p1 = u.f[1];
    return s;
}

/* Record-level failures: what the reference turns into st_errAbort / assert / a crash. */
enum {
    PAFFY_OK = 0,
    PAFFY_ERR_FEW_FIELDS = 1,          /* impl/paf.c:144-172 (NULL token dereferenced)  */
    PAFFY_ERR_STRAND = 2,              /* impl/paf.c:155-157 st_errAbort                */
    PAFFY_ERR_TP_ASSERT = 3,           /* impl/paf.c:190 assert                         */
    PAFFY_ERR_CIGAR_CHAR = 4,          /* impl/paf.c:102 st_errAbort                    */
    PAFFY_ERR_CHECK_QSTART = 5,        /* impl/paf.c:428                                */
    PAFFY_ERR_CHECK_QEND = 6,          /* impl/paf.c:431                                */
    PAFFY_ERR_CHECK_TSTART = 7,        /* impl/paf.c:434                                */
    PAFFY_ERR_CHECK_TEND = 8,          /* impl/paf.c:437                                */
    PAFFY_ERR_CHECK_CIGAR_Q = 9,       /* impl/paf.c:452                                */
    PAFFY_ERR_CHECK_CIGAR_T = 10,      /* impl/paf.c:456                                */
    PAFFY_ERR_SHATTER_ZERO_LEN = 11,   /* impl/paf.c:635 assert                         */
    PAFFY_ERR_SHATTER_BAD_OP = 12,     /* impl/paf.c:650 assert                         */
    PAFFY_ERR_SHATTER_END = 13,        /* impl/paf.c:654-660 asserts                    */
    PAFFY_ERR_TRIM_IDENTITY_ASSERT = 14, /* impl/paf.c:952 assert                       */
    PAFFY_ERR_TRIM_FIXED_ASSERT = 15,  /* impl/paf.c:591 assert                         */
    PAFFY_ERR_NULL_CIGAR = 16,         /* impl/paf.c:520 (NULL cigar dereferenced)      */
    PAFFY_ERR_MISSING_QUERY_SEQ = 17,  /* impl/paf_add_mismatches.c:117-120 exit(1)     */
    PAFFY_ERR_MISSING_TARGET_SEQ = 18, /* impl/paf_add_mismatches.c:123-127 exit(1)     */
    PAFFY_ERR_TILE_ASSERT = 19,        /* impl/paf.c:685,698,708; impl/paf_tile.c:57,86,171 */
    PAFFY_ERR_SEQ_RANGE = 21,          /* paf_encode_mismatches would read outside a sequence */
    PAFFY_ERR_CHAIN_ASSERT = 22        /* impl/chaining.c:275,278-281 asserts              */
};

/* Call-level failures (negative return values). */
enum {
    PAFFY_E_HIP = -1,         /* a HIP runtime call failed; see paffy_hip_last_error()         */
    PAFFY_E_ARG = -2,         /* bad argument (NULL, misaligned pointer, a batch of 2 GiB - 64 bytes or more) */
    PAFFY_E_UNSUPPORTED = -3, /* stage list this build cannot fuse (run the stages one by one)  */
    PAFFY_E_CAPACITY = -4,    /* output buffer smaller than the planned size                   */
    PAFFY_E_STATE = -5        /* emit without a successful plan                                */
};

typedef struct {
    int32_t code;   /* PAFFY_ERR_*, 0 = none */
    int32_t stage;  /* index into the stage list, -1 = while parsing */
    int64_t record; /* zero-based record in the batch */
    int64_t aux;    /* offending character for STRAND / TP / CIGAR_CHAR */
} paffy_error;

typedef struct {
    int64_t n_records; /* lines in the batch                                              */
    int64_t n_rows;    /* lines emit will write                                           */
    int64_t in_bytes;
    int64_t out_bytes; /* bytes emit will write: all records before the first failing one */
    paffy_error error; /* first failure in stream order (code 0 = none)                   */
} paffy_plan_info;

/* Lifetime. `device` < 0 keeps the current HIP device. */
int paffy_hip_create(paffy_hip_ctx **ctx, int device);
void paffy_hip_destroy(paffy_hip_ctx *ctx);
/* Stream all kernels of this context are launched on (a hipStream_t; NULL = default stream). */
int paffy_hip_set_stream(paffy_hip_ctx *ctx, void *hip_stream);

/*
 * plan: index the lines of d_in[0, in_len) (16-byte aligned device pointer; the allocation
 * must be readable up to the next multiple of 16; in_len < 2 GiB - 64: offsets inside a batch are 32 bits --
 * a stream command takes any input as a sequence of batches, tile / to_bed through the begin / add / run calls below),
 * parse every record, run the stage list,
 * and compute each record's exact output size. Blocks until the numbers are known.
 * A final line without '\n' is a record (impl/paf.c:213).
 */
int paffy_hip_plan(paffy_hip_ctx *ctx, const paffy_stage *stages, int32_t n_stages, const void *d_in, int64_t in_len,
                   paffy_plan_info *info);

/*
 * tile_plan: `paffy tile` (impl/paf_tile.c:156-178) over the whole batch: every record gets its
 * tile level (tl) from per-base coverage counters of its QUERY sequence, visiting records by
 * (s1 desc, AS desc, input order); the output lists all records in that order with the cigar text
 * unchanged. A failing record means no output at all. Followed by paffy_hip_emit().
 */
int paffy_hip_tile_plan(paffy_hip_ctx *ctx, const void *d_in, int64_t in_len, paffy_plan_info *info);
/*
 * The same over an input of any size (the reference reads the whole file, read_pafs, impl/paf.c:492-499): begin, add every batch
 * of text (each a whole number of lines, < 2 GiB - 64 bytes, 16-byte aligned; it must stay where it is until the output has been
 * emitted: the lines are written from it), run. Records are ordered and grouped across all batches; info->error.record counts
 * records from the first batch on. Followed by paffy_hip_emit() or paffy_hip_emit_lines().
 */
int paffy_hip_tile_begin(paffy_hip_ctx *ctx);
int paffy_hip_tile_add(paffy_hip_ctx *ctx, const void *d_in, int64_t in_len);
int paffy_hip_tile_run(paffy_hip_ctx *ctx, paffy_plan_info *info);
/*
 * After a tile run: five int64 per output line, in output order, into device memory -- chain_score, score, input record, bytes of
 * the line, tile level. This is what the ranks of a `paffy tile` sharded by query sequence exchange (an all-gather of 40 bytes per
 * record) to find the place of their lines in the ordered output. Returns the number of lines or a negative error.
 */
int64_t paffy_hip_tile_keys(paffy_hip_ctx *ctx, int64_t cap_lines, void *d_keys);
/*
 * Sharding `paffy tile` by query sequence across GPUs (what `paffy split_file -q` does with files in the reference
 * pipeline, tests/paf_pipeline_test.sh:42, impl/paf_split_file.c:142-173). Names travel as 64-bit hashes.
 *   query_names:    the distinct query names of a batch (hashes[], host) and the bytes of their lines (weights[], host): what a
 *                   partitioner balances. Returns their number (<= cap) or a negative error. The line index built here is KEPT for
 *                   the batch (keyed by d_in and in_len, up to 64 batches) and consumed by the split of the same batch. Contract:
 *                   the bytes of the batch must not change between its query_names and its split -- the kept index is not
 *                   re-validated against the text. A batch that will not be split after all: paffy_hip_drop_index(ctx, d_in)
 *                   (d_in NULL: every kept index). Kept indexes are also dropped by any query_names / split call that fails and
 *                   by paffy_hip_tile_begin / _bed_begin / _chain_begin; paffy_hip_plan and the other index-building calls on the
 *                   same d_in drop that batch's entry.
 *   split_by_owner: the lines of the batch regrouped by part -- part = table_owner[i] for a query name with hash table_hash[i]
 *                   (ascending hashes; a name the table lacks goes to hash % n_parts) -- input order kept inside a part, every line
 *                   newline-terminated, into d_out (out_cap >= in_len + 1). part_bytes / part_records (host, n_parts each) say where
 *                   the parts end; d_rec_index (device, rec_index_cap int64, may be NULL) gets the batch index of every output line.
 *                   This is the send buffer of the all-to-all that gives every rank the records of its sequences.
 *   split_to:       the same, straight into a send buffer that gathers the parts of several batches per destination: part p's lines
 *                   go to d_out + part_dst[p], its record indices (+ rec_base: the batch's first global record) to
 *                   d_rec_index + rec_dst[p] (host arrays of n_parts byte / entry offsets; checked against out_cap / rec_index_cap
 *                   before anything is written: PAFFY_E_CAPACITY). No second copy of the text is needed to build the send buffer.
 *   scatter_lines:  line k = d_src[src_off[k], src_off[k + 1]) goes to d_dst + dst_off[k] (int64 offsets in device memory; n_lines + 1
 *                   source offsets): the ordered write once every line's place in the global output is known.
 */
int64_t paffy_hip_query_names(paffy_hip_ctx *ctx, const void *d_in, int64_t in_len, int64_t cap, uint64_t *hashes, int64_t *weights);
/* the same, with the number of lines of every name (records[], host): with bytes AND records per name a caller can lay out the send
   buffer of the partition before it splits the first batch (paffy_hip_split_to) */
int64_t paffy_hip_query_names_counts(paffy_hip_ctx *ctx, const void *d_in, int64_t in_len, int64_t cap, uint64_t *hashes, int64_t *weights, int64_t *records);
int paffy_hip_split_by_owner(paffy_hip_ctx *ctx, const void *d_in, int64_t in_len, int32_t n_parts, const uint64_t *table_hash, const uint32_t *table_owner,
                             int64_t n_table, void *d_out, int64_t out_cap, int64_t *part_bytes, int64_t *part_records, void *d_rec_index, int64_t rec_index_cap,
                             int64_t *n_records);
int paffy_hip_split_to(paffy_hip_ctx *ctx, const void *d_in, int64_t in_len, int32_t n_parts, const uint64_t *table_hash, const uint32_t *table_owner,
                       int64_t n_table, void *d_out, int64_t out_cap, const int64_t *part_dst, const int64_t *rec_dst, int64_t rec_base, int64_t *part_bytes,
                       int64_t *part_records, void *d_rec_index, int64_t rec_index_cap, int64_t *n_records);
int paffy_hip_drop_index(paffy_hip_ctx *ctx, const void *d_in);
int paffy_hip_scatter_lines(paffy_hip_ctx *ctx, const void *d_src, const void *d_src_off, const void *d_dst_off, int64_t n_lines, void *d_dst);
/*
 * Lines [first, first + n) of a tile / dedupe plan into d_out, the first of them at d_out[0] (16-byte aligned): for hosts that
 * drain an output larger than their staging buffer. *bytes = what was written.
 */
int paffy_hip_emit_lines(paffy_hip_ctx *ctx, int64_t first, int64_t n, void *d_out, int64_t out_cap, int64_t *bytes);

/*
 * dedupe_plan: `paffy dedupe [-a]` (impl/paf_dedupe.c:117-143). A record is kept unless a record kept earlier -- in this
 * batch or in an earlier batch of the same context since the last paffy_hip_dedupe_reset -- has the same query name,
 * target name, strand and four coordinates (with check_inverse: or equals it with query and target swapped; paf_check
 * then runs on the record, as in the reference). Kept records are written in input order with the cigar text verbatim.
 * Keys are compared as 128-bit hashes of those fields (two independent 64-bit hashes). The selection runs on the device (sort by key,
 * first of every run, binary searches in the sorted keys of the earlier batches); the context's memory lives in HBM, 16 bytes per
 * record written. Followed by paffy_hip_emit().
 */
int paffy_hip_dedupe_plan(paffy_hip_ctx *ctx, const void *d_in, int64_t in_len, int check_inverse, paffy_plan_info *info);
int paffy_hip_dedupe_reset(paffy_hip_ctx *ctx);
/* check_inverse value that keeps every record: `paf_read(.., 0)` -> `paf_write` (the normalised line, cigar text verbatim),
 * the read/write pair of `paffy split_file` (impl/paf_split_file.c:131-173). The context's dedupe memory is not touched. */
#define PAFFY_DEDUPE_KEEP_ALL 2

/*
 * bed_plan: `paffy to_bed` (impl/paf_to_bed.c:33-55, 166-190) over the whole batch: per-base coverage counters of every query
 * sequence (with include_inverted also of every target sequence, as the inverted record would count), written as maximal runs
 * "name start end value\n"; binary / exclude_unaligned / exclude_aligned / min_size are the reference's -b -e -f -m. Sequences come
 * in order of first appearance (the reference walks a hash table: no order defined). A failing record means no output at all.
 * Followed by paffy_hip_emit().
 */
typedef struct {
    int32_t binary, exclude_unaligned, exclude_aligned, include_inverted;
    int64_t min_size; /* default 1 */
} paffy_bed_opts;
int paffy_hip_bed_plan(paffy_hip_ctx *ctx, const void *d_in, int64_t in_len, const paffy_bed_opts *opts, paffy_plan_info *info);
/* to_bed over an input of any size, as a sequence of batches (see paffy_hip_tile_begin); the same opts for begin and run */
int paffy_hip_bed_begin(paffy_hip_ctx *ctx, const paffy_bed_opts *opts);
int paffy_hip_bed_add(paffy_hip_ctx *ctx, const void *d_in, int64_t in_len);
int paffy_hip_bed_run(paffy_hip_ctx *ctx, const paffy_bed_opts *opts, paffy_plan_info *info);
/*
 * After a bed run: the per-base counters themselves (what get_alignment_count_array / increase_alignment_level_counts keep per
 * sequence, impl/paf.c:667-712). _sequences = how many sequences the run saw; they are numbered in order of first appearance.
 * _counts copies counters [start, end) of one of them to h_counts; with accumulate != 0 it adds them to the values already in
 * h_counts instead, on the GPU, stopping at INT16_MAX - 1 as the reference's increments do (impl/paf.c:701).
 */
int64_t paffy_hip_bed_sequences(paffy_hip_ctx *ctx);
int paffy_hip_bed_counts(paffy_hip_ctx *ctx, int64_t sequence, int64_t start, int64_t end, uint16_t *h_counts, int accumulate);

/*
 * `paffy chain` (impl/paf_chain.c:123-127, paf_chain impl/chaining.c:266-343): the records of all batches are chained per (query,
 * target, strand) with the affine gap cost of impl/paf_chain.c:36-45 (no gap: 0, else gap_open + gap_extend * (query gap + target
 * gap)); every record gets its chain's id (cn) and score (s1) and the lines come out by descending own score. begin / add / run
 * as for tile; the text stays where it is until the output has been emitted (paffy_hip_emit, paffy_hip_emit_lines).
 * paffy_hip_plan_rows gives the input record of every output line, paffy_hip_chain_tags its two tags.
 * Where the reference's comparators fall back on object addresses (impl/chaining.c:18,47,62) creation order is used.
 */
typedef struct {
    int64_t gap_open, gap_extend; /* defaults of the command: 5000, 1 */
    int64_t max_gap_length;       /* 1000000 */
    float trim_fraction;          /* 1.0: alignments are shortened to their centre while chaining */
} paffy_chain_opts;
int paffy_hip_chain_begin(paffy_hip_ctx *ctx);
int paffy_hip_chain_add(paffy_hip_ctx *ctx, const void *d_in, int64_t in_len);
int paffy_hip_chain_run(paffy_hip_ctx *ctx, const paffy_chain_opts *opts, paffy_plan_info *info);
int64_t paffy_hip_chain_tags(paffy_hip_ctx *ctx, int64_t cap, int64_t *chain_id, int64_t *chain_score);

/*
 * After a tile or dedupe plan: the lines emit will write, in output order -- record[k] = zero-based input record of line k,
 * out_off[k] = its first output byte, out_off[n] = the total (cap >= n + 1 entries each). Returns n, or a negative error.
 * This is what a router such as `paffy split_file` needs to send every written line to the file of its contig.
 */
int64_t paffy_hip_plan_rows(paffy_hip_ctx *ctx, int64_t cap, uint32_t *record, int64_t *out_off);

/*
 * emit: write the planned output to d_out (16-byte aligned, out_cap >= info.out_bytes). Returns
 * after enqueueing; paffy_hip_sync() or any synchronisation of the stream completes it.
 */
int paffy_hip_emit(paffy_hip_ctx *ctx, void *d_out, int64_t out_cap);
int paffy_hip_sync(paffy_hip_ctx *ctx);

/* Host-buffer convenience used by the CLI drivers: H2D, plan, emit, D2H. *h_out is malloc'ed. */
int paffy_hip_run_host(paffy_hip_ctx *ctx, const paffy_stage *stages, int32_t n_stages, const char *h_in, int64_t in_len,
                       char **h_out, int64_t *out_len, paffy_plan_info *info);

/*
 * Streaming through host buffers: what the CLI drivers use (the reference's read / transform / write loop, impl/paf_invert.c:84-89,
 * with H2D, kernels and D2H overlapped on three streams). Two slots of pinned input; the output returns in pinned pieces.
 *   open(chunk_bytes: capacity of an input buffer, < 2 GiB - 64; piece_bytes: size of an output piece)
 *   input(want, keep, &cap): the buffer the next chunk is written into (NULL while both slots hold unread output); with want above its
 *                        capacity it is enlarged, the first `keep` bytes kept (a line longer than the chunk)
 *   submit(in_len, &info): the first in_len bytes (whole lines) go through the stage list; info is complete on return
 *   read(&piece, &len):  the next piece of the oldest submitted chunk; len = 0: that chunk is done (submit the next, or stop).
 *                        A piece stays valid until the call after next (three pinned pieces rotate) or close.
 * Submit chunk k + 1 before reading chunk k and the GPU works on k + 1 while the host drains k.
 */
typedef struct paffy_hip_stream paffy_hip_stream;
int paffy_hip_stream_open(paffy_hip_ctx *ctx, const paffy_stage *stages, int32_t n_stages, int64_t chunk_bytes, int64_t piece_bytes, paffy_hip_stream **stream);
char *paffy_hip_stream_input(paffy_hip_stream *stream, int64_t want, int64_t keep, int64_t *cap);
int paffy_hip_stream_submit(paffy_hip_stream *stream, int64_t in_len, paffy_plan_info *info);
int paffy_hip_stream_read(paffy_hip_stream *stream, const char **piece, int64_t *len);
void paffy_hip_stream_close(paffy_hip_stream *stream);
/* A closed stream leaves the device buffers of its two slots (input text, output) with the context; the context's next stream takes them
   again instead of allocating (a hipMalloc of the tens of GB a slot's output needs takes 16 ms most of the time and seconds right behind
   the hipFree of the stream before). paffy_hip_stream_trim frees them; paffy_hip_destroy does too, and so do the whole-input commands when
   they begin (tile_begin / bed_begin / chain_begin: the memory is theirs to use). A context that is destroyed while a stream of its is still
   open detaches the stream: paffy_hip_stream_close stays valid afterwards (it then frees the stream's own buffers), every other stream call
   needs the context. */
int paffy_hip_stream_trim(paffy_hip_ctx *ctx);

/*
 * Sums of the PAFFY_STATS stages of the last plan, in the argument order of paf_stats_calc (impl/paf.c:236-260): matches (M and =
 * bases), mismatches (X bases), query inserts, query deletes, query insert bases, query delete bases -- over the whole batch;
 * meaningful when no record failed (a failing record ends the reference process before it prints anything).
 */
int paffy_hip_plan_stats(paffy_hip_ctx *ctx, int64_t sums[6]);
/*
 * Diagnostics of the last paffy_hip_plan: the lean pipes (invert / identity trim / shatter / pass) are sized by the flat pass
 * (paffy_amd/csrc/flat_kernel.h); *left = the records it left to the record kernels (-1: the plan did not take the flat pass),
 * reasons[k] = how many for reason k (FLAT_WHY_* there). The outputs do not depend on who sized a record; the tests use this to
 * know that the pass under test really ran.
 */
int paffy_hip_flat_stats(paffy_hip_ctx *ctx, int64_t *left, int64_t reasons[16]);
/* The same six sums for every record of the batch (6 * n_records values, record by record): the numbers of the per-alignment line of
 * `paffy view` (paf_pretty_print, impl/paf.c:269-281). Returns n_records or a negative error. */
int64_t paffy_hip_plan_record_stats(paffy_hip_ctx *ctx, int64_t cap_records, int64_t *sums);

/*
 * The base-level rows of paf_pretty_print(..., include_alignment = true) (impl/paf.c:283-315; `paffy view -a`, impl/paf_view.c:158-160)
 * for records [first, first + count) of the planned batch, as the plan's stages left them: per window of 150 columns the target
 * row, the query row ('-' strand: read backwards and complemented) and the row of '*' under agreeing columns. The bases are shown
 * in the case they were loaded in, so the sequences must have been set after paffy_hip_keep_raw_sequences(ctx, 1).
 * _sizes gives the bytes of each record's block (0: no cigar); _rows writes the blocks to h_out at h_off[i] - h_off[0] (h_off:
 * count + 1 running sums of the sizes, so that a batch can be fetched in pieces). A record whose sequences are missing or shorter
 * than its coordinates (the reference reads outside the strings) is reported in *err (stage -1) and h_out is then not complete.
 * The batch text must still be in place (names are looked up in it when the plan had no PAFFY_ADD_MISMATCHES stage).
 */
int paffy_hip_keep_raw_sequences(paffy_hip_ctx *ctx, int on);
int paffy_hip_plan_alignment_sizes(paffy_hip_ctx *ctx, int64_t first, int64_t count, int64_t *h_bytes);
int paffy_hip_plan_alignment_rows(paffy_hip_ctx *ctx, int64_t first, int64_t count, const int64_t *h_off, char *h_out, paffy_error *err);

/*
 * Records as structs, for hosts that hold `Paf` objects (the per-record API of inc/paf.h:75-269, host/paf_api.c): the text is
 * parsed on the GPU (paf_parse + cigar_parse, impl/paf.c:70-209) and the fields come back as arrays -- one paffy_record per line,
 * names and cigar text as slices of h_in, the ops of all records back to back in the CigarRecord layout of inc/paf.h:61-64
 * (bits 0-55 length, bits 56-63 op). *recs and *ops are malloc'ed. A failing line is reported in info->error and nothing is returned.
 */
typedef struct {
    int64_t query_length, query_start, query_end, target_length, target_start, target_end;
    int64_t score, mapping_quality, num_matches, num_bases, tile_level, chain_id, chain_score;
    int64_t ops_first; /* index of this record's first op in *ops */
    int64_t n_ops;     /* -1: no cigar (Paf.cigar == NULL: no cg tag, or an empty one) */
    uint32_t query_name_off, query_name_len, target_name_off, target_name_len;
    uint32_t cigar_off, cigar_len; /* the cg:Z: value in h_in (cigar_len 0: none) */
    uint8_t same_strand, type, pad[6];
} paffy_record;
int paffy_hip_parse_host(paffy_hip_ctx *ctx, const char *h_in, int64_t in_len, paffy_record **recs, uint64_t **ops, int64_t *n_ops_total,
                         paffy_plan_info *info);

/*
 * Sequences for PAFFY_ADD_MISMATCHES: what the reference loads with fastaReadToFunction into a
 * name -> sequence hash (impl/paf_add_mismatches.c:89-97). names[i] is the NUL-terminated FASTA
 * header used as key, seqs[i] / lens[i] the bases (host memory; copied to HBM here).
 */
int paffy_hip_set_sequences(paffy_hip_ctx *ctx, int64_t n, const char *const *names, const char *const *seqs, const int64_t *lens);

/*
 * Thresholds of `paffy filter` as its main() holds them (impl/paf_filter.c:27-32; -s -t -w pass through atoi, -u -v
 * through atof). A record is kept when score >= min_alignment_score && chain_score >= min_chain_score &&
 * (max_tile_level == -1 || tile_level <= max_tile_level) && identity >= min_identity && identity_with_gaps >=
 * min_identity_with_gaps, the identities being float32 quotients of paf_stats_calc sums (impl/paf_filter.c:129-133);
 * `invert` keeps the others instead. Used by every PAFFY_FILTER stage of later plans; defaults -1, -1, -1.0, -1.0, -1, 0.
 */
typedef struct {
    int64_t min_chain_score;
    int64_t min_alignment_score;
    double min_identity;
    double min_identity_with_gaps;
    int64_t max_tile_level;
    int32_t invert;
} paffy_filter;
int paffy_hip_set_filter(paffy_hip_ctx *ctx, const paffy_filter *f);

/* Exit status the reference process ends with for a record error (1, 134 or 139). */
int paffy_hip_error_exit_status(int32_t code);
const char *paffy_hip_error_string(int32_t code);
const char *paffy_hip_last_error(paffy_hip_ctx *ctx);

/* Per-kernel timing with HIP events on the context's stream (for bench.py's roofline line). */
int paffy_hip_profile_enable(paffy_hip_ctx *ctx, int on);
/* Bracket only the launches of one kernel (its name as paffy_hip_profile_read reports it); NULL or "": every kernel again. Two events per
   launch cost the stream about 8 us each: a step of a dozen kernels measured with all of them bracketed runs 3 % slower than without. */
int paffy_hip_profile_only(paffy_hip_ctx *ctx, const char *kernel);
/* Fills up to cap entries; returns the number of distinct kernels seen since the last reset. */
int paffy_hip_profile_read(paffy_hip_ctx *ctx, const char **names, double *total_ms, int64_t *launches, int cap);
int paffy_hip_profile_reset(paffy_hip_ctx *ctx);

/* Plain device-memory helpers so that C hosts need no HIP headers. */
int paffy_hip_malloc(void **d_ptr, int64_t bytes);
int paffy_hip_free(void *d_ptr);
int paffy_hip_memcpy_h2d(void *d_dst, const void *h_src, int64_t bytes);
int paffy_hip_memcpy_d2h(void *h_dst, const void *d_src, int64_t bytes);
int paffy_hip_device_count(void);

/*
 * Synthetic workload of SURVEY.md section 8d, generated on the device (same bytes as
 * tools/paf_synth.c): records [r0, r0+n) of (seed, mean_ops). With d_out == NULL only the
 * byte count is returned in *bytes.
 */
int paffy_hip_synth(paffy_hip_ctx *ctx, uint64_t seed, uint32_t mean_ops, uint64_t r0, uint64_t n, void *d_out,
                    int64_t out_cap, int64_t *bytes);
/* the same with n_contigs contigs per genome instead of 24 (fewer contigs = deeper coverage for a given record count) */
int paffy_hip_synth_contigs(paffy_hip_ctx *ctx, uint64_t seed, uint32_t mean_ops, uint32_t n_contigs, uint64_t r0, uint64_t n, void *d_out,
                            int64_t out_cap, int64_t *bytes);

/*
 * cfg4 workload (records + two genomes for PAFFY_ADD_MISMATCHES), generated on the device: every
 * contig pair (hs.chr<k>, pt.chr<k>) has one master alignment and a record is a window of it, so the aligned columns pair
 * homologous bases (2 % substitutions; a quarter of the contigs hold the query reverse-complemented and give '-' records).
 * setup builds the master tables for target contigs of tlen_min + hash % (tlen_span + 1) bases and, with with_genomes,
 * writes both genomes into the context's sequence store (replacing paffy_hip_set_sequences' content). synth4 then writes
 * records [r0, r0+n) like paffy_hip_synth. Same bytes as the host build in tools/paf_synth.c.
 */
int paffy_hip_synth4_setup(paffy_hip_ctx *ctx, uint64_t seed, uint32_t mean_ops, uint32_t n_contigs, int64_t tlen_min, int64_t tlen_span,
                           int with_genomes);
int paffy_hip_synth4(paffy_hip_ctx *ctx, uint64_t r0, uint64_t n, void *d_out, int64_t out_cap, int64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif
