/*
 * record_types.h -- HBM data model shared by the kernels and the host API.
 *
 * A batch is the raw PAF text (never copied: names and cigar text are addressed by offset)
 * plus, per record, one RecMeta (fixed fields parsed by k_header) and a few per-record result
 * words written by the sizing pass. Cigar ops are never materialised in HBM for records whose
 * ops fit the LDS store (<= PAFFY_OPS_CAP ops, lengths < 2^29): they are re-parsed from the
 * text by the workgroup that owns the record. Larger / wider records keep 8-byte ops
 * (length << 8 | op, the CigarRecord layout of inc/paf.h:61-64) in a global arena.
 */
#ifndef PAFFY_RECORD_TYPES_H_
#define PAFFY_RECORD_TYPES_H_

#include <stdint.h>

#include "../../include/paffy_hip.h"

#define PAFFY_OPS_CAP 8192u /* 4-byte ops held in LDS per workgroup (32 KiB) */
#define PAFFY_TMPL_MAX 768u /* bytes per pre-rendered line piece held in LDS */
#define PAFFY_HALO 32u

enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_EQ = 3, OP_X = 4 }; /* inc/paf.h:52-58 */

struct RecMeta {
    int64_t qlen, qs, qe, tlen, ts, te, nmatch, nbases, mapq; /* fields 2-4, 7-12 */
    int64_t score, tile_level, chain_id, chain_score;         /* AS tl cn s1 (defaults 0,-1,-1,-1) */
    uint32_t qname_off, qname_len, tname_off, tname_len;      /* slices of the input text */
    uint32_t cg_off, cg_len;                                  /* value of the last cg:Z: tag */
    uint8_t same_strand, type, has_cg, pad0;
    int32_t err;     /* PAFFY_ERR_* found while parsing the fixed fields */
    int32_t err_aux; /* offending character */
    uint32_t pad1;
};

/* Record classes decided by the sizing pass. */
enum { KLASS_LDS = 0, KLASS_ARENA = 1 };

/* One FASTA sequence resident in HBM (add_mismatches). */
struct SeqEntry {
    uint64_t off; /* into seq_base */
    int64_t len;
};

struct DevInfo {
    unsigned long long first_err_key; /* min over failing records of rec<<16 | (stage+1)<<8 | code */
    unsigned long long arena_used;    /* bump pointer (in ops) */
    unsigned long long out_bytes, out_rows;
    uint32_t n_seps, n_lines;
    uint32_t w_count;   /* records routed to the arena kernel */
    uint32_t b_count[2]; /* records routed to the next (bigger LDS store) sizing launch, per level */
    uint32_t internal;  /* internal-limit flags (must stay 0) */
    uint32_t g_count;   /* LDS-class shatter records left to the general row kernel (not eligible for k_emit_rows) */
    unsigned long long stats[6]; /* PAFFY_STATS: matches, mismatches, inserts, deletes, insert bases, delete bases */
    /* the longer first store level (paffy_hip.hip: lvl0_long): records whose ops overflowed the first-level store although their cigar was
       short enough to start there, and -- while the safe bound is in force -- records of the second level that would have */
    uint32_t lvl0_over, lvl0_probe_dense;
    /* flat sizing pass (flat_kernel.h): the records it left to the record kernels */
    uint32_t flat_legacy;
    uint32_t flat_defer; /* records k_flat_lane handed on to k_flat_size */
    uint32_t n_items; /* EmitItem entries written (segments of the records too long for one wave of the row writer) */
    uint32_t flat_reason[16]; /* why: FLAT_WHY_* of flat_kernel.h (diagnostics, paffy_hip_flat_stats) */
    unsigned long long add_scr_total, add_new_total; /* flat_add_kernel.h: words of scratch / of new ops the batch needs */
};

/* What the stage list left of a record; written by the sizing pass, read by the emit pass. */
struct RecPlan {
    int64_t qs, qe, ts, te, sub_lo, sub_hi;
    uint32_t lo, n;
    uint32_t flags; /* bit0 rev, bit1 swp, bit2 query/target swapped, bit3 has_cigar, bit4 shatter, bit5 direct, bit6 k_emit_rows,
                       bit7 dropped by a filter, bits 8-15 type, bit16 k_emit_line,
                       bit17 the 4-byte ops live in the arena block arena_off[rec] (rebuilt by add_mismatches), not in the mirror,
                       bit18 the mirror holds 2-byte words (every length below 8192),
                       bit19 a long shatter record written as EmitItem segments by k_emit_rows,
                       bit20 the 4-byte ops are words of new_ops[] (flat add_mismatches), bit21 the row pieces are in row_pieces[],
                       bit22 the line's cigar is a stretch of the input's text: k_emit_copy (wq[0] = first byte from cg_off, wt[0] = bytes;
                       wq[1] / wt[1] = new length of a first / last op a fixed trim shortened, wq[2] / wt[2] = bytes of that op's text) */
    uint32_t chunk; /* ops per lane in the sizing sweep: wave w owns view ops [64*w*chunk, 64*(w+1)*chunk) */
    /* shatter: query / target bases consumed and output bytes produced before each wave's range */
    int64_t wq[4], wt[4], wo[4]; /* four waves: the emit workgroups; a one-wave sizing workgroup fills entry 0 and zeroes the rest */
};


/* A segment of a long shatter record for the one-wave row writer (k_emit_rows): the view's ops [wb, we), the bases consumed and the
   bytes written in front of it. The flat sizing pass cuts records of more than PAFFY_ROWS_MAX_OPS ops into such segments. */
struct EmitItem {
    uint32_t rec, wb, we, pad;
    int64_t cq0, ct0, wo;
};

struct KParams {
    const uint8_t *in;
    uint32_t in_len;
    uint32_t n_rec;
    const RecMeta *meta;
    paffy_stage stages[PAFFY_MAX_STAGES];
    int32_t n_stages;
    /* per-record results of the sizing pass */
    int64_t *out_len;
    int64_t *out_rows;
    uint32_t *status;   /* code | (stage+1)<<8 | klass<<16 */
    int32_t *err_aux;
    uint32_t *n_ops;
    uint64_t *arena_off;
    void *rec_plan; /* RecPlan[n_rec] */
    uint32_t *ops_mirror; /* ops of LDS-class records from word cg_off / 2 on: 2-byte words (len << 3 | op) when every length of the record is below 8192 (RecPlan flag bit 18), else 4-byte words */
    /* add_mismatches: sequences in HBM and, per record, the index of its query / target sequence (-1: absent) */
    const uint8_t *seq_base; /* upper-cased when loaded (every comparison is of toupper'ed bases, impl/paf.c:752-757) */
    const uint8_t *seq_comp; /* the same bytes complemented (A<->T, C<->G): what the - strand compares, read backwards */
    const SeqEntry *seqs;
    const int32_t *rec_qseq;
    const int32_t *rec_tseq;
    /* emit pass */
    const int64_t *out_off;
    uint8_t *out;
    /* arena class */
    uint64_t *arena;
    uint64_t arena_cap; /* in ops */
    uint32_t *w_list;
    uint32_t *b_list[2]; /* records that did not fit the LDS op store of level 0 / level 1 */
    uint32_t ops_cap;    /* 4-byte ops the sizing workgroup's LDS store holds */
    uint32_t next_cap;   /* store of the next level (0: none, overflow goes to the arena) */
    uint32_t level;      /* 0: whole batch; 1, 2: walk b_list[level - 1] */
    uint32_t lvl0_max;   /* records with more than this many cigar bytes / 2 start at level 1 (k_header queued them) */
    uint32_t lvl0_long_bytes; /* the cigar length the longer first level would take (0: none): level 1 counts what would have overflowed it */
    DevInfo *info;
    paffy_filter filter; /* thresholds of PAFFY_FILTER stages */
    const uint32_t *emit_order; /* records by descending output size (coarse): the one-wave-per-record writers start the long ones first */
    const uint32_t *size_order; /* records by descending cigar length (coarse), for the sizing launch */
    int64_t *rec_stats;         /* PAFFY_STATS: six sums per record (the order of paf_stats_calc's arguments), or NULL */
    uint32_t nocheck_mask;      /* bit i: stage i runs without the paf_check the command loops append (PAFFY_NO_CHECK) */
    uint32_t wave_max_bytes;    /* records with at most this many cigar bytes are sized by the one-wave kernel (0: none): the four-wave kernel skips them */
    const uint32_t *new_ops;    /* flat_add_kernel.h: the rebuilt cigars of all records, 4-byte ops back to back (RecPlan flag bit 20: arena_off counts words of it) */
    EmitItem *items;            /* segments of long records, written by the flat sizing pass, emitted by k_emit_rows in front of the records */
    uint32_t n_items, items_cap;
    uint8_t *row_pieces;        /* flat sizing pass: the three constant pieces of a record's rows, 3 x 48 zero filled bytes per record (RecPlan flag bit 21; NULL: none) */
    const uint8_t *flat_done;   /* flat sizing pass: 1 = the record has been sized there, the record kernels skip it (NULL: no flat pass) */
};

#endif
