/*
 * paffy_host.h -- host drivers of the `paffy <command>` CLI over the gfx950 C-ABI.
 *
 * Same subcommand names, option letters / long names and exit codes as the reference CLI
 * (paffy_main.c:46-84, impl/paf_<cmd>.c getopt tables); each paffy_<cmd>_main replaces the
 * reference's per-record loop with whole-batch calls into include/paffy_hip.h.
 */
#ifndef PAFFY_HOST_H_
#define PAFFY_HOST_H_

#include <stdint.h>
#include <stdio.h>

#include "../include/paffy_hip.h"

int paffy_shatter_main(int argc, char *argv[]);
int paffy_invert_main(int argc, char *argv[]);
int paffy_filter_main(int argc, char *argv[]);
int paffy_dedupe_main(int argc, char *argv[]);
int paffy_split_file_main(int argc, char *argv[]);
int paffy_trim_main(int argc, char *argv[]);
int paffy_add_mismatches_main(int argc, char *argv[]);
int paffy_tile_main(int argc, char *argv[]);
int paffy_chain_main(int argc, char *argv[]);
int paffy_view_main(int argc, char *argv[]);
int paffy_to_bed_main(int argc, char *argv[]);

/* The input of a command: `path` (NULL: stdin). Under the N-GPU launcher (host/paffy_launch.c) a worker reads only its byte range
 * of the file: PAFFY_RANGE="first:end". */
FILE *host_open_input(const char *path);
/* The GPU this worker uses: PAFFY_DEVICE (set by the launcher), else the current device (-1). */
int host_device(void);

/* Log level shared by the drivers: 0 off, 1 info, 2 debug (set from -l/--logLevel). */
void host_set_log_level(const char *s);
void host_log_info(const char *fmt, ...);

/*
 * Stream `in` through a stage list: the input is cut into chunks that end on a line boundary,
 * every chunk is one plan/emit round trip, outputs are written in order. On a failing record
 * everything before it is written, the reference's message is printed and the process ends
 * the way the reference does (exit 1, SIGABRT or SIGSEGV). Returns 0 on success.
 */
int host_stream(const paffy_stage *stages, int n_stages, FILE *in, FILE *out);
/* Thresholds handed to the context that host_stream creates (paffy filter). */
void host_set_filter(const paffy_filter *f);
/* paffy dedupe: host_stream runs paffy_hip_dedupe_plan per chunk on one context (which remembers the records written). */
void host_set_dedupe(int check_inverse);

/* paffy view -s -t: host_stream adds up the PAFFY_STATS sums of the chunks instead of writing lines */
void host_set_stats(int on);
void host_get_stats(int64_t sums[6], int64_t *n_records);
/* paffy view without -t: every record's paf_pretty_print stats line is written to fh as the chunks go by (NULL: off) */
void host_set_stats_lines(FILE *fh);

/* `paffy tile`: reads all of `in`, one tile_plan + emit, writes `out`. */
int host_tile(FILE *in, FILE *out);
/* `paffy to_bed`: reads all of `in`, one bed_plan + emit, writes `out` */
int host_to_bed(FILE *in, FILE *out, const paffy_bed_opts *opts);
/* `paffy chain`: reads all of `in`, chains on the GPU, writes the records with their cn / s1 tags by descending score */
int host_chain(FILE *in, FILE *out, const paffy_chain_opts *opts);
/* paffy split_file: normalised lines (cigar text verbatim) routed to "<prefix><contig>.paf" / "<prefix>small_<k>.paf" */
int host_split_file(FILE *in, const char *prefix, int by_query, int64_t min_length);

/* Sequences handed to the context that host_stream creates (add_mismatches); pointers must stay valid. */
void host_set_sequences(const char *const *names, const char *const *seqs, const int64_t *lens, int64_t n);
/* paffy view -a: keep the bases as loaded beside the upper-cased store (before host_stream), and print the rows under each stats line */
void host_keep_raw_sequences(int on);
void host_set_alignment_rows(int on);

#endif
