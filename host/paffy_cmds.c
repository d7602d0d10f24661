/*
 * paffy_cmds.c -- option tables and entry points of the hot-path subcommands.
 * Flags follow the reference drivers: common -i/--inputFile -o/--outputFile -l/--logLevel
 * -h/--help; trim adds -t/--trimFraction -r/--trimIdentity -f/--fixedTrim
 * (impl/paf_trim.c:53-63); -h prints usage and returns 0, an unknown option returns 1.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <inttypes.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "paffy_host.h"

typedef struct {
    const char *in_path, *out_path, *log_level;
    float trim_fraction, trim_identity; /* floats, as the reference's statics (impl/paf_trim.c:14-16) */
    int fixed_trim;
} cmd_opts;

static void usage_common(const char *cmd, const char *what) {
    fprintf(stderr, "paffy %s [options], MI355X build\n%s\n", cmd, what);
    fprintf(stderr, "-i --inputFile : PAF file to read (default: stdin)\n");
    fprintf(stderr, "-o --outputFile : PAF file to write (default: stdout)\n");
}
static void usage_tail(void) {
    fprintf(stderr, "-l --logLevel : log level (INFO, DEBUG)\n");
    fprintf(stderr, "-h --help : print this message\n");
}

/* returns -1 to continue, otherwise the exit code */
static int parse_opts(int argc, char *argv[], const char *cmd, const char *what, int is_trim, cmd_opts *o) {
    static struct option common[] = {{"logLevel", required_argument, 0, 'l'}, {"inputFile", required_argument, 0, 'i'},
                                     {"outputFile", required_argument, 0, 'o'}, {"help", no_argument, 0, 'h'},
                                     {"trimFraction", required_argument, 0, 't'}, {"trimIdentity", required_argument, 0, 'r'},
                                     {"fixedTrim", no_argument, 0, 'f'}, {0, 0, 0, 0}};
    struct option opts[8];
    memcpy(opts, common, sizeof(common));
    if (!is_trim) memset(&opts[4], 0, sizeof(struct option)); /* terminate after the common four */
    memset(o, 0, sizeof(*o));
    o->trim_fraction = 1.0f;
    o->trim_identity = 0.05f;
    optind = 1;
    for (;;) {
        int idx = 0;
        int key = getopt_long(argc, argv, is_trim ? "l:i:o:ht:r:f" : "l:i:o:h", opts, &idx);
        if (key == -1) break;
        switch (key) {
            case 'l': o->log_level = optarg; break;
            case 'i': o->in_path = optarg; break;
            case 'o': o->out_path = optarg; break;
            case 't': o->trim_fraction = (float)atof(optarg); break;
            case 'r': o->trim_identity = (float)atof(optarg); break;
            case 'f': o->fixed_trim = 1; break;
            case 'h':
            default:
                usage_common(cmd, what);
                if (is_trim) {
                    fprintf(stderr, "-r --trimIdentity : trim tails whose identity is below x - x*r of the alignment identity x "
                                    "(0..1, default %f)\n", 0.05);
                    fprintf(stderr, "-t --trimFraction : fraction of aligned bases to trim per end; with identity trimming the "
                                    "largest tail (0..1, default %f)\n", 1.0);
                    fprintf(stderr, "-f --fixedTrim : trim a constant --trimFraction instead of trimming by identity\n");
                }
                usage_tail();
                return key == 'h' ? 0 : 1;
        }
    }
    return -1;
}

static int run_stream_cmd(const cmd_opts *o, const paffy_stage *stages, int n_stages, const char *name) {
    time_t t0 = time(NULL);
    host_set_log_level(o->log_level);
    host_log_info("Input file string : %s\n", o->in_path ? o->in_path : "(stdin)");
    host_log_info("Output file string : %s\n", o->out_path ? o->out_path : "(stdout)");
    FILE *in = host_open_input(o->in_path);
    if (!in) {
        fprintf(stderr, "paffy %s: cannot open %s\n", name, o->in_path);
        return 1;
    }
    FILE *out = o->out_path ? fopen(o->out_path, "w") : stdout;
    if (!out) {
        fprintf(stderr, "paffy %s: cannot open %s\n", name, o->out_path);
        return 1;
    }
    int rc = host_stream(stages, n_stages, in, out);
    if (o->in_path) fclose(in);
    if (o->out_path) fclose(out);
    host_log_info("Paffy %s is done!, %lld seconds have elapsed\n", name, (long long)(time(NULL) - t0));
    return rc;
}

int paffy_shatter_main(int argc, char *argv[]) {
    cmd_opts o;
    int rc = parse_opts(argc, argv, "shatter", "Break up paf alignments into their gapless match blocks", 0, &o);
    if (rc >= 0) return rc;
    paffy_stage st = {PAFFY_SHATTER, 0.05f, 1.0f};
    return run_stream_cmd(&o, &st, 1, "shatter");
}

int paffy_invert_main(int argc, char *argv[]) {
    cmd_opts o;
    int rc = parse_opts(argc, argv, "invert", "Swap query and target of every alignment", 0, &o);
    if (rc >= 0) return rc;
    paffy_stage st = {PAFFY_INVERT, 0.05f, 1.0f};
    return run_stream_cmd(&o, &st, 1, "invert");
}

int paffy_trim_main(int argc, char *argv[]) {
    cmd_opts o;
    int rc = parse_opts(argc, argv, "trim", "Trim the ends of every alignment", 1, &o);
    if (rc >= 0) return rc;
    host_set_log_level(o.log_level);
    host_log_info("Trim fraction using : %f\n", o.trim_fraction);
    host_log_info("Trim by identity fraction : %f\n", o.trim_identity);
    paffy_stage st = {o.fixed_trim ? PAFFY_TRIM_FIXED : PAFFY_TRIM_IDENTITY, o.trim_identity, o.trim_fraction};
    return run_stream_cmd(&o, &st, 1, "trim");
}

/* paffy filter: options and their conversions as in impl/paf_filter.c:45-100 (-s -t -w through atoi, -u -v through atof). */
int paffy_filter_main(int argc, char *argv[]) {
    static struct option opts[] = {{"logLevel", required_argument, 0, 'l'}, {"inputFile", required_argument, 0, 'i'},
                                   {"outputFile", required_argument, 0, 'o'}, {"minChainScore", required_argument, 0, 's'},
                                   {"minAlignmentScore", required_argument, 0, 't'}, {"minIdentity", required_argument, 0, 'u'},
                                   {"minIdentityWithGaps", required_argument, 0, 'v'}, {"maxTileLevel", required_argument, 0, 'w'},
                                   {"invert", no_argument, 0, 'x'}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    cmd_opts o;
    memset(&o, 0, sizeof(o));
    paffy_filter f = {-1, -1, -1.0, -1.0, -1, 0};
    optind = 1;
    for (;;) {
        int idx = 0;
        int key = getopt_long(argc, argv, "l:i:o:s:t:u:v:w:xh", opts, &idx);
        if (key == -1) break;
        switch (key) {
            case 'l': o.log_level = optarg; break;
            case 'i': o.in_path = optarg; break;
            case 'o': o.out_path = optarg; break;
            case 's': f.min_chain_score = atoi(optarg); break;
            case 't': f.min_alignment_score = atoi(optarg); break;
            case 'u': f.min_identity = atof(optarg); break;
            case 'v': f.min_identity_with_gaps = atof(optarg); break;
            case 'w': f.max_tile_level = atoi(optarg); break;
            case 'x': f.invert = 1; break;
            case 'h':
            default:
                usage_common("filter", "Filter pafs based on alignment stats");
                fprintf(stderr, "-s --minChainScore : Filter alignments with a chain score less than this\n");
                fprintf(stderr, "-t --minAlignmentScore : Filter alignments with an alignment score less than this\n");
                fprintf(stderr, "-u --minIdentity : Filter alignments with an identity less than this, exclude indels\n");
                fprintf(stderr, "-v --minIdentityWithGaps : Filter alignments with an identity less than this, including indels\n");
                fprintf(stderr, "-w --maxTileLevel : Filter alignments with a tile level greater than this\n");
                fprintf(stderr, "-x --invert : Only output alignments that don't pass filters\n");
                usage_tail();
                return key == 'h' ? 0 : 1;
        }
    }
    host_set_log_level(o.log_level);
    host_log_info("Filtering paf with min chain score:%lld min alignment score:%lld min identity:%f min identity with gaps:%f max tile level:%lld invert:%s\n",
                  (long long)f.min_chain_score, (long long)f.min_alignment_score, f.min_identity, f.min_identity_with_gaps,
                  (long long)f.max_tile_level, f.invert ? "True" : "False");
    host_set_filter(&f);
    paffy_stage st = {PAFFY_FILTER, 0.05f, 1.0f};
    return run_stream_cmd(&o, &st, 1, "filter");
}

/* paffy dedupe [-a], impl/paf_dedupe.c:48-100 */
int paffy_dedupe_main(int argc, char *argv[]) {
    static struct option opts[] = {{"logLevel", required_argument, 0, 'l'}, {"inputFile", required_argument, 0, 'i'},
                                   {"outputFile", required_argument, 0, 'o'}, {"checkInverse", no_argument, 0, 'a'},
                                   {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    cmd_opts o;
    memset(&o, 0, sizeof(o));
    int check_inverse = 0;
    optind = 1;
    for (;;) {
        int idx = 0;
        int key = getopt_long(argc, argv, "l:i:o:ha", opts, &idx);
        if (key == -1) break;
        switch (key) {
            case 'l': o.log_level = optarg; break;
            case 'i': o.in_path = optarg; break;
            case 'o': o.out_path = optarg; break;
            case 'a': check_inverse = 1; break;
            case 'h':
            default:
                usage_common("dedupe", "Remove alignments with the same names, strand and coordinates as an earlier one");
                fprintf(stderr, "-a --checkInverse : Also deduplicate alignments that are the same, but with query and target reversed\n");
                usage_tail();
                return key == 'h' ? 0 : 1;
        }
    }
    host_set_dedupe(check_inverse);
    return run_stream_cmd(&o, NULL, 0, "dedupe");
}

/* paffy split_file, impl/paf_split_file.c:58-120 */
int paffy_split_file_main(int argc, char *argv[]) {
    static struct option opts[] = {{"logLevel", required_argument, 0, 'l'}, {"inputFile", required_argument, 0, 'i'},
                                   {"prefix", required_argument, 0, 'p'}, {"query", no_argument, 0, 'q'},
                                   {"minLength", required_argument, 0, 'm'}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    const char *log_level = NULL, *in_path = NULL, *prefix = "split_";
    int by_query = 0;
    int64_t min_length = 0;
    optind = 1;
    for (;;) {
        int idx = 0;
        int key = getopt_long(argc, argv, "l:i:p:qm:h", opts, &idx);
        if (key == -1) break;
        switch (key) {
            case 'l': log_level = optarg; break;
            case 'i': in_path = optarg; break;
            case 'p': prefix = optarg; break;
            case 'q': by_query = 1; break;
            case 'm': min_length = atol(optarg); break;
            case 'h':
            default:
                fprintf(stderr, "paffy split_file [options], MI355X build\nSplit PAF file into separate output files by target (default) or query contig name\n");
                fprintf(stderr, "-i --inputFile : Input paf file. If not specified reads from stdin\n");
                fprintf(stderr, "-p --prefix : Output file prefix (may include directory path). Default: split_\n");
                fprintf(stderr, "-q --query : Split by query contig name instead of target contig name\n");
                fprintf(stderr, "-m --minLength : Contigs shorter than m share <prefix>small_<k>.paf files of at most m bases each. Default: 0 (disabled)\n");
                usage_tail();
                return key == 'h' ? 0 : 1;
        }
    }
    host_set_log_level(log_level);
    host_log_info("Input file string : %s\n", in_path ? in_path : "(stdin)");
    host_log_info("Output prefix : %s\n", prefix);
    host_log_info("Split by : %s\n", by_query ? "query" : "target");
    host_log_info("Min contig length : %lld\n", (long long)min_length);
    FILE *in = host_open_input(in_path);
    if (!in) {
        fprintf(stderr, "paffy split_file: cannot open %s\n", in_path);
        return 1;
    }
    int rc = host_split_file(in, prefix, by_query, min_length);
    if (in_path) fclose(in);
    return rc;
}

/* FASTA -> (header, sequence) pairs. Key = the whole header line after '>', sequence = all
 * non-whitespace characters up to the next header (the fastaReadToFunction /
 * fastaRead_readToMapFunction behaviour assumed in SURVEY Appendix C; parity unpinned). */
typedef struct {
    char **names, **seqs;
    int64_t *lens;
    int64_t n, cap;
} fasta_set;

static void fasta_push(fasta_set *f, char *name, char *seq, int64_t len) {
    if (f->n == f->cap) {
        f->cap = f->cap ? f->cap * 2 : 64;
        f->names = (char **)realloc(f->names, sizeof(char *) * (size_t)f->cap);
        f->seqs = (char **)realloc(f->seqs, sizeof(char *) * (size_t)f->cap);
        f->lens = (int64_t *)realloc(f->lens, sizeof(int64_t) * (size_t)f->cap);
    }
    f->names[f->n] = name;
    f->seqs[f->n] = seq;
    f->lens[f->n] = len;
    f->n++;
}

static int fasta_read(const char *path, fasta_set *f) {
    FILE *fh = fopen(path, "r");
    if (!fh) return -1;
    char *line = NULL, *name = NULL, *seq = NULL;
    size_t lcap = 0;
    int64_t slen = 0, scap = 0;
    ssize_t got;
    while ((got = getline(&line, &lcap, fh)) >= 0) {
        while (got > 0 && (line[got - 1] == '\n' || line[got - 1] == '\r')) line[--got] = '\0';
        if (line[0] == '>') {
            if (name) fasta_push(f, name, seq, slen);
            name = strdup(line + 1);
            seq = NULL;
            slen = scap = 0;
        } else if (name) {
            if (slen + got + 1 > scap) {
                scap = (slen + got + 1) * 2;
                seq = (char *)realloc(seq, (size_t)scap);
            }
            for (ssize_t i = 0; i < got; i++)
                if (line[i] != ' ' && line[i] != '\t') seq[slen++] = line[i];
        }
    }
    if (name) fasta_push(f, name, seq ? seq : strdup(""), slen);
    free(line);
    fclose(fh);
    return 0;
}

/* impl/paf_add_mismatches.c: options l:i:o:h plus -a / --removeMismatches; FASTA files are positional */
int paffy_add_mismatches_main(int argc, char *argv[]) {
    static struct option opts[] = {{"logLevel", required_argument, 0, 'l'}, {"inputFile", required_argument, 0, 'i'},
                                   {"outputFile", required_argument, 0, 'o'}, {"removeMismatches", no_argument, 0, 'a'},
                                   {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    cmd_opts o;
    memset(&o, 0, sizeof(o));
    int remove = 0;
    optind = 1;
    for (;;) {
        int idx = 0;
        int key = getopt_long(argc, argv, "l:i:o:ha", opts, &idx);
        if (key == -1) break;
        switch (key) {
            case 'l': o.log_level = optarg; break;
            case 'i': o.in_path = optarg; break;
            case 'o': o.out_path = optarg; break;
            case 'a': remove = 1; break;
            case 'h':
            default:
                fprintf(stderr, "paffy add_mismatches [fasta_files]xN [options], MI355X build\n"
                                "Replace each M op by runs of = (match) and X (mismatch) against the sequences\n");
                fprintf(stderr, "-i --inputFile : PAF file to read (default: stdin)\n-o --outputFile : PAF file to write (default: stdout)\n");
                fprintf(stderr, "-a --removeMismatches : the reverse: merge = and X (and M) runs into M\n");
                usage_tail();
                return key == 'h' ? 0 : 1;
        }
    }
    host_set_log_level(o.log_level);
    paffy_stage st = {remove ? PAFFY_REMOVE_MISMATCHES : PAFFY_ADD_MISMATCHES, 0.05f, 1.0f};
    if (!remove) {
        fasta_set f;
        memset(&f, 0, sizeof(f));
        for (int i = optind; i < argc; i++) {
            host_log_info("Parsing sequence file : %s\n", argv[i]);
            if (fasta_read(argv[i], &f) != 0) {
                fprintf(stderr, "paffy add_mismatches: cannot open %s\n", argv[i]);
                return 1;
            }
        }
        host_log_info("Read %i sequences from sequence files\n", (int)f.n);
        host_set_sequences((const char *const *)f.names, (const char *const *)f.seqs, f.lens, f.n);
    }
    return run_stream_cmd(&o, &st, 1, "add_mismatches");
}

/* impl/paf_tile.c: whole-file command; the batch is the file */
int paffy_tile_main(int argc, char *argv[]) {
    cmd_opts o;
    int rc = parse_opts(argc, argv, "tile", "Give every alignment a tile level along its query sequence", 0, &o);
    if (rc >= 0) return rc;
    host_set_log_level(o.log_level);
    FILE *in = host_open_input(o.in_path);
    FILE *out = o.out_path ? fopen(o.out_path, "w") : stdout;
    if (!in || !out) {
        fprintf(stderr, "paffy tile: cannot open %s\n", !in ? o.in_path : o.out_path);
        return 1;
    }
    rc = host_tile(in, out);
    if (o.in_path) fclose(in);
    if (o.out_path) fclose(out);
    return rc;
}

/*
 * impl/paf_chain.c:47-153: the records are chained per (query, target, strand) with the affine gap cost of :36-45; every record
 * gets the id (cn) and the score (s1) of its chain and the output is ordered by descending alignment score.
 */
int paffy_chain_main(int argc, char *argv[]) {
    static struct option opts[] = {{"logLevel", required_argument, 0, 'l'}, {"inputFile", required_argument, 0, 'i'}, {"outputFile", required_argument, 0, 'o'},
                                   {"maxGapLength", required_argument, 0, 'g'}, {"trimFraction", required_argument, 0, 't'}, {"chainGapOpen", required_argument, 0, 'd'},
                                   {"chainGapExtend", required_argument, 0, 'e'}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    cmd_opts o;
    memset(&o, 0, sizeof(o));
    paffy_chain_opts c = {5000, 1, 1000000, 1.0f}; /* impl/paf_chain.c:18-21 */
    optind = 1;
    for (;;) {
        int idx = 0;
        int key = getopt_long(argc, argv, "l:i:o:hg:t:d:e:", opts, &idx);
        if (key == -1) break;
        switch (key) {
            case 'l': o.log_level = optarg; break;
            case 'i': o.in_path = optarg; break;
            case 'o': o.out_path = optarg; break;
            case 'g': c.max_gap_length = atoi(optarg); break;
            case 't': c.trim_fraction = (float)atof(optarg); break;
            case 'd': c.gap_open = atoi(optarg); break;
            case 'e': c.gap_extend = atoi(optarg); break;
            case 'h':
            default:
                fprintf(stderr, "paffy chain [options], MI355X build\nChains the records in the PAF file into chains, rescoring them as chains.\nChains are indicated with the cn tag.\n");
                fprintf(stderr, "-i --inputFile : Input paf file. If not specified reads from stdin\n-o --outputFile : Output paf file. If not specified outputs to stdout\n");
                fprintf(stderr, "-g --maxGapLength [INT] : The maximum allowable length of a gap in either sequence to chain (default:1000000bp)\n");
                fprintf(stderr, "-d --chainGapOpen [INT] : The cost of opening a chain gap (default:5000)\n-e --chainGapExtend [INT] : The cost of extending a chain gap (default:1)\n");
                fprintf(stderr, "-t --trimFraction : Fraction (from 0 to 1) of aligned bases to discount from the ends of the alignments when chaining (default:1.0)\n");
                usage_tail();
                return key == 'h' ? 0 : 1;
        }
    }
    host_set_log_level(o.log_level);
    host_log_info("Input file string : %s\nOutput file string : %s\nMaximum gap length : %lld\nChain gap open : %lld\nChain gap extend : %lld\n", o.in_path, o.out_path,
                  (long long)c.max_gap_length, (long long)c.gap_open, (long long)c.gap_extend);
    FILE *in = host_open_input(o.in_path);
    FILE *out = o.out_path ? fopen(o.out_path, "w") : stdout;
    if (!in || !out) {
        fprintf(stderr, "paffy chain: cannot open %s\n", !in ? o.in_path : o.out_path);
        return 1;
    }
    int rc = host_chain(in, out, &c);
    if (o.in_path) fclose(in);
    if (o.out_path) fclose(out);
    return rc;
}

/*
 * impl/paf_view.c:42-213: every record is encoded against the sequences (paf_encode_mismatches); its stats line and, with -a, its
 * base-level rows are printed (paf_pretty_print) and its paf_stats_calc sums are added up for the aggregate line (-s).
 */
int paffy_view_main(int argc, char *argv[]) {
    static struct option opts[] = {{"logLevel", required_argument, 0, 'l'}, {"inputFile", required_argument, 0, 'i'},
                                   {"outputFile", required_argument, 0, 'o'}, {"includeAlignment", no_argument, 0, 'a'},
                                   {"printAggregateStats", no_argument, 0, 's'}, {"noPerAlignmentStats", no_argument, 0, 't'},
                                   {"errorIfIdentityLowerThanX", required_argument, 0, 'u'}, {"errorIfAlignedBasesLowerThanX", required_argument, 0, 'v'},
                                   {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    cmd_opts o;
    memset(&o, 0, sizeof(o));
    int include_alignment = 0, aggregate = 0, per_alignment = 1;
    float min_identity = 0.0f;          /* impl/paf_view.c:67: a float set by atof */
    int64_t min_aligned = 0;            /* :68: set by atoi */
    optind = 1;
    for (;;) {
        int idx = 0;
        int key = getopt_long(argc, argv, "l:i:o:hastu:v:", opts, &idx);
        if (key == -1) break;
        switch (key) {
            case 'l': o.log_level = optarg; break;
            case 'i': o.in_path = optarg; break;
            case 'o': o.out_path = optarg; break;
            case 'a': include_alignment = 1; break;
            case 's': aggregate = 1; break;
            case 't': per_alignment = 0; break;
            case 'u': min_identity = (float)atof(optarg); break;
            case 'v': min_aligned = atoi(optarg); break;
            case 'h':
            default:
                fprintf(stderr, "paffy view [fasta_files]xN [options], MI355X build\nAlignment stats per record and overall (-s)\n-a --includeAlignment : print the base-level alignment under each stats line\n");
                fprintf(stderr, "-i --inputFile : PAF file to read (default: stdin)\n-o --outputFile : file to write (default: stdout)\n");
                fprintf(stderr, "-s --printAggregateStats : print overall stats at the end\n-t --noPerAlignmentStats : no stats per alignment\n");
                fprintf(stderr, "-u --errorIfIdentityLowerThanX : assert the average identity is >= X\n-v --errorIfAlignedBasesLowerThanX : assert the aligned bases are >= X\n");
                usage_tail();
                return key == 'h' ? 0 : 1;
        }
    }
    if (optind >= argc) { /* impl/paf_view.c:108-111 */
        fprintf(stderr, "Expected at least one sequence file\n");
        exit(1);
    }
    host_set_log_level(o.log_level);
    fasta_set f;
    memset(&f, 0, sizeof(f));
    for (int i = optind; i < argc; i++) {
        host_log_info("Parsing sequence file : %s\n", argv[i]);
        if (fasta_read(argv[i], &f) != 0) {
            fprintf(stderr, "paffy view: cannot open %s\n", argv[i]);
            return 1;
        }
    }
    host_keep_raw_sequences(include_alignment); /* the rows show the bases in the case of the files */
    host_set_sequences((const char *const *)f.names, (const char *const *)f.seqs, f.lens, f.n);
    host_set_alignment_rows(include_alignment && per_alignment); /* impl/paf_view.c:158-160: paf_pretty_print runs unless -t */
    const paffy_stage st[2] = {{PAFFY_ADD_MISMATCHES, 0.05f, 1.0f}, {PAFFY_STATS, 0.0f, 0.0f}};
    host_set_stats(1);
    FILE *in = host_open_input(o.in_path);
    FILE *out = o.out_path ? fopen(o.out_path, "w") : stdout;
    if (!in || !out) {
        fprintf(stderr, "paffy view: cannot open %s\n", !in ? o.in_path : o.out_path);
        return 1;
    }
    if (per_alignment) host_set_stats_lines(out); /* paf_pretty_print's stats line per record, impl/paf_view.c:158-160 */
    int rc = host_stream(st, 2, in, out);
    host_set_stats_lines(NULL);
    int64_t t[6], n_alignments = 0; /* matches, mismatches, inserts, deletes, insert bases, delete bases */
    host_get_stats(t, &n_alignments);
    host_set_stats(0);
    if (rc) return rc;
    if (!aggregate) memset(t, 0, sizeof(t)); /* the reference only adds the records up under -s (impl/paf_view.c:163-168) */
    if (aggregate) /* impl/paf_view.c:176-184, same float arithmetic and format */
        fprintf(out, "Total-alignments:%" PRIi64 "\tAvg-Identity:%f\tAvg-Identity-with-gaps:%f\tAligned-bases:%" PRIi64 "\tAligned-bases-with-gaps:%" PRIi64
                     "\tQuery-inserts:%" PRIi64 "\tQuery-deletes:%" PRIi64 "\n",
                n_alignments, (float)t[0] / (t[0] + t[1]), (float)t[0] / (t[0] + t[1] + t[4] + t[5]), t[0] + t[1], t[0] + t[1] + t[4] + t[5], t[2], t[3]);
    fflush(out);
    /* the two sanity asserts, impl/paf_view.c:187-188 (NaN >= x is false: an empty input fails the first one there too) */
    if (!((float)t[0] / (t[0] + t[1]) >= min_identity) || !(t[0] + t[1] >= min_aligned)) {
        fprintf(stderr, "paffy view: Assertion failed (average identity or aligned bases below the requested minimum)\n");
        abort();
    }
    if (o.in_path) fclose(in);
    if (o.out_path) fclose(out);
    return 0;
}

/*
 * impl/paf_to_bed.c:69-213: coverage of the query sequences (with -n of the target sequences too) as BED runs. -q adds, under -f,
 * the sequences of a FASTA file that no alignment names (write_missing_fasta_seqs, impl/paf_to_bed.c:63-67: "name 0 length\t0").
 */
int paffy_to_bed_main(int argc, char *argv[]) {
    static struct option opts[] = {{"logLevel", required_argument, 0, 'l'}, {"inputFile", required_argument, 0, 'i'},
                                   {"outputFile", required_argument, 0, 'o'}, {"binary", no_argument, 0, 'b'},
                                   {"excludeUnaligned", no_argument, 0, 'e'}, {"excludeAligned", no_argument, 0, 'f'},
                                   {"minSize", required_argument, 0, 'm'}, {"includeInverted", no_argument, 0, 'n'},
                                   {"queryFastaFile", required_argument, 0, 'q'}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    cmd_opts o;
    memset(&o, 0, sizeof(o));
    paffy_bed_opts b;
    memset(&b, 0, sizeof(b));
    b.min_size = 1;
    const char *query_fasta = NULL;
    optind = 1;
    for (;;) {
        int idx = 0;
        int key = getopt_long(argc, argv, "l:i:o:hbefm:nq:", opts, &idx);
        if (key == -1) break;
        switch (key) {
            case 'l': o.log_level = optarg; break;
            case 'i': o.in_path = optarg; break;
            case 'o': o.out_path = optarg; break;
            case 'b': b.binary = 1; break;
            case 'e': b.exclude_unaligned = 1; break;
            case 'f': b.exclude_aligned = 1; break;
            case 'm': b.min_size = atoi(optarg); break; /* impl/paf_to_bed.c: atoi */
            case 'n': b.include_inverted = 1; break;
            case 'q': query_fasta = optarg; break;
            case 'h':
            default:
                fprintf(stderr, "paffy to_bed [options], MI355X build\nBED file of the alignment coverage of the query sequences\n");
                fprintf(stderr, "-i --inputFile : PAF file to read (default: stdin)\n-o --outputFile : BED file to write (default: stdout)\n");
                fprintf(stderr, "-b --binary : 0 / 1 instead of the number of alignments\n-e --excludeUnaligned : no intervals without alignments\n");
                fprintf(stderr, "-f --excludeAligned : no intervals with alignments\n-m --minSize : no intervals shorter than this\n");
                fprintf(stderr, "-n --includeInverted : count the target sequences too\n-q --queryFastaFile : with -f, also list sequences no alignment names\n");
                usage_tail();
                return key == 'h' ? 0 : 1;
        }
    }
    host_set_log_level(o.log_level);
    FILE *in = host_open_input(o.in_path);
    FILE *out = o.out_path ? fopen(o.out_path, "w") : stdout;
    if (!in || !out) {
        fprintf(stderr, "paffy to_bed: cannot open %s\n", !in ? o.in_path : o.out_path);
        return 1;
    }
    int rc;
    if (b.exclude_aligned && query_fasta) {
        /* the names the alignments use are needed afterwards: keep the text */
        size_t cap = 1 << 20, have = 0;
        char *buf = (char *)malloc(cap);
        for (;;) {
            if (have == cap) buf = (char *)realloc(buf, cap *= 2);
            size_t got = fread(buf + have, 1, cap - have, in);
            if (got == 0) break;
            have += got;
        }
        FILE *mem = fmemopen(buf, have ? have : 1, "r");
        if (have == 0) fgetc(mem);
        rc = host_to_bed(mem, out, &b);
        fclose(mem);
        fasta_set f;
        memset(&f, 0, sizeof(f));
        if (rc == 0 && fasta_read(query_fasta, &f) == 0) {
            for (int64_t k = 0; k < f.n; k++) {
                const size_t nl = strlen(f.names[k]);
                int seen = 0;
                for (const char *p = buf, *end = buf + have; p < end && !seen;) { /* a sequence is known if a line names it as query (or, with -n, as target) */
                    const char *le = memchr(p, '\n', (size_t)(end - p));
                    if (!le) le = end;
                    const char *t1 = memchr(p, '\t', (size_t)(le - p));
                    if (t1 && (size_t)(t1 - p) == nl && memcmp(p, f.names[k], nl) == 0) seen = 1;
                    if (!seen && b.include_inverted && t1) {
                        const char *q = t1;
                        for (int col = 1; col < 5 && q; col++) q = memchr(q + 1, '\t', (size_t)(le - q - 1));
                        if (q) {
                            const char *t6 = memchr(q + 1, '\t', (size_t)(le - q - 1));
                            if (t6 && (size_t)(t6 - q - 1) == nl && memcmp(q + 1, f.names[k], nl) == 0) seen = 1;
                        }
                    }
                    p = le + 1;
                }
                if (!seen) fprintf(out, "%s 0 %" PRIi64 "\t0\n", f.names[k], f.lens[k]);
            }
        }
        free(buf);
    } else {
        rc = host_to_bed(in, out, &b);
    }
    if (o.in_path) fclose(in);
    if (o.out_path) fclose(out);
    else fflush(out);
    return rc;
}
