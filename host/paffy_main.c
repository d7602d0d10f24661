/*
 * paffy_main.c -- `paffy <command> [options]` dispatcher of the MI355X build.
 * Contract of the reference dispatcher (paffy_main.c:46-84): no arguments -> usage, status 0;
 * unknown command -> message + usage, status 1; otherwise the command's own status. The hot-path
 * commands (shatter, invert, trim, add_mismatches, tile) run on the GPU; the others are outside
 * this build's scope and say so with status 1.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "paffy_host.h"

typedef int (*cmd_fn)(int, char **);
static const struct {
    const char *name;
    cmd_fn fn;
    const char *help;
} COMMANDS[] = {
    {"add_mismatches", paffy_add_mismatches_main, "Replace Ms with =/Xs in the cigar (or -a: the reverse)"},
    {"chain", paffy_chain_main, "Chain alignments: every record gets the id and score of its chain"},
    {"dechunk", NULL, "Map chunk coordinates back (not in this build)"},
    {"dedupe", paffy_dedupe_main, "Drop duplicate alignments"},
    {"filter", paffy_filter_main, "Filter alignments on their stats"},
    {"invert", paffy_invert_main, "Switch query and target coordinates"},
    {"shatter", paffy_shatter_main, "Break alignments into gapless blocks"},
    {"tile", paffy_tile_main, "Give alignments tile levels along the query"},
    {"to_bed", paffy_to_bed_main, "Coverage of the query sequences in BED format"},
    {"trim", paffy_trim_main, "Slice off lower identity tails"},
    {"upconvert", NULL, "Convert coordinates to extracted subsequences (not in this build)"},
    {"split_file", paffy_split_file_main, "Split a PAF file per contig"},
    {"view", paffy_view_main, "Alignment stats per record and overall, with -a the base-level alignments"},
};

static void usage(void) {
    fprintf(stderr, "paffy: toolkit for working with PAF files (MI355X hot-path build)\n\n");
    fprintf(stderr, "usage: paffy <command> [options]\n\navailable commands:\n");
    for (size_t i = 0; i < sizeof(COMMANDS) / sizeof(COMMANDS[0]); i++) fprintf(stderr, "    %-24s %s\n", COMMANDS[i].name, COMMANDS[i].help);
    fprintf(stderr, "\n");
}

/* `paffy a | paffy b`: a 64 KiB pipe makes writer and reader trade places every few microseconds; ask for 1 MiB on both ends */
static void widen_pipes(void) {
    for (int fd = 0; fd <= 1; fd++) {
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISFIFO(st.st_mode)) (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);
    }
}

int main(int argc, char *argv[]) {
    widen_pipes();
    if (argc < 2) {
        usage();
        return 0;
    }
    for (size_t i = 0; i < sizeof(COMMANDS) / sizeof(COMMANDS[0]); i++) {
        if (strcmp(argv[1], COMMANDS[i].name) == 0) {
            if (!COMMANDS[i].fn) {
                fprintf(stderr, "paffy %s is outside the scope of this build (hot path only)\n", argv[1]);
                return 1;
            }
            return COMMANDS[i].fn(argc - 1, argv + 1);
        }
    }
    fprintf(stderr, "%s is not a valid paffy command\n", argv[1]);
    usage();
    return 1;
}
