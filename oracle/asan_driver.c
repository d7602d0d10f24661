/*
 * asan_driver.c -- TEST INFRASTRUCTURE ONLY: the oracle's entry points behind a command line, for the sanitizer build
 * (`make -C oracle asan`: paf_oracle.c compiled with -fsanitize=address,undefined into oracle/_san/oracle_asan). tests/test_sanitizers.py
 * runs the known-answer inputs and the reference's fixture through it and compares the bytes with the plain build's.
 *   oracle_asan run <k1,k2,...> in.paf out    stage kinds as numbers (PO_* of paf_oracle.h), default trim parameters
 *   oracle_asan tile|dedupe|dedupe-a|chain|to_bed in.paf out
 * Exit status: 0, or 100 + error code of the failing record (the output then holds what the records before it produced).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "paf_oracle.h"

static char *slurp(const char *path, int64_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *b = (char *)malloc((size_t)n + 1);
    if (b && fread(b, 1, (size_t)n, f) != (size_t)n) {
        free(b);
        b = NULL;
    }
    fclose(f);
    if (b) b[n] = 0;
    *len = n;
    return b;
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const char *cmd = argv[1];
    const int is_run = !strcmp(cmd, "run");
    if (is_run && argc < 5) return 2;
    int64_t in_len = 0;
    char *in = slurp(argv[is_run ? 3 : 2], &in_len);
    if (!in) return 2;
    char *out = NULL;
    int64_t out_len = 0, fresh = 0;
    po_error err;
    memset(&err, 0, sizeof(err));
    if (is_run) {
        po_stage st[16];
        int n = 0;
        char *spec = strdup(argv[2]);
        for (char *t = strtok(spec, ","); t && n < 16; t = strtok(NULL, ",")) {
            st[n].kind = atoi(t);
            st[n].p0 = 0.05f;
            st[n].p1 = 1.0f;
            n++;
        }
        free(spec);
        po_run(st, n, in, in_len, NULL, 0, &out, &out_len, &err);
    } else if (!strcmp(cmd, "tile")) {
        po_tile(in, in_len, &out, &out_len, &err);
    } else if (!strcmp(cmd, "dedupe") || !strcmp(cmd, "dedupe-a")) {
        po_dedupe(in, in_len, !strcmp(cmd, "dedupe-a"), &out, &out_len, &err);
    } else if (!strcmp(cmd, "chain")) {
        po_chain(in, in_len, 5000, 1, 1000000, 1.0f, &out, &out_len, &fresh, &err);
    } else if (!strcmp(cmd, "to_bed")) {
        po_to_bed(in, in_len, 0, 0, 0, 1, 1, &out, &out_len, &err);
    } else {
        free(in);
        return 2;
    }
    FILE *f = fopen(argv[is_run ? 4 : 3], "wb");
    if (!f) return 2;
    if (out_len > 0) fwrite(out, 1, (size_t)out_len, f);
    fclose(f);
    po_free(out);
    free(in);
    return err.code ? 100 + err.code : 0;
}
