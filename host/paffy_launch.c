/*
 * paffy_launch.c -- `bin/paffy`: the front door of the MI355X build. It never touches the GPU (it is not even linked against
 * HIP): with one GPU it replaces itself by the GPU worker `bin/paffy_gpu` (the reference's subcommands over the C-ABI,
 * host/paffy_main.c); with PAFFY_GPUS=N it is the N-GPU path behind the kept CLI -- it starts one worker process per GPU, before
 * anything has initialised a GPU, and puts their outputs together in the order one process would have written them.
 *
 * The reference's own parallelism is at this level: the user splits the input per contig and runs one `paffy` per split
 * (tests/paf_pipeline_test.sh:42-67, impl/paf_split_file.c:142-173). Here:
 *   stream commands (invert, trim, shatter, add_mismatches, filter): records are independent (impl/paf_invert.c:84-89), so worker r
 *     reads the r-th of N contiguous byte ranges of the input, cut at line ends (PAFFY_RANGE; no copy of the input), and writes a
 *     spool file; the spools are concatenated in rank order. A failing record ends the run as one process would: everything before it
 *     is written, the worker's exit status (or signal) becomes ours.
 *   tile: its state is per QUERY sequence (impl/paf_tile.c:160-175, impl/paf.c:675-688). The launcher makes two passes over the input
 *     (bytes of every query name; heaviest name to the lightest worker), routes every line to its worker's spool together with its
 *     global line number, each worker tiles its sequences, and the outputs -- each already in (s1 desc, AS desc, input order) order
 *     (paf_cmp_by_descending_score, impl/paf_tile.c:28-34) -- are merged by that same key. A failing record means no output at all,
 *     as in the reference (it writes after the last record).
 * Everything between the workers goes through files under PAFFY_TMPDIR (default /dev/shm, else TMPDIR, else /tmp): host-mediated, no
 * GPU-to-GPU traffic -- a CLI's input comes from the host and its output goes back there. Other commands run on one GPU.
 *
 * Environment: PAFFY_GPUS=N; PAFFY_ONE_DEVICE=1 (rehearsal: every worker uses device 0); PAFFY_WORKER=path (another worker binary:
 * the CPU tests put a stand-in there); PAFFY_TMPDIR.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <getopt.h>
#include <limits.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#define MAX_RANKS 64

static char g_worker[PATH_MAX];
static char g_tmpdir[PATH_MAX];  /* where the private spool directory is made */
static char g_spooldir[PATH_MAX]; /* mkdtemp(<tmpdir>/paffy.XXXXXX), mode 0700: nobody else can plant a link under a name we open */
static char g_spool[MAX_RANKS][4][PATH_MAX]; /* per rank: input, output, rows, index */
static char g_stdin_spool[PATH_MAX];
static int g_n = 0;
static volatile pid_t g_pids[MAX_RANKS]; /* workers that are running (0: none) */

static void cleanup(void) {
    for (int r = 0; r < g_n; r++)
        for (int k = 0; k < 4; k++)
            if (g_spool[r][k][0]) unlink(g_spool[r][k]);
    if (g_stdin_spool[0]) unlink(g_stdin_spool);
    if (g_spooldir[0]) rmdir(g_spooldir);
}

/* SIGINT / SIGTERM / SIGHUP: the workers go with us and the spools (RAM-backed under /dev/shm) are removed */
static void on_signal(int sig) {
    for (int r = 0; r < MAX_RANKS; r++)
        if (g_pids[r] > 0) kill(g_pids[r], SIGTERM);
    cleanup(); /* unlink / rmdir only: async-signal-safe */
    signal(sig, SIG_DFL);
    raise(sig);
}

static int make_spooldir(void) {
    snprintf(g_spooldir, sizeof(g_spooldir), "%s/paffy.XXXXXX", g_tmpdir);
    if (!mkdtemp(g_spooldir)) {
        g_spooldir[0] = 0;
        return -1;
    }
    return 0;
}

static void find_worker(void) {
    const char *e = getenv("PAFFY_WORKER");
    if (e && *e) {
        snprintf(g_worker, sizeof(g_worker), "%s", e);
        return;
    }
    char self[PATH_MAX];
    ssize_t n = readlink("/proc/self/exe", self, sizeof(self) - 1);
    if (n <= 0) {
        snprintf(g_worker, sizeof(g_worker), "paffy_gpu");
        return;
    }
    self[n] = 0;
    char *slash = strrchr(self, '/');
    if (slash) *slash = 0;
    snprintf(g_worker, sizeof(g_worker), "%s/paffy_gpu", slash ? self : ".");
}

static void find_tmpdir(void) {
    const char *cands[] = {getenv("PAFFY_TMPDIR"), "/dev/shm", getenv("TMPDIR"), "/tmp"};
    for (size_t i = 0; i < sizeof(cands) / sizeof(cands[0]); i++)
        if (cands[i] && *cands[i] && access(cands[i], W_OK | X_OK) == 0) {
            snprintf(g_tmpdir, sizeof(g_tmpdir), "%s", cands[i]);
            return;
        }
    snprintf(g_tmpdir, sizeof(g_tmpdir), ".");
}

static int is_stream_cmd(const char *c) {
    return !strcmp(c, "invert") || !strcmp(c, "trim") || !strcmp(c, "shatter") || !strcmp(c, "add_mismatches") || !strcmp(c, "filter");
}

/*
 * The command line of a sharded command, parsed the way the worker will parse it: getopt_long with the subcommand's own option string
 * and long options (/root/reference/impl/paf_invert.c:41-76, paf_trim.c:45-100, paf_add_mismatches.c:40-85, paf_filter.c:50-115,
 * paf_tile.c:100-150) -- clustered short flags (`trim -fi in.paf`), abbreviated long options (`--input x`), an option's value that
 * looks like an option (`-l -i`) all mean here what they mean there. The worker's command line is rebuilt from the parse: every
 * option but -i / -o as the worker would have seen it, then the positional arguments, then our own -i / -o. Anything getopt_long
 * rejects, and -h, leaves the command to a single worker (which prints what the reference prints).
 */
typedef struct {
    const char *in_path, *out_path;
    char *opts[256]; /* "-x" or "-x", "value" */
    int n_opts;
    char **pos; /* positional arguments (in argv order) */
    int n_pos;
    int ok;
    char **copy; /* the permuted copy of argv that pos points into (freed by main) */
} CmdLine;

static const struct option k_common[] = {{"logLevel", required_argument, 0, 'l'}, {"inputFile", required_argument, 0, 'i'}, {"outputFile", required_argument, 0, 'o'},
                                         {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
static const struct option k_trim[] = {{"logLevel", required_argument, 0, 'l'}, {"inputFile", required_argument, 0, 'i'}, {"outputFile", required_argument, 0, 'o'},
                                       {"help", no_argument, 0, 'h'}, {"trimFraction", required_argument, 0, 't'}, {"trimIdentity", required_argument, 0, 'r'},
                                       {"fixedTrim", no_argument, 0, 'f'}, {0, 0, 0, 0}};
static const struct option k_add[] = {{"logLevel", required_argument, 0, 'l'}, {"inputFile", required_argument, 0, 'i'}, {"outputFile", required_argument, 0, 'o'},
                                      {"removeMismatches", no_argument, 0, 'a'}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
static const struct option k_filter[] = {{"logLevel", required_argument, 0, 'l'}, {"inputFile", required_argument, 0, 'i'}, {"outputFile", required_argument, 0, 'o'},
                                         {"minChainScore", required_argument, 0, 's'}, {"minAlignmentScore", required_argument, 0, 't'},
                                         {"minIdentity", required_argument, 0, 'u'}, {"minIdentityWithGaps", required_argument, 0, 'v'},
                                         {"maxTileLevel", required_argument, 0, 'w'}, {"invert", no_argument, 0, 'x'}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};

static void parse_cmdline(int argc, char **argv, CmdLine *cl) {
    memset(cl, 0, sizeof(*cl));
    const char *cmd = argv[1];
    const char *optstring = "l:i:o:h";
    const struct option *lopts = k_common;
    if (!strcmp(cmd, "trim")) { optstring = "l:i:o:ht:r:f"; lopts = k_trim; }
    else if (!strcmp(cmd, "add_mismatches")) { optstring = "l:i:o:ha"; lopts = k_add; }
    else if (!strcmp(cmd, "filter")) { optstring = "l:i:o:s:t:u:v:w:xh"; lopts = k_filter; }
    /* getopt_long permutes the array it is given: a copy of argv[1..] (argv[1], the subcommand, stands where the program name would) */
    char **v = (char **)calloc((size_t)argc + 1, sizeof(char *));
    for (int i = 1; i < argc; i++) v[i - 1] = argv[i];
    const int vc = argc - 1;
    static char flag[64][3];
    int n_flag = 0;
    opterr = 0;
    optind = 1;
    cl->ok = 1;
    for (;;) {
        int idx = 0;
        const int key = getopt_long(vc, v, optstring, lopts, &idx);
        if (key == -1) break;
        if (key == '?' || key == ':' || key == 'h' || cl->n_opts + 2 >= (int)(sizeof(cl->opts) / sizeof(cl->opts[0])) || n_flag >= 64) {
            cl->ok = 0;
            break;
        }
        if (key == 'i') { cl->in_path = optarg; continue; }
        if (key == 'o') { cl->out_path = optarg; continue; }
        flag[n_flag][0] = '-'; flag[n_flag][1] = (char)key; flag[n_flag][2] = 0;
        cl->opts[cl->n_opts++] = flag[n_flag++];
        if (optarg) cl->opts[cl->n_opts++] = optarg;
    }
    cl->copy = v;
    cl->pos = v + optind; /* what getopt_long moved behind the options */
    cl->n_pos = cl->ok ? vc - optind : 0;
}

static int copy_fd(int from, int to) {
    static char buf[1 << 22];
    for (;;) {
        ssize_t n = read(from, buf, sizeof(buf));
        if (n < 0) {
            if (errno == EINTR) continue;
            return -1;
        }
        if (n == 0) return 0;
        for (ssize_t o = 0; o < n;) {
            ssize_t w = write(to, buf + o, (size_t)(n - o));
            if (w < 0) {
                if (errno == EINTR) continue;
                return -1;
            }
            o += w;
        }
    }
}

/* argv of a worker: the subcommand, the options as parsed (without -i / -o), the positional arguments behind "--", then our -i / -o */
static char **worker_argv(const char *cmd, const CmdLine *cl, const char *in_path, const char *out_path) {
    char **v = (char **)calloc((size_t)cl->n_opts + (size_t)cl->n_pos + 9, sizeof(char *));
    int k = 0;
    v[k++] = g_worker;
    v[k++] = (char *)cmd;
    for (int i = 0; i < cl->n_opts; i++) v[k++] = cl->opts[i];
    v[k++] = (char *)"-i";
    v[k++] = (char *)in_path;
    v[k++] = (char *)"-o";
    v[k++] = (char *)out_path;
    if (cl->n_pos) v[k++] = (char *)"--";
    for (int i = 0; i < cl->n_pos; i++) v[k++] = cl->pos[i];
    v[k] = NULL;
    return v;
}

static pid_t spawn(char **wargv, int rank, int world, int one_device, const char *range, const char *rows_path) {
    pid_t pid = fork();
    if (pid != 0) return pid;
    char b[64];
    snprintf(b, sizeof(b), "%d", rank);
    setenv("PAFFY_RANK", b, 1);
    snprintf(b, sizeof(b), "%d", world);
    setenv("PAFFY_WORLD", b, 1);
    snprintf(b, sizeof(b), "%d", one_device ? 0 : rank);
    setenv("PAFFY_DEVICE", b, 1);
    unsetenv("PAFFY_GPUS");
    if (range) setenv("PAFFY_RANGE", range, 1);
    else unsetenv("PAFFY_RANGE");
    if (rows_path) setenv("PAFFY_ROWS_FILE", rows_path, 1);
    else unsetenv("PAFFY_ROWS_FILE");
    execv(wargv[0], wargv);
    fprintf(stderr, "paffy: cannot start the worker %s: %s\n", wargv[0], strerror(errno));
    _exit(127);
}

/* ends this process the way worker `st` ended (after our own cleanup) */
static int status_of(int st) {
    if (WIFSIGNALED(st)) {
        cleanup();
        signal(WTERMSIG(st), SIG_DFL);
        raise(WTERMSIG(st));
        return 128 + WTERMSIG(st);
    }
    return WEXITSTATUS(st);
}

/* ---------------- stream commands ---------------- */

static int run_stream(const char *cmd, const CmdLine *cl, int n, int one_device, const char *in_path, const char *out_path) {
    int fd = open(in_path, O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0) {
        fprintf(stderr, "paffy %s: cannot open %s\n", cmd, in_path);
        return 1;
    }
    const int64_t size = (int64_t)sb.st_size;
    int64_t cut[MAX_RANKS + 1];
    cut[0] = 0;
    for (int r = 1; r < n; r++) { /* the first line end at or after size * r / n */
        int64_t pos = size / n * r;
        if (pos < cut[r - 1]) pos = cut[r - 1];
        char blk[65536];
        int64_t found = size;
        for (int64_t at = pos; at < size;) {
            ssize_t got = pread(fd, blk, sizeof(blk), (off_t)at);
            if (got <= 0) break;
            char *nl = (char *)memchr(blk, '\n', (size_t)got);
            if (nl) {
                found = at + (nl - blk) + 1;
                break;
            }
            at += got;
        }
        cut[r] = found;
    }
    cut[n] = size;
    close(fd);
    pid_t pids[MAX_RANKS];
    for (int r = 0; r < n; r++) {
        snprintf(g_spool[r][1], PATH_MAX, "%s/%d.out", g_spooldir, r);
        char range[64];
        snprintf(range, sizeof(range), "%lld:%lld", (long long)cut[r], (long long)cut[r + 1]);
        char **wv = worker_argv(cmd, cl, in_path, g_spool[r][1]);
        pids[r] = spawn(wv, r, n, one_device, range, NULL);
        free(wv);
        if (pids[r] < 0) {
            fprintf(stderr, "paffy: fork failed\n");
            for (int q = 0; q < r; q++) kill(pids[q], SIGTERM);
            for (int q = 0; q < r; q++) waitpid(pids[q], NULL, 0);
            return 1;
        }
        g_pids[r] = pids[r];
    }
    int out_fd = out_path ? open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0666) : 1;
    if (out_fd < 0) {
        fprintf(stderr, "paffy %s: cannot open %s\n", cmd, out_path);
        for (int q = 0; q < n; q++) kill(pids[q], SIGTERM);
        for (int q = 0; q < n; q++) waitpid(pids[q], NULL, 0);
        return 1;
    }
    int rc = 0;
    for (int r = 0; r < n; r++) { /* in rank order: the output of worker r follows that of worker r - 1 */
        int st = 0;
        while (waitpid(pids[r], &st, 0) < 0 && errno == EINTR) {}
        g_pids[r] = 0;
        int sfd = open(g_spool[r][1], O_RDONLY | O_NOFOLLOW);
        if (sfd >= 0) {
            if (copy_fd(sfd, out_fd) != 0) rc = 1;
            close(sfd);
        }
        const int failed = WIFSIGNALED(st) || WEXITSTATUS(st) != 0;
        if (failed) { /* the records before the failing one are out; nothing after them is written */
            for (int q = r + 1; q < n; q++) kill(pids[q], SIGTERM);
            for (int q = r + 1; q < n; q++) {
                waitpid(pids[q], NULL, 0);
                g_pids[q] = 0;
            }
            if (out_fd != 1) close(out_fd);
            return status_of(st);
        }
    }
    if (out_fd != 1) close(out_fd);
    return rc;
}

/* ---------------- tile ---------------- */

typedef struct {
    uint64_t hash;
    int64_t weight;
    int32_t owner, used;
} NameSlot;

static uint64_t name_hash(const char *s, size_t len) { /* cov_name_hash (coverage_kernel.h), shard.name_hash */
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < len; i++) h = (h ^ (unsigned char)s[i]) * 0x100000001b3ull;
    h = (h ^ (0x100u + (uint64_t)len)) * 0x100000001b3ull;
    return h ^ (h >> 29);
}

typedef struct {
    NameSlot *slot;
    size_t cap, used;
} NameTab;

static NameSlot *tab_find(NameTab *t, uint64_t h, int insert) {
    if (insert && (t->used + 1) * 2 > t->cap) { /* grow */
        NameTab nt = {NULL, t->cap ? t->cap * 2 : 1024, 0};
        nt.slot = (NameSlot *)calloc(nt.cap, sizeof(NameSlot));
        for (size_t i = 0; i < t->cap; i++)
            if (t->slot[i].used) {
                NameSlot *d = tab_find(&nt, t->slot[i].hash, 1);
                *d = t->slot[i];
            }
        free(t->slot);
        *t = nt;
    }
    if (!t->cap) return NULL;
    for (size_t i = (size_t)(h % t->cap);; i = (i + 1) % t->cap) {
        if (!t->slot[i].used) {
            if (!insert) return NULL;
            t->slot[i].used = 1;
            t->slot[i].hash = h;
            t->slot[i].weight = 0;
            t->slot[i].owner = 0;
            t->used++;
            return &t->slot[i];
        }
        if (t->slot[i].hash == h) return &t->slot[i];
    }
}

static int by_weight_desc(const void *a, const void *b) {
    const NameSlot *x = *(NameSlot *const *)a, *y = *(NameSlot *const *)b;
    if (x->weight != y->weight) return x->weight > y->weight ? -1 : 1;
    return x->hash < y->hash ? -1 : (x->hash > y->hash ? 1 : 0);
}

/* s1 and AS of an output line (impl/paf.c:343-365: tags in the order tp AS tl cn s1 cg); what paf_cmp_by_descending_score compares */
static void line_keys(const char *p, const char *e, int64_t *s1, int64_t *as) {
    *s1 = -1;
    *as = 0;
    int tabs = 0;
    while (p < e && tabs < 12) {
        const char *t = (const char *)memchr(p, '\t', (size_t)(e - p));
        if (!t) return;
        p = t + 1;
        tabs++;
    }
    while (p < e) {
        if (e - p >= 5 && p[2] == ':' && p[4] == ':') {
            if (p[0] == 'c' && p[1] == 'g') return; /* the cigar is the last tag */
            if (p[0] == 'A' && p[1] == 'S') *as = strtoll(p + 5, NULL, 10);
            if (p[0] == 's' && p[1] == '1') *s1 = strtoll(p + 5, NULL, 10);
        }
        const char *t = (const char *)memchr(p, '\t', (size_t)(e - p));
        if (!t) return;
        p = t + 1;
    }
}

typedef struct {
    const char *out, *end, *p; /* the worker's output lines */
    const uint32_t *rows;      /* local record of output line k */
    const uint64_t *idx;       /* global line number of local record j */
    int64_t k, n_rows;
    int64_t s1, as;
    uint64_t gidx;
    const char *line_end;
    size_t out_len, rows_len, idx_len;
} Cursor;

static void cursor_load(Cursor *c) {
    if (c->p >= c->end) {
        c->line_end = NULL;
        return;
    }
    const char *nl = (const char *)memchr(c->p, '\n', (size_t)(c->end - c->p));
    c->line_end = nl ? nl + 1 : c->end;
    line_keys(c->p, c->line_end, &c->s1, &c->as);
    c->gidx = c->k < c->n_rows ? c->idx[c->rows[c->k]] : ~0ull;
}

static const void *map_file(const char *path, size_t *len) {
    *len = 0;
    int fd = open(path, O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0) {
        if (fd >= 0) close(fd);
        return NULL;
    }
    *len = (size_t)sb.st_size;
    const void *p = *len ? mmap(NULL, *len, PROT_READ, MAP_PRIVATE, fd, 0) : (const void *)"";
    close(fd);
    return p == MAP_FAILED ? NULL : p;
}

static int run_tile(const CmdLine *cl, int n, int one_device, const char *in_path, const char *out_path) {
    size_t in_len = 0;
    const char *in = (const char *)map_file(in_path, &in_len);
    if (!in) {
        fprintf(stderr, "paffy tile: cannot open %s\n", in_path);
        return 1;
    }
    /* pass 1: the bytes of every query name's lines */
    NameTab tab = {NULL, 0, 0};
    for (const char *p = in, *end = in + in_len; p < end;) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl + 1 : end;
        const char *t = (const char *)memchr(p, '\t', (size_t)(le - p));
        const size_t nlen = t ? (size_t)(t - p) : (size_t)((nl ? nl : le) - p);
        tab_find(&tab, name_hash(p, nlen), 1)->weight += (int64_t)(le - p);
        p = le;
    }
    /* heaviest name to the lightest worker (contig_partition of paffy_amd/shard.py) */
    {
        NameSlot **order = (NameSlot **)malloc(sizeof(NameSlot *) * (tab.used + 1));
        size_t m = 0;
        for (size_t i = 0; i < tab.cap; i++)
            if (tab.slot[i].used) order[m++] = &tab.slot[i];
        qsort(order, m, sizeof(NameSlot *), by_weight_desc);
        int64_t load[MAX_RANKS] = {0};
        for (size_t i = 0; i < m; i++) {
            int best = 0;
            for (int r = 1; r < n; r++)
                if (load[r] < load[best]) best = r;
            order[i]->owner = best;
            load[best] += order[i]->weight;
        }
        free(order);
    }
    /* pass 2: every line to its worker's spool, with its global line number */
    FILE *fin[MAX_RANKS], *fidx[MAX_RANKS];
    for (int r = 0; r < n; r++) {
        const char *ext[4] = {"in", "out", "rows", "idx"};
        for (int k = 0; k < 4; k++) snprintf(g_spool[r][k], PATH_MAX, "%s/%d.%s", g_spooldir, r, ext[k]);
        fin[r] = fopen(g_spool[r][0], "wx"); /* O_EXCL: inside our own 0700 directory nothing can be there */
        fidx[r] = fopen(g_spool[r][3], "wx");
        if (!fin[r] || !fidx[r]) {
            fprintf(stderr, "paffy tile: cannot write under %s\n", g_tmpdir);
            return 1;
        }
        setvbuf(fin[r], NULL, _IOFBF, 1 << 22);
    }
    uint64_t line_no = 0;
    int64_t spooled[MAX_RANKS] = {0}; /* lines routed to each worker */
    for (const char *p = in, *end = in + in_len; p < end; line_no++) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl + 1 : end;
        const char *t = (const char *)memchr(p, '\t', (size_t)(le - p));
        const size_t nlen = t ? (size_t)(t - p) : (size_t)((nl ? nl : le) - p);
        const int r = tab_find(&tab, name_hash(p, nlen), 0)->owner;
        fwrite(p, 1, (size_t)(le - p), fin[r]);
        if (!nl) fputc('\n', fin[r]); /* a last line without its newline is a record all the same (impl/paf.c:213) */
        fwrite(&line_no, sizeof(line_no), 1, fidx[r]);
        spooled[r]++;
        p = le;
    }
    int werr = 0;
    for (int r = 0; r < n; r++) werr |= fclose(fin[r]) | fclose(fidx[r]);
    munmap((void *)in, in_len);
    free(tab.slot);
    if (werr) {
        fprintf(stderr, "paffy tile: writing the spools under %s failed\n", g_tmpdir);
        return 1;
    }
    /* the workers: one per GPU that has lines to tile. Fewer query names than GPUs (one per-contig split of the input is the reference's
       own workflow, tests/paf_pipeline_test.sh:42-67) or an empty input leave workers without a line: they are not started, and the
       merge below has nothing to take from them -- the reference writes an empty output and exits 0 for an empty input */
    pid_t pids[MAX_RANKS];
    for (int r = 0; r < n; r++) {
        pids[r] = 0;
        if (spooled[r] == 0) continue;
        char **wv = worker_argv("tile", cl, g_spool[r][0], g_spool[r][1]);
        pids[r] = spawn(wv, r, n, one_device, NULL, g_spool[r][2]);
        free(wv);
        if (pids[r] < 0) {
            fprintf(stderr, "paffy: fork failed\n");
            for (int q = 0; q < r; q++)
                if (pids[q] > 0) kill(pids[q], SIGTERM);
            for (int q = 0; q < r; q++)
                if (pids[q] > 0) waitpid(pids[q], NULL, 0);
            return 1;
        }
        g_pids[r] = pids[r];
    }
    int bad_st = 0, any_bad = 0;
    for (int r = 0; r < n; r++) {
        if (pids[r] <= 0) continue;
        int st = 0;
        while (waitpid(pids[r], &st, 0) < 0 && errno == EINTR) {}
        g_pids[r] = 0;
        if (!any_bad && (WIFSIGNALED(st) || WEXITSTATUS(st) != 0)) {
            any_bad = 1;
            bad_st = st;
        }
    }
    if (any_bad) return status_of(bad_st); /* a failing record: nothing is written (impl/paf_tile.c:156-178 writes last) */
    /* merge: every worker's lines are in (s1 desc, AS desc, input order) order already */
    Cursor cur[MAX_RANKS];
    memset(cur, 0, sizeof(cur));
    for (int r = 0; r < n; r++) {
        Cursor *c = &cur[r];
        if (pids[r] <= 0) continue; /* no lines, no worker: line_end stays NULL */
        c->out = (const char *)map_file(g_spool[r][1], &c->out_len);
        c->rows = (const uint32_t *)map_file(g_spool[r][2], &c->rows_len);
        c->idx = (const uint64_t *)map_file(g_spool[r][3], &c->idx_len);
        if (c->out && c->out_len == 0 && !c->rows) { /* a worker that ended well and wrote nothing had nothing to list either */
            c->rows = (const uint32_t *)"";
            c->rows_len = 0;
        }
        if (!c->out || !c->rows || !c->idx) {
            fprintf(stderr, "paffy tile: worker %d left no output\n", r);
            return 1;
        }
        c->p = c->out;
        c->end = c->out + c->out_len;
        c->n_rows = (int64_t)(c->rows_len / sizeof(uint32_t));
        cursor_load(c);
    }
    FILE *out = out_path ? fopen(out_path, "w") : stdout;
    if (!out) {
        fprintf(stderr, "paffy tile: cannot open %s\n", out_path);
        return 1;
    }
    setvbuf(out, NULL, _IOFBF, 1 << 22);
    for (;;) {
        int best = -1;
        for (int r = 0; r < n; r++) {
            const Cursor *c = &cur[r];
            if (!c->line_end) continue;
            if (best < 0) {
                best = r;
                continue;
            }
            const Cursor *b = &cur[best];
            if (c->s1 != b->s1 ? c->s1 > b->s1 : (c->as != b->as ? c->as > b->as : c->gidx < b->gidx)) best = r;
        }
        if (best < 0) break;
        Cursor *c = &cur[best];
        fwrite(c->p, 1, (size_t)(c->line_end - c->p), out);
        c->p = c->line_end;
        c->k++;
        cursor_load(c);
    }
    int rc = fflush(out) != 0;
    if (out != stdout) rc |= fclose(out) != 0;
    return rc;
}

int main(int argc, char **argv) {
    find_worker();
    const char *g = getenv("PAFFY_GPUS");
    int n = g ? atoi(g) : 1;
    if (n > MAX_RANKS) n = MAX_RANKS;
    int shard = n > 1 && argc >= 2 && (is_stream_cmd(argv[1]) || !strcmp(argv[1], "tile"));
    CmdLine cl;
    memset(&cl, 0, sizeof(cl));
    if (shard) {
        parse_cmdline(argc, argv, &cl);
        shard = cl.ok; /* -h, or something getopt_long would reject: the one worker says what the reference says */
    }
    if (!shard) { /* one GPU (or a command that does not shard): this process becomes the worker; it has not touched a GPU */
        free(cl.copy);
        argv[0] = g_worker;
        execv(g_worker, argv);
        fprintf(stderr, "paffy: cannot start %s: %s\n", g_worker, strerror(errno));
        return 127;
    }
    find_tmpdir();
    if (make_spooldir() != 0) {
        fprintf(stderr, "paffy: cannot make a spool directory under %s\n", g_tmpdir);
        return 1;
    }
    g_n = n;
    atexit(cleanup);
    {
        struct sigaction sa;
        memset(&sa, 0, sizeof(sa));
        sa.sa_handler = on_signal;
        sigaction(SIGINT, &sa, NULL);
        sigaction(SIGTERM, &sa, NULL);
        sigaction(SIGHUP, &sa, NULL);
    }
    const int one_device = getenv("PAFFY_ONE_DEVICE") && atoi(getenv("PAFFY_ONE_DEVICE")) != 0;
    const char *in_path = cl.in_path, *out_path = cl.out_path;
    if (!in_path) { /* stdin: to a spool file first, the workers read ranges of it */
        snprintf(g_stdin_spool, sizeof(g_stdin_spool), "%s/stdin", g_spooldir);
        int fd = open(g_stdin_spool, O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
        if (fd < 0 || copy_fd(0, fd) != 0) {
            fprintf(stderr, "paffy: cannot spool the input under %s\n", g_tmpdir);
            return 1;
        }
        close(fd);
        in_path = g_stdin_spool;
    }
    const int rc = !strcmp(argv[1], "tile") ? run_tile(&cl, n, one_device, in_path, out_path) : run_stream(argv[1], &cl, n, one_device, in_path, out_path);
    free(cl.copy);
    return rc;
}
