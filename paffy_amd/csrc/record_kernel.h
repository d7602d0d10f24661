/*
 * record_kernel.h -- one workgroup per PAF record: cigar text -> ops in LDS -> transforms on a
 * view -> byte-exact lines through an LDS ring -> coalesced 16-byte global stores.
 *
 * The same code runs twice per batch: the sizing pass (EMIT = false) computes every record's
 * exact output length and first failing check; after a prefix sum over the lengths the emit
 * pass (EMIT = true) re-derives the record and writes it at its offset. Nothing but the raw
 * text is read from HBM and nothing but the final text is written (algorithmic bytes only).
 *
 * Reference behaviour restated here (paths relative to /root/reference):
 *   cigar_parse impl/paf.c:70-111 | paf_check :427-461 | paf_invert :469-490 |
 *   paf_trim_end_fraction/paf_trim_ends/cigar_trim(_back) :518-598 | paf_shatter :600-663 |
 *   paf_trim_unreliable_tails/_prefix/_ends2/paf_trim_upto :811-953 | paf_write_to_buffer :317-389
 * Transforms never move ops: invert / trims are edits of a View (window, direction, I<->D
 * relabel, shortened end ops) over the parsed array.
 */
/* Included once per workgroup size by record_groups.h (PAFFY_NT, PAFFY_NWAVE, namespace PAFFY_NS); no include guard on purpose. */
namespace PAFFY_NS {

#define INTERNAL_TMPL_TOO_LONG 1u
#define INTERNAL_ROW_TOO_LONG 2u
#define INTERNAL_DIGIT_RUN 4u

__device__ __forceinline__ uint32_t dec_len_u32(uint32_t x);
/* the same for values that are almost always below 10^5 (run lengths): four compares, the other five only when some lane needs them */
__device__ __forceinline__ uint32_t dec_len_short(uint32_t x) {
    uint32_t n = 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u);
    if (__any(x >= 100000u)) n += (x >= 100000u) + (x >= 1000000u) + (x >= 10000000u) + (x >= 100000000u) + (x >= 1000000000u);
    return n;
}

/* ---------------- op stores ---------------- */

struct OpsLds { /* 4-byte ops in LDS (len << 3 | op, len < 2^29), mirrored to HBM for the emit pass */
    uint32_t *p;
    uint32_t *g;     /* HBM mirror of this record's ops */
    uint32_t g_cap;  /* entries of the mirror that belong to this record */
    static constexpr bool kNarrow = true;
    __device__ __forceinline__ void get(uint32_t i, int64_t &len, int &op) const {
        uint32_t w = p[i];
        op = (int)(w & 7u);
        len = (int64_t)(w >> 3);
    }
    __device__ __forceinline__ void set(uint32_t i, int64_t len, int op) const {
        uint32_t w = ((uint32_t)len << 3) | (uint32_t)op;
        p[i] = w;
        if (i < g_cap) g[i] = w;
    }
};
struct OpsGlobal { /* the HBM mirror, read by the emit pass */
    const uint32_t *p;
    bool half; /* round 3: the ops of this record are 2-byte words (len << 3 | op with every length below 8192), at the same base address */
    static constexpr bool kNarrow = true;
    typedef uint32_t raw_t;
    __device__ __forceinline__ raw_t raw(uint32_t i) const { return half ? (raw_t) reinterpret_cast<const uint16_t *>(p)[i] : p[i]; }
    static __device__ __forceinline__ void decode(raw_t w, int64_t &len, int &op) {
        op = (int)(w & 7u);
        len = (int64_t)(w >> 3);
    }
    __device__ __forceinline__ void get(uint32_t i, int64_t &len, int &op) const {
        const uint32_t w = raw(i);
        op = (int)(w & 7u);
        len = (int64_t)(w >> 3);
    }
};
struct OpsCoherent { /* 4-byte ops another wave of this workgroup has just written to HBM: read past the L1 */
    const uint32_t *p;
    static constexpr bool kNarrow = true;
    __device__ __forceinline__ void get(uint32_t i, int64_t &len, int &op) const {
        const uint32_t w = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        op = (int)(w & 7u);
        len = (int64_t)(w >> 3);
    }
};
struct OpsArena { /* 8-byte ops in HBM: the CigarRecord layout, inc/paf.h:61-64 */
    uint64_t *p;
    static constexpr bool kNarrow = false;
    typedef uint64_t raw_t;
    __device__ __forceinline__ raw_t raw(uint32_t i) const { return p[i]; }
    static __device__ __forceinline__ void decode(raw_t w, int64_t &len, int &op) {
        op = (int)(w & 0xffu);
        len = (int64_t)w >> 8;
    }
    __device__ __forceinline__ void get(uint32_t i, int64_t &len, int &op) const {
        uint64_t w = p[i];
        op = (int)(w & 0xffu);
        len = (int64_t)w >> 8; /* 56-bit signed */
    }
    __device__ __forceinline__ void set(uint32_t i, int64_t len, int op) const { p[i] = ((uint64_t)len << 8) | (uint64_t)op; }
};

template <class OPS>
struct View {
    OPS ops;
    uint32_t lo, n;
    bool rev, swp;
    int64_t sub_lo, sub_hi; /* bases cut from the raw first / last op of the window (fixed trim) */
    /*
     * Running sums over the window, kept current by the transforms so that paf_check and the
     * identity statistics need no sweep: tm = bases of M and = ops, tx = bases of X, I and D ops
     * (the matches / mismatches of impl/paf.c:823-828), tq = bases of ops other than D, tt = other than I.
     */
    int64_t tm, tx, tq, tt;
    bool totals_ok;
    __device__ __forceinline__ void get(uint32_t i, int64_t &len, int &op) const {
        uint32_t raw = rev ? lo + n - 1 - i : lo + i;
        ops.get(raw, len, op);
        if (swp) op ^= (int)((0x6u >> op) & 1u) * 3; /* I <-> D */
        if ((sub_lo | sub_hi) != 0) {
            if (raw == lo) len -= sub_lo;
            if (raw == lo + n - 1) len -= sub_hi;
        }
    }
    /* get() for the 4-byte stores, 32-bit and without branches (the sweeps of the identity trim run once per op and thread: the
       general form above costs ~30 instructions there, this one ~10): length and code of view index i */
    __device__ __forceinline__ void get32(uint32_t i, uint32_t &len, uint32_t &op) const {
        static_assert(!std::is_same<OPS, OpsGlobal>::value, "the mirror may hold 2-byte words: read it through raw()");
        const uint32_t raw = rev ? lo + n - 1 - i : lo + i;
        const uint32_t w = ops.p[raw];
        uint32_t o = w & 7u;
        o ^= swp ? ((0x6u >> o) & 1u) * 3u : 0u; /* I <-> D */
        op = o;
        len = (w >> 3) - (raw == lo ? (uint32_t)sub_lo : 0u) - (raw == lo + n - 1 ? (uint32_t)sub_hi : 0u);
    }
    /* sums of the match-type (M, =) and the other lengths of view indices [b, e): order and relabelling do not matter */
    __device__ __forceinline__ void class_sums32(uint32_t b, uint32_t e, uint32_t &m, uint32_t &x) const {
        static_assert(!std::is_same<OPS, OpsGlobal>::value, "the mirror may hold 2-byte words: read it through raw()");
        uint32_t sm = 0, sx = 0;
        if (e > b) {
            const uint32_t r0 = rev ? lo + n - e : lo + b, r1 = r0 + (e - b); /* the same ops as raw indices [r0, r1) */
            for (uint32_t r = r0; r < r1; r++) {
                const uint32_t w = ops.p[r];
                const uint32_t is_m = 0u - ((0x9u >> (w & 7u)) & 1u);
                sm += (w >> 3) & is_m;
                sx += (w >> 3) & ~is_m;
            }
            if ((sub_lo | sub_hi) != 0) { /* shortened end ops (fixed trim before this stage) */
                if (r0 <= lo && lo < r1) {
                    const uint32_t is_m = 0u - ((0x9u >> (ops.p[lo] & 7u)) & 1u);
                    sm -= (uint32_t)sub_lo & is_m;
                    sx -= (uint32_t)sub_lo & ~is_m;
                }
                if (r0 <= lo + n - 1 && lo + n - 1 < r1) {
                    const uint32_t is_m = 0u - ((0x9u >> (ops.p[lo + n - 1] & 7u)) & 1u);
                    sm -= (uint32_t)sub_hi & is_m;
                    sx -= (uint32_t)sub_hi & ~is_m;
                }
            }
        }
        m = sm;
        x = sx;
    }
    /* split form of get(): issue the load early, decode when the value is needed */
    __device__ __forceinline__ uint32_t raw_index(uint32_t i) const { return rev ? lo + n - 1 - i : lo + i; }
    template <class RAW>
    __device__ __forceinline__ void decode(RAW w, uint32_t raw, int64_t &len, int &op) const {
        OPS::decode(w, len, op);
        if (swp) op ^= (int)((0x6u >> op) & 1u) * 3;
        if ((sub_lo | sub_hi) != 0) {
            if (raw == lo) len -= sub_lo;
            if (raw == lo + n - 1) len -= sub_hi;
        }
    }
    __device__ __forceinline__ void drop_front(uint32_t k) {
        if (k == 0) return;
        if (!rev) {
            lo += k;
            sub_lo = 0;
        } else {
            sub_hi = 0;
        }
        n -= k;
        if (n == 0) sub_lo = sub_hi = 0;
    }
    __device__ __forceinline__ void shorten_front(int64_t amt) {
        if (!rev) sub_lo += amt;
        else sub_hi += amt;
    }
    __device__ __forceinline__ void reset(const OPS &o, uint32_t count) {
        ops = o; lo = 0; n = count; rev = false; swp = false; sub_lo = sub_hi = 0;
        tm = tx = tq = tt = 0;
        totals_ok = false;
    }
};

struct RecState {
    int64_t qlen, qs, qe, tlen, ts, te, nmatch, nbases, mapq, score, tile_level, chain_id, chain_score;
    uint32_t qn_off, qn_len, tn_off, tn_len;
    bool same;
    uint8_t type;
    bool has_cigar;
};

struct Shared { /* small workgroup-shared words */
    uint32_t err_pos;
    uint32_t flags;
    int64_t bcast[4];
};

__device__ __forceinline__ int op_code_of(uint32_t c) { /* impl/paf.c:96-103 */
    /* without branches (the compiler turns a chain of comparisons into a tree of divergent branches, ~80 instructions per call in the
       parser's loop): bits 2..4 of the five letters differ -- '=' 7, 'D' 1, 'I' 2, 'M' 3, 'X' 6 -- so they index a nibble table of
       the codes and a byte table of the letters themselves, against which c is checked */
    const uint32_t h = (c >> 2) & 7u;
    const uint32_t code = (0x34ff012fu >> (4u * h)) & 0xfu; /* h: 7 6 5 4 3 2 1 0 -> = X - - M I D - */
    const uint32_t want = ((h & 4u ? 0x3d580000u : 0x4d494400u) >> (8u * (h & 3u))) & 0xffu;
    const bool ok = (c == want) & (want != 0u);
    return ok ? (int)code : -1;
}
__device__ __forceinline__ uint32_t op_char_of(int op) { /* impl/paf.c:372-379 */
    return op == OP_M ? 'M' : op == OP_I ? 'I' : op == OP_D ? 'D' : op == OP_EQ ? '=' : op == OP_X ? 'X' : 'N';
}

/* chunk of the view owned by this thread for whole-record sweeps (odd stride: no LDS bank conflicts) */
__device__ __forceinline__ void sweep_bounds(uint32_t n, uint32_t &b, uint32_t &e) {
    uint32_t c = ((n + PAFFY_NT - 1) / PAFFY_NT) | 1u;
    uint64_t bb = (uint64_t)threadIdx.x * c;
    b = bb < n ? (uint32_t)bb : n;
    e = (bb + c) < n ? (uint32_t)(bb + c) : n;
}

/* ---------------- cigar text -> ops ---------------- */

/*
 * cigar_parse, impl/paf.c:70-111, workgroup-parallel: 4 KiB text tiles staged in LDS; every byte
 * that is not a digit ends an op; the digits before it are read backwards (mod 2^64, like the
 * reference's forward accumulate). Returns the op count; *fits is false when the ops do not
 * fit `cap` or (narrow store) a length needs more than 29 bits. err_pos: smallest text offset
 * holding a character outside MID=X, or cg_off+cg_len for a trailing digit run.
 */
template <class OPS>
__device__ __forceinline__ uint32_t parse_cigar(const uint8_t *in, uint32_t cg_off, uint32_t cg_len, const OPS &ops, uint32_t cap, uint8_t *txt,
                                BlockComm &bc, Shared *sh, bool *fits, uint32_t *err_pos, int64_t (&sums)[4]) {
    const uint32_t tid = threadIdx.x;
    const uint32_t end = cg_off + cg_len;
    const uint32_t a0 = cg_off & ~15u;
    if (tid == 0) {
        sh->err_pos = 0xffffffffu;
        sh->flags = 0;
    }
    uint32_t n = 0;
    int64_t acc_m = 0, acc_x = 0, acc_q = 0, acc_t = 0; /* this lane's share of the view totals (tm, tx, tq, tt) */
    for (uint32_t tb = a0; tb < end; tb += PAFFY_NT * 16) {
        uint4 h = make_uint4(0, 0, 0, 0);
        if (tb != a0 && tid < 2) h = reinterpret_cast<uint4 *>(txt + PAFFY_HALO + PAFFY_NT * 16 - 32)[tid];
        __syncthreads();
        if (tb != a0 && tid < 2) reinterpret_cast<uint4 *>(txt)[tid] = h;
        const uint32_t g = tb + tid * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g < end) v = *reinterpret_cast<const uint4 *>(in + g);
        reinterpret_cast<uint4 *>(txt + PAFFY_HALO)[tid] = v;
        __syncthreads();
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t opmask = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t c = (w[j >> 2] >> ((j & 3) * 8)) & 0xffu;
            uint32_t pos = g + j;
            bool inr = pos >= cg_off && pos < end;
            bool dig = (c - '0') < 10u;
            if (inr && !dig) opmask |= 1u << j;
            if (inr && dig && pos == end - 1) atomicMin(&sh->err_pos, end); /* trailing digits: switch sees NUL */
        }
        int64_t cnt[1] = {(int64_t)__popc(opmask)}, tot[1];
        block_excl_scan<1>(cnt, tot, bc);
        uint32_t idx = n + (uint32_t)cnt[0];
        while (opmask) {
            int j = __ffs((int)opmask) - 1;
            opmask &= opmask - 1;
            uint32_t c = (w[j >> 2] >> ((j & 3) * 8)) & 0xffu;
            int code = op_code_of(c);
            if (code < 0) {
                atomicMin(&sh->err_pos, g + j);
                code = 0;
            }
            int p = (int)(PAFFY_HALO + tid * 16 + j) - 1;
            uint32_t avail = g + j - cg_off; /* cigar bytes before this character */
            uint32_t reach = (uint32_t)(p + 1);
            if (tb == a0 && reach > tid * 16 + j) reach = tid * 16 + j; /* no halo before the first tile */
            uint32_t lim = avail < reach ? avail : reach;
            uint64_t len = 0, pw = 1;
            uint32_t k = 0;
            bool ended = false; /* a non-digit (the previous op letter) closed the run */
            if (lim >= 4) { /* usual case: the four bytes before the letter in one go (independent LDS reads) */
                uint32_t d0 = (uint32_t)txt[p] - '0', d1 = (uint32_t)txt[p - 1] - '0', d2 = (uint32_t)txt[p - 2] - '0',
                         d3 = (uint32_t)txt[p - 3] - '0';
                if (d0 > 9u) { ended = true; }
                else if (d1 > 9u) { len = d0; k = 1; ended = true; }
                else if (d2 > 9u) { len = d0 + 10 * d1; k = 2; ended = true; }
                else if (d3 > 9u) { len = d0 + 10 * d1 + 100 * d2; k = 3; ended = true; }
                else { len = d0 + 10 * d1 + 100 * d2 + 1000 * d3; k = 4; pw = 10000; }
            }
            for (; !ended && k < lim; k++) {
                uint32_t d = (uint32_t)txt[p - (int)k] - '0';
                if (d > 9u) {
                    ended = true;
                    break;
                }
                len += d * pw;
                pw *= 10;
            }
            if (!ended && lim < avail) atomicOr(&sh->flags, INTERNAL_DIGIT_RUN); /* run longer than the halo */
            int64_t l56 = (int64_t)(len << 8) >> 8;
            if (OPS::kNarrow && (l56 < 0 || l56 >= (1ll << 29))) atomicOr(&sh->flags, 0x100u);
            if (idx < cap) ops.set(idx, l56, code);
            idx++;
            if (code == OP_M || code == OP_EQ) acc_m += l56;
            else acc_x += l56;
            if (code != OP_D) acc_q += l56;
            if (code != OP_I) acc_t += l56;
        }
        n += (uint32_t)tot[0];
    }
    sums[0] = acc_m; sums[1] = acc_x; sums[2] = acc_q; sums[3] = acc_t;
    block_sum<4>(sums, bc);
    __syncthreads();
    *err_pos = sh->err_pos;
    uint32_t fl = sh->flags;
    *fits = n <= cap && !(fl & 0x100u);
    __syncthreads();
    return n | ((fl & INTERNAL_DIGIT_RUN) ? 0x80000000u : 0u);
}

/*
 * cigar_parse for every cigar whose numbers have at most seven digits (anything else: the general parser above): the text is staged
 * in LDS 4 KiB at a time, 16 bytes per thread at its natural alignment, and every thread converts the number in front of each op
 * letter in its 16 bytes from the eight bytes before the letter (three aligned LDS reads, SWAR digits; device_util.h) -- one
 * iteration per op instead of one per byte. An op without digits has length 0, as in the reference (impl/paf.c:92-95).
 * Ops go to LDS; the HBM mirror is a coalesced copy at the end. *plain: every op is M, I or D with a length of at least 1 (what
 * paf_shatter's asserts demand, impl/paf.c:635,649).
 * Returns the op count, or 0xffffffff when a number has eight digits or more (nothing done here matters then).
 * (Measured on cfg3, k_size_lds per 131 072 records: byte-serial register parser for cigars up to 8 KiB + tile parser beyond 4.68 ms;
 * this parser 4.26 ms; the same with 32 bytes per thread staged in the top of the op store 4.39 ms -- more barriers per record.)
 */
#define PARSE_LDS_TEXT (16u * PAFFY_NT) /* bytes per round */
__device__ __forceinline__ uint32_t parse_cigar_lds(const uint8_t *in, uint32_t cg_off, uint32_t cg_len, const OpsLds &ops, uint32_t cap, uint8_t *txt, BlockComm &bc,
                                                    Shared *sh, bool *fits, uint32_t *err_pos, int64_t (&sums)[4], bool *plain, bool *mirror16) {
    const uint32_t tid = threadIdx.x;
    const uint32_t end = cg_off + cg_len;
    const uint32_t a0 = cg_off & ~15u;
    if (tid < PAFFY_HALO / 4) reinterpret_cast<uint32_t *>(txt)[tid] = 0; /* nothing in front of the first round */
    if (tid == 0) {
        sh->err_pos = 0xffffffffu;
        sh->flags = 0;
    }
    uint32_t n = 0;
    uint32_t sm = 0, sx = 0, sq = 0, st = 0, flags = 0; /* flags: 1 = a number of eight digits or more, 2 = not plain */
    uint32_t bad_at = 0xffffffffu;                      /* smallest text offset of a letter outside MID=X this thread saw */
    /* the text of a round is requested a round ahead: a round of 16 bytes per thread is short next to the latency of its load */
    uint4 v_ahead = make_uint4(0, 0, 0, 0);
    if (a0 + tid * 16 < end) v_ahead = *reinterpret_cast<const uint4 *>(in + a0 + tid * 16);
    for (uint32_t tb = a0; tb < end; tb += PARSE_LDS_TEXT) {
        uint4 h = make_uint4(0, 0, 0, 0);
        if (tb != a0 && tid < 2) h = reinterpret_cast<uint4 *>(txt + PARSE_LDS_TEXT)[tid]; /* the last 32 bytes become the halo */
        __syncthreads();
        if (tb != a0 && tid < 2) reinterpret_cast<uint4 *>(txt)[tid] = h;
        const uint32_t g = tb + tid * 16;
        const uint4 v = v_ahead;
        reinterpret_cast<uint4 *>(txt + PAFFY_HALO)[tid] = v;
        v_ahead = make_uint4(0, 0, 0, 0);
        if (g + PARSE_LDS_TEXT < end) v_ahead = *reinterpret_cast<const uint4 *>(in + g + PARSE_LDS_TEXT);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t opmask = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t nd = nondigit4(w[j]) >> 7;
            nd = (nd | (nd >> 7) | (nd >> 14) | (nd >> 21)) & 0xfu;
            opmask |= nd << (4 * j);
        }
        {
            const uint32_t first = g < cg_off ? (cg_off - g < 16u ? cg_off - g : 16u) : 0u;
            const uint32_t last = g + 16u > end ? (end > g ? end - g : 0u) : 16u;
            /* a cigar that ends in digits: the reference's switch sees the NUL (impl/paf.c:96-103) */
            if (last > first && g + last == end && !((opmask >> (last - 1)) & 1u)) atomicMin(&sh->err_pos, end);
            opmask &= (last >= 16u ? 0xffffu : ((1u << last) - 1u)) & ~((1u << first) - 1u);
        }
        uint32_t c1[1] = {(uint32_t)__popc(opmask)}, tot[1];
        block_excl_scan_u32<1>(c1, tot, bc); /* its barrier also publishes the staged text */
        /* Two steps per round (round 3; one thread used to convert the numbers of its own 16 bytes in a loop over its letters -- as
           many iterations as the densest 16 bytes of the wave hold letters, up to 16, for a mean of five ops per thread): first every
           thread only leaves the LDS positions of its letters in the op store, at the ops' indices; then the threads take the ops of
           the round in index order, one op per thread and step (ceil(ops / threads) steps, all lanes busy), read the position, convert
           the number in front of it and put the op word where the position was -- and into the HBM mirror, coalesced. */
        uint32_t idx = n + c1[0];
        const uint32_t q0 = PAFFY_HALO + tid * 16u;
        while (opmask) {
            const uint32_t j = (uint32_t)__ffs((int)opmask) - 1u;
            opmask &= opmask - 1u;
            if (idx < cap) ops.p[idx] = q0 + j;
            idx++;
        }
        __syncthreads();
        const uint32_t top = n + tot[0] < cap ? n + tot[0] : cap;
        for (uint32_t i = n + tid; i < top; i += PAFFY_NT) {
            const uint32_t pos = ops.p[i];
            uint32_t kk, c; /* the letter comes with the words the number is read from */
            const uint32_t len = number_before(txt, pos, &kk, &c);
            int code = op_code_of(c);
            const bool bad = code < 0; /* remembered per thread, reported once after the rounds: no atomic in the loop */
            const uint32_t at = tb + (pos - PAFFY_HALO);
            bad_at = bad && at < bad_at ? at : bad_at;
            code = bad ? 0 : code;
            flags |= (kk >= 8u ? 1u : 0u) | (((len == 0u) | (code > OP_D)) ? 2u : 0u);
            const uint32_t word = (len << 3) | (uint32_t)code;
            ops.p[i] = word;
            /* the mirror in 2-byte words (round 3: half the bytes written here and read by the writers); a length of 8192 or more
               anywhere in the record and the mirror is written again below, in 4-byte words */
            if (i < ops.g_cap) reinterpret_cast<uint16_t *>(ops.g)[i] = (uint16_t)word;
            flags |= word > 0xffffu ? 4u : 0u;
            /* masks, not branches: code is 0..4 here (M I D = X) */
            const uint32_t is_m = 0u - ((0x9u >> code) & 1u); /* M or = */
            sm += len & is_m;
            sx += len & ~is_m;
            sq += len & (0u - (uint32_t)(code != OP_D));
            st += len & (0u - (uint32_t)(code != OP_I));
        }
        n += tot[0];
    }
    if (bad_at != 0xffffffffu) atomicMin(&sh->err_pos, bad_at);
    /* one collective: the four sums and the flags (a thread's own sums stay below 2^32: a thread converts at most 36 864 / 64 numbers of seven digits) */
    sums[0] = (int64_t)((uint64_t)sm | ((uint64_t)(flags & 1u) << 42) | ((uint64_t)((flags >> 1) & 1u) << 52));
    sums[1] = (int64_t)((uint64_t)sx | ((uint64_t)((flags >> 2) & 1u) << 42)); /* sums stay below 2^42: 36 864 lengths of seven digits */
    sums[2] = sq; sums[3] = st;
    block_sum<4>(sums, bc);
    const uint64_t packed = (uint64_t)sums[0], packed1 = (uint64_t)sums[1];
    sums[0] = (int64_t)(packed & ((1ull << 42) - 1ull));
    sums[1] = (int64_t)(packed1 & ((1ull << 42) - 1ull));
    *err_pos = sh->err_pos; /* written before the barrier of the collective */
    if ((packed >> 42) & 0x3ffull) return 0xffffffffu;
    *plain = ((packed >> 52) & 0x3ffull) == 0;
    *fits = n <= cap;
    const bool wide = ((packed1 >> 42) & 0x3ffull) != 0;
    *mirror16 = !wide;
    if (wide && n <= cap) { /* every op is still in LDS: the mirror once more, as 4-byte words (behind the collective's barrier: the 2-byte stores are done) */
        const uint32_t top = n < ops.g_cap ? n : ops.g_cap;
        for (uint32_t i = tid; i < top; i += PAFFY_NT) ops.g[i] = ops.p[i];
    }
    __syncthreads(); /* the text area and the error word are free again */
    return n;
}

/* Sequential fallback for digit runs longer than the LDS halo (leading zeros etc.). */
template <class OPS>
__device__ __forceinline__ uint32_t parse_cigar_serial(const uint8_t *in, uint32_t cg_off, uint32_t cg_len, const OPS &ops, uint32_t cap, Shared *sh,
                                       bool *fits, uint32_t *err_pos) {
    if (threadIdx.x == 0) {
        uint32_t n = 0, ep = 0xffffffffu, wide = 0;
        uint32_t p = cg_off, end = cg_off + cg_len;
        while (p < end) {
            uint64_t len = 0;
            while (p < end && (uint32_t)(in[p] - '0') < 10u) len = len * 10 + (uint32_t)(in[p++] - '0');
            int code = p < end ? op_code_of(in[p]) : -1;
            if (code < 0) {
                if (ep == 0xffffffffu) ep = p;
                code = 0;
                if (p >= end) break;
            }
            int64_t l56 = (int64_t)(len << 8) >> 8;
            if (OPS::kNarrow && (l56 < 0 || l56 >= (1ll << 29))) wide = 1;
            if (n < cap) ops.set(n, l56, code);
            n++;
            p++;
        }
        sh->err_pos = ep;
        sh->flags = wide;
        sh->bcast[0] = n;
    }
    __syncthreads();
    uint32_t n = (uint32_t)sh->bcast[0];
    *err_pos = sh->err_pos;
    *fits = n <= cap && !sh->flags;
    __syncthreads();
    return n;
}

/* ---------------- transforms on a view ---------------- */

__device__ __forceinline__ void invert_state(RecState &s) { /* paf_invert, impl/paf.c:469-474 */
    int64_t t;
    uint32_t u;
    t = s.qs; s.qs = s.ts; s.ts = t;
    t = s.qe; s.qe = s.te; s.te = t;
    t = s.qlen; s.qlen = s.tlen; s.tlen = t;
    u = s.qn_off; s.qn_off = s.tn_off; s.tn_off = u;
    u = s.qn_len; s.qn_len = s.tn_len; s.tn_len = u;
}
template <class OPS>
__device__ __forceinline__ void invert_view(const RecState &s, View<OPS> &v) { /* impl/paf.c:476-489 */
    v.swp = !v.swp;
    if (!s.same) v.rev = !v.rev;
    int64_t t = v.tq; /* I and D trade places */
    v.tq = v.tt;
    v.tt = t;
}

/* one sweep that (re)establishes the running sums of the view */
template <class OPS>
__device__ __forceinline__ void ensure_totals(View<OPS> &v, BlockComm &bc) {
    if (v.totals_ok) return;
    uint32_t b, e;
    sweep_bounds(v.n, b, e);
    int64_t a[4] = {0, 0, 0, 0};
    for (uint32_t i = b; i < e; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        if (op == OP_EQ || op == OP_M) a[0] += len;
        else a[1] += len;
        if (op != OP_D) a[2] += len;
        if (op != OP_I) a[3] += len;
    }
    block_sum<4>(a, bc);
    v.tm = a[0]; v.tx = a[1]; v.tq = a[2]; v.tt = a[3];
    v.totals_ok = true;
}

/* paf_check, impl/paf.c:427-461. Returns 0 or the PAFFY_ERR_CHECK_* code. */
template <class OPS>
__device__ __forceinline__ int check_record(const RecState &s, View<OPS> &v, BlockComm &bc) {
    if (s.qs < 0 || s.qs >= s.qlen) return PAFFY_ERR_CHECK_QSTART;
    if (s.qs > s.qe || s.qe > s.qlen) return PAFFY_ERR_CHECK_QEND;
    if (s.ts < 0 || s.ts >= s.tlen) return PAFFY_ERR_CHECK_TSTART;
    if (s.ts > s.te || s.te > s.tlen) return PAFFY_ERR_CHECK_TEND;
    if (s.has_cigar) {
        ensure_totals(v, bc);
        if (v.tq != s.qe - s.qs) return PAFFY_ERR_CHECK_CIGAR_Q;
        if (v.tt != s.te - s.ts) return PAFFY_ERR_CHECK_CIGAR_T;
    }
    return 0;
}

/* float32 quotient widened to double: `((float)a)/(a + b)` of impl/paf.c:832,886,923,937 */
__device__ __forceinline__ double ratio_f32(int64_t num, int64_t den) {
    /* both conversions round to nearest even; below 2^32 one v_cvt_f32_u32 each does the same job */
    if ((((uint64_t)num | (uint64_t)den) >> 32) == 0)
        return (double)__fdiv_rn(__uint2float_rn((uint32_t)num), __uint2float_rn((uint32_t)den));
    return (double)__fdiv_rn(__ll2float_rn(num), __ll2float_rn(den));
}

/* matches / mismatches of the whole view: paf_trim_unreliable_ends2(.., 0, 1, -1), impl/paf.c:811-840 */
template <class OPS>
__device__ __forceinline__ void match_stats(View<OPS> &v, int64_t &m, int64_t &x, BlockComm &bc) {
    ensure_totals(v, bc);
    m = v.tm;
    x = v.tx;
}

/* paf_trim_unreliable_prefix + paf_trim_upto, impl/paf.c:842-904 (thresholds arrive as float32). */
template <class OPS>
__device__ __forceinline__ void trim_prefix(RecState &s, View<OPS> &v, float thr_f, float id_f, int64_t max_trim, BlockComm &bc, Shared *sh) {
    const double thr = (double)thr_f, idd = (double)id_f;
    uint32_t b, e;
    sweep_bounds(v.n, b, e);
    /* sweep A: per-thread (matches, mismatches) -> exclusive prefix */
    int64_t c[2] = {0, 0}, tot[2];
    for (uint32_t i = b; i < e; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        if (op == OP_EQ || op == OP_M) c[0] += len;
        else c[1] += len;
    }
    const int64_t chunk_x = c[1];
    block_excl_scan<2>(c, tot, bc);
    /* sweep B: last index (while cumulative <= max_trim) whose prefix identity < threshold.
       Inside a chunk every prefix identity is at least m0 / (m0 + x0 + chunk_x) (m0, x0: sums before the chunk;
       worst case all of the chunk's mismatches first). float32 conversions and the divide are off by < 2e-7
       relative, so a chunk whose bound clears the threshold by 1e-5 cannot hold a hit: whole waves deep inside a
       record skip the sweep and its divides. */
    int64_t cm = c[0], cx = c[1], found = -1;
    const bool may_hit = e > b && !(c[0] > 0 && (double)c[0] >= thr * 1.00001 * (double)(c[0] + c[1] + chunk_x));
    if (may_hit)
    for (uint32_t i = b; i < e; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        if (op == OP_EQ || op == OP_M) cm += len;
        else cx += len;
        if (max_trim >= 0 && cm + cx > max_trim) break;
        if (ratio_f32(cm, cm + cx) < thr) found = i;
    }
    int64_t trim_idx = block_max_i64(found, bc);
    if (trim_idx < 0) return;
    /* inclusive cumulative at trim_idx, broadcast by its owner */
    if (trim_idx >= (int64_t)b && trim_idx < (int64_t)e) {
        int64_t am = c[0], ax = c[1];
        for (uint32_t i = b; i <= (uint32_t)trim_idx; i++) {
            int64_t len;
            int op;
            v.get(i, len, op);
            if (op == OP_EQ || op == OP_M) am += len;
            else ax += len;
        }
        sh->bcast[0] = am;
        sh->bcast[1] = ax;
    }
    __syncthreads();
    const int64_t tm = sh->bcast[0], tx = sh->bcast[1];
    __syncthreads();
    /* sweep C: smallest i <= trim_idx whose suffix [i, trim_idx] has identity >= identity */
    int64_t em = c[0], ex = c[1], best = INT64_MAX;
    for (uint32_t i = b; i < e && (int64_t)i <= trim_idx; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        int64_t sm = tm - em, sx = tx - ex;
        if (best == INT64_MAX && ratio_f32(sm, sm + sx) >= idd) best = i;
        if (op == OP_EQ || op == OP_M) em += len;
        else ex += len;
    }
    best = block_min_i64(best, bc);
    int64_t count = best != INT64_MAX ? best : trim_idx + 1;
    if (count <= 0) return;
    /* paf_trim_upto: advance coordinates over the dropped ops (and keep the running sums current) */
    int64_t d[4] = {0, 0, 0, 0};
    for (uint32_t i = b; i < e && (int64_t)i < count; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        if (op != OP_I) d[0] += len;
        if (op != OP_D) d[1] += len;
        if (op == OP_EQ || op == OP_M) d[2] += len;
        else d[3] += len;
    }
    block_sum<4>(d, bc);
    s.ts += d[0];
    if (s.same) s.qs += d[1];
    else s.qe -= d[1];
    v.tt -= d[0]; v.tq -= d[1]; v.tm -= d[2]; v.tx -= d[3];
    v.drop_front((uint32_t)count);
}

/* The same pass for records whose match + mismatch total fits 31 bits: 32-bit sums, scans and quotient operands. */
__device__ __forceinline__ double ratio_f32_u32(uint32_t num, uint32_t den) {
    return (double)__fdiv_rn(__uint2float_rn(num), __uint2float_rn(den));
}
template <class OPS>
__device__ __forceinline__ void trim_prefix32(RecState &s, View<OPS> &v, float thr_f, float id_f, int64_t max_trim, BlockComm &bc, Shared *sh) {
    const double thr = (double)thr_f, idd = (double)id_f;
    uint32_t b, e;
    sweep_bounds(v.n, b, e);
    uint32_t c[2], tot[2];
    v.class_sums32(b, e, c[0], c[1]);
    const uint32_t chunk_x = c[1];
    block_excl_scan_u32<2>(c, tot, bc);
    const uint32_t lane = threadIdx.x & 63u;
    int32_t found = -1;
    const bool may_hit = e > b && !(c[0] > 0 && (double)c[0] >= thr * 1.00001 * (double)(c[0] + c[1] + chunk_x));
    /* The chunks that may hold a hit (usually the first one or two of the record) are walked by the whole wave, one op per lane --
       a thread walking its own chunk made the wave pay chunk-length iterations of a loop with a float division for one or two busy
       lanes (round 3: that loop and the one below were most of the trim's time). The last index with cumulative <= max_trim and
       prefix identity < threshold is wanted (impl/paf.c:820-838: the cumulative sums only grow, so "break once above max_trim"
       excludes exactly the indices whose own cumulative sum is above it): chunks from the last one down, first hit wins. */
    {
        unsigned long long flagged = __ballot(may_hit);
        while (flagged) {
            const int t = 63 - __clzll((long long)flagged);
            flagged &= ~(1ull << t);
            const uint32_t cb = (uint32_t)__shfl((int)b, t), ce = (uint32_t)__shfl((int)e, t);
            uint32_t pm = (uint32_t)__shfl((int)c[0], t), px = (uint32_t)__shfl((int)c[1], t);
            int32_t hit = -1;
            for (uint32_t p0 = cb; p0 < ce; p0 += 64u) {
                const uint32_t i = p0 + lane;
                uint32_t len = 0, op = 0;
                if (i < ce) v.get32(i, len, op);
                const uint32_t is_m = 0u - ((0x9u >> op) & 1u);
                const uint32_t im = wave_incl_scan_u32(len & is_m), ix = wave_incl_scan_u32(len & ~is_m);
                const uint32_t cm = pm + im, cx = px + ix;
                const bool ok = i < ce && !(max_trim >= 0 && (int64_t)(cm + cx) > max_trim) && ratio_f32_u32(cm, cm + cx) < thr;
                const unsigned long long hb = __ballot(ok);
                if (hb) hit = (int32_t)(p0 + 63u - (uint32_t)__clzll((long long)hb));
                pm += wave_last_u32(im);
                px += wave_last_u32(ix);
            }
            if (hit >= 0) {
                found = hit;
                break;
            }
        }
    }
    const int32_t trim_idx = block_max_idx(found, bc);
    if (trim_idx < 0) return;
    if (trim_idx >= (int32_t)b && trim_idx < (int32_t)e) {
        uint32_t am, ax;
        v.class_sums32(b, (uint32_t)trim_idx + 1u, am, ax);
        sh->bcast[0] = c[0] + am;
        sh->bcast[1] = c[1] + ax;
    }
    __syncthreads();
    const uint32_t tm = (uint32_t)sh->bcast[0], tx = (uint32_t)sh->bcast[1];
    __syncthreads();
    /* smallest index <= trim_idx whose suffix [i, trim_idx] has identity >= identity (impl/paf.c:879-890): the same way, chunks from
       the first one up, first hit wins */
    uint32_t best = 0xffffffffu;
    {
        unsigned long long flagged = __ballot(e > b && (int32_t)b <= trim_idx);
        while (flagged) {
            const int t = __ffsll((long long)flagged) - 1;
            flagged &= flagged - 1ull;
            const uint32_t cb = (uint32_t)__shfl((int)b, t);
            uint32_t ce = (uint32_t)__shfl((int)e, t);
            if ((int32_t)ce > trim_idx + 1) ce = (uint32_t)trim_idx + 1u;
            uint32_t pm = (uint32_t)__shfl((int)c[0], t), px = (uint32_t)__shfl((int)c[1], t);
            uint32_t hit = 0xffffffffu;
            for (uint32_t p0 = cb; p0 < ce && hit == 0xffffffffu; p0 += 64u) {
                const uint32_t i = p0 + lane;
                uint32_t len = 0, op = 0;
                if (i < ce) v.get32(i, len, op);
                const uint32_t is_m = 0u - ((0x9u >> op) & 1u);
                const uint32_t vm = len & is_m, vx = len & ~is_m;
                const uint32_t im = wave_incl_scan_u32(vm), ix = wave_incl_scan_u32(vx);
                const uint32_t sm = tm - (pm + im - vm), sx = tx - (px + ix - vx); /* sums of [i, trim_idx] */
                const bool ok = i < ce && ratio_f32_u32(sm, sm + sx) >= idd;
                const unsigned long long hb = __ballot(ok);
                if (hb) hit = p0 + (uint32_t)__ffsll((long long)hb) - 1u;
                pm += wave_last_u32(im);
                px += wave_last_u32(ix);
            }
            if (hit != 0xffffffffu) {
                best = hit;
                break;
            }
        }
    }
    best = block_min_u32(best, bc);
    const uint32_t count = best != 0xffffffffu ? best : (uint32_t)trim_idx + 1u;
    if (count == 0) return;
    uint32_t d[4] = {0, 0, 0, 0};
    for (uint32_t i = b; i < e && i < count; i++) {
        uint32_t len, op;
        v.get32(i, len, op);
        const uint32_t is_m = 0u - ((0x9u >> op) & 1u);
        d[0] += len & (0u - (uint32_t)(op != (uint32_t)OP_I));
        d[1] += len & (0u - (uint32_t)(op != (uint32_t)OP_D));
        d[2] += len & is_m;
        d[3] += len & ~is_m;
    }
    block_sum_u32<4>(d, bc);
    s.ts += d[0];
    if (s.same) s.qs += d[1];
    else s.qe -= d[1];
    v.tt -= d[0]; v.tq -= d[1]; v.tm -= d[2]; v.tx -= d[3];
    v.drop_front(count);
}

/* paf_trim_unreliable_tails, impl/paf.c:906-953. Returns 0 or PAFFY_ERR_TRIM_IDENTITY_ASSERT. */
template <class OPS>
__device__ __forceinline__ int trim_identity(RecState &s, View<OPS> &v, float score_fraction, float max_fraction, BlockComm &bc, Shared *sh) {
    int64_t m, x;
    match_stats(v, m, x, bc);
    const double identity = ratio_f32(m, m + x);
    const double thr = __dsub_rn(identity, __dmul_rn(identity, (double)score_fraction));
    const int64_t max_trim = __float2ll_rz(__fmul_rn(__ll2float_rn(m + x), max_fraction));
    const float thr_f = __double2float_rn(thr), id_f = __double2float_rn(identity);
    /* paf_invert only reverses the cigar of '-' records (impl/paf.c:487): for a '+' record the second
       pass scans the very same op sequence with the same thresholds, so when the first pass removed
       nothing the second one cannot either */
    const uint32_t n_before = v.n;
    const bool small_sums = OPS::kNarrow && m >= 0 && x >= 0 && m + x < 0x7fffffffll; /* 4-byte ops have lengths >= 0, all in these sums: every partial sum fits 31 bits */
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            if (s.same && v.n == n_before) break;
            invert_state(s);
            invert_view(s, v);
        }
        if (small_sums) trim_prefix32(s, v, thr_f, id_f, max_trim, bc, sh);
        else trim_prefix(s, v, thr_f, id_f, max_trim, bc, sh);
        if (pass == 1) {
            invert_state(s);
            invert_view(s, v);
        }
    }
    int64_t m2, x2;
    match_stats(v, m2, x2, bc);
    const double final_identity = ratio_f32(m2, m2 + x2);
    return final_identity >= identity ? 0 : PAFFY_ERR_TRIM_IDENTITY_ASSERT;
}

__device__ __forceinline__ bool is_aligned_op(int op) { return op == OP_M || op == OP_EQ || op == OP_X; }

/*
 * cigar_trim / cigar_trim_back, impl/paf.c:518-576, on the front of the view (call on the
 * reversed view for the back). Ops are popped while the front op is an indel or fewer than
 * `end` aligned bases are gone; the op that crosses `end` is shortened. dq/dt: bases consumed.
 */
template <class OPS>
__device__ __forceinline__ void trim_front_fixed(View<OPS> &v, int64_t end, int64_t &dq, int64_t &dt, BlockComm &bc, Shared *sh) {
    uint32_t b, e;
    sweep_bounds(v.n, b, e);
    int64_t c[1] = {0}, tot[1];
    for (uint32_t i = b; i < e; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        if (is_aligned_op(op)) c[0] += len;
    }
    block_excl_scan<1>(c, tot, bc);
    int64_t tb = c[0], stop = INT64_MAX, stop_tb = 0;
    for (uint32_t i = b; i < e; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        if (is_aligned_op(op)) {
            if (!(tb < end) || tb + len > end) {
                stop = i;
                stop_tb = tb;
                break;
            }
            tb += len;
        }
    }
    int64_t s_idx = block_min_i64(stop, bc);
    if (s_idx != INT64_MAX && s_idx >= (int64_t)b && s_idx < (int64_t)e) sh->bcast[0] = stop_tb;
    __syncthreads();
    int64_t tb_s = s_idx != INT64_MAX ? sh->bcast[0] : 0;
    __syncthreads();
    uint32_t drop = s_idx == INT64_MAX ? v.n : (uint32_t)s_idx;
    int64_t d[2] = {0, 0};
    for (uint32_t i = b; i < e && i < drop; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        if (op != OP_D) d[0] += len;
        if (op != OP_I) d[1] += len;
    }
    block_sum<2>(d, bc);
    dq = d[0];
    dt = d[1];
    v.drop_front(drop);
    if (s_idx != INT64_MAX && tb_s < end) {
        int64_t amt = end - tb_s;
        v.shorten_front(amt);
        dq += amt;
        dt += amt;
    }
}

/* paf_trim_end_fraction + paf_trim_ends, impl/paf.c:578-598. */
template <class OPS>
__device__ __forceinline__ int trim_fixed(RecState &s, View<OPS> &v, float pct, BlockComm &bc, Shared *sh, bool by_count = false, int64_t count = 0) {
    if (!by_count && !(pct >= 0.0f && pct <= 1.0f)) return PAFFY_ERR_TRIM_FIXED_ASSERT;
    uint32_t b, e;
    sweep_bounds(v.n, b, e);
    int64_t a[1] = {0};
    for (uint32_t i = b; i < e; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        if (is_aligned_op(op)) a[0] += len;
    }
    block_sum<1>(a, bc);
    const int64_t end = by_count ? count : __double2ll_rz((double)__fmul_rn(__ll2float_rn(a[0]), pct) / 2.0); /* paf_trim_ends / paf_trim_end_fraction */
    if (!s.has_cigar) return PAFFY_ERR_NULL_CIGAR;
    int64_t dq, dt;
    trim_front_fixed(v, end, dq, dt, bc, sh);
    if (s.same) s.qs += dq;
    else s.qe -= dq;
    s.ts += dt;
    v.rev = !v.rev;
    trim_front_fixed(v, end, dq, dt, bc, sh);
    v.rev = !v.rev;
    if (s.same) s.qe -= dq;
    else s.qs += dq;
    s.te -= dt;
    v.totals_ok = false;
    return 0;
}

/* ---------------- stages that rebuild the op array ---------------- */

struct MirrorDst { /* 4-byte ops into the HBM mirror of an LDS-class record */
    uint32_t *g;
    uint32_t cap;
    static constexpr bool kNarrow = true;
    __device__ __forceinline__ void set(uint32_t i, int64_t len, int op) const {
        if (i < cap) g[i] = ((uint32_t)len << 3) | (uint32_t)op;
    }
};

/*
 * paf_remove_mismatches, impl/paf.c:786-809: every maximal run of M / = / X ops becomes one M
 * (lengths summed in the 56-bit field), I and D ops are copied. Workgroup-parallel: an op is
 * a head unless it and its predecessor are both match-type; heads are numbered by a scan and
 * each match-type head walks its run. Writes the new array to `dst`; returns its length.
 * *narrow_ok is cleared when a merged length does not fit a 4-byte op.
 */
template <class OPS, class DST>
__device__ __forceinline__ uint32_t merge_match_runs(const View<OPS> &v, const DST &dst, BlockComm &bc, Shared *sh, bool *narrow_ok) {
    if (threadIdx.x == 0) sh->flags = 0;
    uint32_t b, e;
    sweep_bounds(v.n, b, e);
    bool prev_mt = false;
    if (b > 0 && b < v.n) {
        int64_t len;
        int op;
        v.get(b - 1, len, op);
        prev_mt = is_aligned_op(op);
    }
    int64_t cnt[1] = {0}, tot[1];
    {
        bool pm = prev_mt;
        for (uint32_t i = b; i < e; i++) {
            int64_t len;
            int op;
            v.get(i, len, op);
            bool mt = is_aligned_op(op);
            if (!(mt && pm)) cnt[0]++;
            pm = mt;
        }
    }
    block_excl_scan<1>(cnt, tot, bc);
    uint32_t o = (uint32_t)cnt[0];
    bool pm = prev_mt, wide = false;
    for (uint32_t i = b; i < e; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        bool mt = is_aligned_op(op);
        if (!(mt && pm)) {
            if (mt) {
                int64_t sum = len;
                for (uint32_t j = i + 1; j < v.n; j++) {
                    int64_t l2;
                    int o2;
                    v.get(j, l2, o2);
                    if (!is_aligned_op(o2)) break;
                    sum += l2;
                }
                sum = (int64_t)((uint64_t)sum << 8) >> 8;
                if (DST::kNarrow && (sum < 0 || sum >= (1ll << 29))) wide = true;
                dst.set(o, sum, OP_M);
            } else {
                dst.set(o, len, op);
            }
            o++;
        }
        pm = mt;
    }
    if (wide) atomicOr(&sh->flags, 0x100u);
    __syncthreads();
    *narrow_ok = !(sh->flags & 0x100u);
    __syncthreads();
    return (uint32_t)tot[0];
}

/* stString_reverseComplementChar [sonLib, absent]: A<->T, C<->G in both cases, identity otherwise
 * (parity unpinned for other letters). */
__device__ __forceinline__ uint32_t rc_base(uint32_t c) {
    switch (c) {
        case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
        case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
        default: return c;
    }
}
__device__ __forceinline__ uint32_t up_base(uint32_t c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; } /* toupper, C locale */

/*
 * Match mask of up to eight alignment columns: bit j = toupper(T[tj + j]) == toupper(query base j), the query base being
 * Q[qoff + j] on the + strand and the complement of Q[qoff - j] on the - strand (impl/paf.c:752-757). Eight bytes of each
 * sequence per load (byte-aligned loads; the sequence store is padded behind its last byte); only a - strand chunk that
 * would reach in front of the store falls back to single bytes.
 */
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
__device__ __forceinline__ uint32_t match_mask8(const uint8_t *base, const uint8_t *Q, const uint8_t *T, int64_t qoff, int64_t tj, uint32_t nb, bool same) {
    const uint64_t tw = *reinterpret_cast<const u64_unaligned *>(T + tj);
    uint32_t m = 0;
    if (same) {
        const uint64_t qw = *reinterpret_cast<const u64_unaligned *>(Q + qoff);
#pragma unroll
        for (int j = 0; j < 8; j++) m |= (up_base((uint32_t)(tw >> (8 * j)) & 0xffu) == up_base((uint32_t)(qw >> (8 * j)) & 0xffu) ? 1u : 0u) << j;
    } else if (Q + qoff - 7 >= base) {
        const uint64_t qw = *reinterpret_cast<const u64_unaligned *>(Q + qoff - 7); /* byte 7 - j is Q[qoff - j] */
#pragma unroll
        for (int j = 0; j < 8; j++)
            m |= (up_base((uint32_t)(tw >> (8 * j)) & 0xffu) == up_base(rc_base((uint32_t)(qw >> (8 * (7 - j))) & 0xffu)) ? 1u : 0u) << j;
    } else {
        for (uint32_t j = 0; j < nb; j++) m |= (up_base((uint32_t)(tw >> (8 * j)) & 0xffu) == up_base(rc_base(Q[qoff - (int64_t)j])) ? 1u : 0u) << j;
    }
    return m & ((1u << nb) - 1u);
}

/*
 * paf_encode_mismatches, impl/paf.c:739-784: each M op becomes its maximal runs of matching (=)
 * and mismatching (X) columns against the two sequences; other ops are copied. Two walks over the
 * bases (count, then fill) with one lane per op chunk. The new array goes to a fresh arena block.
 * Returns 0 or PAFFY_ERR_SEQ_RANGE; *n_out = new op count; *blk = arena offset (UINT64_MAX: no room).
 */
template <class OPS>
__device__ __forceinline__ int encode_mismatch_runs(const KParams &P, const RecState &s, const View<OPS> &v, const uint8_t *Q, int64_t qseq_len,
                                    const uint8_t *T, int64_t tseq_len, BlockComm &bc, Shared *sh, uint32_t *n_out, uint64_t *blk) {
    uint32_t b, e;
    sweep_bounds(v.n, b, e);
    int64_t c[2] = {0, 0}, tot[2];
    for (uint32_t i = b; i < e; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        if (op != OP_D) c[0] += len;
        if (op != OP_I) c[1] += len;
    }
    block_excl_scan<2>(c, tot, bc);
    /* walk 1: count */
    int64_t cnt[1] = {0}, ctot[1], bad = INT64_MAX;
    {
        int64_t qi = c[0], tj = s.ts + c[1];
        for (uint32_t i = b; i < e; i++) {
            int64_t len;
            int op;
            v.get(i, len, op);
            if (op == OP_M) {
                const int64_t qoff = s.same ? s.qs + qi : s.qe - (qi + 1);
                const bool in_range = len <= 0 || (tj >= 0 && tj + len <= tseq_len &&
                                                   (s.same ? (qoff >= 0 && qoff + len <= qseq_len) : (qoff < qseq_len && qoff - (len - 1) >= 0)));
                if (!in_range) {
                    if (bad == INT64_MAX) bad = i;
                } else {
                    uint32_t prev = 0; /* match bit of the column before the chunk */
                    for (int64_t k = 0; k < len; k += 8) {
                        const uint32_t nb = len - k < 8 ? (uint32_t)(len - k) : 8u;
                        const uint32_t m = match_mask8(P.seq_base, Q, T, s.same ? qoff + k : qoff - k, tj + k, nb, s.same);
                        uint32_t starts = (m ^ ((m << 1) | prev)) & ((1u << nb) - 1u); /* columns that differ from the one before */
                        if (k == 0) starts |= 1u;
                        cnt[0] += __popc(starts);
                        prev = (m >> (nb - 1)) & 1u;
                    }
                }
            } else {
                cnt[0]++;
            }
            if (op != OP_D) qi += len;
            if (op != OP_I) tj += len;
        }
    }
    bad = block_min_i64(bad, bc);
    block_excl_scan<1>(cnt, ctot, bc);
    *n_out = (uint32_t)ctot[0];
    if (bad != INT64_MAX) return PAFFY_ERR_SEQ_RANGE;
    if (threadIdx.x == 0) sh->bcast[3] = (int64_t)atomicAdd(&P.info->arena_used, (unsigned long long)ctot[0]);
    __syncthreads();
    const uint64_t off = (uint64_t)sh->bcast[3];
    __syncthreads();
    if (off + (uint64_t)ctot[0] > P.arena_cap) { /* no room: the host grows the arena and repeats the pass */
        *blk = ~0ull;
        return 0;
    }
    *blk = off;
    OpsArena dst{P.arena + off};
    /* walk 2: fill */
    {
        uint32_t o = (uint32_t)cnt[0];
        int64_t qi = c[0], tj = s.ts + c[1];
        for (uint32_t i = b; i < e; i++) {
            int64_t len;
            int op;
            v.get(i, len, op);
            if (op == OP_M) {
                const int64_t qoff = s.same ? s.qs + qi : s.qe - (qi + 1);
                uint32_t prev = 0;
                int64_t run_start = 0;
                for (int64_t k = 0; k < len; k += 8) {
                    const uint32_t nb = len - k < 8 ? (uint32_t)(len - k) : 8u;
                    const uint32_t m = match_mask8(P.seq_base, Q, T, s.same ? qoff + k : qoff - k, tj + k, nb, s.same);
                    uint32_t starts = (m ^ ((m << 1) | prev)) & ((1u << nb) - 1u);
                    if (k == 0) starts &= ~1u; /* the first column opens the first run: nothing to close */
                    while (starts) { /* a run ends in front of every other start */
                        const int j = __ffs((int)starts) - 1;
                        starts &= starts - 1;
                        const uint32_t before = j ? (m >> (j - 1)) & 1u : prev;
                        dst.set(o++, k + j - run_start, before ? OP_EQ : OP_X);
                        run_start = k + j;
                    }
                    prev = (m >> (nb - 1)) & 1u;
                }
                if (len > 0) dst.set(o++, len - run_start, prev ? OP_EQ : OP_X);
            } else {
                dst.set(o++, len, op);
            }
            if (op != OP_D) qi += len;
            if (op != OP_I) tj += len;
        }
    }
    __syncthreads();
    return 0;
}

/*
 * The same encoding for LDS-class records, wave-parallel over 16-column chunks.
 *
 * A wave owns the view ops [64 * w * chunk, 64 * (w + 1) * chunk) and walks them in windows of 64 (one lane per op: offsets
 * on both sequences by a wave scan, range check). The window's work is a list of ITEMS in op order -- every M op contributes
 * ceil(len / 16) chunks of 16 alignment columns, every other op one item -- and the lanes take 64 consecutive items at a
 * time (item -> op by a binary search over the lanes' item offsets), so that long and short ops fill the wave alike.
 * A chunk's match mask comes from two 16-byte loads compared as packed bytes (toupper and complement as SWAR on 32-bit
 * words). A column starts a run when its match bit differs from the column before (the chunk before is the lane before;
 * the first column of an op always starts one), so the number of runs is a sum over items and the run that ENDS in front of
 * a start needs only the position of the previous start: the nearest lane before with any start (a max scan), carried
 * over iterations. Walk 1 counts, walk 2 writes 4-byte ops at their final index.
 */
__device__ __forceinline__ uint32_t upper4(uint32_t w) { /* toupper of four bytes (C locale: a-z only) */
    const uint32_t hm = w & 0x7f7f7f7fu;
    const uint32_t m = (hm + 0x1f1f1f1fu) & ~(hm + 0x05050505u) & ~w & 0x80808080u; /* 0x61 <= byte <= 0x7a */
    return w ^ (m >> 2);
}
__device__ __forceinline__ uint32_t nonzero4(uint32_t x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; }
__device__ __forceinline__ uint32_t comp4(uint32_t x) { /* complement of four upper-case bytes: A<->T, C<->G, others kept */
    const uint32_t at = ~(nonzero4(x ^ 0x41414141u) & nonzero4(x ^ 0x54545454u)) & 0x80808080u;
    const uint32_t cg = ~(nonzero4(x ^ 0x43434343u) & nonzero4(x ^ 0x47474747u)) & 0x80808080u;
    const uint32_t a = at >> 7;
    return x ^ (a | (a << 2) | (a << 4)) ^ (cg >> 5);
}
__device__ __forceinline__ uint32_t equal4(uint32_t a, uint32_t b) { /* bit j = byte j equal */
    const uint32_t r = (~nonzero4(a ^ b) & 0x80808080u) >> 7; /* bits 0, 8, 16, 24 */
    return (r * 0x01020408u) >> 24 & 0xfu;                       /* bit 8 k -> bit 24 + k; the other partial products stay below bit 24 or leave the word */
}
/* bit j = T[j] == query base of column j, j < 16, on the canonical copies of the sequences (upper case; for the - strand the
 * complemented copy): q points at column 0 (+ strand: columns go up; - strand: column j is q[-j]). `lowest`: first byte of the
 * store q points into (nothing in front of it is read). */
__device__ __forceinline__ uint32_t match_mask16(const uint8_t *lowest, const uint8_t *q, const uint8_t *t, bool same) {
    const uint64_t t0 = *reinterpret_cast<const u64_unaligned *>(t), t1 = *reinterpret_cast<const u64_unaligned *>(t + 8);
    uint32_t qw[4];
    if (same) {
        const uint64_t q0 = *reinterpret_cast<const u64_unaligned *>(q), q1 = *reinterpret_cast<const u64_unaligned *>(q + 8);
        qw[0] = (uint32_t)q0; qw[1] = (uint32_t)(q0 >> 32); qw[2] = (uint32_t)q1; qw[3] = (uint32_t)(q1 >> 32);
    } else if (q - 15 >= lowest) {
        const uint64_t q0 = *reinterpret_cast<const u64_unaligned *>(q - 15), q1 = *reinterpret_cast<const u64_unaligned *>(q - 7);
        /* byte 15 - j of the 16 loaded is column j */
        qw[0] = __builtin_bswap32((uint32_t)(q1 >> 32)); qw[1] = __builtin_bswap32((uint32_t)q1);
        qw[2] = __builtin_bswap32((uint32_t)(q0 >> 32)); qw[3] = __builtin_bswap32((uint32_t)q0);
    } else { /* the chunk would reach in front of the store: single bytes (the caller masks the columns beyond the op) */
        qw[0] = qw[1] = qw[2] = qw[3] = 0;
        for (int j = 0; j < 16; j++)
            if (q - j >= lowest) qw[j >> 2] |= (uint32_t)*(q - j) << (8 * (j & 3));
    }
    return equal4((uint32_t)t0, qw[0]) | (equal4((uint32_t)(t0 >> 32), qw[1]) << 4) | (equal4((uint32_t)t1, qw[2]) << 8) | (equal4((uint32_t)(t1 >> 32), qw[3]) << 12);
}
__device__ __forceinline__ int64_t wave_first_i64(int64_t x) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)x >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
/* inclusive max over the lanes before and including this one (DPP; lanes without a source keep their own value) */
__device__ __forceinline__ int32_t wave_incl_max_i32(int32_t v) {
    int32_t x = v;
#define PAFFY_MAX32_STEP(CTRL, RM, BM, SRC)                                                              \
    {                                                                                                    \
        int32_t t = __builtin_amdgcn_update_dpp((int)x, (int)(SRC), CTRL, RM, BM, false);                 \
        x = t > x ? t : x;                                                                               \
    }
    PAFFY_MAX32_STEP(DPP_ROW_SHR(1), 0xf, 0xf, v)
    PAFFY_MAX32_STEP(DPP_ROW_SHR(2), 0xf, 0xf, v)
    PAFFY_MAX32_STEP(DPP_ROW_SHR(3), 0xf, 0xf, v)
    PAFFY_MAX32_STEP(DPP_ROW_SHR(4), 0xf, 0xe, x)
    PAFFY_MAX32_STEP(DPP_ROW_SHR(8), 0xf, 0xc, x)
    PAFFY_MAX32_STEP(DPP_BCAST15, 0xa, 0xf, x)
    PAFFY_MAX32_STEP(DPP_BCAST31, 0xc, 0xf, x)
#undef PAFFY_MAX32_STEP
    return x;
}

/*
 * Walk 1 over this wave's ops [wb, we): every item gets one 32-bit word in items[] (the wave's share of the scratch), in op order:
 *   chunk of an M op:  OP_M | columns << 3 | match mask << 8 | first chunk << 24 | last chunk << 25
 *   any other op:      its 4-byte op word (len << 3 | op, op != OP_M)
 * Returns the number of ops the range becomes; *n_items = words written; *bad = first op whose bases lie outside a sequence
 * (INT64_MAX: none). q0 / t0: bases consumed on the query / target before the wave's first op (below 2^30 for the whole record).
 */
template <class OPS>
__device__ __forceinline__ uint32_t mismatch_count_wave(const KParams &P, const RecState &s, const View<OPS> &v, const uint8_t *Q, int64_t qseq_len,
                                                        const uint8_t *T, int64_t tseq_len, uint32_t wb, uint32_t we, int64_t q0, int64_t t0,
                                                        uint32_t *items, uint32_t *nm_list, uint32_t *n_items_out, uint32_t *n_nm_out, int64_t *bad) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t qpos = (uint32_t)q0, tpos = (uint32_t)t0; /* wave-uniform */
    uint32_t lane_cnt = 0, written = 0;
    uint32_t nm_written = 0; /* ops other than M seen so far: they are kept as they are, in nm_list */
    uint32_t nm_run = 0;     /* how many of them since the last M op (wave-uniform) */
    int64_t first_bad = INT64_MAX;
    /* ops in the HBM mirror (flat_add_kernel.h: one wave per piece): the op words of a window are requested a window ahead -- a window is
       the op load, the scans and one or two rounds of sequence loads, each a round trip the wave would otherwise sit through in turn */
    typename std::conditional<std::is_same<OPS, OpsGlobal>::value, uint32_t, int>::type raw_ahead = 0;
    if constexpr (std::is_same<OPS, OpsGlobal>::value)
        if (wb + lane < we) raw_ahead = v.ops.raw(v.raw_index(wb + lane));
    for (uint32_t base = wb; base < we; base += 64) {
        const uint32_t i = base + lane;
        int64_t len = 0;
        int op = OP_I;
        if constexpr (std::is_same<OPS, OpsGlobal>::value) {
            const uint32_t raw_now = raw_ahead;
            raw_ahead = 0;
            if (i + 64u < we) raw_ahead = v.ops.raw(v.raw_index(i + 64u));
            if (i < we) v.decode(raw_now, v.raw_index(i), len, op);
        } else {
            if (i < we) v.get(i, len, op);
        }
        const uint32_t dq = (i < we && op != OP_D) ? (uint32_t)len : 0u, dt = (i < we && op != OP_I) ? (uint32_t)len : 0u;
        const uint32_t qinc = wave_incl_scan_u32(dq), tinc = wave_incl_scan_u32(dt);
        const uint32_t qrel = qinc - dq, trel = tinc - dt; /* columns in front of this op, from the window's first */
        /* the window's first column on both sequences (wave-uniform); the - strand walks the query downwards from qe - 1 */
        const int64_t tj0 = s.ts + (int64_t)tpos, qoff0 = s.same ? s.qs + (int64_t)qpos : s.qe - 1 - (int64_t)qpos;
        bool is_m = i < we && op == OP_M && len > 0;
        if (is_m) {
            const int64_t tj = tj0 + trel, qoff = s.same ? qoff0 + qrel : qoff0 - qrel;
            const bool in_range = tj >= 0 && tj + len <= tseq_len && (s.same ? (qoff >= 0 && qoff + len <= qseq_len) : (qoff < qseq_len && qoff - (len - 1) >= 0));
            if (!in_range) {
                if (first_bad == INT64_MAX) first_bad = i;
                is_m = false;
            }
        }
        qpos += wave_last_u32(qinc);
        tpos += wave_last_u32(tinc);
        /* items of this op: the chunks of an M op; none for an M op of length 0 (impl/paf.c:747-779 writes nothing). Any other op
           passes through: its word goes to nm_list, and the first chunk of the next M op says how many such ops stand in front of it */
        const bool is_nm = i < we && op != OP_M;
        const uint32_t nch = is_m ? (uint32_t)((len + 15) >> 4) : 0u;
        const uint32_t iinc = wave_incl_scan_u32(nch);
        const uint32_t ioff = iinc - nch, n_items = wave_last_u32(iinc);
        const uint32_t opw = ((uint32_t)len << 3) | (uint32_t)op;
        const uint32_t nminc = wave_incl_scan_u32(is_nm ? 1u : 0u), nm_here = wave_last_u32(nminc);
        const uint32_t nmex = nminc - (is_nm ? 1u : 0u);
        if (is_nm) {
            nm_list[nm_written + nmex] = opw;
            lane_cnt++;
        }
        /* ops other than M between the M op before (the nearest lane before with items, or the windows before) and this one */
        const int32_t m_incl = wave_incl_max_i32(nch ? (int32_t)lane : -1);
        int32_t m_src = __shfl_up(m_incl, 1);
        if (lane == 0) m_src = -1;
        const uint32_t nm_at_src = __shfl(nmex, m_src < 0 ? 0 : m_src);
        uint32_t gap = m_src < 0 ? nm_run + nmex : nmex - nm_at_src;
        if (nch && gap > 62u) { /* does not fit the item word: the general encoder takes the record */
            gap = 62u;
            first_bad = -1;
        }
        {
            const int32_t last_m = __builtin_amdgcn_readlane(m_incl, 63);
            nm_run = last_m < 0 ? nm_run + nm_here : nm_here - (uint32_t)__builtin_amdgcn_readlane((int)nmex, last_m);
        }
        nm_written += nm_here;
        const uint8_t *Tw = T + tj0, *Qw = (s.same ? Q : P.seq_comp + (Q - P.seq_base)) + qoff0; /* - strand: the complemented copy */
        uint32_t carry_m = 0; /* mask of the last item of the iteration before */
#if defined(PAFFY_ABL) && (PAFFY_ABL == 32 || PAFFY_ABL == 33) /* timing only: no item loop in the count walk */
        for (uint32_t c0 = 0; c0 < (n_items & 0u); c0 += 64) {
#else
        for (uint32_t c0 = 0; c0 < n_items; c0 += 64) {
#endif
            const uint32_t c = c0 + lane;
            const bool act = c < n_items;
            uint32_t ol = 0; /* the op of item c: the last lane whose first item is <= c */
#pragma unroll
            for (uint32_t step = 32; step; step >>= 1) {
                const uint32_t cand = ol + step;
                const uint32_t vv = __shfl(ioff, (int)(cand & 63u));
                if (cand < 64 && vv <= c) ol = cand;
            }
            const uint32_t ow = __shfl(opw, (int)ol);
            const uint32_t o_first = __shfl(ioff, (int)ol);
            const uint32_t oq = __shfl(qrel, (int)ol), ot = __shfl(trel, (int)ol); /* by every lane: the source lanes must be live */
            const uint32_t og = __shfl(gap, (int)ol);
            const uint32_t olen = ow >> 3;
            const uint32_t k = (c - o_first) << 4;
            uint32_t m = 0;
            if (act) { /* every item is a chunk of an M op */
                const uint32_t nb = olen - k < 16u ? olen - k : 16u;
                m = match_mask16(P.seq_comp, s.same ? Qw + (oq + k) : Qw - (oq + k), Tw + (ot + k), s.same) & ((1u << nb) - 1u);
                items[written + c] = (uint32_t)OP_M | (nb << 3) | (m << 8) | (k == 0 ? (1u << 24) | (og << 26) : 0u) | (k + 16u >= olen ? 1u << 25 : 0u);
                uint32_t pm = __shfl_up(m, 1); /* only the bit of a chunk of the same op is used: k != 0 */
                if (lane == 0) pm = carry_m;
                /* a column starts a run when its match bit differs from the column before; the chunk before is full */
                uint32_t starts = (m ^ ((m << 1) | ((pm >> 15) & 1u))) & ((1u << nb) - 1u);
                if (k == 0) starts |= 1u;
                lane_cnt += (uint32_t)__popc(starts);
            }
            carry_m = __builtin_amdgcn_readlane((int)m, 63);
        }
        written += n_items;
    }
    *bad = first_bad;
    *n_items_out = written;
    *n_nm_out = nm_written;
    return wave_last_u32(wave_incl_scan_u32(lane_cnt));
}

/*
 * Loads that are waited for by hand. One counter (vmcnt) tracks a wave's loads and stores in issue order, so the compiler's wait
 * for a load is a wait for every store issued before it -- microseconds when the write queues are full. A word needed in the next
 * iteration is therefore requested before this iteration's stores and waited for with s_waitcnt vmcnt(N), N = a lower bound of
 * the store instructions issued since: everything older than the newest N operations, the load included, is complete.
 * sc1: past the L1 (the words were written by this wave's other walk; a line may predate those stores).
 */
#define PAFFY_LOAD32_AHEAD(dst, ptr) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(dst) : "v"(ptr) : "memory")
/* the wait and the first read of the two loaded registers in one statement: nothing can slip a copy of them in front of the wait */
#define PAFFY_WAIT_TAKE2(N, out0, out1, in0, in1) \
    asm volatile("s_waitcnt vmcnt(" #N ")\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(out0), "=&v"(out1) : "v"(in0), "v"(in1) : "memory")
__device__ __forceinline__ void wait_all_but_newest(uint32_t n, uint32_t a, uint32_t b, uint32_t &out_a, uint32_t &out_b) { /* a, b: the registers the loads fill */
    if (n >= 32) PAFFY_WAIT_TAKE2(32, out_a, out_b, a, b);
    else if (n >= 16) PAFFY_WAIT_TAKE2(16, out_a, out_b, a, b);
    else if (n >= 12) PAFFY_WAIT_TAKE2(12, out_a, out_b, a, b);
    else if (n >= 8) PAFFY_WAIT_TAKE2(8, out_a, out_b, a, b);
    else if (n == 7) PAFFY_WAIT_TAKE2(7, out_a, out_b, a, b);
    else if (n == 6) PAFFY_WAIT_TAKE2(6, out_a, out_b, a, b);
    else if (n == 5) PAFFY_WAIT_TAKE2(5, out_a, out_b, a, b);
    else if (n == 4) PAFFY_WAIT_TAKE2(4, out_a, out_b, a, b);
    else if (n == 3) PAFFY_WAIT_TAKE2(3, out_a, out_b, a, b);
    else if (n == 2) PAFFY_WAIT_TAKE2(2, out_a, out_b, a, b);
    else if (n == 1) PAFFY_WAIT_TAKE2(1, out_a, out_b, a, b);
    else PAFFY_WAIT_TAKE2(0, out_a, out_b, a, b);
}

/*
 * Walk 2: the wave's item words, 64 at a time, become the ops blk[out_base ..): no op is looked at again.
 *
 * A run of equal match bits starts at a column whose bit differs from the one before, and at an op's first column. The lanes hold
 * one item (16 columns) each and know its run starts as a bit mask; what an op needs is one lane per RUN. So the starts of the
 * 64 items are first spread out: every lane walks its own mask (the walk does nothing but write a 4-byte word per start to the
 * wave's LDS window, slot = runs before it), then lane v takes start v - 1 and its successor from LDS: the run is as long as the
 * distance to the next start (less the unused columns of an op's last chunk when that start opens an op), its letter is the match
 * bit at its own first column, its place in blk[] the number of runs and passed-through ops before it. The last start of a batch
 * waits for the first of the next (`pend`).
 * The ops other than M are copied from nm_list: the first chunk of an M op carries the number of them that stand right in front
 * of it, the rest follow the last M op. The item words and the next 64 words of nm_list are requested one iteration ahead.
 *   start word: bits 0-9 column in the batch (16 * lane + bit), 10 match bit, 11-15 unused columns in front (op's first start only),
 *               16-31 passed-through ops in front of the run, counted from the batch's first
 */
#define PAFFY_FILL_SLOTS 256u /* starts per pass: 1 KiB of the wave's text staging area */
__device__ __forceinline__ uint32_t mismatch_fill_wave(const uint32_t *items, uint32_t n_items, const uint32_t *nm_list, uint32_t n_nm, uint32_t out_base, uint32_t *blk,
                                                        uint32_t *slots) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t text = 0; /* bytes of cigar text of the ops this lane wrote (digits + letter) */
    uint32_t done = out_base; /* ops placed so far (the pending run included), the passed-through ones too; wave-uniform */
    uint32_t nm_done = 0;     /* ops of nm_list placed so far; wave-uniform */
    uint32_t carry_w = 0;     /* word of the last item of the iteration before */
    bool pend = false;        /* a run whose end is not known yet: first column (16 * item + bit), match bit, place */
    uint32_t pend_col = 0, pend_bit = 0, pend_out = 0;
    uint32_t w_next = 0, nm_next = 0; /* item word c0 + lane and nm_list[nm_done + lane] of the coming iteration */
    uint32_t since = 0;               /* store instructions issued after those two loads (a lower bound) */
#if defined(PAFFY_ABL) && (PAFFY_ABL == 31 || PAFFY_ABL == 33) /* timing only: no fill walk */
    n_items = 0;
#endif
    if (n_items) {
        if (lane < n_items) PAFFY_LOAD32_AHEAD(w_next, items + lane);
        if (lane < n_nm) PAFFY_LOAD32_AHEAD(nm_next, nm_list + lane);
    }
    for (uint32_t c0 = 0; c0 < n_items; c0 += 64) {
        const uint32_t c = c0 + lane;
        const bool act = c < n_items;
        uint32_t w_got, nmw;
        wait_all_but_newest(since, w_next, nm_next, w_got, nmw);
        const uint32_t w = act ? w_got : 0u;
        const uint32_t m = (w >> 8) & 0xffffu, nb = (w >> 3) & 31u;
        const bool first = (w >> 24) & 1u;
        const uint32_t gap = first ? w >> 26 : 0u;
        const uint32_t ginc = wave_incl_scan_u32(gap), gtot = wave_last_u32(ginc);
        /* the words of the iteration after this one, before any store of this one */
        since = 0;
        w_next = 0;
        nm_next = 0;
        if (c0 + 64 < n_items) {
            if (c + 64 < n_items) PAFFY_LOAD32_AHEAD(w_next, items + c + 64);
            if (nm_done + gtot + lane < n_nm) PAFFY_LOAD32_AHEAD(nm_next, nm_list + nm_done + gtot + lane);
        }
        uint32_t pw = __shfl_up(w, 1); /* the item before: its last match bit, and how many of its 16 columns it used */
        if (lane == 0) pw = carry_w;
        const uint32_t prevbit = (pw >> 23) & 1u;
        uint32_t starts = 0;
        if (act) starts = ((m ^ ((m << 1) | prevbit)) & ((1u << nb) - 1u)) | (first ? 1u : 0u);
        const uint32_t cnt = (uint32_t)__popc(starts);
        const uint32_t rinc = wave_incl_scan_u32(cnt), rtot = wave_last_u32(rinc);
        const uint32_t rex = rinc - cnt; /* starts of the batch in front of this item's */
        /* the ops that stand between the M op before and this one: word ginc - gap + t of the 64 held by the lanes; they go right
           in front of the item's first run */
        const uint32_t gmax = (uint32_t)__builtin_amdgcn_readlane(wave_incl_max_i32((int32_t)gap), 63);
        if (gtot <= 64u) {
            for (uint32_t t = 0; t < gmax; t++) {
                const uint32_t ow = __shfl(nmw, (int)((ginc - gap + t) & 63u)); /* by every lane: the source lanes must be live */
                if (t < gap) {
                    blk[done + rex + ginc - gap + t] = ow;
                    text += dec_len_short(ow >> 3) + 1u;
                }
                since++;
            }
        } else { /* more of them than the lanes hold (hundreds of ops other than M in a row of short M ops) */
            for (uint32_t t = 0; t < gap; t++) {
                const uint32_t ow = __hip_atomic_load(nm_list + nm_done + ginc - gap + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                blk[done + rex + ginc - gap + t] = ow;
                text += dec_len_short(ow >> 3) + 1u;
            }
        }
        nm_done += gtot;
        const uint32_t pad = first ? 16u - ((pw >> 3) & 31u) : 0u; /* columns the op before left unused in its last chunk */
        for (uint32_t s0 = 0; s0 < rtot; s0 += PAFFY_FILL_SLOTS) { /* one pass unless the batch has more than 256 starts */
            __builtin_amdgcn_wave_barrier();
            {
                uint32_t st = starts, k = rex;
                while (st) {
                    const uint32_t j = (uint32_t)__ffs((int)st) - 1u;
                    st &= st - 1u;
                    if (k - s0 < PAFFY_FILL_SLOTS) slots[k - s0] = (16u * lane + j) | (((m >> j) & 1u) << 10) | ((k == rex ? pad : 0u) << 11) | (ginc << 16);
                    k++;
                }
            }
            __builtin_amdgcn_wave_barrier(); /* a wave's LDS operations execute in order */
            const uint32_t here = rtot - s0 < PAFFY_FILL_SLOTS ? rtot - s0 : PAFFY_FILL_SLOTS;
            /* virtual run v of the pass: the run that ends at start s0 + v; v = 0 is the pending one */
            for (uint32_t v0 = 0; v0 < here; v0 += 64) {
                const uint32_t v = v0 + lane;
                if (v < here) {
                    const uint32_t nxt = slots[v];
                    const uint32_t end = 16u * c0 + (nxt & 1023u) - ((nxt >> 11) & 31u);
                    uint32_t col, bit, out;
                    bool have = true;
                    if (v == 0) {
                        col = pend_col; bit = pend_bit; out = pend_out;
                        have = pend;
                    } else {
                        const uint32_t cur = slots[v - 1u];
                        col = 16u * c0 + (cur & 1023u);
                        bit = (cur >> 10) & 1u;
                        out = done + (s0 + v - 1u) + (cur >> 16);
                    }
                    if (have) {
                        const uint32_t rl = end - col;
                        blk[out] = (rl << 3) | (bit ? (uint32_t)OP_EQ : (uint32_t)OP_X);
                        text += dec_len_short(rl) + 1u;
                    }
                }
                since += (v0 > 0 || here > 1u || pend) ? 1u : 0u; /* a round of one start with nothing pending stores nothing */
            }
            { /* the pass's last start waits for its successor */
                const uint32_t last = slots[here - 1u]; /* every lane reads the same word */
                pend = true;
                pend_col = 16u * c0 + (last & 1023u);
                pend_bit = (last >> 10) & 1u;
                pend_out = done + (s0 + here - 1u) + (last >> 16);
            }
        }
        done += rtot + gtot;
        carry_w = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)(n_items - c0 > 64u ? 63u : n_items - c0 - 1u)); /* the last item so far */
    }
    if (pend && lane == 0) { /* the last run ends with its op: the used columns of the wave's last item */
        const uint32_t rl = 16u * (n_items - 1u) + ((carry_w >> 3) & 31u) - pend_col;
        blk[pend_out] = (rl << 3) | (pend_bit ? (uint32_t)OP_EQ : (uint32_t)OP_X);
        text += dec_len_short(rl) + 1u;
    }
    /* what follows the last M op */
    for (uint32_t t = nm_done + lane; t < n_nm; t += 64) {
        const uint32_t ow = __hip_atomic_load(nm_list + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        blk[done + (t - nm_done)] = ow;
        text += dec_len_short(ow >> 3) + 1u;
    }
    return wave_last_u32(wave_incl_scan_u32(text));
}

#if defined(PAFFY_ABL) && PAFFY_ABL == 23 /* phase clock of the LDS-class mismatch encoder (device printf per sampled record) */
#define ABL23_DECL unsigned long long a23_t0 = __builtin_readcyclecounter(), a23[5] = {0, 0, 0, 0, 0};
#define ABL23_MARK(k) { unsigned long long t1_ = __builtin_readcyclecounter(); a23[k] += t1_ - a23_t0; a23_t0 = t1_; }
#define ABL23_PRINT if ((blockIdx.x & 8191u) == 77u && (threadIdx.x & 63u) == 0) printf("blk %u wave %u ops %u -> %u items %lld: sweep %llu count %llu alloc %llu fill %llu copy %llu\n", blockIdx.x, threadIdx.x >> 6, v.n, ctot[0], (long long)tot[2], a23[0], a23[1], a23[2], a23[3], a23[4]);
#else
#define ABL23_DECL
#define ABL23_MARK(k)
#define ABL23_PRINT
#endif
__device__ __forceinline__ int encode_mismatch_runs_lds(const KParams &P, const RecState &s, View<OpsLds> &v, OpsLds &ops, uint32_t cap, const uint8_t *Q,
                                                        int64_t qseq_len, const uint8_t *T, int64_t tseq_len, BlockComm &bc, Shared *sh, uint8_t *stage,
                                                        uint32_t *n_out, uint64_t *blk_off, bool keep_lds, int64_t *text_out) {
    ABL23_DECL
    /* an even split of the ops over the threads, so that the four waves get a quarter each whatever the record's size */
    const uint32_t b = (uint32_t)((uint64_t)v.n * threadIdx.x / PAFFY_NT), e = (uint32_t)((uint64_t)v.n * (threadIdx.x + 1) / PAFFY_NT);
    int64_t c[4] = {0, 0, 0, 0}, tot[4]; /* query bases, target bases, items (16-column chunks of M ops), the other ops */
    for (uint32_t i = b; i < e; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        if (op != OP_D) c[0] += len;
        if (op != OP_I) c[1] += len;
        if (op == OP_M) c[2] += (len + 15) >> 4;
        else c[3]++;
    }
    block_excl_scan<4>(c, tot, bc);
    ABL23_MARK(0)
    if (tot[0] >= (1ll << 30) || tot[1] >= (1ll << 30)) return -1; /* 32-bit columns and op counts below */
    /* scratch for the item words and the words of the ops that pass through: 4 bytes each, in the arena */
    const uint64_t mslots = ((uint64_t)tot[2] + (uint64_t)tot[3] + 1) >> 1;
    if (threadIdx.x == 0) sh->bcast[3] = (int64_t)atomicAdd(&P.info->arena_used, (unsigned long long)mslots);
    __syncthreads();
    const uint64_t moff = (uint64_t)sh->bcast[3];
    __syncthreads();
    if (moff + mslots > P.arena_cap) return -2;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t wb = (uint32_t)((uint64_t)v.n * (64u * wave) / PAFFY_NT), we = (uint32_t)((uint64_t)v.n * (64u * wave + 64u) / PAFFY_NT);
    const int64_t q0 = wave_first_i64(c[0]), t0 = wave_first_i64(c[1]);
    uint32_t *items = reinterpret_cast<uint32_t *>(P.arena + moff) + wave_first_i64(c[2]);
    uint32_t *nm_list = reinterpret_cast<uint32_t *>(P.arena + moff) + tot[2] + wave_first_i64(c[3]);
    int64_t bad = INT64_MAX;
    uint32_t n_items = 0, n_nm = 0;
    const uint32_t wcnt = mismatch_count_wave(P, s, v, Q, qseq_len, T, tseq_len, wb, we, q0, t0, items, nm_list, &n_items, &n_nm, &bad);
    ABL23_MARK(1)
    bad = block_min_i64(wave_min(bad), bc);
    if (bad == -1) return -1; /* a run of more than 62 ops other than M: the general encoder (arena class) */
    uint32_t cw[1] = {lane == 0 ? wcnt : 0u}, ctot[1];
    block_excl_scan_u32<1>(cw, ctot, bc);
    const uint32_t out_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)cw[0]);
    *n_out = ctot[0];
    if (bad != INT64_MAX) return PAFFY_ERR_SEQ_RANGE;
    if (keep_lds && ctot[0] > cap) return -3;
    const uint64_t slots = ((uint64_t)ctot[0] + 1) >> 1; /* 8-byte arena slots holding 4-byte ops */
    if (threadIdx.x == 0) sh->bcast[3] = (int64_t)atomicAdd(&P.info->arena_used, (unsigned long long)slots);
    __syncthreads();
    const uint64_t off = (uint64_t)sh->bcast[3];
    __syncthreads();
    if (off + slots > P.arena_cap) return -2;
    uint32_t *blk = reinterpret_cast<uint32_t *>(P.arena + off);
    ABL23_MARK(2)
    /* every wave has 1 KiB of the text staging area (free since the cigar was parsed) for the starts of a batch */
    const uint32_t wtext = mismatch_fill_wave(items, n_items, nm_list, n_nm, out_base, blk, reinterpret_cast<uint32_t *>(stage) + wave * PAFFY_FILL_SLOTS);
    ABL23_MARK(3)
    __syncthreads(); /* every op of the old array has been read, every op of the new one written */
    *blk_off = off;
    if (!keep_lds) { /* last stage: LDS keeps the old ops (the record check runs on them), the line is sized from the walk */
        uint32_t tw[1] = {lane == 0 ? wtext : 0u}, ttot[1];
        block_excl_scan_u32<1>(tw, ttot, bc);
        *text_out = ttot[0];
        ABL23_MARK(4)
        ABL23_PRINT
        return 0;
    }
    for (uint32_t i = threadIdx.x; i < ctot[0]; i += PAFFY_NT) ops.p[i] = __hip_atomic_load(blk + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    ABL23_MARK(4)
    ABL23_PRINT
    ops.g = blk;
    ops.g_cap = ctot[0];
    return 0;
}

/* ---------------- line pieces ---------------- */

/*
 * Wave-cooperative LDS byte builder for the per-record line pieces: all 64 lanes of one wave call
 * every method with the same arguments and lane k writes byte k of what is appended (no serial
 * digit loops, no per-lane arrays).
 */
struct Piece {
    uint8_t *p;
    uint32_t n, cap;
    bool over;
    __device__ __forceinline__ uint32_t lane() const { return threadIdx.x & 63u; }
    __device__ __forceinline__ void put_at(uint32_t k, uint32_t c) {
        if (n + k < cap) p[n + k] = (uint8_t)c;
        else over = true;
    }
    __device__ __forceinline__ void ch(uint32_t c) {
        if (lane() == 0) put_at(0, c);
        n++;
    }
    __device__ __forceinline__ void str(const char *s, uint32_t len) { /* len <= 64 */
        if (lane() < len) put_at(lane(), (uint8_t)s[lane()]);
        n += len;
    }
    __device__ __forceinline__ void num(int64_t v) { /* int64_to_str, impl/paf.c:10-34 */
        DecText d;
        dec_text(v, d);
        const uint32_t total = text_len(d);
        uint32_t k = lane();
        if (k < total) {
            uint32_t c;
            uint32_t kk = k;
            if (d.neg_separate) kk = k - 1;
            if (d.neg_separate && k == 0) {
                c = '-';
            } else if (kk < d.ntop) {
                c = (uint32_t)(d.top >> (8 * kk)) & 0xffu;
            } else {
                const uint32_t k2 = kk - d.ntop, g = k2 >> 3, b = k2 & 7u;
                const uint32_t grp = (d.groups == 2 && g == 0) ? d.g1 : d.g0;
                c = (uint32_t)(ascii8(grp) >> (8 * b)) & 0xffu;
            }
            put_at(k, c);
        }
        n += total;
    }
    __device__ __forceinline__ void name(const uint8_t *in, uint32_t off, uint32_t len) {
        for (uint32_t i = lane(); i < len; i += 64) put_at(i, in[off + i]);
        n += len;
    }
    __device__ __forceinline__ void pad_to(uint32_t m) { /* zero fill so that whole words can be read back */
        for (uint32_t i = n + lane(); i < m; i += 64)
            if (i < cap) p[i] = 0;
    }
};

/* Optional tags in the fixed order of impl/paf.c:343-365; `s1` is the chain_score to print. */
__device__ __forceinline__ void piece_tags(Piece &w, const RecState &s, int64_t s1) {
    if (s.type != 0 || s.tile_level != -1) {
        uint32_t t = s.type;
        if (t == 0) t = s.tile_level > 1 ? 'S' : 'P';
        w.str("\ttp:A:", 6);
        w.ch(t);
    }
    if (s.score != 2147483647ll) { /* INT_MAX guard, impl/paf.c:349 */
        w.str("\tAS:i:", 6);
        w.num(s.score);
    }
    if (s.tile_level != -1) { w.str("\ttl:i:", 6); w.num(s.tile_level); }
    if (s.chain_id != -1) { w.str("\tcn:i:", 6); w.num(s.chain_id); }
    if (s1 != -1) { w.str("\ts1:i:", 6); w.num(s1); }
}
__device__ __forceinline__ uint32_t tags_len(const RecState &s, int64_t s1) {
    uint32_t n = 0;
    if (s.type != 0 || s.tile_level != -1) n += 7;
    if (s.score != 2147483647ll) n += 6 + dec_len(s.score);
    if (s.tile_level != -1) n += 6 + dec_len(s.tile_level);
    if (s.chain_id != -1) n += 6 + dec_len(s.chain_id);
    if (s1 != -1) n += 6 + dec_len(s1);
    return n;
}

/* ---------------- ring -> HBM ---------------- */

/*
 * Progress of one contiguous output range through an LDS ring of R bytes (multiple of 16), ring
 * index = output offset mod R. The range is produced by a group of lanes: the whole workgroup
 * (GROUP = PAFFY_NT, barriers between the phases) or a single wave (GROUP = 64: the LDS operations
 * of one wave execute in order, so the phases need no barrier).
 */
template <int GROUP, uint32_t R>
struct Emitter {
    uint8_t *ring;
    uint8_t *out;
    uint64_t begin;   /* first byte of the range */
    uint64_t pos;     /* next byte to produce */
    uint64_t flushed; /* multiple of 16: everything below is in HBM */
    uint32_t pos_r, flushed_r; /* ring indices of pos and flushed */
    __device__ __forceinline__ void start(uint8_t *r, uint8_t *o, uint64_t off) {
        ring = r;
        out = o;
        begin = pos = off;
        flushed = off & ~15ull;
        pos_r = (uint32_t)(off % R);
        flushed_r = pos_r - (pos_r & 15u);
    }
    __device__ __forceinline__ void sync() {
        if (GROUP == 64) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
    }
    __device__ __forceinline__ void store_chunk(uint64_t c, const uint4 &v, uint64_t hi) {
        if (c >= begin && c + 16 <= hi) {
            *reinterpret_cast<uint4 *>(out + c) = v;
        } else {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int b = 0; b < 16; b++)
                if (c + b >= begin && c + b < hi) out[c + b] = (uint8_t)(w[b >> 2] >> ((b & 3) * 8));
        }
    }
    /* store ring chunks [flushed, to) to HBM; bytes outside [begin, hi) are not ours */
    __device__ __forceinline__ void flush_to(uint64_t to, uint64_t hi) {
        const uint32_t li = threadIdx.x & (GROUP - 1);
        const uint32_t nch = (uint32_t)((to - flushed) >> 4);
        for (uint32_t ch = li; ch < nch; ch += 2 * GROUP) { /* two chunks per step: both LDS reads fly before the stores */
            const uint32_t ch2 = ch + GROUP;
            uint32_t ri = flushed_r + 16 * ch, ri2 = flushed_r + 16 * ch2;
            if (ri >= R) ri -= R;
            if (ri2 >= R) ri2 -= R;
            const bool two = ch2 < nch;
            uint4 v = *reinterpret_cast<const uint4 *>(ring + ri);
            uint4 v2 = make_uint4(0, 0, 0, 0);
            if (two) v2 = *reinterpret_cast<const uint4 *>(ring + ri2);
            store_chunk(flushed + 16ull * ch, v, hi);
            if (two) store_chunk(flushed + 16ull * ch2, v2, hi);
        }
        uint32_t adv = flushed_r + 16 * nch;
        flushed_r = adv >= R ? adv - R : adv;
        flushed = to;
    }
    /*
     * A window: every lane put() its full words (phase 1); then em.sync(); rw.tail(); em.commit(bytes).
     * commit waits for the tails and pushes the complete 16-byte chunks out. The ring bytes flushed
     * here are only overwritten by a later window's deposits (bytes < R - 32 per window).
     */
    __device__ __forceinline__ void commit(uint32_t bytes) {
        sync();
        const uint64_t np = pos + bytes, to = np & ~15ull;
        if (to > flushed) flush_to(to, np);
        pos = np;
        uint32_t pr = pos_r + bytes;
        pos_r = pr >= R ? pr - R : pr;
        if (GROUP == 64) __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void finish() {
        if (pos > flushed) flush_to((pos + 15) & ~15ull, pos);
    }
};

/* ---------------- terminals ---------------- */

struct ShatterConst {
    uint32_t lenA, lenB, lenC;
    uint32_t row_const, row_max;
};

/* The five numbers of a row as text, and the row's byte count. */
struct RowText {
    DecText q0, q1, t0, t1, l;
    uint32_t bytes;
};
__device__ __forceinline__ void row_text(const ShatterConst &k, int64_t q0, int64_t t0, int64_t len, RowText &r) {
    dec_text(q0, r.q0);
    dec_text(q0 + len, r.q1);
    dec_text(t0, r.t0);
    dec_text(t0 + len, r.t1);
    dec_text(len, r.l);
    r.bytes = k.row_const + text_len(r.q0) + text_len(r.q1) + text_len(r.t0) + text_len(r.t1) + 3 * text_len(r.l);
}
__device__ __forceinline__ uint32_t row_len(const ShatterConst &k, int64_t q0, int64_t t0, int64_t len) {
    return k.row_const + dec_len(q0) + dec_len(q0 + len) + dec_len(t0) + dec_len(t0 + len) + 3 * dec_len(len);
}

/* Line pieces A, B, C of a record held in (wave-uniform) registers when they are short enough. */
struct RowPieces {
    uint64_t a[4], b[4], c[6];
    const uint64_t *A, *B, *C; /* LDS copies, any length */
    bool in_regs;
};
__device__ __forceinline__ uint64_t uniform_u64(uint64_t x) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ void load_pieces(RowPieces &t, const ShatterConst &k, const uint64_t *A, const uint64_t *B, const uint64_t *C) {
    t.A = A; t.B = B; t.C = C;
    t.in_regs = k.lenA <= 32 && k.lenB <= 32 && k.lenC <= 48;
    if (t.in_regs) {
#pragma unroll
        for (int i = 0; i < 4; i++) t.a[i] = uniform_u64(A[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) t.b[i] = uniform_u64(B[i]);
#pragma unroll
        for (int i = 0; i < 6; i++) t.c[i] = uniform_u64(C[i]);
    }
}
template <int N, class SINK>
__device__ __forceinline__ void put_regs(SINK &w, const uint64_t (&t)[N], uint32_t len) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (len >= 8u * (i + 1)) w.put8(t[i]);
        else if (len > 8u * i) w.put(t[i] & ((1ull << (8 * (len - 8u * i))) - 1ull), len - 8u * i);
    }
}

/* Row of paf_shatter2 + paf_write: A qs \t qe B ts \t te \t L \t L C L "M\n" */
template <class SINK>
__device__ __forceinline__ void put_row(SINK &w, const RowPieces &t, const ShatterConst &k, const RowText &r) {
    if (t.in_regs) put_regs<4>(w, t.a, k.lenA);
    else put_lds(w, t.A, k.lenA);
    put_text(w, r.q0, 0);
    put_text(w, r.q1, '\t');
    if (t.in_regs) put_regs<4>(w, t.b, k.lenB);
    else put_lds(w, t.B, k.lenB);
    put_text(w, r.t0, 0);
    put_text(w, r.t1, '\t');
    put_text(w, r.l, '\t');
    put_text(w, r.l, '\t');
    if (t.in_regs) put_regs<6>(w, t.c, k.lenC);
    else put_lds(w, t.C, k.lenC);
    if (r.l.groups == 0 && !r.l.neg_separate && r.l.ntop <= 6) { /* digits + "M\n" in one word */
        w.put(r.l.top | ((uint64_t)'M' << (8 * r.l.ntop)) | ((uint64_t)'\n' << (8 * r.l.ntop + 8)), r.l.ntop + 2);
    } else {
        put_text(w, r.l, 0);
        w.put((uint64_t)'M' | ((uint64_t)'\n' << 8), 2);
    }
}

/*
 * paf_shatter, impl/paf.c:629-663, sizing: total bytes / rows, or the first failing assert /
 * child paf_check in op order (key = op index * 32 + code). Also records, for the emit pass,
 * the bases consumed and bytes produced before each wave's share of the ops (lane chunks of
 * `chunk` ops, so wave w owns [64*w*chunk, 64*(w+1)*chunk)).
 */
template <class OPS>
__device__ __forceinline__ int shatter_size(const RecState &s, const View<OPS> &v, const ShatterConst &k, int64_t &bytes, int64_t &rows,
                            RecPlan *plan_out, bool checked, BlockComm &bc) {
    uint32_t b, e;
    sweep_bounds(v.n, b, e);
    {
        /*
         * One-sweep path. After a passed paf_check (coordinates inside the sequences, cigar sums equal
         * to the spans) every child block of a cigar with lengths >= 1 and ops in {M, I, D} lies inside
         * [start, end], so the child checks and the final asserts of paf_shatter cannot fail; and when
         * start and end print with the same number of digits a row's size depends on L only.
         */
        const uint32_t dq0 = dec_len(s.qs), dt0 = dec_len(s.ts);
        if (OPS::kNarrow && checked && dq0 == (uint32_t)dec_len(s.qe) && dt0 == (uint32_t)dec_len(s.te) && s.qe - s.qs < 0x7fffffffll &&
            s.te - s.ts < 0x7fffffffll) { /* the same sweep with 32-bit sums: 4-byte ops, spans (= the cigar's sums) below 2^31 */
            const uint32_t fixed = k.row_const + 2 * dq0 + 2 * dt0;
            uint32_t a[4] = {0, 0, 0, 0}, at[4], err = 0xffffffffu;
            for (uint32_t i = b; i < e; i++) {
                int64_t len;
                int op;
                v.get(i, len, op);
                int code = 0;
                if (!(len >= 1)) code = PAFFY_ERR_SHATTER_ZERO_LEN;
                else if (op == OP_M) {
                    a[2] += fixed + 3 * dec_len_u32((uint32_t)len);
                    a[3] += 1;
                } else if (op != OP_I && op != OP_D) code = PAFFY_ERR_SHATTER_BAD_OP;
                if (code && err == 0xffffffffu) err = i * 32u + (uint32_t)code;
                if (op != OP_D) a[0] += (uint32_t)len;
                if (op != OP_I) a[1] += (uint32_t)len;
            }
            block_excl_scan4_min_u32(a, at, err, bc);
            bytes = at[2];
            rows = at[3];
            if ((threadIdx.x & 63) == 0) {
                const uint32_t w = threadIdx.x >> 6;
                plan_out->wq[w] = a[0];
                plan_out->wt[w] = a[1];
                plan_out->wo[w] = a[2];
            }
            return err != 0xffffffffu ? (int)(err & 31u) : 0;
        }
        if (checked && dq0 == (uint32_t)dec_len(s.qe) && dt0 == (uint32_t)dec_len(s.te)) {
            const uint32_t fixed = k.row_const + 2 * dq0 + 2 * dt0;
            int64_t a[4] = {0, 0, 0, 0}, at[4], err = INT64_MAX;
            for (uint32_t i = b; i < e; i++) {
                int64_t len;
                int op;
                v.get(i, len, op);
                int code = 0;
                if (!(len >= 1)) code = PAFFY_ERR_SHATTER_ZERO_LEN;
                else if (op == OP_M) {
                    a[2] += fixed + 3 * dec_len(len);
                    a[3] += 1;
                } else if (op != OP_I && op != OP_D) code = PAFFY_ERR_SHATTER_BAD_OP;
                if (code && err == INT64_MAX) err = (int64_t)i * 32 + code;
                if (op != OP_D) a[0] += len;
                if (op != OP_I) a[1] += len;
            }
            err = block_min_i64(err, bc);
            block_excl_scan<4>(a, at, bc);
            bytes = at[2];
            rows = at[3];
            if ((threadIdx.x & 63) == 0) {
                const uint32_t w = threadIdx.x >> 6;
                plan_out->wq[w] = a[0];
                plan_out->wt[w] = a[1];
                plan_out->wo[w] = a[2];
            }
            return err != INT64_MAX ? (int)(err & 31) : 0;
        }
    }
    int64_t c[2] = {0, 0}, tot[2];
    for (uint32_t i = b; i < e; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        if (op != OP_D) c[0] += len;
        if (op != OP_I) c[1] += len;
    }
    block_excl_scan<2>(c, tot, bc);
    int64_t cq = c[0], ct = c[1], err = INT64_MAX;
    int64_t acc[2] = {0, 0}, acct[2];
    /* rows of a valid record have coordinates inside [start, end]: when both ends print with the same
       number of digits every row does, and only the digits of L vary */
    const uint32_t dqs = dec_len(s.qs), dts = dec_len(s.ts);
    const bool uniform_digits = s.qs >= 0 && s.ts >= 0 && dqs == (uint32_t)dec_len(s.qe) && dts == (uint32_t)dec_len(s.te);
    const uint32_t fixed_len = k.row_const + 2 * dqs + 2 * dts;
    for (uint32_t i = b; i < e; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        int code = 0;
        if (!(len >= 1)) code = PAFFY_ERR_SHATTER_ZERO_LEN;
        else if (op == OP_M) {
            int64_t q0, t0 = s.ts + ct;
            if (s.same) q0 = s.qs + cq;
            else q0 = s.qe - (cq + len);
            /* paf_check on the child (impl/paf.c:624) */
            if (q0 < 0 || q0 >= s.qlen) code = PAFFY_ERR_CHECK_QSTART;
            else if (q0 + len > s.qlen) code = PAFFY_ERR_CHECK_QEND;
            else if (t0 < 0 || t0 >= s.tlen) code = PAFFY_ERR_CHECK_TSTART;
            else if (t0 + len > s.tlen) code = PAFFY_ERR_CHECK_TEND;
            acc[0] += (uniform_digits && !code) ? fixed_len + 3 * dec_len(len) : row_len(k, q0, t0, len);
            acc[1] += 1;
        } else if (op != OP_I && op != OP_D) code = PAFFY_ERR_SHATTER_BAD_OP;
        if (code && err == INT64_MAX) err = (int64_t)i * 32 + code;
        if (op != OP_D) cq += len;
        if (op != OP_I) ct += len;
    }
    err = block_min_i64(err, bc);
    block_excl_scan<2>(acc, acct, bc);
    bytes = acct[0];
    rows = acct[1];
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = threadIdx.x >> 6;
        plan_out->wq[w] = c[0];
        plan_out->wt[w] = c[1];
        plan_out->wo[w] = acc[0];
    }
    if (err != INT64_MAX) return (int)(err & 31);
    if (tot[1] != s.te - s.ts) return PAFFY_ERR_SHATTER_END;
    if (tot[0] != s.qe - s.qs) return PAFFY_ERR_SHATTER_END;
    return 0;
}

/*
 * Emit the rows of a (validated) record. Every wave owns a contiguous share of the ops whose
 * starting coordinates and output byte come from the sizing pass; it runs its own windows of up
 * to 128 ops (two per lane, loaded from HBM, DPP scans, its own LDS ring, coalesced 16-byte
 * flushes). No workgroup barrier anywhere.
 */
#ifndef PAFFY_ROWS_MAX_OPS
#define PAFFY_ROWS_MAX_OPS 32768u /* one wave formats at most this many ops (256 windows) on its own; 16 384 until the last session of round 3: the
                                     four-wave writers of the records in between cost cfg4 2.6 % and cfg3 1-2 % of a step */
#endif
#ifndef PAFFY_WAVE_RING
#define PAFFY_WAVE_RING 9216u
#endif /* bytes of LDS ring per wave: 64 rows of the usual ~130-byte lines */
template <class OPS>
__device__ __forceinline__ void shatter_emit(const RecState &s, const View<OPS> &v, const ShatterConst &k, const RowPieces &pieces, const RecPlan &pl,
                             uint8_t *ring, uint8_t *out, uint64_t rec_off) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t span = 64ull * pl.chunk;
    const uint32_t wb = span * wave < v.n ? (uint32_t)(span * wave) : v.n;
    const uint32_t we = span * (wave + 1) < v.n ? (uint32_t)(span * (wave + 1)) : v.n;
    int64_t cq = pl.wq[wave], ct = pl.wt[wave];
    Emitter<64, PAFFY_WAVE_RING> em;
    em.start(ring + wave * PAFFY_WAVE_RING, out, rec_off + (uint64_t)pl.wo[wave]);
    const uint32_t cap_bytes = PAFFY_WAVE_RING - 32;
    const uint32_t rows_cap = cap_bytes / k.row_max; /* >= 1, checked by the sizing pass */
    const uint32_t w_safe = rows_cap < 128 ? rows_cap : 128;
    const uint32_t w_full = 2 * rows_cap < 128 ? 2 * rows_cap : 128;
    uint32_t i = wb, w_try = w_full;
    /* raw op words are loaded one window ahead so that the HBM latency hides behind the formatting */
    typename OPS::raw_t nraw0 = 0, nraw1 = 0;
    uint32_t n_i = 0xffffffffu, n_w = 0; /* window the prefetched pair belongs to */
    while (i < we) {
        const uint32_t w = we - i < w_try ? we - i : w_try;
        /* this lane's two ops of the window */
        const uint32_t j0 = i + 2 * lane;
        const bool has0 = j0 < i + w, has1 = j0 + 1 < i + w;
        const uint32_t r0 = v.raw_index(j0), r1 = v.raw_index(j0 + 1);
        typename OPS::raw_t raw0 = 0, raw1 = 0;
        if (n_i == i && n_w == w) {
            raw0 = nraw0;
            raw1 = nraw1;
        } else {
            if (has0) raw0 = v.ops.raw(r0);
            if (has1) raw1 = v.ops.raw(r1);
        }
        { /* issue the loads of the following window now; they are decoded next iteration */
            n_i = i + w;
            n_w = we - n_i < w_full ? we - n_i : w_full;
            const uint32_t k0 = n_i + 2 * lane;
            nraw0 = nraw1 = 0;
            if (k0 < n_i + n_w) nraw0 = v.ops.raw(v.raw_index(k0));
            if (k0 + 1 < n_i + n_w) nraw1 = v.ops.raw(v.raw_index(k0 + 1));
        }
        int64_t len0 = 0, len1 = 0;
        int op0 = -1, op1 = -1;
        if (has0) v.decode(raw0, r0, len0, op0);
        if (has1) v.decode(raw1, r1, len1, op1);
        int64_t c[2], tot[2];
        c[0] = (op0 >= 0 && op0 != OP_D ? len0 : 0) + (op1 >= 0 && op1 != OP_D ? len1 : 0);
        c[1] = (op0 >= 0 && op0 != OP_I ? len0 : 0) + (op1 >= 0 && op1 != OP_I ? len1 : 0);
        wave_excl_scan<2>(c, tot);
        const int64_t pq = cq + c[0], pt = ct + c[1];
        /* row text of the first M op is kept; a second M op in the pair is rare and recomputed */
        RowText rt;
        int cached = -1;
        int64_t nb[1] = {0}, nbt[1];
        {
            int64_t q = pq, t = pt;
#pragma unroll 1
            for (int sl = 0; sl < 2; sl++) {
                const int64_t len = sl ? len1 : len0;
                const int op = sl ? op1 : op0;
                if (op == OP_M) {
                    const int64_t q0 = s.same ? s.qs + q : s.qe - (q + len);
                    if (cached < 0) {
                        row_text(k, q0, s.ts + t, len, rt);
                        nb[0] += rt.bytes;
                        cached = sl;
                    } else {
                        nb[0] += row_len(k, q0, s.ts + t, len);
                    }
                }
                if (op >= 0 && op != OP_D) q += len;
                if (op >= 0 && op != OP_I) t += len;
            }
        }
        wave_excl_scan<1>(nb, nbt);
        if (nbt[0] > (int64_t)cap_bytes && w > w_safe) { /* unusually dense window: retry with the safe size */
            w_try = w_safe;
            continue;
        }
        RingWriter rw;
        rw.init(em.ring, PAFFY_WAVE_RING, em.pos_r, (uint32_t)nb[0]);
        {
            int64_t q = pq, t = pt;
#pragma unroll 1
            for (int sl = 0; sl < 2; sl++) {
                const int64_t len = sl ? len1 : len0;
                const int op = sl ? op1 : op0;
                if (op == OP_M) {
                    if (sl != cached) row_text(k, s.same ? s.qs + q : s.qe - (q + len), s.ts + t, len, rt);
#if !defined(PAFFY_ABL) || PAFFY_ABL != 2
                    put_row(rw, pieces, k, rt);
#else
                    rw.put(rt.q0.top ^ rt.q1.top ^ rt.t0.top ^ rt.t1.top ^ rt.l.top, 8);
#endif
                }
                if (op >= 0 && op != OP_D) q += len;
                if (op >= 0 && op != OP_I) t += len;
            }
        }
        em.sync();
        rw.tail();
#if !defined(PAFFY_ABL) || PAFFY_ABL != 3
        em.commit((uint32_t)nbt[0]);
#else
        em.pos += (uint32_t)nbt[0]; { uint32_t pr = em.pos_r + (uint32_t)nbt[0]; em.pos_r = pr >= PAFFY_WAVE_RING ? pr - PAFFY_WAVE_RING : pr; }
#endif
        cq += tot[0];
        ct += tot[1];
        i += w;
        w_try = w_full;
    }
    em.finish();
}

/* ---------------- rows through byte-aligned LDS stores ---------------- */
#if defined(PAFFY_ABL) && (PAFFY_ABL == 20 || PAFFY_ABL == 21)
#define PT_DECL unsigned long long pt_t0 = __builtin_readcyclecounter(), pt_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned pt_n = 0;
#define PT_MARK(k) { unsigned long long pt_t1 = __builtin_readcyclecounter(); pt_acc[k] += pt_t1 - pt_t0; pt_t0 = pt_t1; }
#else
#define PT_DECL
#define PT_MARK(k)
#endif


/*
 * gfx950 executes ds_write_b128 at any byte address (the kernel driver runs the LDS in unaligned mode);
 * a wave instruction with misaligned lanes is serialised to one lane per cycle, 16 bytes each
 * (tools/probes/lds_unaligned.hip: 64 cycles per instruction whatever the width). A lane can therefore
 * drop the pieces of its row at their byte positions with one store per piece instead of funnelling
 * the row through 64-bit shifts. One such store costs the CU as much as ~64 VALU instructions, so
 * neighbouring pieces are first merged in registers whenever they are known to fit 16 bytes.
 *
 * Stores are 16 bytes wide whatever the piece's length. The surplus lands on bytes that are written
 * later: the following pieces of the same row, or -- past the row's end -- the first bytes of the next
 * row's piece A, which is why the first 16 bytes of every A are written last, after all rows of the
 * window (lenA >= 16 is the condition for this path).
 */
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
__device__ __forceinline__ void store16(uint8_t *p, u32x4 v) { *reinterpret_cast<u32x4_unaligned *>(p) = v; }
__device__ __forceinline__ void store16(uint8_t *p, uint64_t lo, uint64_t hi) {
    const u32x4 v = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
    store16(p, v);
}

struct Txt16 { /* up to 16 characters, first character in the lowest byte, unused bytes zero */
    uint64_t lo, hi;
    uint32_t n;
};
/* decimal text of 0 <= v < 10^15 followed by the n_tail (0..2) characters of `tail` */
__device__ __forceinline__ void txt16(uint64_t v, uint32_t tail, uint32_t n_tail, Txt16 &d) {
    uint32_t top, low = 0;
    bool two;
    if ((v >> 32) == 0) {
        const uint32_t x = (uint32_t)v;
        two = x >= 100000000u;
        const uint32_t q = x / 100000000u;
        top = two ? q : x;
        low = x - q * 100000000u;
    } else {
        const uint64_t q = v / 100000000ull;
        top = (uint32_t)q;
        low = (uint32_t)(v - q * 100000000ull);
        two = true;
    }
    uint32_t nt;
    d.lo = ascii_upto8(top, &nt);
    d.hi = 0;
    d.n = nt;
    if (two) { /* top < 10^7: one to seven characters, then eight digits */
        const uint64_t g = ascii8(low);
        d.lo |= g << (8 * nt);
        d.hi = g >> (64 - 8 * nt);
        d.n = nt + 8;
    }
    const uint32_t sh = 8 * (d.n & 7u);
    const uint64_t w = (uint64_t)tail << sh;
    if (d.n < 8) {
        d.lo |= w;
        d.hi |= sh > 48 ? (uint64_t)tail >> (64 - sh) : 0;
    } else {
        d.hi |= w;
    }
    d.n += n_tail;
}

/*
 * Coordinates of one window are base .. base + total with total small, so they share all digits but
 * the last four (or those of the next ten-thousand): base = P * 10^4 + W, and a value base + d prints
 * as text(P + e) followed by the four digits of W + d - 10^4 e, e in {0, 1}, as long as W + d < 2 * 10^4.
 * The texts of P and P + 1 are wave-uniform and only change when a window crosses a multiple of 10^4.
 */
struct DigitBase {
    uint32_t P, W;
    uint64_t t0, t1; /* text(P) (empty for P = 0), text(P + 1); at most 7 characters: values < 10^11 */
    uint32_t n0, n1;
    __device__ __forceinline__ void texts() {
        t0 = 0;
        n0 = 0;
        if (P) t0 = ascii_upto8(P, &n0);
        t1 = ascii_upto8(P + 1, &n1);
        /* wave-uniform, but only ever operands of per-lane selects: kept in vector registers (there are spare ones),
           the scalar file is what overflows in the row kernel */
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(n0), "+v"(n1));
    }
    __device__ __forceinline__ void set(uint64_t v) {
        const uint64_t q = v / 10000ull;
        P = (uint32_t)q;
        W = (uint32_t)(v - q * 10000ull);
        texts();
    }
    __device__ __forceinline__ uint64_t value() const { return (uint64_t)P * 10000ull + W; }
    __device__ __forceinline__ void advance(uint32_t d) {
        W += d;
        if (W >= 10000u) {
            const uint32_t c = W / 10000u;
            W -= c * 10000u;
            P += c;
            texts();
        }
    }
    __device__ __forceinline__ void retreat(uint32_t d) {
        if (W >= d) {
            W -= d;
        } else {
            const uint32_t c = (d - W + 9999u) / 10000u;
            W = W + c * 10000u - d;
            P -= c;
            texts();
        }
    }
    /* text of base + d (W + d < 20000) followed by `tail` (n_tail <= 2 characters) */
    __device__ __forceinline__ void num(uint32_t d, uint32_t tail, uint32_t n_tail, Txt16 &o) const {
        uint32_t x = W + d;
        const bool e = x >= 10000u;
        if (e) x -= 10000u;
        uint32_t w4 = bcd4(x);
        const uint64_t pt = e ? t1 : t0;
        const uint32_t np = e ? n1 : n0;
        uint64_t v5;
        uint32_t n4 = 4;
        if (np == 0) { /* value < 10^4: no leading zeros */
            const uint32_t z = w4 ? ((uint32_t)__ffs((int)w4) - 1) >> 3 : 3u;
            n4 = 4 - z;
            v5 = (uint64_t)((w4 + 0x30303030u) >> (8 * z)) | ((uint64_t)tail << (8 * n4));
        } else {
            v5 = (uint64_t)(w4 + 0x30303030u) | ((uint64_t)tail << 32);
        }
        const uint32_t sh = 8 * np;
        o.lo = pt | (v5 << sh);
        o.hi = (v5 >> 1) >> (63 - sh);
        o.n = np + n4 + n_tail;
    }
};

/* a wave-uniform piece rest (r = 1..15 bytes, zero above) followed by t, r + t.n <= 16 */
__device__ __forceinline__ void store_rest_then(uint8_t *p, uint64_t rest_lo, uint64_t rest_hi, uint32_t r, const Txt16 &t) {
    if (r < 8) {
        const uint32_t sh = 8 * r;
        store16(p, rest_lo | (t.lo << sh), (t.hi << sh) | (t.lo >> (64 - sh)));
    } else {
        store16(p, rest_lo, rest_hi | (t.lo << (8 * (r - 8))));
    }
}

/*
 * One wave's contiguous output range through a linear LDS buffer: buffer byte 0 is output byte `base`
 * (a multiple of 16); the incomplete last 16 bytes of a window are carried to the front. The flush of a
 * window is issued after the sizing arithmetic of the next one (stage()/flush() are separate), so that
 * a wave does not sit on the LDS queue behind its own stores.
 */
#define PAFFY_FLUSH_GRAN 128u /* bytes: a window leaves for HBM in whole granules (cache lines), the rest is carried */
struct WaveLinear {
    uint8_t *buf, *out;
    uint64_t begin, base; /* wave-uniform */
    uint32_t phase;       /* bytes of the buffer already holding output (< PAFFY_FLUSH_GRAN) */
    uint32_t pending;     /* bytes staged behind `phase` and not flushed yet */
    __device__ __forceinline__ void start(uint8_t *b, uint8_t *o, uint64_t off) {
        buf = b;
        out = o;
        begin = off;
        base = off & ~(uint64_t)(PAFFY_FLUSH_GRAN - 1u);
        phase = (uint32_t)(off & (PAFFY_FLUSH_GRAN - 1u));
        pending = 0;
    }
    /* buffer offset at which the next window's rows start (valid before the pending flush has run) */
    __device__ __forceinline__ uint32_t next_phase() const { return (phase + pending) & (PAFFY_FLUSH_GRAN - 1u); }
    __device__ __forceinline__ void stage(uint32_t bytes) { pending = bytes; }
    /* push the staged window's whole granules to HBM (tools/probes/rowstore_global.hip: a flush that ends on a cache-line boundary
       instead of any 16-byte boundary is worth 7 % of the write rate: no line is written in two halves by two store instructions
       that are a whole window's arithmetic apart) */
    __device__ __forceinline__ void flush() {
        if (pending == 0) return;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_setprio(3); /* the LDS reads and stores of a flush in front of other waves' arithmetic: the LDS pipe and the store
                                          stream are what the kernel is short of (k_emit_rows 3.78 -> 3.755 ms, A/B on one box) */
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t total = phase + pending, nch = (total / PAFFY_FLUSH_GRAN) * (PAFFY_FLUSH_GRAN / 16u);
        uint8_t *dst = out + base; /* wave-uniform: scalar base, 32-bit lane offsets */
        uint32_t first = 0;
        if (base < begin && nch > 0) { /* first granule of the range: the bytes below `begin` belong to the previous wave or record */
            const uint32_t head = (uint32_t)(begin - base);
            first = (head + 15u) >> 4;
            if (lane >= (head & 15u) && lane < 16 && (head & 15u)) dst[(head & ~15u) + lane] = buf[(head & ~15u) + lane];
        }
        for (uint32_t ch = first + lane; ch < nch; ch += 256) { /* four chunks per step: the LDS reads fly together, one wait */
            const uint32_t c1 = ch + 64, c2 = ch + 128, c3 = ch + 192;
            uint4 v0 = *reinterpret_cast<const uint4 *>(buf + 16 * ch), v1, v2, v3;
            if (c1 < nch) v1 = *reinterpret_cast<const uint4 *>(buf + 16 * c1);
            if (c2 < nch) v2 = *reinterpret_cast<const uint4 *>(buf + 16 * c2);
            if (c3 < nch) v3 = *reinterpret_cast<const uint4 *>(buf + 16 * c3);
            *reinterpret_cast<uint4 *>(dst + 16 * ch) = v0; /* plain stores: as write-through stores (sc0 sc1) the kernel takes 4.08 instead of 3.83 ms */
            if (c1 < nch) *reinterpret_cast<uint4 *>(dst + 16 * c1) = v1;
            if (c2 < nch) *reinterpret_cast<uint4 *>(dst + 16 * c2) = v2;
            if (c3 < nch) *reinterpret_cast<uint4 *>(dst + 16 * c3) = v3;
        }
        const uint32_t rest = total - 16u * nch; /* < PAFFY_FLUSH_GRAN: carried to the front (nch >= 8, so the chunks read and written are apart) */
        if (nch > 0 && 16u * lane < rest) {
            const uint4 t = *reinterpret_cast<const uint4 *>(buf + 16 * (nch + lane));
            *reinterpret_cast<uint4 *>(buf + 16 * lane) = t;
        }
        base += 16ull * nch;
        phase = rest;
        pending = 0;
        __builtin_amdgcn_s_setprio(0);
#if defined(PAFFY_STORE_WINDOW)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PAFFY_STORE_WINDOW) : "memory");
#endif
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void finish() {
        flush();
        const uint32_t lane = threadIdx.x & 63;
        for (uint32_t b = lane; b < phase; b += 64)
            if (base + b >= begin) out[base + b] = buf[b];
    }
};

/* wave-uniform description of the constant pieces of a record's rows (the pieces themselves stay in LDS) */
struct RowConst {
    const u32x4 *A16, *B16, *C16;
    uint32_t lenA, lenB, lenC;
    bool fuseA, fuseB; /* rest of A + q0 + tab / rest of B + t0 + tab always fit one store */
    uint32_t dt;
};
/* one row's numbers: offsets from the window's digit bases (near) or the values themselves */
struct RowNums {
    uint32_t dq, dt, len;
};
__device__ __forceinline__ uint32_t dec_len_u32(uint32_t x) {
    return 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) + (x >= 10000000u) + (x >= 100000000u) +
           (x >= 1000000000u);
}
/* digits of base + d, W + d < 20000 */
__device__ __forceinline__ uint32_t near_len(const DigitBase &b, uint32_t d) {
    uint32_t x = b.W + d;
    const bool e = x >= 10000u;
    const uint32_t np = e ? b.n1 : b.n0;
    return np ? np + 4u : 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u);
}
template <bool NEAR>
__device__ __forceinline__ uint32_t row_bytes(const RowConst &c, const DigitBase &bq, const DigitBase &bt, const RowNums &r) {
    const uint32_t nl = dec_len_u32(r.len);
    uint32_t n;
    if (NEAR) {
        n = near_len(bq, r.dq) + near_len(bq, r.dq + r.len) + near_len(bt, r.dt) + near_len(bt, r.dt + r.len);
    } else {
        const uint64_t q0 = bq.value() + r.dq, t0 = bt.value() + r.dt;
        n = dec_len((int64_t)q0) + dec_len((int64_t)(q0 + r.len)) + dec_len((int64_t)t0) + dec_len((int64_t)(t0 + r.len));
    }
    return c.lenA + c.lenB + c.lenC + 6u + n + 3u * nl;
}
template <bool NEAR>
__device__ __forceinline__ void coord_txt(const DigitBase &b, uint32_t d, uint32_t tail, uint32_t n_tail, Txt16 &o) {
    if (NEAR) b.num(d, tail, n_tail, o);
    else txt16(b.value() + d, tail, n_tail, o);
}
/*
 * A row at p, everything but the first 16 bytes of piece A: each number is formatted right before its
 * store so that no text stays live. `small`: every L of the window has at most two digits.
 */
/* The chunks of the three pieces for the usual shape (A 16..31 bytes, B at most 31, C at most 47), read from LDS at the top of a
   window's formatting: a read issued between the byte-aligned stores would wait for all of them (the LDS executes a wave's
   operations in order and those stores take 64 cycles each). */
struct RowPre {
    u32x4 a_rest, b0, b_rest, c0, c1, c_rest; /* rest: the chunk holding a piece's last len % 16 bytes */
};
__device__ __forceinline__ bool row_pre_ok(const RowConst &c) { return c.lenA < 32 && c.lenB < 32 && c.lenC < 48; }
__device__ __forceinline__ void row_pre_load(const RowConst &c, RowPre &r) {
    r.a_rest = c.A16[1];
    r.b0 = c.B16[0]; r.b_rest = c.B16[c.lenB >> 4];
    r.c0 = c.C16[0]; r.c1 = c.C16[1]; r.c_rest = c.C16[c.lenC >> 4];
}
/* the chunks of a piece, from LDS (wave-uniform addresses: broadcast reads) */
#define PIECE_FULL_CHUNKS(P, PTR, LEN, FROM)                                                                  \
    {                                                                                                         \
        const uint32_t full_ = (LEN) >> 4;                                                                    \
        _Pragma("unroll 1") for (uint32_t j = (FROM); j < full_; j++) store16((P) + 16 * j, c.PTR[j]);        \
    }
#define PIECE_REST_CHUNK(PTR, LEN) (c.PTR[(LEN) >> 4])

template <bool NEAR, bool PRE>
__device__ __forceinline__ void put_row_body(uint8_t *p, const RowConst &c, const DigitBase &bq, const DigitBase &bt, const RowNums &r, bool small) {
    RowPre pre;
    if (PRE) row_pre_load(c, pre); /* before the row's first byte-aligned store */
    Txt16 t;
    const uint32_t fullA = c.lenA >> 4, rA = c.lenA & 15u;
    if (!PRE) PIECE_FULL_CHUNKS(p, A16, c.lenA, 1)
    coord_txt<NEAR>(bq, r.dq, '\t', 1, t);
    if (c.fuseA) {
        const u32x4 a = PRE ? pre.a_rest : PIECE_REST_CHUNK(A16, c.lenA);
        store_rest_then(p + 16 * fullA, a.x | ((uint64_t)a.y << 32), a.z | ((uint64_t)a.w << 32), rA, t);
    } else {
        if (rA) store16(p + 16 * fullA, PRE ? pre.a_rest : PIECE_REST_CHUNK(A16, c.lenA));
        store16(p + c.lenA, t.lo, t.hi);
    }
    p += c.lenA + t.n;
    coord_txt<NEAR>(bq, r.dq + r.len, 0, 0, t);
    store16(p, t.lo, t.hi);
    p += t.n;
    const uint32_t fullB = c.lenB >> 4, rB = c.lenB & 15u;
    if (PRE) {
        if (fullB) store16(p, pre.b0);
    } else {
        PIECE_FULL_CHUNKS(p, B16, c.lenB, 0)
    }
    coord_txt<NEAR>(bt, r.dt, '\t', 1, t);
    if (c.fuseB) {
        const u32x4 b = PRE ? pre.b_rest : PIECE_REST_CHUNK(B16, c.lenB);
        store_rest_then(p + 16 * fullB, b.x | ((uint64_t)b.y << 32), b.z | ((uint64_t)b.w << 32), rB, t);
    } else {
        if (rB) store16(p + 16 * fullB, PRE ? pre.b_rest : PIECE_REST_CHUNK(B16, c.lenB));
        store16(p + c.lenB, t.lo, t.hi);
    }
    p += c.lenB + t.n;
    coord_txt<NEAR>(bt, r.dt + r.len, '\t', 1, t);
    Txt16 lt;
    txt16(r.len, '\t', 1, lt); /* "L\t" */
    const uint32_t nl = lt.n - 1;
    if (small && c.dt <= 10) { /* "t1\tL\tL": at most 11 + 5 characters in one store */
        const uint64_t ltl = lt.lo | ((lt.lo & 0xffffull) << (8 * lt.n)); /* a second tab (one-digit L) sits where C's first byte, a tab, follows */
        if (t.n < 8) {
            t.lo |= ltl << (8 * t.n);
            t.hi |= (ltl >> 1) >> (63 - 8 * t.n);
        } else {
            t.hi |= ltl << (8 * (t.n - 8));
        }
        store16(p, t.lo, t.hi);
    } else {
        store16(p, t.lo, t.hi);
        store16(p + t.n, lt.lo, lt.hi);
        store16(p + t.n + lt.n, lt.lo, lt.hi); /* its tab is C's first byte */
    }
    p += t.n + lt.n + nl;
    const uint32_t fullC = c.lenC >> 4, rC = c.lenC & 15u;
    if (PRE) {
        if (fullC >= 1) store16(p, pre.c0);
        if (fullC >= 2) store16(p + 16, pre.c1);
    } else {
        PIECE_FULL_CHUNKS(p, C16, c.lenC, 0)
    }
    /* "LM\n" from "L\t": the tab becomes 'M', then '\n' */
    if (nl < 8) {
        const uint32_t sh = 8 * nl;
        lt.lo ^= (uint64_t)('\t' ^ 'M') << sh;
        if (sh < 56) lt.lo |= (uint64_t)'\n' << (sh + 8);
        else lt.hi = (uint64_t)'\n';
    } else {
        const uint32_t sh = 8 * (nl - 8);
        lt.hi = (lt.hi ^ ((uint64_t)('\t' ^ 'M') << sh)) | ((uint64_t)'\n' << (sh + 8));
    }
    lt.n = nl + 2;
    if (small && rC != 0 && rC <= 12) { /* rest of C + "LM\n" (at most four characters) in one store */
        const u32x4 cr = PRE ? pre.c_rest : PIECE_REST_CHUNK(C16, c.lenC);
        store_rest_then(p + 16 * fullC, cr.x | ((uint64_t)cr.y << 32), cr.z | ((uint64_t)cr.w << 32), rC, lt);
    } else {
        if (rC) store16(p + 16 * fullC, PRE ? pre.c_rest : PIECE_REST_CHUNK(C16, c.lenC));
        store16(p + c.lenC, lt.lo, lt.hi);
    }
}
__device__ __forceinline__ bool shatter_fast_ok(const RecState &s, const ShatterConst &k) {
    /* rows of a valid record have 0 <= coordinates <= sequence length (child paf_check, impl/paf.c:624),
       and the cigar's sums equal the spans, so window sums fit 32 bits when the spans do */
    /* the bases of the digit arithmetic are the record's own coordinates: they must be sane themselves (shatter does not
       check its parent: a record with target_start -1 whose first op is a deletion still has valid rows) */
    return k.lenA >= 16 && k.row_max <= PAFFY_WAVE_RING - 48u - PAFFY_FLUSH_GRAN && s.qlen < 100000000000ll && s.tlen < 100000000000ll && s.qs >= 0 && s.qs <= s.qe && s.qe <= s.qlen && s.ts >= 0 &&
           s.ts <= s.te && s.te <= s.tlen && s.qe - s.qs < 0x7f000000ll && s.te - s.ts < 0x7f000000ll;
}

/*
 * Same work split as shatter_emit (every wave owns a contiguous share of the ops, its start taken from
 * the sizing pass; windows of up to 128 ops, two per lane; no workgroup barrier), rows written with
 * byte-aligned 16-byte LDS stores, coordinates as 32-bit offsets from a wave-uniform base.
 */
template <class OPS>
__device__ __forceinline__ void shatter_emit_fast(const RecState &s, const View<OPS> &v, const ShatterConst &k, const u32x4 *A16, const u32x4 *B16,
                                                  const u32x4 *C16, uint32_t wb, uint32_t we, int64_t cq0, int64_t ct0, uint8_t *buf, uint8_t *out,
                                                  uint64_t out_start) {
    /* this wave emits the rows of ops [wb, we): cq0 / ct0 bases lie before op wb, its first output byte is out_start */
    const uint32_t lane = threadIdx.x & 63;
    WaveLinear em;
    em.start(buf, out, out_start);
    RowConst rc;
    rc.A16 = A16; rc.B16 = B16; rc.C16 = C16;
    rc.lenA = k.lenA; rc.lenB = k.lenB; rc.lenC = k.lenC;
    rc.dt = dec_len(s.tlen);
    rc.fuseA = (k.lenA & 15u) != 0 && (k.lenA & 15u) + dec_len(s.qlen) + 1 <= 16;
    rc.fuseB = (k.lenB & 15u) != 0 && (k.lenB & 15u) + rc.dt + 1 <= 16;
    const bool pre_ok = row_pre_ok(rc);
    /* digit bases: forward strand q = (qs + cq) + dq; reverse strand q0 = (qe - cq - total) + (total - dq - len) */
    DigitBase bq, bt;
#if defined(PAFFY_ABL) && PAFFY_ABL == 43 /* 43: without the two digit bases' divisions and texts (their cost; the rows come out wrong) */
    bq.P = 1; bq.W = (uint32_t)cq0; bq.t0 = 0x31; bq.t1 = 0x32; bq.n0 = bq.n1 = 1;
    bt.P = 1; bt.W = (uint32_t)ct0; bt.t0 = 0x31; bt.t1 = 0x32; bt.n0 = bt.n1 = 1;
#else
    bq.set(s.same ? (uint64_t)(s.qs + cq0) : (uint64_t)(s.qe - cq0));
    bt.set((uint64_t)(s.ts + ct0));
#endif
    const uint32_t cap_bytes = PAFFY_WAVE_RING - 48 - PAFFY_FLUSH_GRAN; /* a carried rest of up to a granule in front */
    const uint32_t rows_cap = cap_bytes / k.row_max; /* >= 1: shatter_fast_ok */
    const uint32_t w_safe = rows_cap < 128 ? rows_cap : 128;
    const uint32_t w_full = 2 * rows_cap < 128 ? 2 * rows_cap : 128;
    uint32_t i = wb, w_try = w_full;
    /*
     * Raw ops of five windows at a time, two per lane and window, in registers, and the block after it already on
     * its way. One counter (vmcnt) tracks loads and stores in issue order, so an ordinary wait for a load is also a
     * wait for every store issued before it -- and a flushed window is acknowledged only after ~10 us when the write
     * queues are full. The next block's loads are therefore issued a whole block ahead (inline asm: the compiler must
     * not put its own vmcnt(0) in front of their use) and waited for with s_waitcnt vmcnt(N), N = a lower bound of the
     * store instructions issued since: everything older than the newest N operations, the loads included, is done.
     */
    typedef typename OPS::raw_t raw_t;
    constexpr bool kAhead = std::is_same<OPS, OpsGlobal>::value; /* 4-byte ops in the HBM mirror */
    raw_t b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0, b8 = 0, b9 = 0;
    raw_t n0 = 0, n1 = 0, n2 = 0, n3 = 0, n4 = 0, n5 = 0, n6 = 0, n7 = 0, n8 = 0, n9 = 0; /* the block after */
    uint32_t slot = 5, blk_next = 0xffffffffu; /* next unused window of the block and the op index it starts at */
    uint32_t ahead_at = 0xffffffffu;           /* op index the block in n0..n9 starts at (none: 0xffffffff) */
    uint32_t stores_since = 0;                 /* store instructions issued after the loads of n0..n9 */
#define PAFFY_LOAD_AHEAD(dst, idx)                                                                                   \
    {                                                                                                                \
        dst = 0;                                                                                                     \
        if ((idx) < we) {                                                                                            \
            if (v.ops.half) { /* wave-uniform: 2-byte words, zero-extended by the load */                            \
                const uint16_t *ptr_ = reinterpret_cast<const uint16_t *>(v.ops.p) + v.raw_index(idx);               \
                asm volatile("global_load_ushort %0, %1, off" : "=v"(dst) : "v"(ptr_) : "memory");                  \
            } else {                                                                                                 \
                const uint32_t *ptr_ = v.ops.p + v.raw_index(idx);                                                   \
                asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(ptr_) : "memory");                   \
            }                                                                                                        \
        }                                                                                                            \
    }
    PT_DECL
    while (i < we) {
        PT_MARK(5)
        const uint32_t w = we - i < w_try ? we - i : w_try;
        const uint32_t j0 = i + 2 * lane;
        const bool has0 = j0 < i + w, has1 = j0 + 1 < i + w;
        const uint32_t r0 = v.raw_index(j0), r1 = v.raw_index(j0 + 1);
        raw_t raw0 = 0, raw1 = 0;
        if (w_try != w_full) { /* safe-size retry: plain loads */
            if (has0) raw0 = v.ops.raw(r0);
            if (has1) raw1 = v.ops.raw(r1);
            slot = 5;
        } else {
            if (slot >= 5 || blk_next != i) {
                if constexpr (kAhead) {
                    if (ahead_at == i) { /* the block is (almost) here */
                        if (stores_since >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                        else if (stores_since >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                        else if (stores_since >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        b0 = n0; b1 = n1; b2 = n2; b3 = n3; b4 = n4; b5 = n5; b6 = n6; b7 = n7; b8 = n8; b9 = n9;
                    } else {
                        if (ahead_at != 0xffffffffu) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* a stale block in flight */
                        uint32_t k0 = j0;
                        PAFFY_LOAD_AHEAD(b0, k0) PAFFY_LOAD_AHEAD(b1, k0 + 1)
                        k0 += w_full;
                        PAFFY_LOAD_AHEAD(b2, k0) PAFFY_LOAD_AHEAD(b3, k0 + 1)
                        k0 += w_full;
                        PAFFY_LOAD_AHEAD(b4, k0) PAFFY_LOAD_AHEAD(b5, k0 + 1)
                        k0 += w_full;
                        PAFFY_LOAD_AHEAD(b6, k0) PAFFY_LOAD_AHEAD(b7, k0 + 1)
                        k0 += w_full;
                        PAFFY_LOAD_AHEAD(b8, k0) PAFFY_LOAD_AHEAD(b9, k0 + 1)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    /* the block after this one */
                    ahead_at = 0xffffffffu;
                    if ((uint64_t)i + 5ull * w_full < we) {
                        ahead_at = i + 5 * w_full;
                        stores_since = 0;
                        uint32_t k0 = ahead_at + 2 * lane;
                        PAFFY_LOAD_AHEAD(n0, k0) PAFFY_LOAD_AHEAD(n1, k0 + 1)
                        k0 += w_full;
                        PAFFY_LOAD_AHEAD(n2, k0) PAFFY_LOAD_AHEAD(n3, k0 + 1)
                        k0 += w_full;
                        PAFFY_LOAD_AHEAD(n4, k0) PAFFY_LOAD_AHEAD(n5, k0 + 1)
                        k0 += w_full;
                        PAFFY_LOAD_AHEAD(n6, k0) PAFFY_LOAD_AHEAD(n7, k0 + 1)
                        k0 += w_full;
                        PAFFY_LOAD_AHEAD(n8, k0) PAFFY_LOAD_AHEAD(n9, k0 + 1)
                    }
                } else {
                    uint32_t k0 = j0;
                    b0 = k0 < we ? v.ops.raw(v.raw_index(k0)) : (raw_t)0;
                    b1 = k0 + 1 < we ? v.ops.raw(v.raw_index(k0 + 1)) : (raw_t)0;
                    k0 += w_full;
                    b2 = k0 < we ? v.ops.raw(v.raw_index(k0)) : (raw_t)0;
                    b3 = k0 + 1 < we ? v.ops.raw(v.raw_index(k0 + 1)) : (raw_t)0;
                    k0 += w_full;
                    b4 = k0 < we ? v.ops.raw(v.raw_index(k0)) : (raw_t)0;
                    b5 = k0 + 1 < we ? v.ops.raw(v.raw_index(k0 + 1)) : (raw_t)0;
                    k0 += w_full;
                    b6 = k0 < we ? v.ops.raw(v.raw_index(k0)) : (raw_t)0;
                    b7 = k0 + 1 < we ? v.ops.raw(v.raw_index(k0 + 1)) : (raw_t)0;
                    k0 += w_full;
                    b8 = k0 < we ? v.ops.raw(v.raw_index(k0)) : (raw_t)0;
                    b9 = k0 + 1 < we ? v.ops.raw(v.raw_index(k0 + 1)) : (raw_t)0;
                }
                slot = 0;
            }
            raw0 = slot == 0 ? b0 : slot == 1 ? b2 : slot == 2 ? b4 : slot == 3 ? b6 : b8;
            raw1 = slot == 0 ? b1 : slot == 1 ? b3 : slot == 2 ? b5 : slot == 3 ? b7 : b9;
            slot++;
            blk_next = i + w;
        }
        int64_t len0_64 = 0, len1_64 = 0;
        int op0 = -1, op1 = -1;
        if (has0) v.decode(raw0, r0, len0_64, op0);
        if (has1) v.decode(raw1, r1, len1_64, op1);
        const uint32_t len0 = (uint32_t)len0_64, len1 = (uint32_t)len1_64;
        const uint32_t q_0 = (op0 >= 0 && op0 != OP_D) ? len0 : 0, t_0 = (op0 >= 0 && op0 != OP_I) ? len0 : 0;
        const uint32_t q_1 = (op1 >= 0 && op1 != OP_D) ? len1 : 0, t_1 = (op1 >= 0 && op1 != OP_I) ? len1 : 0;
        PT_MARK(0)
        const uint32_t iq = wave_incl_scan_u32(q_0 + q_1), it = wave_incl_scan_u32(t_0 + t_1);
        const uint32_t totq = wave_last_u32(iq), tott = wave_last_u32(it);
        const uint32_t eq = iq - (q_0 + q_1), et = it - (t_0 + t_1);
        /* a lane's first M op is its first row; two M ops in one pair (a second row) are rare */
        const bool m0 = op0 == OP_M, m1 = op1 == OP_M;
        const bool prim = m0 || m1, sec = m0 && m1;
        const bool small = __all(!prim || ((m0 ? len0 : len1) < 100u && (!sec || len1 < 100u))) != 0;
        const bool any_sec = __any(sec) != 0;
        if (!s.same) bq.retreat(totq); /* the window's lowest query coordinate */
        const bool near = bq.W + totq < 20000u && bt.W + tott < 20000u; /* wave-uniform */
        RowNums rn0, rn1;
        rn0.len = m0 ? len0 : len1;
        rn0.dq = m0 ? eq : eq + q_0;
        rn0.dt = m0 ? et : et + t_0;
        rn1.len = len1;
        rn1.dq = eq + q_0;
        rn1.dt = et + t_0;
        if (!s.same) {
            rn0.dq = totq - rn0.dq - rn0.len;
            rn1.dq = totq - rn1.dq - rn1.len;
        }
        uint32_t bytes0 = 0, bytes1 = 0;
        if (prim) bytes0 = near ? row_bytes<true>(rc, bq, bt, rn0) : row_bytes<false>(rc, bq, bt, rn0);
        if (sec) bytes1 = near ? row_bytes<true>(rc, bq, bt, rn1) : row_bytes<false>(rc, bq, bt, rn1);
        const uint32_t mine = bytes0 + bytes1;
        const uint32_t inc = wave_incl_scan_u32(mine), total = wave_last_u32(inc);
        if (total > cap_bytes && w > w_safe) { /* unusually dense window: retry with the safe size */
            if (!s.same) bq.advance(totq);
            w_try = w_safe;
            continue;
        }
        PT_MARK(1)
        const uint32_t o = em.next_phase() + inc - mine;
        em.flush();
        PT_MARK(2)
 /* the previous window: its stores were issued before this window's arithmetic */
#pragma unroll 1
        for (int r = 0; r < 2; r++) {
            if (r && !any_sec) break;
            const bool on = r ? sec : prim;
            if (on) {
                uint8_t *p = em.buf + o + (r ? bytes0 : 0u);
                RowNums x;
                x.len = r ? rn1.len : rn0.len;
                x.dq = r ? rn1.dq : rn0.dq;
                x.dt = r ? rn1.dt : rn0.dt;
                if (pre_ok) {
                    if (near) put_row_body<true, true>(p, rc, bq, bt, x, small);
                    else put_row_body<false, true>(p, rc, bq, bt, x, small);
                } else {
                    if (near) put_row_body<true, false>(p, rc, bq, bt, x, small);
                    else put_row_body<false, false>(p, rc, bq, bt, x, small);
                }
            }
        }
        /* the head of piece A of every row, last: it repairs what the previous row's wide stores spilled */
        if (prim) {
            const u32x4 a_head = A16[0];
            store16(em.buf + o, a_head);
            if (sec) store16(em.buf + o + bytes0, a_head);
        }
        em.stage(total);
        PT_MARK(3)
        if (s.same) bq.advance(totq);
        bt.advance(tott);
        i += w;
        w_try = w_full;
    }
#undef PAFFY_LOAD_AHEAD
    em.finish();
#if defined(PAFFY_ABL) && PAFFY_ABL == 20
    PT_MARK(4)
    if ((blockIdx.x & 8191u) == 77u && (threadIdx.x & 63) == 0)
        printf("rec %u wave %u ops %u: load+decode %llu scans+sizes %llu flush %llu format+stores %llu finish %llu loop-top %llu\n", blockIdx.x, (unsigned)(threadIdx.x >> 6), we - wb,
               pt_acc[0], pt_acc[1], pt_acc[2], pt_acc[3], pt_acc[4], pt_acc[5]);
#endif
}

/* Per-lane serial writers for the rare records whose pieces do not fit the LDS staging (names of
 * many hundreds of bytes): bytes go straight to HBM, the pieces are re-derived from the input text. */
struct DirectWriter {
    uint8_t *p;
    __device__ __forceinline__ void put(uint64_t w, uint32_t k) {
        for (uint32_t b = 0; b < k; b++) p[b] = (uint8_t)(w >> (8 * b));
        p += k;
    }
    __device__ __forceinline__ void put8(uint64_t w) { put(w, 8); }
    __device__ __forceinline__ void bytes(const uint8_t *src, uint32_t n) {
        for (uint32_t i = 0; i < n; i++) p[i] = src[i];
        p += n;
    }
    __device__ __forceinline__ void lit(const char *s, uint32_t n) {
        for (uint32_t i = 0; i < n; i++) p[i] = (uint8_t)s[i];
        p += n;
    }
};
__device__ __forceinline__ void direct_tags(DirectWriter &w, const RecState &s, int64_t s1) { /* impl/paf.c:343-365 */
    if (s.type != 0 || s.tile_level != -1) {
        uint32_t t = s.type;
        if (t == 0) t = s.tile_level > 1 ? 'S' : 'P';
        w.lit("\ttp:A:", 6);
        w.put(t, 1);
    }
    if (s.score != 2147483647ll) { w.lit("\tAS:i:", 6); put_dec(w, s.score); }
    if (s.tile_level != -1) { w.lit("\ttl:i:", 6); put_dec(w, s.tile_level); }
    if (s.chain_id != -1) { w.lit("\tcn:i:", 6); put_dec(w, s.chain_id); }
    if (s1 != -1) { w.lit("\ts1:i:", 6); put_dec(w, s1); }
}
template <class OPS>
__device__ __forceinline__ void shatter_emit_direct(const KParams &P, const RecState &s, const View<OPS> &v, const ShatterConst &k,
                                                    const RecPlan &pl, uint64_t rec_off) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t span = 64ull * pl.chunk;
    const uint32_t wb = span * wave < v.n ? (uint32_t)(span * wave) : v.n;
    const uint32_t we = span * (wave + 1) < v.n ? (uint32_t)(span * (wave + 1)) : v.n;
    int64_t cq = pl.wq[wave], ct = pl.wt[wave];
    uint64_t pos = rec_off + (uint64_t)pl.wo[wave];
    for (uint32_t i = wb; i < we; i += 64) { /* one op per lane */
        const uint32_t j = i + lane;
        int64_t len = 0;
        int op = -1;
        if (j < we) v.get(j, len, op);
        int64_t c[2] = {op >= 0 && op != OP_D ? len : 0, op >= 0 && op != OP_I ? len : 0}, tot[2];
        wave_excl_scan<2>(c, tot);
        const int64_t q0 = s.same ? s.qs + cq + c[0] : s.qe - (cq + c[0] + len), t0 = s.ts + ct + c[1];
        int64_t nb[1] = {op == OP_M ? (int64_t)row_len(k, q0, t0, len) : 0}, nbt[1];
        wave_excl_scan<1>(nb, nbt);
        if (op == OP_M) {
            DirectWriter w{P.out + pos + (uint64_t)nb[0]};
            w.bytes(P.in + s.qn_off, s.qn_len);
            w.put('\t', 1); put_dec(w, s.qlen);
            w.put('\t', 1); put_dec(w, q0);
            w.put('\t', 1); put_dec(w, q0 + len);
            w.put('\t', 1); w.put(s.same ? '+' : '-', 1); w.put('\t', 1);
            w.bytes(P.in + s.tn_off, s.tn_len);
            w.put('\t', 1); put_dec(w, s.tlen);
            w.put('\t', 1); put_dec(w, t0);
            w.put('\t', 1); put_dec(w, t0 + len);
            w.put('\t', 1); put_dec(w, len);
            w.put('\t', 1); put_dec(w, len);
            w.put('\t', 1); put_dec(w, s.mapq);
            direct_tags(w, s, 0);
            w.lit("\tcg:Z:", 6);
            put_dec(w, len);
            w.put((uint64_t)'M' | ((uint64_t)'\n' << 8), 2);
        }
        pos += (uint64_t)nbt[0];
        cq += tot[0];
        ct += tot[1];
    }
}

/* Header of paf_write_to_buffer up to (and including) "\tcg:Z:" -- impl/paf.c:317-368. */
/*
 * The line in front of the cigar (paf_write, impl/paf.c:317-368), one ITEM per lane (round 3): the twelve fields, the five optional
 * tags, the cigar's tag name and the newline are nineteen items; lane k turns item k into text -- one decimal conversion per lane
 * instead of twelve to seventeen wave-uniform ones, each computed by all 64 lanes (about 2 000 instructions per record: a third of the
 * one-wave line writer's time on cfg3 records, most of the tile writer's) --, a wave scan of the lengths places the items, every lane
 * drops its bytes, the two names are copied by all lanes.
 */
struct HeaderItem {
    int64_t val;      /* numeric items */
    uint64_t pre;     /* up to seven bytes in front of the number: the tab, a tag's name */
    uint32_t plen;
    uint32_t name_len; /* items 0 and 5: the query / target name follows the prefix */
    bool numeric, present;
};
__device__ constexpr uint64_t header_tag6(char a, char b, char t) {
    return 0x09ull | ((uint64_t)(uint8_t)a << 8) | ((uint64_t)(uint8_t)b << 16) | ((uint64_t)':' << 24) | ((uint64_t)(uint8_t)t << 32) | ((uint64_t)':' << 40);
}
__device__ __forceinline__ HeaderItem header_item(const RecState &s, bool newline, uint32_t k) {
    HeaderItem it;
    it.val = 0; it.pre = '\t'; it.plen = 1; it.name_len = 0; it.numeric = true; it.present = k < 12;
    const int64_t f[12] = {0, s.qlen, s.qs, s.qe, 0, 0, s.tlen, s.ts, s.te, s.nmatch, s.nbases, s.mapq};
#pragma unroll
    for (uint32_t j = 1; j < 12; j++)
        if (k == j) it.val = f[j];
    if (k == 0) { it.numeric = false; it.plen = 0; it.name_len = s.qn_len; }
    if (k == 4) { it.numeric = false; it.pre = (uint64_t)'\t' | ((uint64_t)(s.same ? '+' : '-') << 8); it.plen = 2; }
    if (k == 5) { it.numeric = false; it.name_len = s.tn_len; }
    if (k == 12) { /* impl/paf.c:343-348 */
        uint32_t t = s.type;
        if (t == 0) t = s.tile_level > 1 ? 'S' : 'P';
        it.present = s.type != 0 || s.tile_level != -1;
        it.numeric = false;
        it.pre = header_tag6('t', 'p', 'A') | ((uint64_t)t << 48);
        it.plen = 7;
    }
    if (k == 13) { it.present = s.score != 2147483647ll; it.pre = header_tag6('A', 'S', 'i'); it.plen = 6; it.val = s.score; } /* INT_MAX guard, impl/paf.c:349 */
    if (k == 14) { it.present = s.tile_level != -1; it.pre = header_tag6('t', 'l', 'i'); it.plen = 6; it.val = s.tile_level; }
    if (k == 15) { it.present = s.chain_id != -1; it.pre = header_tag6('c', 'n', 'i'); it.plen = 6; it.val = s.chain_id; }
    if (k == 16) { it.present = s.chain_score != -1; it.pre = header_tag6('s', '1', 'i'); it.plen = 6; it.val = s.chain_score; }
    if (k == 17) { it.present = s.has_cigar; it.numeric = false; it.pre = header_tag6('c', 'g', 'Z'); it.plen = 6; }
    if (k == 18) { it.present = newline; it.numeric = false; it.pre = '\n'; }
    return it;
}
__device__ __forceinline__ void header_put(uint8_t *p, uint32_t cap, uint32_t at, uint64_t w, uint32_t n) { /* n <= 8 bytes of w at p[at ..) */
#pragma unroll
    for (uint32_t b = 0; b < 8; b++)
        if (b < n && at + b < cap) p[at + b] = (uint8_t)(w >> (8 * b));
}
__device__ __forceinline__ void build_header(Piece &w, const RecState &s, const uint8_t *in, bool newline) {
    const uint32_t lane = threadIdx.x & 63u;
    const HeaderItem it = header_item(s, newline, lane);
    DecText d;
    dec_text(it.val, d);
    const uint32_t len = it.present ? it.plen + it.name_len + (it.numeric ? text_len(d) : 0u) : 0u;
    const uint32_t inc = wave_incl_scan_u32(len), total = wave_last_u32(inc);
    const uint32_t at0 = w.n + inc - len;
    if (w.n + total > w.cap) w.over = true;
    if (it.present) {
        uint32_t at = at0;
        header_put(w.p, w.cap, at, it.pre, it.plen);
        at += it.plen;
        if (it.numeric) {
            if (d.neg_separate) {
                header_put(w.p, w.cap, at, '-', 1);
                at += 1;
            }
            header_put(w.p, w.cap, at, d.top, d.ntop);
            at += d.ntop;
            if (d.groups == 2) {
                header_put(w.p, w.cap, at, ascii8(d.g1), 8);
                at += 8;
            }
            if (d.groups >= 1) header_put(w.p, w.cap, at, ascii8(d.g0), 8);
        }
    }
    /* the names, by all lanes: the query name opens the line, the target name follows item 5's tab */
    const uint32_t t_at = w.n + (uint32_t)__shfl((int)(inc - len), 5) + 1u;
    for (uint32_t i = lane; i < s.qn_len; i += 64)
        if (w.n + i < w.cap) w.p[w.n + i] = in[s.qn_off + i];
    for (uint32_t i = lane; i < s.tn_len; i += 64)
        if (t_at + i < w.cap) w.p[t_at + i] = in[s.tn_off + i];
    w.n += total;
}
/* the same items' lengths, one per lane: for the callers that are whole waves */
__device__ __forceinline__ uint32_t header_len_wave(const RecState &s, bool newline) {
    const HeaderItem it = header_item(s, newline, threadIdx.x & 63u);
    const uint32_t len = it.present ? it.plen + it.name_len + (it.numeric ? (uint32_t)dec_len(it.val) : 0u) : 0u;
    return wave_last_u32(wave_incl_scan_u32(len));
}
__device__ uint32_t header_len(const RecState &s, bool newline) {
    uint32_t n = s.qn_len + s.tn_len + 12 + dec_len(s.qlen) + dec_len(s.qs) + dec_len(s.qe) + dec_len(s.tlen) + dec_len(s.ts) +
                 dec_len(s.te) + dec_len(s.nmatch) + dec_len(s.nbases) + dec_len(s.mapq) + tags_len(s, s.chain_score);
    if (s.has_cigar) n += 6;
    if (newline) n += 1;
    return n;
}

/*
 * Bytes of the cigar text of the view: sum of digits + 1 per op (impl/paf.c:369-380). Also leaves,
 * for the emit pass, the text bytes in front of each wave's share of the ops (wave w owns lane
 * chunks [64*w*chunk, 64*(w+1)*chunk), as in sweep_bounds()).
 */
template <class OPS>
__device__ __forceinline__ int64_t cigar_text_len(const View<OPS> &v, RecPlan *plan_out, BlockComm &bc) {
    uint32_t b, e;
    sweep_bounds(v.n, b, e);
    int64_t a[1] = {0}, at[1];
    for (uint32_t i = b; i < e; i++) {
        int64_t len;
        int op;
        v.get(i, len, op);
        if constexpr (OPS::kNarrow) a[0] += dec_len_short((uint32_t)len) + 1; /* 4-byte ops: 0 <= length < 2^29 */
        else a[0] += dec_len(len) + 1;
    }
    block_excl_scan<1>(a, at, bc);
    if ((threadIdx.x & 63) == 0) plan_out->wo[threadIdx.x >> 6] = a[0];
    return at[0];
}

/*
 * One whole line (paf_write, impl/paf.c:317-389): the header piece from LDS, then the ops, then '\n'.
 * Same structure as shatter_emit: every wave owns a contiguous share of the ops (starting byte from
 * the sizing pass) and streams it through its own LDS ring in windows of WRITE_PER ops per lane, with
 * no workgroup barrier; wave 0 first sends the header in 1 KiB windows.
 */
/*
 * Cigar text of the one-wave line writer, one op per lane and step (round 3). The funnel below gives every lane sixteen consecutive
 * ops and a 64-bit accumulator: three sweeps over them, a store decision per op -- about 130 instructions per op and lane, 1.8 ms for
 * the 1.2 GB of a cfg4 batch. Here the 64 lanes take 64 consecutive ops: length and letter become at most five characters (lengths
 * below 10^4: one 4-digit BCD conversion), an inclusive scan of the character counts places them, and the characters are dropped at
 * their bytes of the ring with ds_write_b8 -- the one LDS store that runs at full rate at any byte address. Sixteen steps make a
 * window (the op words of all of them are requested together, then waited for once); the window leaves through the emitter as before.
 * Returns the first op it did not write: a window with a length of five digits or more is left to the funnel.
 */
#define TEXT_STEPS 16u
template <bool WRAP>
__device__ __forceinline__ uint32_t text_window_steps(const View<OpsGlobal> &v, uint32_t i, uint32_t w, const uint32_t (&raw)[TEXT_STEPS], uint8_t *ring, uint32_t pos_r) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t done = 0; /* bytes of the steps before (wave-uniform) */
    const uint32_t steps = (w + 63u) >> 6;
#pragma unroll
    for (uint32_t st = 0; st < TEXT_STEPS; st++) {
        if (st >= steps) continue; /* wave-uniform; no break: the loop must unroll for raw[] to stay in registers */
        const uint32_t j = i + 64u * st + lane;
        const bool has = j < i + w;
        const uint32_t r = v.raw_index(j);
        uint32_t op = raw[st] & 7u;
        op ^= v.swp ? ((0x6u >> op) & 1u) * 3u : 0u; /* I <-> D */
        const uint32_t x = (raw[st] >> 3) - (r == v.lo ? (uint32_t)v.sub_lo : 0u) - (r == v.lo + v.n - 1 ? (uint32_t)v.sub_hi : 0u);
        const uint32_t nd = 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u);
        const uint32_t digits = (bcd4(x) + 0x30303030u) >> (8u * (4u - nd)); /* most significant digit in byte 0 */
        const uint32_t letter = op < 4u ? (0x3d44494du >> (8u * op)) & 0xffu : (op == 4u ? (uint32_t)'X' : (uint32_t)'N'); /* M I D = X, impl/paf.c:372-379 */
        const bool nl = has && j + 1u == v.n; /* the record's last op takes the newline along */
        const uint32_t n = has ? nd + 1u + (nl ? 1u : 0u) : 0u;
        const uint32_t inc = wave_incl_scan_u32(n);
        uint32_t a = pos_r + done + inc - n;
        if (WRAP && a >= PAFFY_WAVE_RING) a -= PAFFY_WAVE_RING;
        done += wave_last_u32(inc);
        if (has) {
#define TEXT_PUT(k, val)                                                      \
    {                                                                         \
        uint32_t a_ = a + (k);                                                \
        if (WRAP && a_ >= PAFFY_WAVE_RING) a_ -= PAFFY_WAVE_RING;             \
        ring[a_] = (uint8_t)(val);                                            \
    }
            TEXT_PUT(0u, digits)
            if (nd > 1u) TEXT_PUT(1u, digits >> 8)
            if (nd > 2u) TEXT_PUT(2u, digits >> 16)
            if (nd > 3u) TEXT_PUT(3u, digits >> 24)
            TEXT_PUT(nd, letter)
            if (nl) TEXT_PUT(nd + 1u, '\n')
#undef TEXT_PUT
        }
    }
    return done;
}
template <class EM>
__device__ __forceinline__ uint32_t cigar_text_fast(const View<OpsGlobal> &v, uint32_t wb, uint32_t we, EM &em) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t i = wb;
    while (i < we) {
        const uint32_t w = we - i < 64u * TEXT_STEPS ? we - i : 64u * TEXT_STEPS;
        uint32_t raw[TEXT_STEPS];
        bool short_lens = true;
#pragma unroll
        for (uint32_t st = 0; st < TEXT_STEPS; st++) {
            const uint32_t j = i + 64u * st + lane;
            raw[st] = j < i + w ? v.ops.raw(v.raw_index(j)) : 0u;
        }
#pragma unroll
        for (uint32_t st = 0; st < TEXT_STEPS; st++) short_lens = short_lens && (raw[st] >> 3) < 10000u; /* the ends' cuts only shorten */
        if (!__all(short_lens)) break;
        uint32_t bytes;
        if (em.pos_r + 6u * w + 8u <= PAFFY_WAVE_RING) bytes = text_window_steps<false>(v, i, w, raw, em.ring, em.pos_r);
        else bytes = text_window_steps<true>(v, i, w, raw, em.ring, em.pos_r);
        em.commit(bytes); /* its wave barrier orders the byte stores before the flush's reads */
        i += w;
    }
    return i;
}

#define WRITE_PER 16u
template <class OPS>
__device__ __forceinline__ void write_emit_range(const View<OPS> &v, uint32_t wb, uint32_t we, bool with_header, const uint64_t *H, uint32_t lenH,
                                                 uint8_t *ring, uint8_t *out, uint64_t start_off) {
    /* this wave writes [the header and] the text of ops [wb, we) from output byte start_off on, through its own LDS ring */
    const uint32_t lane = threadIdx.x & 63;
    Emitter<64, PAFFY_WAVE_RING> em;
    em.start(ring, out, start_off);
    if (with_header) {
        for (uint32_t base = 0; base < lenH; base += 16 * 64) {
            const uint32_t left = lenH - base, wbytes = left < 16 * 64 ? left : 16 * 64;
            const uint32_t mine_off = 16 * lane;
            RingWriter rw;
            if (mine_off < wbytes) {
                const uint32_t mine = wbytes - mine_off < 16 ? wbytes - mine_off : 16;
                rw.init(em.ring, PAFFY_WAVE_RING, em.pos_r, mine_off);
                put_lds(rw, H + ((base + mine_off) >> 3), mine);
            } else {
                rw.init(em.ring, PAFFY_WAVE_RING, em.pos_r, 0);
                rw.nacc = rw.head = 0; /* nothing to write */
            }
            em.sync();
            rw.tail();
            em.commit(wbytes);
        }
    }
    const uint32_t cap_bytes = PAFFY_WAVE_RING - 32;
    const uint32_t w_full = 64 * WRITE_PER, w_safe = cap_bytes / 21; /* an op prints as at most 20 + 1 bytes */
    uint32_t i = wb, w_try = w_full;
#ifndef PAFFY_NO_TEXT_FAST
    if constexpr (std::is_same<OPS, OpsGlobal>::value) i = cigar_text_fast(v, wb, we, em); /* what it leaves (a length of five digits or more) follows below */
#endif
    while (i < we) {
        const uint32_t w = we - i < w_try ? we - i : w_try;
        const uint32_t per = (w + 63) / 64;
        uint32_t b = i + lane * per, e = b + per;
        if (b > i + w) b = i + w;
        if (e > i + w) e = i + w;
        const bool last = (i + w == v.n);
        /* the lane's ops of this window, loaded together (independent loads, one wait) and kept in registers for both passes */
        typename OPS::raw_t raw[WRITE_PER];
#pragma unroll
        for (int jj = 0; jj < WRITE_PER; jj++) {
            raw[jj] = 0;
            if (b + jj < e) raw[jj] = v.ops.raw(v.raw_index(b + jj));
        }
        int64_t nb[1] = {0}, nbt[1];
        /* the usual window: every length has at most four digits -- 24-bit multiplies instead of the 64-bit digit text */
        bool short_lens = true;
#pragma unroll
        for (int jj = 0; jj < WRITE_PER; jj++) {
            if (b + jj < e) {
                int64_t len;
                int op;
                v.decode(raw[jj], v.raw_index(b + jj), len, op);
                short_lens = short_lens && len >= 0 && len < 10000;
            }
        }
        short_lens = __all(short_lens) != 0;
#pragma unroll
        for (int jj = 0; jj < WRITE_PER; jj++) {
            if (b + jj < e) {
                int64_t len;
                int op;
                v.decode(raw[jj], v.raw_index(b + jj), len, op);
                if (short_lens) {
                    const uint32_t x = (uint32_t)len;
                    nb[0] += 2 + (x >= 10u) + (x >= 100u) + (x >= 1000u);
                } else {
                    nb[0] += dec_len(len) + 1;
                }
            }
        }
        if (last && e == v.n && b < e) nb[0] += 1; /* '\n' goes with the last op */
        wave_excl_scan<1>(nb, nbt);
        if (nbt[0] > (int64_t)cap_bytes && w > w_safe) {
            w_try = w_safe;
            continue;
        }
        RingWriter rw;
        rw.init(em.ring, PAFFY_WAVE_RING, em.pos_r, (uint32_t)nb[0]);
#pragma unroll 2
        for (int jj = 0; jj < WRITE_PER; jj++) {
            if (b + jj < e) {
                int64_t len;
                int op;
                v.decode(raw[jj], v.raw_index(b + jj), len, op);
                if (short_lens) {
                    const uint32_t x = (uint32_t)len;
                    const uint32_t nd = 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u);
                    const uint32_t digits = (bcd4(x) + 0x30303030u) >> (8u * (4u - nd)); /* most significant digit in byte 0 */
                    rw.put((uint64_t)digits | ((uint64_t)op_char_of(op) << (8u * nd)), nd + 1u);
                    continue;
                }
                DecText d;
                dec_text(len, d);
                if (d.groups == 0 && !d.neg_separate && d.ntop <= 7) { /* digits + op letter in one word */
                    rw.put(d.top | ((uint64_t)op_char_of(op) << (8 * d.ntop)), d.ntop + 1);
                } else {
                    put_text(rw, d, 0);
                    rw.put(op_char_of(op), 1);
                }
            }
        }
        if (last && e == v.n && b < e) rw.put('\n', 1);
        em.sync();
        rw.tail();
        em.commit((uint32_t)nbt[0]);
        i += w;
        w_try = w_full;
    }
    em.finish();
}
template <class OPS>
__device__ __forceinline__ void write_emit(const View<OPS> &v, bool has_cigar, const uint64_t *H, uint32_t lenH, const RecPlan &pl,
                                           uint8_t *ring, uint8_t *out, uint64_t rec_off, bool header_done) {
    /* four waves: wave 0 starts at the record's first byte (header first); the others after the header and the text before them */
    const uint32_t wave = threadIdx.x >> 6;
    const bool with_ops = has_cigar && v.n > 0;
    const uint64_t span = 64ull * pl.chunk;
    const uint32_t wb = !with_ops ? 0 : (span * wave < v.n ? (uint32_t)(span * wave) : v.n);
    const uint32_t we = !with_ops ? 0 : (span * (wave + 1) < v.n ? (uint32_t)(span * (wave + 1)) : v.n);
    const bool with_header = wave == 0 && !header_done;
    write_emit_range(v, wb, we, with_header, H, lenH, ring + wave * PAFFY_WAVE_RING, out,
                     rec_off + (with_header ? 0 : lenH + (with_ops ? (uint64_t)pl.wo[wave] : 0)));
}

/* ---------------- the record program ---------------- */

struct RecLds {
    uint8_t *ring;    /* output ring(s); in the sizing kernel the cigar text staging area */
    uint64_t *pieces; /* 3 * PAFFY_TMPL_MAX bytes */
    mutable BlockComm bc; /* 64 words of LDS */
    Shared *sh;
};

__device__ __forceinline__ void report(const KParams &P, uint32_t rec, int code, int stage, int aux, uint32_t klass) {
    if (threadIdx.x == 0) {
        P.status[rec] = (uint32_t)code | ((uint32_t)((stage + 1) & 0xff) << 8) | (klass << 16);
        P.err_aux[rec] = aux;
        P.out_len[rec] = 0;
        P.out_rows[rec] = 0;
        if (code)
            atomicMin(&P.info->first_err_key,
                      ((unsigned long long)rec << 16) | ((unsigned long long)((stage + 1) & 0xff) << 8) | (unsigned long long)code);
    }
}

__device__ __forceinline__ void shatter_consts(const RecState &s, ShatterConst &k) {
    /* line pieces shared by every row of a record (paf_shatter2, impl/paf.c:600-627) */
    k.lenA = s.qn_len + 2 + dec_len(s.qlen);
    k.lenB = s.tn_len + 5 + dec_len(s.tlen);
    k.lenC = 1 + dec_len(s.mapq) + tags_len(s, 0) + 6; /* children carry s1:i:0 (calloc, impl/paf.c:601) */
    k.row_const = k.lenA + k.lenB + k.lenC + 6;
    uint32_t dq = dec_len(s.qlen), dt = dec_len(s.tlen);
    k.row_max = k.row_const + 2 * dq + 2 * dt + 3 * (dq < dt ? dq : dt);
}
__device__ __forceinline__ bool shatter_fits(const ShatterConst &k) {
    return k.lenA <= PAFFY_TMPL_MAX && k.lenB <= PAFFY_TMPL_MAX && k.lenC <= PAFFY_TMPL_MAX && k.row_max <= PAFFY_WAVE_RING - 32;
}

__device__ __forceinline__ void load_state(const RecMeta &m, RecState &s) {
    s.qlen = m.qlen; s.qs = m.qs; s.qe = m.qe; s.tlen = m.tlen; s.ts = m.ts; s.te = m.te;
    s.nmatch = m.nmatch; s.nbases = m.nbases; s.mapq = m.mapq; s.score = m.score;
    s.tile_level = m.tile_level; s.chain_id = m.chain_id; s.chain_score = m.chain_score;
    s.qn_off = m.qname_off; s.qn_len = m.qname_len; s.tn_off = m.tname_off; s.tn_len = m.tname_len;
    s.same = m.same_strand != 0;
    s.type = m.type;
    s.has_cigar = m.has_cg && m.cg_len > 0; /* cigar_parse("") == NULL, impl/paf.c:71-73 */
}

/*
 * Sizing pass for record `rec`: parse the cigar into `ops` (capacity `cap`), run the stage list,
 * record the exact output size, the first failing check and the RecPlan the emit pass resumes
 * from. Returns false when the record does not fit this op store (LDS class only).
 */
/*
 * MASK: the stage kinds this instantiation knows (bit = kind). The host picks the lean instantiation when the
 * pipe only has invert / identity trim / shatter / pass stages: a kernel without the other transforms is a third
 * smaller (instruction cache, registers).
 */
#define PAFFY_MASK_ALL 0xffffffffu
#define PAFFY_MASK_LEAN ((1u << PAFFY_INVERT) | (1u << PAFFY_TRIM_IDENTITY) | (1u << PAFFY_SHATTER) | (1u << PAFFY_PASS))
#define PAFFY_MASK_ADD (PAFFY_MASK_LEAN | (1u << PAFFY_ADD_MISMATCHES))
/* everything but the mismatch encoder, the stats sums and the trim by count: the pipes of filter, fixed trim, add_mismatches -a */
#define PAFFY_MASK_PLAIN (PAFFY_MASK_ALL & ~((1u << PAFFY_ADD_MISMATCHES) | (1u << PAFFY_STATS) | (1u << PAFFY_TRIM_ENDS)))
/* the selecting / measuring commands around the lean kinds (`filter`, `trim -f`, `view -s`, a bare paf_check): an instantiation of their own keeps
   them free of the spills that the op-array rebuilding stages (remove_mismatches) bring into the general build */
#define PAFFY_MASK_SEL (PAFFY_MASK_LEAN | (1u << PAFFY_FILTER) | (1u << PAFFY_TRIM_FIXED) | (1u << PAFFY_CHECK) | (1u << PAFFY_STATS))
#define STAGE_ON(kind) ((MASK >> (kind)) & 1u)
template <class OPS, uint32_t MASK = PAFFY_MASK_ALL>
__device__ __forceinline__ bool size_record(const KParams &P, uint32_t rec, OPS &ops, uint32_t cap, const RecLds &L, uint32_t klass,
                            uint32_t *n_ops_out, uint32_t *need_out = nullptr) {
    const RecMeta m = P.meta[rec];
    if (m.err) {
        report(P, rec, m.err, -1, m.err_aux, klass);
        return true;
    }
    PT_DECL
    RecState s;
    load_state(m, s);
    uint32_t n = 0;
    int64_t parse_sums[4] = {0, 0, 0, 0};
    bool have_sums = false;
    bool parse_plain = false; /* parsed here and every op is M / I / D with a length >= 1 */
    bool mirror16 = false;    /* the HBM mirror of this record holds 2-byte words */
    if (s.has_cigar) {
        bool fits;
        uint32_t err_pos;
        uint32_t r;
        r = 0xffffffffu;
        if constexpr (std::is_same<OPS, OpsLds>::value) {
            r = parse_cigar_lds(P.in, m.cg_off, m.cg_len, ops, cap, L.ring, L.bc, L.sh, &fits, &err_pos, parse_sums, &parse_plain, &mirror16);
            if (r == 0xffffffffu) {
                parse_plain = false;
                mirror16 = false; /* the general parser below writes the mirror again, in 4-byte words */
            }
        }
        if (r == 0xffffffffu) r = parse_cigar(P.in, m.cg_off, m.cg_len, ops, cap, L.ring, L.bc, L.sh, &fits, &err_pos, parse_sums);
        have_sums = true;
        if (r & 0x80000000u) {
            r = parse_cigar_serial(P.in, m.cg_off, m.cg_len, ops, cap, L.sh, &fits, &err_pos);
            have_sums = false;
        }
        n = r;
        *n_ops_out = n;
        if (err_pos != 0xffffffffu) { /* st_errAbort, impl/paf.c:102 */
            report(P, rec, PAFFY_ERR_CIGAR_CHAR, -1, err_pos < m.cg_off + m.cg_len ? P.in[err_pos] : 0, klass);
            return true;
        }
        if (!fits) return false;
    }
    *n_ops_out = n;
    PT_MARK(0)
    View<OPS> v;
    v.reset(ops, n);
    if (have_sums) { /* the parse already summed every op */
        v.tm = parse_sums[0]; v.tx = parse_sums[1]; v.tq = parse_sums[2]; v.tt = parse_sums[3];
        v.totals_ok = true;
    }
    bool swapped = false, shatter = false;
    bool ops_in_arena = false; /* LDS class whose op array was rebuilt: 4-byte ops in an arena block instead of the mirror */
    bool ops_block_only = false; /* ... by the last stage: the new ops are in that block only, LDS still holds the old ones */
    int64_t block_text = 0;      /* ... and their cigar text takes this many bytes */
    uint64_t arena_block = 0;
    bool checked = false; /* a paf_check has passed since the record last changed */
    int32_t si = 0;
    for (; si < P.n_stages; si++) {
        const paffy_stage st = P.stages[si];
        if (si > 0) { /* what `paf_write | paf_parse` between two processes does to the record */
            if (s.has_cigar && v.n == 0) s.has_cigar = false;
            if (s.type == 0 && s.tile_level != -1) s.type = s.tile_level > 1 ? 'S' : 'P';
        }
        int rc = 0;
        const bool do_check = !((P.nocheck_mask >> si) & 1u); /* the command loops check after every transform; the library calls of inc/paf.h do not */
        if (STAGE_ON(PAFFY_INVERT) && st.kind == PAFFY_INVERT) {
            invert_state(s);
            invert_view(s, v);
            swapped = !swapped;
            if (do_check) rc = check_record(s, v, L.bc);
            PT_MARK(1)
        } else if (STAGE_ON(PAFFY_TRIM_IDENTITY) && st.kind == PAFFY_TRIM_IDENTITY) {
            rc = trim_identity(s, v, st.p0, st.p1, L.bc, L.sh);
            PT_MARK(2)
            if (!rc && do_check) rc = check_record(s, v, L.bc);
            PT_MARK(3)
        } else if (STAGE_ON(PAFFY_TRIM_FIXED) && st.kind == PAFFY_TRIM_FIXED) {
            rc = trim_fixed(s, v, st.p1, L.bc, L.sh);
            if (!rc && do_check) rc = check_record(s, v, L.bc);
        } else if (STAGE_ON(PAFFY_TRIM_ENDS) && st.kind == PAFFY_TRIM_ENDS) {
            const int64_t count = (int64_t)(((uint64_t)__float_as_uint(st.p1) << 32) | (uint64_t)__float_as_uint(st.p0));
            rc = trim_fixed(s, v, 0.0f, L.bc, L.sh, true, count);
            if (!rc && do_check) rc = check_record(s, v, L.bc);
        } else if (STAGE_ON(PAFFY_CHECK) && st.kind == PAFFY_CHECK) {
            rc = check_record(s, v, L.bc); /* paf_check alone, impl/paf.c:427-461 */
        } else if (STAGE_ON(PAFFY_REMOVE_MISMATCHES) && st.kind == PAFFY_REMOVE_MISMATCHES) {
            parse_plain = false; /* the op array is rebuilt */
            mirror16 = false;    /* ... in 4-byte words (MirrorDst) */
            if (s.has_cigar) {
                bool narrow_ok = true;
                uint32_t n2;
                if constexpr (OPS::kNarrow) { /* LDS class: new array via the HBM mirror, then back into LDS */
                    MirrorDst dst{ops.g, ops.g_cap};
                    n2 = merge_match_runs(v, dst, L.bc, L.sh, &narrow_ok);
                    if (!narrow_ok || v.n > ops.g_cap) return false; /* needs 8-byte ops: arena class */
                    for (uint32_t i = threadIdx.x; i < n2; i += PAFFY_NT) ops.p[i] = ops.g[i];
                    __syncthreads();
                } else { /* arena class: new array in a fresh arena block */
                    if (threadIdx.x == 0) L.sh->bcast[3] = (int64_t)atomicAdd(&P.info->arena_used, (unsigned long long)v.n);
                    __syncthreads();
                    const uint64_t off = (uint64_t)L.sh->bcast[3];
                    __syncthreads();
                    if (off + v.n > P.arena_cap) return true; /* no room: the host grows the arena and repeats the pass */
                    OpsArena dst{P.arena + off};
                    n2 = merge_match_runs(v, dst, L.bc, L.sh, &narrow_ok);
                    ops.p = dst.p;
                }
                v.reset(ops, n2);
            }
            if (do_check) rc = check_record(s, v, L.bc);
        } else if (STAGE_ON(PAFFY_ADD_MISMATCHES) && st.kind == PAFFY_ADD_MISMATCHES) {
            parse_plain = false;
            if constexpr (std::is_same<OPS, OpsLds>::value) {
                const int32_t qi = swapped ? P.rec_tseq[rec] : P.rec_qseq[rec], ti = swapped ? P.rec_qseq[rec] : P.rec_tseq[rec];
                if (qi < 0) rc = PAFFY_ERR_MISSING_QUERY_SEQ;   /* impl/paf_add_mismatches.c:117-120 */
                else if (ti < 0) rc = PAFFY_ERR_MISSING_TARGET_SEQ; /* :123-127 */
                else if (s.has_cigar) {
                    uint32_t n2 = 0;
                    const bool last = si == P.n_stages - 1; /* nothing reads the new ops but the line writer: they need not fit LDS */
                    const int r = encode_mismatch_runs_lds(P, s, v, ops, cap, P.seq_base + P.seqs[qi].off, P.seqs[qi].len, P.seq_base + P.seqs[ti].off,
                                                           P.seqs[ti].len, L.bc, L.sh, L.ring, &n2, &arena_block, !last, &block_text);
                    if (r == -1) return false; /* wider than the narrow path: arena class */
                    if (r == -2) return true;  /* arena full: repeated by the host */
                    if (r == -3) {             /* more ops than this store holds */
                        if (need_out) *need_out = n2;
                        return false;
                    }
                    if (r > 0) rc = r;
                    else {
                        if (last) rc = check_record(s, v, L.bc); /* on the ops still in LDS: the runs of an M op add up to it */
                        v.reset(ops, n2);
                        ops_in_arena = true;
                        ops_block_only = last;
                        if (last && !rc) goto add_done;
                    }
                }
                if (!rc) rc = check_record(s, v, L.bc);
            add_done:;
            } else if constexpr (OPS::kNarrow) {
                return false;
            } else {
                const int32_t qi = swapped ? P.rec_tseq[rec] : P.rec_qseq[rec], ti = swapped ? P.rec_qseq[rec] : P.rec_tseq[rec];
                if (qi < 0) rc = PAFFY_ERR_MISSING_QUERY_SEQ;   /* impl/paf_add_mismatches.c:117-120 */
                else if (ti < 0) rc = PAFFY_ERR_MISSING_TARGET_SEQ; /* :123-127 */
                else if (s.has_cigar) {
                    uint32_t n2 = 0;
                    uint64_t blk = 0;
                    rc = encode_mismatch_runs(P, s, v, P.seq_base + P.seqs[qi].off, P.seqs[qi].len, P.seq_base + P.seqs[ti].off,
                                              P.seqs[ti].len, L.bc, L.sh, &n2, &blk);
                    if (!rc) {
                        if (blk == ~0ull) return true; /* arena full: repeated by the host */
                        ops.p = P.arena + blk;
                        v.reset(ops, n2);
                    }
                }
                if (!rc) rc = check_record(s, v, L.bc);
            }
        } else if (STAGE_ON(PAFFY_STATS) && st.kind == PAFFY_STATS) {
            /* paf_stats_calc(.., zero_counts = 0), impl/paf.c:236-260, into the batch's sums (`paffy view -s`, impl/paf_view.c:163-168) */
            if (s.has_cigar) {
                uint32_t b, e;
                sweep_bounds(v.n, b, e);
                int64_t a[3] = {0, 0, 0}, c2[3] = {0, 0, 0}; /* matches, mismatches, insert bases | delete bases, inserts, deletes */
                for (uint32_t i = b; i < e; i++) {
                    int64_t len;
                    int op;
                    v.get(i, len, op);
                    if (op == OP_M || op == OP_EQ) a[0] += len;
                    else if (op == OP_X) a[1] += len;
                    else if (op == OP_I) { a[2] += len; c2[1]++; }
                    else { c2[0] += len; c2[2]++; }
                }
                block_sum<3>(a, L.bc);
                block_sum<3>(c2, L.bc);
                /* the record's six sums; the batch's sums are their reduction (k_stats_reduce behind the sizing launches) -- six atomic
                   adds per record on one cache line made `view -s` eight times slower than `filter` */
                if (threadIdx.x == 0 && P.rec_stats) {
                    int64_t *o = P.rec_stats + 6ull * rec;
                    o[0] = a[0]; o[1] = a[1]; o[2] = c2[1]; o[3] = c2[2]; o[4] = a[2]; o[5] = c2[0];
                }
            } else if (threadIdx.x == 0 && P.rec_stats) {
                for (int k = 0; k < 6; k++) P.rec_stats[6ull * rec + k] = 0; /* cigar_count(NULL) == 0 */
            }
        } else if (STAGE_ON(PAFFY_FILTER) && st.kind == PAFFY_FILTER) {
            /* paffy filter, impl/paf_filter.c:120-156: paf_stats_calc sums from the view's running totals
               (matches = M and =; tx = X + I + D, I = all - tt, D = all - tq) */
            int64_t mm = 0, mx = 0;
            if (s.has_cigar) match_stats(v, mm, mx, L.bc);
            const int64_t all = mm + mx;
            const int64_t ins = s.has_cigar ? all - v.tt : 0, del = s.has_cigar ? all - v.tq : 0;
            const double identity = ratio_f32(mm, mm + (mx - ins - del));
            const double identity_with_gaps = ratio_f32(mm, all);
            /* field by field: a reference to the struct inside the kernel arguments makes the compiler keep a private copy (36 bytes of scratch per lane) */
            const int64_t f_as = P.filter.min_alignment_score, f_cs = P.filter.min_chain_score, f_tl = P.filter.max_tile_level;
            const double f_id = P.filter.min_identity, f_idg = P.filter.min_identity_with_gaps;
            const bool f_inv = P.filter.invert != 0;
            const bool pass = s.score >= f_as && s.chain_score >= f_cs && (f_tl == -1 || s.tile_level <= f_tl) && identity >= f_id && identity_with_gaps >= f_idg;
            if (pass == f_inv) { /* dropped: no output, later stages never see the record */
                if (threadIdx.x == 0) {
                    RecPlan *dp = static_cast<RecPlan *>(P.rec_plan) + rec;
                    P.status[rec] = klass << 16;
                    P.out_len[rec] = 0;
                    P.out_rows[rec] = 0;
                    dp->flags = 128u;
                    dp->n = 0;
                }
                return true;
            }
        } else if (STAGE_ON(PAFFY_SHATTER) && st.kind == PAFFY_SHATTER) {
            shatter = true;
            break;
        }
        if (rc) {
            report(P, rec, rc, si, 0, klass);
            return true;
        }
        if (st.kind != PAFFY_STATS) checked = st.kind != PAFFY_PASS && st.kind != PAFFY_FILTER && (do_check || st.kind == PAFFY_CHECK); /* a passed paf_check ends the stage */
    }
    RecPlan *plan = static_cast<RecPlan *>(P.rec_plan) + rec;
    int64_t bytes, rows;
    bool direct = false, rows_kernel = false, line_kernel = false;
    if (shatter) {
        ShatterConst k;
        shatter_consts(s, k);
        direct = !shatter_fits(k);
        rows_kernel = OPS::kNarrow && !direct && shatter_fast_ok(s, k) && k.lenA <= 48 && k.lenB <= 48 && k.lenC <= 48 &&
                      v.n <= PAFFY_ROWS_MAX_OPS; /* emitted by k_emit_rows; longer records are split over the four waves of k_emit_lds */ /* pieces too long for the LDS staging: the emit pass writes this record's rows straight to HBM */
        int rc = 0;
        bool sized = false;
        if constexpr (std::is_same<OPS, OpsLds>::value) {
            /* plain ops, a passed paf_check, start and end coordinates of equal digit counts, the window's end ops whole, written by the
               one-wave row kernel (which needs no per-wave starts): a row's size depends on L only and nothing can fail -- one lean sweep */
            const uint32_t dq0 = dec_len(s.qs), dt0 = dec_len(s.ts);
            if (parse_plain && rows_kernel && checked && (v.sub_lo | v.sub_hi) == 0 && dq0 == (uint32_t)dec_len(s.qe) && dt0 == (uint32_t)dec_len(s.te) &&
                s.qe - s.qs < 0x7fffffffll && s.te - s.ts < 0x7fffffffll) {
                const uint32_t fixed = k.row_const + 2 * dq0 + 2 * dt0;
                uint32_t b, e, a[2] = {0, 0};
                sweep_bounds(v.n, b, e);
                uint32_t extra = 0; /* digits of the lengths beyond the first, summed over the M ops */
                for (uint32_t i = b; i < e; i++) {
                    const uint32_t w = ops.p[v.lo + i]; /* the order of the ops does not matter here */
                    const uint32_t is_m = (w & 7u) == (uint32_t)OP_M ? 1u : 0u, x = w >> 3;
                    const uint32_t mk = 0u - is_m;
                    a[1] += is_m;
                    extra += mk & ((x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u));
                    if (__any(x >= 100000u)) extra += mk & ((x >= 100000u) + (x >= 1000000u) + (x >= 10000000u) + (x >= 100000000u)); /* lengths stay below 2^29 */
                }
                a[0] = a[1] * (fixed + 3u) + 3u * extra;
                block_sum_u32<2>(a, L.bc);
                bytes = a[0];
                rows = a[1];
                sized = true;
            }
        }
        if (!sized) rc = shatter_size(s, v, k, bytes, rows, plan, checked, L.bc);
        PT_MARK(4)
        if (rc) {
            report(P, rec, rc, si, 0, klass);
            return true;
        }
    } else {
        const bool nl_in_header = !(s.has_cigar && v.n > 0);
        const uint32_t lenH = header_len(s, nl_in_header); /* not header_len_wave: its items cost the mismatch encoder's build eleven registers (and a wave per SIMD) */
        direct = lenH > 3 * PAFFY_TMPL_MAX; /* header too long for the LDS staging: built straight in HBM */
        line_kernel = OPS::kNarrow && lenH + 8 <= PAFFY_TMPL_MAX && v.n <= PAFFY_ROWS_MAX_OPS; /* written by k_emit_line */
        bytes = lenH;
        if (!nl_in_header) {
            if (ops_block_only && line_kernel) { /* one wave writes the line: no per-wave starts needed */
                bytes += block_text + 1;
            } else if (ops_block_only) {
                View<OpsCoherent> gv;
                gv.reset(OpsCoherent{reinterpret_cast<const uint32_t *>(P.arena + arena_block)}, v.n);
                bytes += cigar_text_len(gv, plan, L.bc) + 1;
            } else {
                bytes += cigar_text_len(v, plan, L.bc) + 1;
            }
        }
        rows = 1;
    }
    if (threadIdx.x == 0) {
        if (klass == KLASS_LDS && (shatter ? !rows_kernel : !line_kernel)) atomicAdd(&P.info->g_count, 1u);
        P.status[rec] = klass << 16;
        P.out_len[rec] = bytes;
        P.out_rows[rec] = rows;
        plan->qs = s.qs; plan->qe = s.qe; plan->ts = s.ts; plan->te = s.te; plan->sub_lo = v.sub_lo; plan->sub_hi = v.sub_hi;
        plan->lo = v.lo; plan->n = v.n;
        plan->flags = (v.rev ? 1u : 0u) | (v.swp ? 2u : 0u) | (swapped ? 4u : 0u) | (s.has_cigar ? 8u : 0u) | ((uint32_t)s.type << 8) |
                      (shatter ? 16u : 0u) | (direct ? 32u : 0u) | (rows_kernel ? 64u : 0u) | (line_kernel ? 0x10000u : 0u) |
                      (ops_in_arena ? 0x20000u : 0u) | (mirror16 && !ops_in_arena ? 0x40000u : 0u);
        if (ops_in_arena) P.arena_off[rec] = arena_block;
        plan->chunk = ((v.n + PAFFY_NT - 1) / PAFFY_NT) | 1u; /* = sweep_bounds() */
#if PAFFY_NWAVE == 1
        /* sized by one wave: it owns all the ops (chunk covers them), the other three waves of a four-wave writer get empty shares */
        for (int w = 1; w < 4; w++) plan->wq[w] = plan->wt[w] = plan->wo[w] = 0;
#endif
    }
#if defined(PAFFY_ABL) && PAFFY_ABL == 21
    PT_MARK(5)
    if ((blockIdx.x & 8191u) == 77u && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == (PAFFY_NWAVE == 1 ? 0 : 1))
        printf("g%d rec %u ops %u: parse %llu invert+check %llu trim %llu check %llu shatter_size %llu plan %llu\n", PAFFY_NT, blockIdx.x, n, pt_acc[0], pt_acc[1], pt_acc[2],
               pt_acc[3], pt_acc[4], pt_acc[5]);
#endif
    return true;
}

/*
 * Emit pass: the record is known to be valid; resume from its RecPlan with the ops in HBM.
 * SHATTER selects the terminal at compile time (the host knows the pipe), so the row kernel
 * carries none of the whole-line writer and vice versa.
 */
template <class OPS, bool SHATTER>
__device__ __forceinline__ void emit_record(const KParams &P, uint32_t rec, const OPS &ops, const RecLds &L) {
    /* by reference: a by-value copy of these structs (indexed per wave below) would live in scratch memory */
    const RecMeta &m = P.meta[rec];
    const RecPlan &pl = static_cast<const RecPlan *>(P.rec_plan)[rec];
    if (pl.flags & 128u) return; /* dropped by a filter stage */
    RecState s;
    load_state(m, s);
    if (pl.flags & 4u) invert_state(s);
    s.qs = pl.qs; s.qe = pl.qe; s.ts = pl.ts; s.te = pl.te;
    s.has_cigar = (pl.flags & 8u) != 0;
    s.type = (uint8_t)(pl.flags >> 8);
    View<OPS> v;
    v.reset(ops, pl.n);
    v.lo = pl.lo; v.rev = pl.flags & 1u; v.swp = (pl.flags & 2u) != 0;
    v.sub_lo = pl.sub_lo; v.sub_hi = pl.sub_hi;
    if constexpr (SHATTER) {
        ShatterConst k;
        shatter_consts(s, k);
        if (pl.flags & 32u) {
            shatter_emit_direct(P, s, v, k, pl, (uint64_t)P.out_off[rec]);
            return;
        }
        uint64_t *A = L.pieces, *B = L.pieces + PAFFY_TMPL_MAX / 8, *C = L.pieces + 2 * (PAFFY_TMPL_MAX / 8);
        const uint32_t wave = threadIdx.x >> 6;
        if (wave == 0) {
            Piece w{(uint8_t *)A, 0, PAFFY_TMPL_MAX, false};
            w.name(P.in, s.qn_off, s.qn_len);
            w.ch('\t'); w.num(s.qlen); w.ch('\t');
            w.pad_to(((w.n + 31u) & ~15u) < 32u ? 32u : ((w.n + 31u) & ~15u));
        } else if (wave == 1) {
            Piece w{(uint8_t *)B, 0, PAFFY_TMPL_MAX, false};
            w.ch('\t'); w.ch(s.same ? '+' : '-'); w.ch('\t');
            w.name(P.in, s.tn_off, s.tn_len);
            w.ch('\t'); w.num(s.tlen); w.ch('\t');
            w.pad_to(((w.n + 31u) & ~15u) < 32u ? 32u : ((w.n + 31u) & ~15u));
        } else if (wave == 2) {
            Piece w{(uint8_t *)C, 0, PAFFY_TMPL_MAX, false};
            w.ch('\t'); w.num(s.mapq);
            piece_tags(w, s, 0);
            w.str("\tcg:Z:", 6);
            w.pad_to(((w.n + 31u) & ~15u) < 48u ? 48u : ((w.n + 31u) & ~15u));
        }
        __syncthreads();
        if (shatter_fast_ok(s, k)) {
            const uint64_t span = 64ull * pl.chunk;
            const uint32_t wb = span * wave < v.n ? (uint32_t)(span * wave) : v.n;
            const uint32_t we = span * (wave + 1) < v.n ? (uint32_t)(span * (wave + 1)) : v.n;
            shatter_emit_fast(s, v, k, reinterpret_cast<const u32x4 *>(A), reinterpret_cast<const u32x4 *>(B), reinterpret_cast<const u32x4 *>(C), wb, we,
                              pl.wq[wave], pl.wt[wave], L.ring + wave * PAFFY_WAVE_RING, P.out,
                              uniform_u64((uint64_t)P.out_off[rec] + (uint64_t)pl.wo[wave]));
            return;
        }
        RowPieces pieces;
        load_pieces(pieces, k, A, B, C);
        shatter_emit(s, v, k, pieces, pl, L.ring, P.out, (uint64_t)P.out_off[rec]);
    } else {
        const bool nl_in_header = !(s.has_cigar && v.n > 0);
        const uint32_t lenH = header_len_wave(s, nl_in_header);
        const bool direct = (pl.flags & 32u) != 0;
        if (threadIdx.x < 64) { /* wave 0 builds the header: in LDS, or straight in the output when it is too long */
            Piece w{direct ? P.out + P.out_off[rec] : (uint8_t *)L.pieces, 0, direct ? lenH : 3 * PAFFY_TMPL_MAX, false};
            build_header(w, s, P.in, nl_in_header);
        }
        __syncthreads();
        write_emit(v, s.has_cigar, L.pieces, lenH, pl, L.ring, P.out, (uint64_t)P.out_off[rec], direct);
    }
}

/* LDS budgets: both kernels fit four workgroups per CU (160 KiB). */
#define PAFFY_SIZE_LDS_BYTES_FOR(cap) ((cap) * 4 + (PAFFY_HALO + PAFFY_NT * 16) + 64 * 8 + 64)
#define PAFFY_SIZE_LDS_BYTES PAFFY_SIZE_LDS_BYTES_FOR(PAFFY_OPS_CAP)
#ifndef PAFFY_OPS_CAP_MID
#define PAFFY_OPS_CAP_MID 16384u /* second sizing level: 64 KiB of ops, two workgroups per CU */
#endif
#define PAFFY_OPS_CAP_BIG 36864u /* third sizing level: 144 KiB of ops, one workgroup per CU */
#define PAFFY_EMIT_LDS_BYTES (PAFFY_NWAVE * PAFFY_WAVE_RING + 3 * PAFFY_TMPL_MAX + 64 * 8 + 64)

__device__ __forceinline__ RecLds carve_size_lds(uint8_t *smem, uint32_t **ops_lds, uint32_t cap = PAFFY_OPS_CAP) {
    RecLds L;
    *ops_lds = reinterpret_cast<uint32_t *>(smem);
    L.ring = smem + cap * 4; /* text staging */
    L.pieces = nullptr;
    L.bc.scratch = reinterpret_cast<int64_t *>(smem + cap * 4 + PAFFY_HALO + PAFFY_NT * 16);
    L.bc.flip = 0;
    L.sh = reinterpret_cast<Shared *>(smem + cap * 4 + PAFFY_HALO + PAFFY_NT * 16 + 64 * 8);
    return L;
}
__device__ __forceinline__ RecLds carve_emit_lds(uint8_t *smem) {
    RecLds L;
    L.ring = smem;
    L.pieces = reinterpret_cast<uint64_t *>(smem + PAFFY_NWAVE * PAFFY_WAVE_RING);
    L.bc.scratch = reinterpret_cast<int64_t *>(smem + PAFFY_NWAVE * PAFFY_WAVE_RING + 3 * PAFFY_TMPL_MAX);
    L.bc.flip = 0;
    L.sh = reinterpret_cast<Shared *>(smem + PAFFY_NWAVE * PAFFY_WAVE_RING + 3 * PAFFY_TMPL_MAX + 64 * 8);
    return L;
}

/* Where the HBM mirror of a record's 4-byte ops lives: cigars do not overlap and an op takes at
 * least two text bytes (digits + letter), so index cg_off / 2 is private to the record. */
__device__ __forceinline__ uint32_t mirror_index(const RecMeta &m) { return m.cg_off >> 1; }
/* the 4-byte ops the emit pass reads for an LDS-class record */
__device__ __forceinline__ const uint32_t *emit_ops_of(const KParams &P, uint32_t rec, const RecMeta &m, const RecPlan &pl) {
    /* bit 20: the ops stand in new_ops[] (flat_add_kernel.h: the new cigars of all records back to back), arena_off counts its 4-byte words */
    if (pl.flags & 0x100000u) return P.new_ops + P.arena_off[rec];
    return (pl.flags & 0x20000u) ? reinterpret_cast<const uint32_t *>(P.arena + P.arena_off[rec]) : P.ops_mirror + mirror_index(m);
}
__device__ __forceinline__ bool emit_ops_half(const RecPlan &pl) { return (pl.flags & 0x60000u) == 0x40000u; } /* the mirror, in 2-byte words */

/*
 * Sizing, LDS class: one workgroup per record, ops parsed from the text into LDS (and mirrored).
 * Launched three times with growing LDS stores: the whole batch with 8192 ops (4 workgroups per
 * CU), what did not fit with 16384 ops (2 per CU), then with 36864 ops (1 per CU). What still does
 * not fit (lengths >= 2^29, rebuilt op arrays) goes to the arena kernel.
 */
template <uint32_t MASK>
__device__ __forceinline__ void size_lds_one(const KParams &P, uint32_t rec, uint32_t *ops_lds, const RecLds &L) {
    const RecMeta &m = P.meta[rec];
    OpsLds ops{ops_lds, P.ops_mirror + mirror_index(m), (m.cg_len + 1) >> 1};
    uint32_t n_ops = 0, need = 0; /* need: ops of an array rebuilt by add_mismatches that did not fit this level's store */
    bool ok = size_record<OpsLds, MASK>(P, rec, ops, P.ops_cap, L, KLASS_LDS, &n_ops, &need);
    if (ok && n_ops > ((m.cg_len + 1) >> 1)) ok = false; /* digit-less ops overran the mirror: arena class */
#if PAFFY_NWAVE == 1
    /* the one-wave build: more ops than its store holds -> the four-wave build takes the record (it runs after this kernel) */
    const bool to_four = !ok && need == 0 && n_ops > P.ops_cap && n_ops <= ((m.cg_len + 1) >> 1);
    if (threadIdx.x == 0) P.n_ops[rec] = to_four ? 0xffffffffu : 0u;
    if (to_four) return;
#endif
#if PAFFY_NWAVE == 4
    if (threadIdx.x == 0 && need == 0) {
        if (P.level == 0 && !ok && n_ops > P.ops_cap) atomicAdd(&P.info->lvl0_over, 1u); /* denser than the longer first level assumed */
        if (P.level == 1 && P.lvl0_long_bytes && m.cg_len <= P.lvl0_long_bytes && n_ops > PAFFY_OPS_CAP) atomicAdd(&P.info->lvl0_probe_dense, 1u);
    }
#endif
    if (!ok && threadIdx.x == 0) {
        P.out_len[rec] = 0;
        P.out_rows[rec] = 0;
        /* too many ops for this level's store only: try the next, bigger store; anything else needs the arena */
        const bool try_next = P.next_cap != 0 && (need ? need <= PAFFY_OPS_CAP_BIG
                                                        : (n_ops > P.ops_cap && n_ops <= PAFFY_OPS_CAP_BIG && n_ops <= ((m.cg_len + 1) >> 1)));
        if (try_next) {
            P.b_list[P.level][atomicAdd(&P.info->b_count[P.level], 1u)] = rec;
        } else {
            P.status[rec] = (uint32_t)KLASS_ARENA << 16;
            P.w_list[atomicAdd(&P.info->w_count, 1u)] = rec;
        }
    }
}
#ifndef PAFFY_SIZE_OCC
#define PAFFY_SIZE_OCC 4
#endif
template <uint32_t MASK>
__global__ __launch_bounds__(PAFFY_NT, PAFFY_SIZE_OCC) void k_size_lds(KParams P) { /* the whole batch, 8192-op store */
    extern __shared__ uint4 smem4[];
    uint32_t *ops_lds;
    RecLds L = carve_size_lds(reinterpret_cast<uint8_t *>(smem4), &ops_lds, P.ops_cap);
    const uint32_t rec = P.size_order ? P.size_order[blockIdx.x] : blockIdx.x; /* long cigars first */
    if (P.flat_done && P.flat_done[rec]) return; /* sized by the flat pass (flat_kernel.h) */
    /* records whose cigar text promises more ops than this store holds were queued for level 1 by k_header */
    if ((P.meta[rec].cg_len >> 1) > P.lvl0_max && P.meta[rec].err == 0) return;
    /* short cigars belong to the one-wave build of this kernel (g64, launched first), everything else -- and what did not fit the
       wave's op store, marked in n_ops -- to the four-wave build */
    const bool is_short = P.meta[rec].cg_len <= P.wave_max_bytes;
    if (PAFFY_NWAVE == 1 ? !is_short : (is_short && P.n_ops[rec] != 0xffffffffu)) return;
    size_lds_one<MASK>(P, rec, ops_lds, L);
}
#if PAFFY_NWAVE == 4 /* the long-record levels, every emit kernel and the arena class exist in the four-wave build only */
template <uint32_t MASK>
__global__ __launch_bounds__(PAFFY_NT, 2) void k_size_lds_long(KParams P) { /* two workgroups per CU at most (LDS): room for 256 registers, no spills */ /* levels 1 and 2: the queued long records */
    extern __shared__ uint4 smem4[];
    uint32_t *ops_lds;
    RecLds L = carve_size_lds(reinterpret_cast<uint8_t *>(smem4), &ops_lds, P.ops_cap);
    const uint32_t count = P.info->b_count[P.level - 1];
    for (uint32_t li = blockIdx.x; li < count; li += gridDim.x) {
        const uint32_t rec = P.b_list[P.level - 1][li];
        if (!(P.flat_done && P.flat_done[rec])) size_lds_one<MASK>(P, rec, ops_lds, L); /* not sized by the flat pass (flat_kernel.h) */
        __syncthreads();
    }
}

/* Emit, LDS class. */
#ifndef PAFFY_EMIT_OCC
#define PAFFY_EMIT_OCC 4
#endif
template <bool SHATTER>
__global__ __launch_bounds__(PAFFY_NT, PAFFY_EMIT_OCC) void k_emit_lds(KParams P) {
    extern __shared__ uint4 smem4[];
    RecLds L = carve_emit_lds(reinterpret_cast<uint8_t *>(smem4));
    const uint32_t rec = blockIdx.x;
    if (rec >= (uint32_t)(P.info->first_err_key >> 16)) return; /* nothing at or after the first failure */
    if ((P.status[rec] >> 16) != KLASS_LDS) return;
    if (static_cast<const RecPlan *>(P.rec_plan)[rec].flags & (SHATTER ? (64u | 0x80000u) : (0x10000u | 0x80000u | 0x400000u))) return; /* k_emit_rows / k_emit_line / k_emit_copy has it */
    OpsGlobal ops{emit_ops_of(P, rec, P.meta[rec], static_cast<const RecPlan *>(P.rec_plan)[rec]), emit_ops_half(static_cast<const RecPlan *>(P.rec_plan)[rec])};
#if defined(PAFFY_ABL) && PAFFY_ABL == 22
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
#endif
    emit_record<OpsGlobal, SHATTER>(P, rec, ops, L);
#if defined(PAFFY_ABL) && PAFFY_ABL == 22
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if ((rec & 4095u) == 99u && threadIdx.x == 0) printf("rec %u: %llu core cycles, %llu wall ticks (100 MHz) -> %.0f MHz\n", rec, c1 - c0, w1 - w0, (double)(c1 - c0) / (double)(w1 - w0) * 100.0);
#endif
}

/*
 * Row kernel: one WAVE per record for the usual shatter record (pieces of at most 48 bytes, coordinates below
 * 10^11, ops in the HBM mirror). The per-record preparation runs once instead of once per wave of a workgroup,
 * there is no barrier at all, and the kernel carries none of the general paths.
 */
/*
 * The three constant pieces of a record's rows (paf_shatter2 + paf_write, impl/paf.c:600-627, 317-368) for the one-wave row kernel, 64 zero
 * filled bytes each at p, p + 64, p + 128 -- one item per lane like build_header: thirteen items, one decimal conversion per lane
 * (the Piece form converted the three to five numbers one after the other, each in all 64 lanes). Pieces are at most 48 bytes here.
 */
__device__ __forceinline__ void row_pieces_lanes(uint8_t *p, const RecState &s, const uint8_t *in) {
    const uint32_t k = threadIdx.x & 63u;
    if (k < 12) reinterpret_cast<uint4 *>(p)[k] = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_wave_barrier(); /* a wave's LDS operations execute in order */
    HeaderItem it;
    it.val = 0; it.pre = '\t'; it.plen = 1; it.name_len = 0; it.numeric = false; it.present = k < 13;
    if (k == 0) { it.plen = 0; it.name_len = s.qn_len; }
    if (k == 1) { it.numeric = true; it.val = s.qlen; }
    if (k == 3) { it.pre = (uint64_t)'\t' | ((uint64_t)(s.same ? '+' : '-') << 8) | ((uint64_t)'\t' << 16); it.plen = 3; it.name_len = s.tn_len; }
    if (k == 4) { it.numeric = true; it.val = s.tlen; }
    if (k == 6) { it.numeric = true; it.val = s.mapq; }
    if (k >= 7 && k <= 11) { /* the tags of header_item(), children carry s1:i:0 (calloc, impl/paf.c:601) */
        RecState c = s;
        c.chain_score = 0;
        it = header_item(c, false, k + 5u);
    }
    if (k == 12) { it.pre = header_tag6('c', 'g', 'Z'); it.plen = 6; }
    DecText d;
    dec_text(it.val, d);
    const uint32_t len = it.present ? it.plen + it.name_len + (it.numeric ? text_len(d) : 0u) : 0u;
    const uint32_t inc = wave_incl_scan_u32(len), ex = inc - len;
    const uint32_t seg = k < 3 ? 0u : (k < 6 ? 1u : 2u);
    const uint32_t seg_first = seg == 0 ? 0u : (seg == 1 ? 3u : 6u);
    const uint32_t at0 = ex - (uint32_t)__shfl((int)ex, (int)seg_first);
    uint8_t *q = p + 64u * seg;
    if (it.present) {
        uint32_t at = at0;
        header_put(q, 64, at, it.pre, it.plen);
        at += it.plen;
        if (it.numeric) {
            if (d.neg_separate) {
                header_put(q, 64, at, '-', 1);
                at += 1;
            }
            header_put(q, 64, at, d.top, d.ntop);
            at += d.ntop;
            if (d.groups == 2) {
                header_put(q, 64, at, ascii8(d.g1), 8);
                at += 8;
            }
            if (d.groups >= 1) header_put(q, 64, at, ascii8(d.g0), 8);
        }
    }
    const uint32_t t_at = (uint32_t)__shfl((int)at0, 3) + 3u; /* the target name follows "\t+\t" */
    for (uint32_t i = k; i < s.qn_len; i += 64)
        if (i < 64) p[i] = in[s.qn_off + i];
    for (uint32_t i = k; i < s.tn_len; i += 64)
        if (t_at + i < 64) p[64 + t_at + i] = in[s.tn_off + i];
}
#define PAFFY_ROWS_LDS_BYTES (PAFFY_WAVE_RING + 3 * 64 + 64)
#ifndef PAFFY_ROWS_OCC
#define PAFFY_ROWS_OCC 4
#endif
__global__ __launch_bounds__(64, PAFFY_ROWS_OCC) void k_emit_rows(KParams P) {
    extern __shared__ uint4 smem4[];
    uint8_t *smem = reinterpret_cast<uint8_t *>(smem4);
    /* the first n_items workgroups write the segments of the long records (flat_kernel.h cuts a record of more than PAFFY_ROWS_MAX_OPS ops
       into segments of a quarter of that: any length of record is just more workgroups, and they start first), the others one record each */
    const bool is_item = blockIdx.x < P.n_items;
    uint32_t rec, wb = 0, we = 0;
    int64_t cq0 = 0, ct0 = 0, wo = 0;
    if (is_item) {
        const EmitItem &it = P.items[blockIdx.x];
        rec = it.rec; wb = it.wb; we = it.we; cq0 = it.cq0; ct0 = it.ct0; wo = it.wo;
    } else {
        rec = P.emit_order ? P.emit_order[blockIdx.x - P.n_items] : blockIdx.x - P.n_items; /* long records first */
    }
    if (rec >= (uint32_t)(P.info->first_err_key >> 16)) return; /* nothing at or after the first failure */
    if ((P.status[rec] >> 16) != KLASS_LDS) return;
    const RecPlan &pl = static_cast<const RecPlan *>(P.rec_plan)[rec];
    if (!(pl.flags & (is_item ? 0x80000u : 64u))) return;
    const RecMeta &m = P.meta[rec];
    RecState s;
    load_state(m, s);
    if (pl.flags & 4u) invert_state(s);
    s.qs = pl.qs; s.qe = pl.qe; s.ts = pl.ts; s.te = pl.te;
    s.has_cigar = (pl.flags & 8u) != 0;
    s.type = (uint8_t)(pl.flags >> 8);
    OpsGlobal ops{emit_ops_of(P, rec, m, pl), emit_ops_half(pl)};
    View<OpsGlobal> v;
    v.reset(ops, pl.n);
    v.lo = pl.lo; v.rev = pl.flags & 1u; v.swp = (pl.flags & 2u) != 0;
    v.sub_lo = pl.sub_lo; v.sub_hi = pl.sub_hi;
    if (!is_item) we = v.n;
#if defined(PAFFY_ABL) && PAFFY_ABL == 42 /* 42: one window per record (what a record costs before and after its rows) */
    if (we > wb + 128u) we = wb + 128u;
#endif
    ShatterConst k;
    shatter_consts(s, k);
    uint8_t *A = smem + PAFFY_WAVE_RING, *B = A + 64, *C = B + 64;
#if !defined(PAFFY_ABL) || PAFFY_ABL != 41 /* 41: without the per-record pieces (their cost) */
    if (pl.flags & 0x200000u) { /* the flat sizing pass left the pieces in HBM (row_pieces_serial, flat_kernel.h) */
        if (threadIdx.x < 48) /* 3 x 48 bytes there, 3 x 64 zero filled bytes here */
            reinterpret_cast<uint32_t *>(A)[threadIdx.x] =
                (threadIdx.x & 15u) < 12u ? reinterpret_cast<const uint32_t *>(P.row_pieces + 144ull * rec)[12u * (threadIdx.x >> 4) + (threadIdx.x & 15u)] : 0u;
    } else
    row_pieces_lanes(A, s, P.in); /* A = qname \t qlen \t | B = \t strand \t tname \t tlen \t | C = \t mapq tags \tcg:Z: -- 64 bytes each, zero filled */
#endif
    __builtin_amdgcn_wave_barrier();
#if defined(PAFFY_ABL) && PAFFY_ABL == 24 /* core clock against the 100 MHz wall clock inside the row kernel */
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
#endif
    shatter_emit_fast(s, v, k, reinterpret_cast<const u32x4 *>(A), reinterpret_cast<const u32x4 *>(B), reinterpret_cast<const u32x4 *>(C), wb, we, cq0, ct0,
                      smem, P.out, (uint64_t)P.out_off[rec] + (uint64_t)wo);
#if defined(PAFFY_ABL) && PAFFY_ABL == 24
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if ((blockIdx.x & 16383u) == 99u && threadIdx.x == 0)
        printf("block %u (%u ops): %llu core cycles, %llu wall ticks (100 MHz) -> %.0f MHz\n", blockIdx.x, v.n, c1 - c0, w1 - w0, (double)(c1 - c0) / (double)(w1 - w0) * 100.0);
#endif
}

/*
 * Line kernel: one WAVE per record for the whole-line pipes (invert, trim, filter, add_mismatches -a ...) when the header
 * fits 768 bytes and the ops are in the HBM mirror: preparation once per record, no barrier.
 */
#define PAFFY_LINE_LDS_BYTES (PAFFY_WAVE_RING + PAFFY_TMPL_MAX + 64)
__global__ __launch_bounds__(64, PAFFY_EMIT_OCC) void k_emit_line(KParams P) {
    extern __shared__ uint4 smem4[];
    uint8_t *smem = reinterpret_cast<uint8_t *>(smem4);
    /* the first n_items workgroups write the segments of the long lines (flat_add_kernel.h cuts the cigar of a record that becomes more
       than PAFFY_ROWS_MAX_OPS ops into segments: the header goes with the first), the others one record each */
    const bool is_item = blockIdx.x < P.n_items;
    uint32_t rec, wb = 0, we = 0;
    int64_t wo = 0;
    if (is_item) {
        const EmitItem &it = P.items[blockIdx.x];
        rec = it.rec; wb = it.wb; we = it.we; wo = it.wo;
    } else {
        rec = P.emit_order ? P.emit_order[blockIdx.x - P.n_items] : blockIdx.x - P.n_items; /* long records first */
    }
    if (rec >= (uint32_t)(P.info->first_err_key >> 16)) return;
    if ((P.status[rec] >> 16) != KLASS_LDS) return;
    const RecPlan &pl = static_cast<const RecPlan *>(P.rec_plan)[rec];
    if (!(pl.flags & (is_item ? 0x80000u : 0x10000u))) return; /* long header or dropped: k_emit_lds<line> */
    const RecMeta &m = P.meta[rec];
    RecState s;
    load_state(m, s);
    if (pl.flags & 4u) invert_state(s);
    s.qs = pl.qs; s.qe = pl.qe; s.ts = pl.ts; s.te = pl.te;
    s.has_cigar = (pl.flags & 8u) != 0;
    s.type = (uint8_t)(pl.flags >> 8);
    OpsGlobal ops{emit_ops_of(P, rec, m, pl), emit_ops_half(pl)};
    View<OpsGlobal> v;
    v.reset(ops, pl.n);
    v.lo = pl.lo; v.rev = pl.flags & 1u; v.swp = (pl.flags & 2u) != 0;
    v.sub_lo = pl.sub_lo; v.sub_hi = pl.sub_hi;
    const bool nl_in_header = !(s.has_cigar && v.n > 0);
    const uint32_t lenH = header_len_wave(s, nl_in_header);
    uint64_t *H = reinterpret_cast<uint64_t *>(smem + PAFFY_WAVE_RING);
    const bool with_header = !is_item || wb == 0;
    if (with_header) {
        Piece w{(uint8_t *)H, 0, PAFFY_TMPL_MAX, false};
        build_header(w, s, P.in, nl_in_header);
    }
    __builtin_amdgcn_wave_barrier();
    if (!is_item) we = nl_in_header ? 0u : v.n;
    write_emit_range(v, wb, we, with_header, H, lenH, smem, P.out, (uint64_t)P.out_off[rec] + (with_header ? 0ull : (uint64_t)lenH + (uint64_t)wo));
}

/*
 * Copy kernel: one wave per record for the whole-line pipes of the flat pass whose window of ops is not reversed (every pipe's + strand
 * records; all records when no invert stands in the pipe): the line's cigar is then a stretch of the input's own cigar text -- what the
 * flat pass keeps is text as paf_write prints it (impl/paf.c:381-385: no leading zeros, lengths below 8 192, letters MID=X) -- with I and D
 * swapped under an invert (impl/paf.c:469-490). No op is looked at: the header as k_emit_line builds it, then 16 bytes per lane and
 * step from the text (RecPlan flag bit 22; wq[0] = first byte from the cigar's start, wt[0] = bytes), a newline. k_emit_line formats
 * 64 ops (176 bytes) in 107 instructions; this moves a kilobyte in about fifty.
 */
#define PAFFY_COPY_LDS_BYTES (PAFFY_TMPL_MAX + 64)
__device__ __forceinline__ uint32_t swap_id4(uint32_t w) { /* 'I' (0x49) <-> 'D' (0x44) in four bytes of cigar text */
    const uint32_t hit = (~nonzero4(w ^ 0x49494949u) | ~nonzero4(w ^ 0x44444444u)) & 0x80808080u;
    const uint32_t m = hit >> 7;
    return w ^ (m | (m << 2) | (m << 3)); /* 0x0d per hit */
}
__global__ __launch_bounds__(64, 8) void k_emit_copy(KParams P) {
    extern __shared__ uint4 smem4[];
    uint8_t *H = reinterpret_cast<uint8_t *>(smem4);
    const uint32_t lane = threadIdx.x;
    const uint32_t rec = P.emit_order ? P.emit_order[blockIdx.x] : blockIdx.x;
    if (rec >= (uint32_t)(P.info->first_err_key >> 16)) return;
    if ((P.status[rec] >> 16) != KLASS_LDS) return;
    const RecPlan &pl = static_cast<const RecPlan *>(P.rec_plan)[rec];
    if (!(pl.flags & 0x400000u)) return;
    const RecMeta &m = P.meta[rec];
    RecState s;
    load_state(m, s);
    if (pl.flags & 4u) invert_state(s);
    s.qs = pl.qs; s.qe = pl.qe; s.ts = pl.ts; s.te = pl.te;
    s.has_cigar = true;
    s.type = (uint8_t)(pl.flags >> 8);
    const uint32_t lenH = header_len_wave(s, false);
    Piece w{H, 0, PAFFY_TMPL_MAX, false};
    build_header(w, s, P.in, false);
    __builtin_amdgcn_wave_barrier();
    uint8_t *out = P.out + P.out_off[rec];
    for (uint32_t b = lane * 16u; b < lenH; b += 1024u) {
        if (b + 16u <= lenH) {
            *reinterpret_cast<u32x4_unaligned *>(out + b) = *reinterpret_cast<const u32x4 *>(H + b);
        } else {
            for (uint32_t k = b; k < lenH; k++) out[k] = H[k];
        }
    }
    const uint8_t *src = P.in + m.cg_off + (uint32_t)pl.wq[0];
    uint32_t n = (uint32_t)pl.wt[0];
    const bool swp = (pl.flags & 2u) != 0;
    uint8_t *dst = out + lenH;
    /* end ops a fixed trim has shortened (wq[1] / wt[1] = new length of the first / last op, wq[2] / wt[2] = bytes of its text in the
       input): written anew by lane 0, the stretch between them copied */
    const uint32_t head_len = (uint32_t)pl.wq[1], head_in = (uint32_t)pl.wq[2], tail_len = (uint32_t)pl.wt[1], tail_in = (uint32_t)pl.wt[2];
    if (head_len | tail_len) {
        auto put_op = [&](uint8_t *p, uint32_t len, uint8_t letter) -> uint32_t { /* lengths below 8 192 */
            const uint32_t d = 1u + (len >= 10u) + (len >= 100u) + (len >= 1000u);
            uint32_t x = len;
            for (uint32_t k = d; k-- > 0;) { p[k] = (uint8_t)('0' + x % 10u); x /= 10u; }
            p[d] = swp ? (letter == 'I' ? (uint8_t)'D' : (letter == 'D' ? (uint8_t)'I' : letter)) : letter;
            return d + 1u;
        };
        uint32_t head_out = 0;
        if (head_len) {
            const uint8_t letter = src[head_in - 1u];
            head_out = 1u + (head_len >= 10u) + (head_len >= 100u) + (head_len >= 1000u) + 1u;
            if (lane == 0) put_op(dst, head_len, letter);
        }
        const uint32_t n_mid = n - head_in - tail_in;
        if (tail_len && lane == 0) put_op(dst + head_out + n_mid, tail_len, src[n - 1u]);
        const uint32_t tail_out = tail_len ? 1u + (tail_len >= 10u) + (tail_len >= 100u) + (tail_len >= 1000u) + 1u : 0u;
        if (lane == 0) dst[head_out + n_mid + tail_out] = '\n';
        src += head_in;
        dst += head_out;
        n = n_mid;
    } else if (lane == 0) {
        dst[n] = '\n';
    }
    for (uint32_t i0 = 0; i0 < n; i0 += 4096u) { /* four loads of a lane in flight */
        u32x4 t[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t i = i0 + k * 1024u + lane * 16u;
            t[k] = u32x4{0, 0, 0, 0};
            if (i + 16u <= n) t[k] = *reinterpret_cast<const u32x4_unaligned *>(src + i);
        }
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t i = i0 + k * 1024u + lane * 16u;
            if (i + 16u <= n) {
                if (swp) t[k] = u32x4{swap_id4(t[k].x), swap_id4(t[k].y), swap_id4(t[k].z), swap_id4(t[k].w)};
                *reinterpret_cast<u32x4_unaligned *>(dst + i) = t[k];
            } else if (i < n) { /* the text's last bytes */
                for (uint32_t j = i; j < n; j++) {
                    const uint8_t c = src[j];
                    dst[j] = swp ? (c == 'I' ? (uint8_t)'D' : (c == 'D' ? (uint8_t)'I' : c)) : c;
                }
            }
        }
    }
}

/* Arena class: records whose ops do not fit LDS; persistent workgroups walk the list. */
__global__ __launch_bounds__(PAFFY_NT) void k_arena_size(KParams P) {
    extern __shared__ uint4 smem4[];
    uint32_t *ops_lds;
    RecLds L = carve_size_lds(reinterpret_cast<uint8_t *>(smem4), &ops_lds);
    const uint32_t count = P.info->w_count;
    for (uint32_t li = blockIdx.x; li < count; li += gridDim.x) {
        const uint32_t rec = P.w_list[li];
        /* upper bound for the allocation: one op per cigar byte */
        const uint32_t cg_len = P.meta[rec].cg_len;
        if (threadIdx.x == 0) L.sh->bcast[3] = (int64_t)atomicAdd(&P.info->arena_used, (unsigned long long)cg_len);
        __syncthreads();
        const uint64_t off = (uint64_t)L.sh->bcast[3];
        __syncthreads();
        if (off + cg_len <= P.arena_cap) {
            OpsArena ops{P.arena + off};
            uint32_t n_ops = 0;
            size_record<OpsArena>(P, rec, ops, cg_len, L, KLASS_ARENA, &n_ops);
            if (threadIdx.x == 0) {
                P.n_ops[rec] = n_ops;
                P.arena_off[rec] = (uint64_t)(ops.p - P.arena); /* a stage may have moved the ops to a new block */
            }
        }
        __syncthreads();
    }
}
template <bool SHATTER>
__global__ __launch_bounds__(PAFFY_NT) void k_arena_emit(KParams P) {
    extern __shared__ uint4 smem4[];
    RecLds L = carve_emit_lds(reinterpret_cast<uint8_t *>(smem4));
    const uint32_t count = P.info->w_count;
    const uint32_t first_err = (uint32_t)(P.info->first_err_key >> 16);
    for (uint32_t li = blockIdx.x; li < count; li += gridDim.x) {
        const uint32_t rec = P.w_list[li];
        if (rec < first_err && (P.status[rec] & 0xff) == 0) {
            OpsArena ops{P.arena + P.arena_off[rec]};
            emit_record<OpsArena, SHATTER>(P, rec, ops, L);
        }
        __syncthreads();
    }
}

#endif /* PAFFY_NWAVE == 4 */
} /* namespace */
