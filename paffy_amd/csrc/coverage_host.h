/*
 * coverage_host.h -- host side of `paffy tile` and `paffy to_bed` over any number of text batches (included by paffy_hip.hip).
 *
 * The reference reads the whole file first (read_pafs, impl/paf.c:492-499 / the loop of impl/paf_to_bed.c:166-190) and keeps it in
 * memory; here every batch of text (< 2 GiB each, 32-bit offsets inside a batch) stays resident in HBM, the records of all batches
 * are ordered and grouped on the device (radix sorts), and the coverage walk runs over chunks of entries whose bitmaps fit a budget.
 */
#ifndef PAFFY_COVERAGE_HOST_H_
#define PAFFY_COVERAGE_HOST_H_

#include <rocprim/rocprim.hpp>

struct CovBatch {
    const uint8_t *in;
    uint32_t len, n;
    uint64_t first; /* index of its first record in the concatenated arrays */
};

struct CovState {
    std::vector<CovBatch> batches;
    uint64_t n_rec = 0;
    DevBuf meta, batch_ptrs, key_score, key_chain, idx, info;
    DevBuf tmp, k64a, k64b, v32a, v32b, order, entries, name_hash, seq_len, flags, scan32, first_entry, contig_len, contig_cov, contig_slice0;
    DevBuf bm_words, n_pairs, bm_off, pair_off, aligned, bitmap, pairs, pairs2, item_start, item_key, item_idx, item_key2, item_order;
    DevBuf slots, arena, arena_used, cov, backup, level, err_aux, out_len, out_off, names, name_tab;
    /* result of the last run */
    uint32_t n_contigs = 0;
    uint64_t cov_total = 0;
    std::vector<uint32_t> appearance; /* sequence ids in order of first appearance */
    std::vector<int64_t> h_contig_len;
    std::vector<uint64_t> h_contig_cov;
    uint64_t n_out = 0; /* lines emit writes */
    uint32_t sides = 1; /* entries per record: 2 for to_bed -n */
    uint32_t name_salt = 0; /* salt of the name hash the last run's grouping passed its byte check with (0 unless names collided) */
};

static int ensure_keep(paffy_hip_ctx *c, DevBuf &b, size_t bytes, size_t used) {
    if (bytes <= b.cap) return 0;
    void *p = nullptr;
    size_t want = bytes + bytes / 2 + 4096;
    if (hipMalloc(&p, want) != hipSuccess) { /* no room for the slack: the exact size must still work */
        (void)hipGetLastError();
        p = nullptr;
        want = bytes;
        HIPCHK(c, hipMalloc(&p, want));
    }
    if (b.p && used) HIPCHK(c, hipMemcpyAsync(p, b.p, used, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = p;
    b.cap = want;
    return 0;
}

#define RPCHK(ctx, call)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);           \
            return PAFFY_E_HIP;                                                              \
        }                                                                                    \
    } while (0)

/* stable sort of (key, value) pairs by key, ascending; result in (kout, vout) */
static int cov_sort_pairs(paffy_hip_ctx *c, CovState &S, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, size_t n) {
    size_t bytes = 0;
    RPCHK(c, rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, 64, c->stream));
    if (ensure(c, S.tmp, bytes + 16)) return PAFFY_E_HIP;
    RPCHK(c, rocprim::radix_sort_pairs(S.tmp.p, bytes, kin, kout, vin, vout, n, 0, 64, c->stream));
    return 0;
}
static int cov_sort_keys(paffy_hip_ctx *c, CovState &S, const uint64_t *kin, uint64_t *kout, size_t n) {
    size_t bytes = 0;
    RPCHK(c, rocprim::radix_sort_keys(nullptr, bytes, kin, kout, n, 0, 64, c->stream));
    if (ensure(c, S.tmp, bytes + 16)) return PAFFY_E_HIP;
    RPCHK(c, rocprim::radix_sort_keys(S.tmp.p, bytes, kin, kout, n, 0, 64, c->stream));
    return 0;
}
static int cov_sort_pairs32(paffy_hip_ctx *c, CovState &S, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n) {
    size_t bytes = 0;
    RPCHK(c, rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, 32, c->stream));
    if (ensure(c, S.tmp, bytes + 16)) return PAFFY_E_HIP;
    RPCHK(c, rocprim::radix_sort_pairs(S.tmp.p, bytes, kin, kout, vin, vout, n, 0, 32, c->stream));
    return 0;
}
static int cov_incl_scan32(paffy_hip_ctx *c, CovState &S, const uint32_t *in, uint32_t *out, size_t n) {
    size_t bytes = 0;
    RPCHK(c, rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::plus<uint32_t>(), c->stream));
    if (ensure(c, S.tmp, bytes + 16)) return PAFFY_E_HIP;
    RPCHK(c, rocprim::inclusive_scan(S.tmp.p, bytes, in, out, n, rocprim::plus<uint32_t>(), c->stream));
    return 0;
}
/* exclusive scan of n values into out[0..n-1], the total in out[n] */
static int cov_excl_scan64(paffy_hip_ctx *c, CovState &S, uint64_t *in_with_zero_at_n, uint64_t *out, size_t n) {
    size_t bytes = 0;
    RPCHK(c, rocprim::exclusive_scan(nullptr, bytes, in_with_zero_at_n, out, (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), c->stream));
    if (ensure(c, S.tmp, bytes + 16)) return PAFFY_E_HIP;
    RPCHK(c, rocprim::exclusive_scan(S.tmp.p, bytes, in_with_zero_at_n, out, (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), c->stream));
    return 0;
}

__global__ __launch_bounds__(PAFFY_NT) void k_u32_to_u64(const uint32_t *in, uint32_t n, uint64_t *out) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n) out[i] = in[i];
    if (i == 0) out[n] = 0;
}
__global__ __launch_bounds__(PAFFY_NT) void k_iota32(uint32_t *out, uint32_t n) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n) out[i] = i;
}
/* name slice of every sequence (batch, offset, length of its first entry's name) */
struct CovName { uint32_t batch, off, len, pad; };
__global__ __launch_bounds__(PAFFY_NT) void k_cov_contig_names(CovParams P, const uint32_t *first_entry, uint32_t n_contigs, CovName *out) {
    const uint32_t c = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (c >= n_contigs) return;
    const CovEntry &E = P.entries[first_entry[c]];
    const RecMeta &m = P.meta[E.rec];
    CovName nm;
    nm.batch = m.pad1;
    nm.off = E.side ? m.tname_off : m.qname_off;
    nm.len = E.side ? m.tname_len : m.qname_len;
    nm.pad = first_entry[c];
    out[c] = nm;
}
__global__ __launch_bounds__(PAFFY_NT) void k_cov_copy_names(const uint8_t *const *batch_in, const CovName *nm, const uint32_t *blob_off, uint32_t n_contigs, uint8_t *blob) {
    const uint32_t c = blockIdx.x;
    if (c >= n_contigs) return;
    const uint8_t *src = batch_in[nm[c].batch] + nm[c].off;
    for (uint32_t i = threadIdx.x; i < nm[c].len; i += PAFFY_NT) blob[blob_off[c] + i] = src[i];
}

/* ---- the lines of `paffy tile` / dedupe / split_file: header + the cigar text as it was read (impl/paf.c:381-385) ---- */

__global__ __launch_bounds__(PAFFY_NT) void k_line_size(const RecMeta *meta, const uint32_t *order, const int64_t *level, uint64_t n, uint64_t *out_len) {
    const uint64_t k = (uint64_t)blockIdx.x * PAFFY_NT + threadIdx.x;
    if (k >= n) {
        if (k == n) out_len[n] = 0;
        return;
    }
    const RecMeta &m = meta[order[k]];
    RecState s;
    tile_state(m, level[order[k]], s);
    out_len[k] = (uint64_t)header_len(s, false) + (m.has_cg ? 6u + (uint64_t)m.cg_len : 0u) + 1u;
}
/* one WAVE per line (four lines per workgroup, no barrier): the header is built in the wave's LDS; the line leaves as 16-byte stores
   at 16-byte aligned output addresses, the cigar read with 16-byte loads at whatever alignment it has in the input */
#define LINE_HDR_BYTES (3 * PAFFY_TMPL_MAX + 16)
__global__ __launch_bounds__(PAFFY_NT) void k_line_emit(const uint8_t *const *batch_in, const RecMeta *meta, const uint32_t *order, const int64_t *level,
                                                         const uint64_t *out_off, uint64_t first_line, uint64_t n_lines, uint64_t base_off, uint8_t *out) {
    __shared__ uint8_t hdr_all[PAFFY_NWAVE][LINE_HDR_BYTES];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t li = (uint64_t)blockIdx.x * PAFFY_NWAVE + wave;
    if (li >= n_lines) return;
    uint8_t *hdr = hdr_all[wave];
    const uint64_t k = first_line + li;
    const RecMeta &m = meta[order[k]];
    const uint8_t *in = batch_in[m.pad1];
    uint8_t *o = out + (out_off[k] - base_off);
    RecState s;
    tile_state(m, level[order[k]], s);
    const uint32_t hl = header_len_wave(s, false) + (m.has_cg ? 6u : 0u), cl = m.has_cg ? m.cg_len : 0;
    const bool direct = hl > 3 * PAFFY_TMPL_MAX; /* header longer than the LDS staging: built in place */
    {
        Piece w{direct ? o : hdr, 0, direct ? hl : 3 * PAFFY_TMPL_MAX, false};
        build_header(w, s, in, false);
        if (m.has_cg) w.str("\tcg:Z:", 6);
    }
    __builtin_amdgcn_wave_barrier(); /* a wave's LDS operations execute in order */
    const uint32_t total = hl + cl + 1u;
    const uint8_t *cg = in + m.cg_off;
    auto byte_at = [&](uint32_t x) -> uint8_t { return x < hl ? hdr[x] : (x < hl + cl ? cg[x - hl] : (uint8_t)'\n'); };
    const uint32_t skip = direct ? hl : 0u; /* a header built in place is not copied again */
    /* three stretches: the header and the cigar's first bytes up to a 16-byte boundary of the output (byte by byte, the lanes side by
       side), whole 16-byte chunks of cigar text (aligned stores, loads at whatever alignment the input has, four in flight per lane),
       the last bytes and the newline (byte by byte) */
    const uint32_t head = (uint32_t)((16u - (((uintptr_t)o + hl) & 15u)) & 15u); /* cigar bytes in front of the first aligned chunk */
    const uint32_t x0 = hl + head < total ? hl + head : total;
    const uint32_t n_ch = x0 + 16u <= hl + cl ? (hl + cl - x0) >> 4 : 0u;
    const uint32_t x1 = x0 + 16u * n_ch;
    for (uint32_t x = skip + lane; x < x0; x += 64) o[x] = byte_at(x);
    const uint8_t *src = cg + (x0 - hl);
    uint8_t *dst = o + x0;
    uint32_t ch = lane;
    for (; ch + 192u < n_ch; ch += 256u) {
        const u32x4 v0 = *reinterpret_cast<const u32x4_unaligned *>(src + 16u * ch), v1 = *reinterpret_cast<const u32x4_unaligned *>(src + 16u * (ch + 64u));
        const u32x4 v2 = *reinterpret_cast<const u32x4_unaligned *>(src + 16u * (ch + 128u)), v3 = *reinterpret_cast<const u32x4_unaligned *>(src + 16u * (ch + 192u));
        *reinterpret_cast<u32x4 *>(dst + 16u * ch) = v0;
        *reinterpret_cast<u32x4 *>(dst + 16u * (ch + 64u)) = v1;
        *reinterpret_cast<u32x4 *>(dst + 16u * (ch + 128u)) = v2;
        *reinterpret_cast<u32x4 *>(dst + 16u * (ch + 192u)) = v3;
    }
    for (; ch < n_ch; ch += 64u) *reinterpret_cast<u32x4 *>(dst + 16u * ch) = *reinterpret_cast<const u32x4_unaligned *>(src + 16u * ch);
    for (uint32_t x = x1 + lane; x < total; x += 64)
        if (x >= skip) o[x] = byte_at(x);
}

/* ---------------------------------------------------------------------------------------------------------------------- */

static CovState &cov_state(paffy_hip_ctx *c);

static int cov_begin(paffy_hip_ctx *c, uint32_t sides) {
    CovState &S = cov_state(c);
    S.sides = sides;
    S.batches.clear();
    S.n_rec = 0;
    S.n_out = 0;
    if (ensure(c, S.info, sizeof(DevInfo))) return PAFFY_E_HIP;
    DevInfo zero;
    memset(&zero, 0, sizeof(zero));
    zero.first_err_key = ~0ull;
    HIPCHK(c, hipMemcpyAsync(S.info.p, &zero, sizeof(zero), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

/* index + header parse of one more batch of text; d_in must stay valid until the output has been emitted */
static int cov_add(paffy_hip_ctx *c, const void *d_in, int64_t in_len, bool with_sort_keys) {
    CovState &S = cov_state(c);
    if (in_len < 0 || in_len >= (1ll << 31) - 64 || (in_len > 0 && !d_in) || (reinterpret_cast<uintptr_t>(d_in) & 15)) return PAFFY_E_ARG;
    if (in_len == 0) return 0;
    uint32_t n = 0;
    int rc = index_and_parse(c, static_cast<const uint8_t *>(d_in), (uint32_t)in_len, &n);
    if (rc) return rc;
    if (n == 0) return 0;
    if (S.n_rec + n >= (1ull << 31)) {
        c->last_error = "more than 2^31 records";
        return PAFFY_E_UNSUPPORTED;
    }
    const uint64_t first = S.n_rec, total = first + n;
    if (ensure_keep(c, S.meta, sizeof(RecMeta) * total, sizeof(RecMeta) * first)) return PAFFY_E_HIP;
    if (ensure_keep(c, S.idx, sizeof(uint32_t) * total, sizeof(uint32_t) * first)) return PAFFY_E_HIP;
    if (with_sort_keys) {
        if (ensure_keep(c, S.key_score, sizeof(uint64_t) * total, sizeof(uint64_t) * first)) return PAFFY_E_HIP;
        if (ensure_keep(c, S.key_chain, sizeof(uint64_t) * total, sizeof(uint64_t) * first)) return PAFFY_E_HIP;
    }
    LAUNCH(c, "k_cov_collect", k_cov_collect, dim3((n + PAFFY_NT - 1) / PAFFY_NT), dim3(PAFFY_NT), 0, static_cast<const RecMeta *>(c->meta.p), n,
           (uint32_t)S.batches.size(), static_cast<RecMeta *>(S.meta.p) + first, with_sort_keys ? static_cast<uint64_t *>(S.key_score.p) + first : nullptr,
           with_sort_keys ? static_cast<uint64_t *>(S.key_chain.p) + first : nullptr, static_cast<uint32_t *>(S.idx.p) + first, (uint32_t)first, S.sides,
           static_cast<DevInfo *>(S.info.p));
    HIPCHK(c, hipStreamSynchronize(c->stream)); /* c->meta is reused by the next batch */
    S.batches.push_back(CovBatch{static_cast<const uint8_t *>(d_in), (uint32_t)in_len, n, first});
    S.n_rec = total;
    return 0;
}

static uint32_t cov_wave_bytes() { /* experiments: PAFFY_COV_WAVE_BYTES=0 sends every entry to the four-wave shape */
    static const uint32_t v = getenv("PAFFY_COV_WAVE_BYTES") ? (uint32_t)atol(getenv("PAFFY_COV_WAVE_BYTES")) : 9000u;
    return v;
}
static size_t cov_bitmap_budget_words() {
    const char *e = getenv("PAFFY_COV_BITMAP_MB");
    long mb = e ? atol(e) : 0;
    if (!e) { /* 32 GiB when the GPU has the room (a quarter of what is free at most): every chunk loads and saves the counters of the slices
                 it touches, so fewer, larger chunks move less (cfg5 at 2 M records per step: 3 chunks at 4 GiB, 110.2 ms; 1 chunk, 106.6 ms;
                 at 10 M records: 623 ms at 4 GiB, 548 at 16, 521 at 32, 520 at 64) */
        size_t free_b = 0, total_b = 0;
        mb = 4096;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const long quarter = (long)(free_b >> 22);
            mb = quarter > 32768 ? 32768 : (quarter < 256 ? 256 : quarter);
        } else {
            (void)hipGetLastError();
        }
    }
    if (mb < 1) mb = 1;
    return (size_t)mb << 18; /* 32-bit words */
}

static int cov_fetch(paffy_hip_ctx *c, void *dst, const void *src, size_t bytes) {
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

/*
 * mode 0: `paffy tile`: entries = records in visiting order, levels; mode 1: `paffy to_bed`: entries = records (and, with
 * include_inverted, their target sides) in input order, no levels. On success the counters of every sequence are in S.cov.
 * *err gets the first failing record (code 0: none).
 */
static int cov_run(paffy_hip_ctx *c, int mode, paffy_error *err) {
    CovState &S = cov_state(c);
    memset(err, 0, sizeof(*err));
    const uint32_t n = (uint32_t)S.n_rec;
    const uint32_t sides = S.sides;
    S.n_contigs = 0;
    S.cov_total = 0;
    S.appearance.clear();
    if (n == 0) return 0;
    if ((uint64_t)n * sides >= (1ull << 31)) return PAFFY_E_UNSUPPORTED;
    const uint32_t n_ent = n * sides;
    DevInfo hinfo;
    if (cov_fetch(c, &hinfo, S.info.p, sizeof(hinfo))) return PAFFY_E_HIP;
    auto report = [&](unsigned long long key) -> int {
        err->code = (int32_t)(key & 0xff);
        err->stage = (int32_t)((key >> 8) & 0xff) - 1;
        uint64_t rec = key >> 16;
        if (err->stage < 0) {
            rec /= sides; /* the key holds the rank of the record's first entry */
        } else {
            CovEntry E;
            if (cov_fetch(c, &E, static_cast<CovEntry *>(S.entries.p) + rec, sizeof(E))) return PAFFY_E_HIP;
            rec = E.rec;
        }
        err->record = (int64_t)rec;
        if (err->stage < 0) {
            RecMeta m;
            if (cov_fetch(c, &m, static_cast<RecMeta *>(S.meta.p) + rec, sizeof(m))) return PAFFY_E_HIP;
            err->aux = m.err_aux;
        } else {
            int32_t aux = 0;
            if (cov_fetch(c, &aux, static_cast<int32_t *>(S.err_aux.p) + rec, sizeof(aux))) return PAFFY_E_HIP;
            err->aux = aux;
        }
        return 0;
    };
    /* tile: read_pafs parses every line before anything else happens: the first bad line in input order wins */
    if (mode == 0 && hinfo.first_err_key != ~0ull) return report(hinfo.first_err_key);

    /* device table of the batch texts */
    {
        std::vector<const uint8_t *> ptrs;
        for (const CovBatch &b : S.batches) ptrs.push_back(b.in);
        if (ensure(c, S.batch_ptrs, sizeof(void *) * ptrs.size())) return PAFFY_E_HIP;
        HIPCHK(c, hipMemcpyAsync(S.batch_ptrs.p, ptrs.data(), sizeof(void *) * ptrs.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    const uint32_t g_rec = (n + PAFFY_NT - 1) / PAFFY_NT, g_ent = (n_ent + PAFFY_NT - 1) / PAFFY_NT;
    if (ensure(c, S.k64a, sizeof(uint64_t) * ((size_t)n_ent + 1))) return PAFFY_E_HIP;
    if (ensure(c, S.k64b, sizeof(uint64_t) * ((size_t)n_ent + 1))) return PAFFY_E_HIP;
    if (ensure(c, S.v32a, sizeof(uint32_t) * ((size_t)n_ent + 1))) return PAFFY_E_HIP;
    if (ensure(c, S.v32b, sizeof(uint32_t) * ((size_t)n_ent + 1))) return PAFFY_E_HIP;
    if (ensure(c, S.order, sizeof(uint32_t) * ((size_t)n + 1))) return PAFFY_E_HIP;
    uint64_t *k64a = static_cast<uint64_t *>(S.k64a.p), *k64b = static_cast<uint64_t *>(S.k64b.p);
    uint32_t *v32a = static_cast<uint32_t *>(S.v32a.p), *v32b = static_cast<uint32_t *>(S.v32b.p);
    uint32_t *order = static_cast<uint32_t *>(S.order.p);
    if (mode == 0) {
        /* visiting order: paf_cmp_by_descending_score, impl/paf_tile.c:28-34 -- chain_score desc, score desc, ties in input order
           (glibc's qsort is a stable merge sort for these sizes): least significant key first, both sorts stable */
        if (cov_sort_pairs(c, S, static_cast<const uint64_t *>(S.key_score.p), k64a, static_cast<const uint32_t *>(S.idx.p), v32a, n)) return PAFFY_E_HIP;
        LAUNCH(c, "k_gather_u64", k_gather_u64, dim3(g_rec), dim3(PAFFY_NT), 0, static_cast<const uint64_t *>(S.key_chain.p), v32a, n, k64b);
        if (cov_sort_pairs(c, S, k64b, k64a, v32a, order, n)) return PAFFY_E_HIP;
    }
    /* entries */
    if (ensure(c, S.entries, sizeof(CovEntry) * (size_t)n_ent)) return PAFFY_E_HIP;
    if (ensure(c, S.name_hash, sizeof(uint64_t) * (size_t)n_ent)) return PAFFY_E_HIP;
    if (ensure(c, S.seq_len, sizeof(int64_t) * (size_t)n_ent)) return PAFFY_E_HIP;
    if (ensure(c, S.aligned, sizeof(int64_t) * (size_t)n_ent)) return PAFFY_E_HIP;
    if (ensure(c, S.level, sizeof(int64_t) * (size_t)n)) return PAFFY_E_HIP;
    if (ensure(c, S.err_aux, sizeof(int32_t) * (size_t)n)) return PAFFY_E_HIP;
    HIPCHK(c, hipMemsetAsync(S.level.p, 0xff, sizeof(int64_t) * (size_t)n, c->stream));
    HIPCHK(c, hipMemsetAsync(S.aligned.p, 0, sizeof(int64_t) * (size_t)n_ent, c->stream));
    CovParams P;
    memset(&P, 0, sizeof(P));
    P.batch_in = static_cast<const uint8_t *const *>(S.batch_ptrs.p);
    P.meta = static_cast<const RecMeta *>(S.meta.p);
    P.entries = static_cast<CovEntry *>(S.entries.p);
    P.n_entries = n_ent;
    P.aligned = static_cast<int64_t *>(S.aligned.p);
    P.level = static_cast<int64_t *>(S.level.p);
    P.info = static_cast<DevInfo *>(S.info.p);
    P.err_aux = static_cast<int32_t *>(S.err_aux.p);
    if (ensure(c, S.flags, sizeof(uint32_t) * ((size_t)n_ent + 2))) return PAFFY_E_HIP;
    if (ensure(c, S.scan32, sizeof(uint32_t) * ((size_t)n_ent + 1))) return PAFFY_E_HIP;
    uint32_t *flags = static_cast<uint32_t *>(S.flags.p), *scan32 = static_cast<uint32_t *>(S.scan32.p);
    uint32_t n_contigs = 0;
    /* sequences: entries grouped by name hash, the names then checked against their group's first (impl/paf.c:675-688 looks the name up
       by string equality): two names under one hash -> the grouping again with the next salt */
    for (uint32_t salt = 0;; salt++) {
        LAUNCH(c, "k_cov_entry_init", k_cov_entry_init, dim3(g_ent), dim3(PAFFY_NT), 0, P, mode == 0 ? order : nullptr, sides, static_cast<uint64_t *>(S.name_hash.p),
               static_cast<int64_t *>(S.seq_len.p), salt);
        LAUNCH(c, "k_iota32", k_iota32, dim3(g_ent), dim3(PAFFY_NT), 0, v32a, n_ent);
        if (cov_sort_pairs(c, S, static_cast<const uint64_t *>(S.name_hash.p), k64a, v32a, v32b, n_ent)) return PAFFY_E_HIP;
        LAUNCH(c, "k_cov_run_heads", k_cov_run_heads, dim3(g_ent), dim3(PAFFY_NT), 0, k64a, n_ent, flags);
        if (cov_incl_scan32(c, S, flags, scan32, n_ent)) return PAFFY_E_HIP;
        if (cov_fetch(c, &n_contigs, scan32 + (n_ent - 1), sizeof(uint32_t))) return PAFFY_E_HIP;
        if (ensure(c, S.first_entry, sizeof(uint32_t) * (size_t)n_contigs)) return PAFFY_E_HIP;
        LAUNCH(c, "k_cov_assign_contig", k_cov_assign_contig, dim3(g_ent), dim3(PAFFY_NT), 0, P, v32b, static_cast<const uint32_t *>(flags), static_cast<const uint32_t *>(scan32), n_ent,
               static_cast<uint32_t *>(S.first_entry.p));
        uint32_t *collide = flags + n_ent + 1, hit = 0;
        HIPCHK(c, hipMemsetAsync(collide, 0, sizeof(uint32_t), c->stream));
        LAUNCH(c, "k_cov_verify_names", k_cov_verify_names, dim3(g_ent), dim3(PAFFY_NT), 0, P, static_cast<const uint32_t *>(S.first_entry.p), collide);
        if (cov_fetch(c, &hit, collide, sizeof(uint32_t))) return PAFFY_E_HIP;
        S.name_salt = salt;
        if (!hit) break;
        if (salt == 7) {
            c->last_error = "sequence names keep colliding under eight differently salted 64-bit hashes";
            return PAFFY_E_UNSUPPORTED;
        }
    }
    if (ensure(c, S.contig_len, sizeof(int64_t) * (size_t)n_contigs)) return PAFFY_E_HIP;
    if (ensure(c, S.contig_cov, sizeof(uint64_t) * ((size_t)n_contigs + 1))) return PAFFY_E_HIP;
    if (ensure(c, S.contig_slice0, sizeof(uint32_t) * ((size_t)n_contigs + 1))) return PAFFY_E_HIP;
    if (ensure(c, S.name_tab, sizeof(CovName) * (size_t)n_contigs)) return PAFFY_E_HIP;
    const uint32_t g_c = (n_contigs + PAFFY_NT - 1) / PAFFY_NT;
    LAUNCH(c, "k_cov_contig_len", k_cov_contig_len, dim3(g_c), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(S.first_entry.p),
           static_cast<const int64_t *>(S.seq_len.p), n_contigs, static_cast<int64_t *>(S.contig_len.p));
    LAUNCH(c, "k_cov_contig_names", k_cov_contig_names, dim3(g_c), dim3(PAFFY_NT), 0, P, static_cast<const uint32_t *>(S.first_entry.p), n_contigs,
           static_cast<CovName *>(S.name_tab.p));
    std::vector<int64_t> clen(n_contigs);
    std::vector<CovName> cname(n_contigs);
    HIPCHK(c, hipMemcpyAsync(clen.data(), S.contig_len.p, sizeof(int64_t) * n_contigs, hipMemcpyDeviceToHost, c->stream));
    if (cov_fetch(c, cname.data(), S.name_tab.p, sizeof(CovName) * n_contigs)) return PAFFY_E_HIP;
    /* counters: sequences in order of first appearance, each at a multiple of 8 counters with at least 8 counters of padding behind it */
    std::vector<uint32_t> appearance(n_contigs);
    for (uint32_t i = 0; i < n_contigs; i++) appearance[i] = i;
    std::sort(appearance.begin(), appearance.end(), [&](uint32_t a, uint32_t b) { return cname[a].pad < cname[b].pad; });
    std::vector<uint64_t> ccov(n_contigs + 1);
    std::vector<uint32_t> cslice(n_contigs + 1);
    uint64_t cov_total = 0, slices = 0;
    for (uint32_t k = 0; k < n_contigs; k++) {
        const uint32_t ci = appearance[k];
        const int64_t L = clen[ci] > 0 ? clen[ci] : 0;
        if (L > (1ll << 40)) {
            c->last_error = "a sequence longer than 2^40 bases";
            return PAFFY_E_UNSUPPORTED;
        }
        ccov[ci] = cov_total;
        cslice[ci] = (uint32_t)slices;
        cov_total += ((uint64_t)L + 8u + 7u) & ~7ull;
        slices += ((uint64_t)L + COV_SLICE - 1) >> COV_SLICE_SHIFT;
        if (slices >= (1ull << 32) - 1 || cov_total >= (1ull << 38)) {
            c->last_error = "sequences too long for the coverage counters of this build";
            return PAFFY_E_UNSUPPORTED;
        }
    }
    ccov[n_contigs] = cov_total;
    cslice[n_contigs] = (uint32_t)slices;
    HIPCHK(c, hipMemcpyAsync(S.contig_cov.p, ccov.data(), sizeof(uint64_t) * (n_contigs + 1), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(S.contig_slice0.p, cslice.data(), sizeof(uint32_t) * (n_contigs + 1), hipMemcpyHostToDevice, c->stream));
    if (ensure(c, S.cov, sizeof(uint16_t) * (size_t)(cov_total + 64))) return PAFFY_E_HIP;
    HIPCHK(c, hipMemsetAsync(S.cov.p, 0, sizeof(uint16_t) * (size_t)cov_total, c->stream));
    P.contig_len = static_cast<const int64_t *>(S.contig_len.p);
    P.contig_cov = static_cast<const uint64_t *>(S.contig_cov.p);
    P.contig_slice0 = static_cast<const uint32_t *>(S.contig_slice0.p);
    P.cov = static_cast<uint16_t *>(S.cov.p);
    /* sizes and offsets of every entry's bitmap and pairs */
    if (ensure(c, S.bm_words, sizeof(uint64_t) * ((size_t)n_ent + 1))) return PAFFY_E_HIP;
    if (ensure(c, S.n_pairs, sizeof(uint32_t) * ((size_t)n_ent + 1))) return PAFFY_E_HIP;
    if (ensure(c, S.bm_off, sizeof(uint64_t) * ((size_t)n_ent + 1))) return PAFFY_E_HIP;
    if (ensure(c, S.pair_off, sizeof(uint64_t) * ((size_t)n_ent + 1))) return PAFFY_E_HIP;
    uint64_t *bm_words = static_cast<uint64_t *>(S.bm_words.p), *bm_off = static_cast<uint64_t *>(S.bm_off.p), *pair_off = static_cast<uint64_t *>(S.pair_off.p);
    LAUNCH(c, "k_cov_entry_sizes", k_cov_entry_sizes, dim3(g_ent), dim3(PAFFY_NT), 0, P, static_cast<const int64_t *>(S.seq_len.p), bm_words,
           static_cast<uint32_t *>(S.n_pairs.p));
    HIPCHK(c, hipMemsetAsync(bm_words + n_ent, 0, sizeof(uint64_t), c->stream));
    if (cov_excl_scan64(c, S, bm_words, bm_off, n_ent)) return PAFFY_E_HIP;
    LAUNCH(c, "k_u32_to_u64", k_u32_to_u64, dim3(g_ent), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(S.n_pairs.p), n_ent, k64b);
    if (cov_excl_scan64(c, S, k64b, pair_off, n_ent)) return PAFFY_E_HIP;
    LAUNCH(c, "k_cov_entry_offsets", k_cov_entry_offsets, dim3(g_ent), dim3(PAFFY_NT), 0, P, static_cast<const uint64_t *>(bm_off), static_cast<const uint64_t *>(pair_off));
    uint64_t tot_words = 0, tot_pairs = 0;
    HIPCHK(c, hipMemcpyAsync(&tot_words, bm_off + n_ent, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    if (cov_fetch(c, &tot_pairs, pair_off + n_ent, sizeof(uint64_t))) return PAFFY_E_HIP;
    if (tot_pairs >= (1ull << 32) - 2) {
        c->last_error = "too many (record, slice) pairs";
        return PAFFY_E_UNSUPPORTED;
    }
    if (mode == 0) {
        if (ensure(c, S.slots, sizeof(CovSlot) * (size_t)(tot_pairs + 1))) return PAFFY_E_HIP;
        HIPCHK(c, hipMemsetAsync(S.slots.p, 0, sizeof(CovSlot) * (size_t)tot_pairs, c->stream));
        if (ensure(c, S.arena_used, sizeof(unsigned long long))) return PAFFY_E_HIP;
        P.slots = static_cast<CovSlot *>(S.slots.p);
        P.arena_used = static_cast<unsigned long long *>(S.arena_used.p);
    }
    /* chunks of entries whose bitmaps fit the budget */
    const size_t budget = cov_bitmap_budget_words();
    uint64_t arena_base = 0; /* arena entries used by the chunks before this one (their slots point there) */
    uint32_t e0 = 0;
    uint64_t w_e0 = 0, p_e0 = 0;
    while (e0 < n_ent) {
        uint32_t e1 = n_ent;
        uint64_t w_e1 = tot_words, p_e1 = tot_pairs;
        if (tot_words - w_e0 > budget) { /* largest e1 with bm_off[e1] - bm_off[e0] <= budget (at least one entry) */
            uint32_t lo = e0 + 1, hi = n_ent;
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo + 1) / 2;
                uint64_t w = 0;
                if (cov_fetch(c, &w, bm_off + mid, sizeof(w))) return PAFFY_E_HIP;
                if (w - w_e0 <= budget) lo = mid;
                else hi = mid - 1;
            }
            e1 = lo;
            if (cov_fetch(c, &w_e1, bm_off + e1, sizeof(uint64_t))) return PAFFY_E_HIP;
            if (cov_fetch(c, &p_e1, pair_off + e1, sizeof(uint64_t))) return PAFFY_E_HIP;
        }
        const uint64_t n_w = w_e1 - w_e0, n_p = p_e1 - p_e0;
        if (ensure(c, S.bitmap, sizeof(uint32_t) * (size_t)(n_w + 64))) return PAFFY_E_HIP;
        HIPCHK(c, hipMemsetAsync(S.bitmap.p, 0, sizeof(uint32_t) * (size_t)(n_w + 64), c->stream));
        P.e0 = e0;
        P.e1 = e1;
        P.bitmap = static_cast<uint32_t *>(S.bitmap.p);
        P.bm_base = w_e0;
        /* short cigars one wave per entry, the rest four (COV_WAVE_BYTES: the longest cigar of the one-wave shape) */
        LAUNCH(c, "k_cov_bitmap_wave", k_cov_bitmap<CovGroup64>, dim3(e1 - e0), dim3(64), COV_BM_LDS_BYTES_OF(CovGroup64), P, mode == 0 ? 1 : 0, 0u, cov_wave_bytes());
        LAUNCH(c, "k_cov_bitmap", k_cov_bitmap<CovGroup256>, dim3(e1 - e0), dim3(256), COV_BM_LDS_BYTES_OF(CovGroup256), P, mode == 0 ? 1 : 0, cov_wave_bytes() + 1u, 0xffffffffu);
        if (n_p > 0) {
            const uint32_t np = (uint32_t)n_p, g_p = (np + PAFFY_NT - 1) / PAFFY_NT;
            if (ensure(c, S.pairs, sizeof(uint64_t) * (size_t)(n_p + 1))) return PAFFY_E_HIP;
            if (ensure(c, S.pairs2, sizeof(uint64_t) * (size_t)(n_p + 1))) return PAFFY_E_HIP;
            uint64_t *pk = static_cast<uint64_t *>(S.pairs.p), *pk2 = static_cast<uint64_t *>(S.pairs2.p);
            LAUNCH(c, "k_cov_pairs", k_cov_pairs, dim3((e1 - e0 + PAFFY_NT - 1) / PAFFY_NT), dim3(PAFFY_NT), 0, P, (uint32_t)p_e0, pk);
            if (cov_sort_keys(c, S, pk, pk2, np)) return PAFFY_E_HIP;
            if (ensure(c, S.flags, sizeof(uint32_t) * ((size_t)np + 1))) return PAFFY_E_HIP;
            if (ensure(c, S.scan32, sizeof(uint32_t) * ((size_t)np + 1))) return PAFFY_E_HIP;
            flags = static_cast<uint32_t *>(S.flags.p);
            scan32 = static_cast<uint32_t *>(S.scan32.p);
            LAUNCH(c, "k_cov_item_heads", k_cov_item_heads, dim3(g_p), dim3(PAFFY_NT), 0, static_cast<const uint64_t *>(pk2), np, flags);
            if (cov_incl_scan32(c, S, flags, scan32, np)) return PAFFY_E_HIP;
            uint32_t n_items = 0;
            if (cov_fetch(c, &n_items, scan32 + (np - 1), sizeof(uint32_t))) return PAFFY_E_HIP;
            if (ensure(c, S.item_start, sizeof(uint32_t) * ((size_t)n_items + 1))) return PAFFY_E_HIP;
            LAUNCH(c, "k_cov_item_starts", k_cov_item_starts, dim3(g_p), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(flags), static_cast<const uint32_t *>(scan32), np,
                   static_cast<uint32_t *>(S.item_start.p));
            /* heavy slices first */
            if (ensure(c, S.item_key, sizeof(uint32_t) * (size_t)n_items)) return PAFFY_E_HIP;
            if (ensure(c, S.item_idx, sizeof(uint32_t) * (size_t)n_items)) return PAFFY_E_HIP;
            if (ensure(c, S.item_key2, sizeof(uint32_t) * (size_t)n_items)) return PAFFY_E_HIP;
            if (ensure(c, S.item_order, sizeof(uint32_t) * (size_t)n_items)) return PAFFY_E_HIP;
            LAUNCH(c, "k_cov_item_sizes", k_cov_item_sizes, dim3((n_items + PAFFY_NT - 1) / PAFFY_NT), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(S.item_start.p), n_items,
                   static_cast<uint32_t *>(S.item_key.p), static_cast<uint32_t *>(S.item_idx.p));
            if (cov_sort_pairs32(c, S, static_cast<const uint32_t *>(S.item_key.p), static_cast<uint32_t *>(S.item_key2.p), static_cast<const uint32_t *>(S.item_idx.p),
                                 static_cast<uint32_t *>(S.item_order.p), n_items))
                return PAFFY_E_HIP;
            P.pairs = pk2;
            P.item_start = static_cast<const uint32_t *>(S.item_start.p);
            P.item_order = static_cast<const uint32_t *>(S.item_order.p);
            P.n_items = n_items;
            if (mode == 1) {
                LAUNCH(c, "k_cov_walk", k_cov_walk<false>, dim3(n_items), dim3(COV_NT), sizeof(CovWalkLds), P, static_cast<uint16_t *>(nullptr), 0);
            } else {
                /* the partial histograms of the chunk go to an arena; when it proves too small the slices' counters are put back
                   (every workgroup saved what it loaded) and the walk runs again with the size it asked for */
                if (ensure(c, S.backup, sizeof(uint16_t) * (size_t)n_items * COV_SLICE)) return PAFFY_E_HIP;
                uint64_t cap = arena_base + (n_p * 24 > (1ull << 20) ? n_p * 24 : (1ull << 20));
                for (int attempt = 0; attempt < 2; attempt++) {
                    if (ensure_keep(c, S.arena, sizeof(uint16_t) * (size_t)(cap + 64), sizeof(uint16_t) * (size_t)arena_base)) return PAFFY_E_HIP;
                    unsigned long long used0 = arena_base;
                    HIPCHK(c, hipMemcpyAsync(S.arena_used.p, &used0, sizeof(used0), hipMemcpyHostToDevice, c->stream));
                    P.arena = static_cast<uint16_t *>(S.arena.p);
                    P.arena_cap = cap;
                    LAUNCH(c, "k_cov_walk", k_cov_walk<true>, dim3(n_items), dim3(COV_NT), sizeof(CovWalkLds), P, static_cast<uint16_t *>(S.backup.p), attempt);
                    unsigned long long used = 0;
                    if (cov_fetch(c, &used, S.arena_used.p, sizeof(used))) return PAFFY_E_HIP;
                    if (used <= cap) {
                        arena_base = used;
                        break;
                    }
                    if (attempt == 1) {
                        c->last_error = "coverage arena demand changed between two passes";
                        return PAFFY_E_HIP;
                    }
                    cap = used;
                }
            }
        } else {
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        e0 = e1;
        w_e0 = w_e1;
        p_e0 = p_e1;
    }
    if (mode == 0) LAUNCH(c, "k_cov_merge", k_cov_merge, dim3((n_ent + PAFFY_NWAVE - 1) / PAFFY_NWAVE), dim3(PAFFY_NT), 0, P);
    if (cov_fetch(c, &hinfo, S.info.p, sizeof(hinfo))) return PAFFY_E_HIP;
    if (c->profile) prof_collect(c);
    if (hinfo.first_err_key != ~0ull) return report(hinfo.first_err_key);
    S.n_contigs = n_contigs;
    S.cov_total = cov_total;
    S.appearance = appearance;
    S.h_contig_len = clen;
    S.h_contig_cov.assign(ccov.begin(), ccov.end());
    /* names of the sequences as one blob (to_bed prints them) */
    if (mode == 1) {
        std::vector<uint32_t> boff(n_contigs + 1);
        uint32_t at = 0;
        for (uint32_t ci = 0; ci < n_contigs; ci++) {
            boff[ci] = at;
            at += cname[ci].len;
        }
        boff[n_contigs] = at;
        if (ensure(c, S.names, (size_t)at + 16 + sizeof(uint32_t) * (n_contigs + 1))) return PAFFY_E_HIP;
        uint32_t *d_boff = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(S.names.p) + (((size_t)at + 15) & ~(size_t)15));
        HIPCHK(c, hipMemcpyAsync(d_boff, boff.data(), sizeof(uint32_t) * (n_contigs + 1), hipMemcpyHostToDevice, c->stream));
        LAUNCH(c, "k_cov_copy_names", k_cov_copy_names, dim3(n_contigs), dim3(PAFFY_NT), 0, P.batch_in, static_cast<const CovName *>(S.name_tab.p),
               static_cast<const uint32_t *>(d_boff), n_contigs, static_cast<uint8_t *>(S.names.p));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

#endif
