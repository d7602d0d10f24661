/*
 * paffy dedupe [-a] on the device (impl/paf_dedupe.c:117-143). The reference walks the records in input order with a hash table of
 * the records it has written: a record is written unless a written record has the same key (query name, target name, strand, four
 * coordinates) -- or, with -a, the key of the record with query and target swapped; with -a every record whose own key was not
 * found also goes through paf_check, which ends the run.
 *
 * "First seen wins" needs no loop: of all records that share a key -- with -a: a key or the swapped key, which is one class, because
 * the swapped record of the swapped record is the record -- exactly the first in input order is written, unless an earlier batch of
 * the same context already wrote one of the class. So per batch:
 *   1. the class key of every record (its own 128-bit key; with -a the smaller of its key and the swapped key), and whether the
 *      context's memory -- the sorted keys of everything written by earlier batches -- holds its key / its swapped key (binary searches);
 *   2. a stable sort of the records by class key (two 64-bit radix passes): the head of every run is the class's first record;
 *   3. per record: written = head of its run and nothing of the class written before; "own key found" (which spares it paf_check) =
 *      found in the memory, or the run's head was written and has the record's orientation; the smallest record index that fails
 *      (parse error, or paf_check where it applies) ends the run as the reference's loop would;
 *   4. the written records in input order (scan of the flags), their keys merged into the memory (sorted again).
 * Included by paffy_hip.hip behind coverage_host.h (RPCHK, rocPRIM).
 */
#pragma once

struct DedupeState {
    DevBuf seen_hi, seen_lo, seen_hi2, seen_lo2; /* keys of the records written so far, sorted by (hi, lo) */
    size_t seen_n = 0;
    DevBuf chi, clo, idx, idx2, k64, flags, keep, pos, first, tmp;
};

static void dedupe_free(paffy_hip_ctx *c) {
    if (!c->dedupe) return;
    DedupeState &D = *c->dedupe;
    DevBuf *db[] = {&D.seen_hi, &D.seen_lo, &D.seen_hi2, &D.seen_lo2, &D.chi, &D.clo, &D.idx, &D.idx2, &D.k64, &D.flags, &D.keep, &D.pos, &D.first, &D.tmp};
    for (DevBuf *b : db)
        if (b->p) (void)hipFree(b->p);
    delete c->dedupe;
    c->dedupe = nullptr;
}

__device__ __forceinline__ bool dd_less(uint64_t ah, uint64_t al, uint64_t bh, uint64_t bl) { return ah < bh || (ah == bh && al < bl); }
__device__ __forceinline__ bool dd_seen(const uint64_t *hi, const uint64_t *lo, uint32_t n, uint64_t kh, uint64_t kl) {
    uint32_t b = 0, e = n; /* first element not less than the key */
    while (b < e) {
        const uint32_t m = b + ((e - b) >> 1);
        if (dd_less(hi[m], lo[m], kh, kl)) b = m + 1;
        else e = m;
    }
    return b < n && hi[b] == kh && lo[b] == kl;
}
/* flags: bit 0 the class was written by an earlier batch, bit 1 the record's own key was, bit 2 the record's own key is its class key */
__global__ __launch_bounds__(PAFFY_NT) void k_dedupe_class(const DedupeKey *keys, uint32_t n, int inverse, const uint64_t *seen_hi, const uint64_t *seen_lo,
                                                         uint32_t seen_n, uint64_t *chi, uint64_t *clo, uint32_t *idx, uint32_t *flags) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i >= n) return;
    const DedupeKey k = keys[i];
    const bool own_is_class = !inverse || !dd_less(k.ia, k.ib, k.a, k.b);
    chi[i] = own_is_class ? k.a : k.ia;
    clo[i] = own_is_class ? k.b : k.ib;
    idx[i] = i;
    const bool own = dd_seen(seen_hi, seen_lo, seen_n, k.a, k.b);
    const bool swapped = inverse && !own && dd_seen(seen_hi, seen_lo, seen_n, k.ia, k.ib);
    flags[i] = ((own || swapped) ? 1u : 0u) | (own ? 2u : 0u) | (own_is_class ? 4u : 0u);
}
__global__ __launch_bounds__(PAFFY_NT) void k_gather_u64_by(const uint64_t *src, const uint32_t *order, uint32_t n, uint64_t *dst) {
    const uint32_t p = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (p < n) dst[p] = src[order[p]];
}
/* position p of the sorted order: head[p] = p when the record opens a run of equal class keys, else 0 (a running maximum gives every
   position its run's head) */
__global__ __launch_bounds__(PAFFY_NT) void k_dedupe_heads(const uint64_t *chi, const uint64_t *clo, const uint32_t *order, uint32_t n, uint32_t *head) {
    const uint32_t p = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (p >= n) return;
    bool h = p == 0;
    if (!h) {
        const uint32_t i = order[p], j = order[p - 1];
        h = chi[i] != chi[j] || clo[i] != clo[j];
    }
    head[p] = h ? p : 0u;
}
__global__ __launch_bounds__(PAFFY_NT) void k_dedupe_decide(const DedupeKey *keys, const uint32_t *order, const uint32_t *head_of, const uint32_t *flags, uint32_t n,
                                                          int mode, uint32_t *keep, uint32_t *first_bad) {
    const uint32_t p = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (p >= n) return;
    const uint32_t i = order[p], h = order[head_of[p]];
    const uint32_t fi = flags[i], fh = flags[h];
    const bool head_written = !(fh & 1u); /* nothing of the class written by an earlier batch: its first record of this batch is */
    const bool written = mode == PAFFY_DEDUPE_KEEP_ALL || (i == h && head_written);
    keep[i] = written ? 1u : 0u;
    const DedupeKey k = keys[i];
    bool bad = k.err != 0;
    if (!bad && mode == 1 && k.check != 0) { /* paf_check for every record whose own key was not found, impl/paf_dedupe.c:121-126 */
        const bool own_found = (fi & 2u) || (i != h && head_written && ((fi ^ fh) & 4u) == 0);
        bad = !own_found;
    }
    if (bad) atomicMin(first_bad, i);
}
/* the written records below the first failing one, in input order; their keys behind the memory's */
__global__ __launch_bounds__(PAFFY_NT) void k_dedupe_collect(const DedupeKey *keys, const uint32_t *keep, const uint32_t *pos, uint32_t n, uint32_t first_bad,
                                                           int remember, uint32_t *kept, uint64_t *seen_hi, uint64_t *seen_lo, uint32_t seen_n) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i >= n || i >= first_bad || !keep[i]) return;
    kept[pos[i]] = i;
    if (remember) {
        seen_hi[seen_n + pos[i]] = keys[i].a;
        seen_lo[seen_n + pos[i]] = keys[i].b;
    }
}
__global__ __launch_bounds__(PAFFY_NT) void k_dedupe_mask_keep(uint32_t *keep, uint32_t n, const uint32_t *first_bad) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i < n && i >= *first_bad) keep[i] = 0;
}

__global__ __launch_bounds__(PAFFY_NT) void k_dedupe_keep_all(const DedupeKey *keys, uint32_t n, uint32_t *keep, uint32_t *first_bad) {
    const uint32_t i = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (i >= n) return;
    keep[i] = 1u;
    if (keys[i].err != 0) atomicMin(first_bad, i);
}

struct DdMax {
    __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

static int dd_sort_pairs(paffy_hip_ctx *c, DedupeState &D, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, size_t n) {
    size_t bytes = 0;
    RPCHK(c, rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, 64, c->stream));
    if (ensure(c, D.tmp, bytes + 16)) return PAFFY_E_HIP;
    RPCHK(c, rocprim::radix_sort_pairs(D.tmp.p, bytes, kin, kout, vin, vout, n, 0, 64, c->stream));
    return 0;
}
static int dd_sort_pairs64(paffy_hip_ctx *c, DedupeState &D, const uint64_t *kin, uint64_t *kout, const uint64_t *vin, uint64_t *vout, size_t n) {
    size_t bytes = 0;
    RPCHK(c, rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, 64, c->stream));
    if (ensure(c, D.tmp, bytes + 16)) return PAFFY_E_HIP;
    RPCHK(c, rocprim::radix_sort_pairs(D.tmp.p, bytes, kin, kout, vin, vout, n, 0, 64, c->stream));
    return 0;
}

/*
 * keys[n] (k_dedupe_keys) -> kept[0 .. *n_kept): the records written, in input order; *first_bad: index of the record that ends the
 * run (n: none). mode: 0 plain, 1 -a, PAFFY_DEDUPE_KEEP_ALL (split_file: every record up to the first parse error; nothing remembered).
 */
static int dedupe_select(paffy_hip_ctx *c, DedupeState &D, const DedupeKey *keys, uint32_t n, int mode, uint32_t *kept, uint32_t *n_kept, uint32_t *first_bad) {
    const uint32_t grid = (n + PAFFY_NT - 1) / PAFFY_NT;
    const bool remember = mode != PAFFY_DEDUPE_KEEP_ALL;
    if (ensure(c, D.chi, sizeof(uint64_t) * (size_t)n) || ensure(c, D.clo, sizeof(uint64_t) * (size_t)n) || ensure(c, D.k64, sizeof(uint64_t) * 2 * (size_t)n) ||
        ensure(c, D.idx, sizeof(uint32_t) * (size_t)n) || ensure(c, D.idx2, sizeof(uint32_t) * 2 * (size_t)n) || ensure(c, D.flags, sizeof(uint32_t) * (size_t)n) ||
        ensure(c, D.keep, sizeof(uint32_t) * (size_t)n) || ensure(c, D.pos, sizeof(uint32_t) * ((size_t)n + 1)) || ensure(c, D.first, 2 * sizeof(uint32_t)))
        return PAFFY_E_HIP;
    if (remember) { /* room for this batch's keys behind the memory's; the copies the sort passes alternate with need no content */
        const size_t want = sizeof(uint64_t) * (D.seen_n + (size_t)n), used = sizeof(uint64_t) * D.seen_n;
        if (ensure_keep(c, D.seen_hi, want, used) || ensure_keep(c, D.seen_lo, want, used) || ensure(c, D.seen_hi2, want) || ensure(c, D.seen_lo2, want)) return PAFFY_E_HIP;
    }
    uint64_t *chi = static_cast<uint64_t *>(D.chi.p), *clo = static_cast<uint64_t *>(D.clo.p), *k64a = static_cast<uint64_t *>(D.k64.p), *k64b = k64a + n;
    uint32_t *idx = static_cast<uint32_t *>(D.idx.p), *v_a = static_cast<uint32_t *>(D.idx2.p), *order = v_a + n;
    uint32_t *flags = static_cast<uint32_t *>(D.flags.p), *keep = static_cast<uint32_t *>(D.keep.p), *pos = static_cast<uint32_t *>(D.pos.p);
    uint32_t *first = static_cast<uint32_t *>(D.first.p);
    const uint32_t none = n;
    HIPCHK(c, hipMemcpyAsync(first, &none, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    if (!remember) {
        LAUNCH(c, "k_dedupe_keep_all", k_dedupe_keep_all, dim3(grid), dim3(PAFFY_NT), 0, keys, n, keep, first);
    } else {
        LAUNCH(c, "k_dedupe_class", k_dedupe_class, dim3(grid), dim3(PAFFY_NT), 0, keys, n, mode == 1 ? 1 : 0, static_cast<const uint64_t *>(D.seen_hi.p),
               static_cast<const uint64_t *>(D.seen_lo.p), (uint32_t)D.seen_n, chi, clo, idx, flags);
        /* stable sort by (hi, lo): by lo, then by hi */
        if (dd_sort_pairs(c, D, clo, k64a, idx, v_a, n)) return PAFFY_E_HIP;
        LAUNCH(c, "k_gather_u64_by", k_gather_u64_by, dim3(grid), dim3(PAFFY_NT), 0, chi, v_a, n, k64a);
        if (dd_sort_pairs(c, D, k64a, k64b, v_a, order, n)) return PAFFY_E_HIP;
        uint32_t *head = idx; /* the identity is not needed any more */
        LAUNCH(c, "k_dedupe_heads", k_dedupe_heads, dim3(grid), dim3(PAFFY_NT), 0, chi, clo, order, n, head);
        {
            size_t bytes = 0;
            RPCHK(c, rocprim::inclusive_scan(nullptr, bytes, head, v_a, n, DdMax(), c->stream));
            if (ensure(c, D.tmp, bytes + 16)) return PAFFY_E_HIP;
            RPCHK(c, rocprim::inclusive_scan(D.tmp.p, bytes, head, v_a, n, DdMax(), c->stream));
        }
        LAUNCH(c, "k_dedupe_decide", k_dedupe_decide, dim3(grid), dim3(PAFFY_NT), 0, keys, order, v_a, flags, n, mode, keep, first);
    }
    LAUNCH(c, "k_dedupe_mask_keep", k_dedupe_mask_keep, dim3(grid), dim3(PAFFY_NT), 0, keep, n, first);
    {
        size_t bytes = 0;
        RPCHK(c, rocprim::exclusive_scan(nullptr, bytes, keep, pos, 0u, n, rocprim::plus<uint32_t>(), c->stream));
        if (ensure(c, D.tmp, bytes + 16)) return PAFFY_E_HIP;
        RPCHK(c, rocprim::exclusive_scan(D.tmp.p, bytes, keep, pos, 0u, n, rocprim::plus<uint32_t>(), c->stream));
    }
    uint32_t h_first = n, last_pos = 0, last_keep = 0;
    HIPCHK(c, hipMemcpyAsync(&h_first, first, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&last_pos, pos + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&last_keep, keep + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint32_t nk = last_pos + last_keep;
    if (nk > 0)
        LAUNCH(c, "k_dedupe_collect", k_dedupe_collect, dim3(grid), dim3(PAFFY_NT), 0, keys, keep, pos, n, h_first, remember ? 1 : 0, kept,
               static_cast<uint64_t *>(D.seen_hi.p), static_cast<uint64_t *>(D.seen_lo.p), (uint32_t)D.seen_n);
    if (remember && nk > 0) { /* the memory sorted again: by lo, then by hi */
        const size_t m = D.seen_n + nk;
        uint64_t *hi = static_cast<uint64_t *>(D.seen_hi.p), *lo = static_cast<uint64_t *>(D.seen_lo.p);
        uint64_t *hi2 = static_cast<uint64_t *>(D.seen_hi2.p), *lo2 = static_cast<uint64_t *>(D.seen_lo2.p);
        if (dd_sort_pairs64(c, D, lo, lo2, hi, hi2, m)) return PAFFY_E_HIP;
        if (dd_sort_pairs64(c, D, hi2, hi, lo2, lo, m)) return PAFFY_E_HIP;
        D.seen_n = m;
    }
    *n_kept = nk;
    *first_bad = h_first;
    return 0;
}
