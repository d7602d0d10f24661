/*
 * chain_host.h -- host side of `paffy chain` (included by paffy_hip.hip after coverage_host.h, whose batch bookkeeping, radix sort
 * wrappers and line writer it shares). Every step runs on the device; the host sizes buffers and reads back three counters.
 */
#ifndef PAFFY_CHAIN_HOST_H_
#define PAFFY_CHAIN_HOST_H_

#include "chain_kernel.h"

struct ChainState {
    DevBuf i64[12]; /* per record: qs qe ts te sc (5), level (kept in CovState), per position: qs qe ts te sc mq best (7) */
    DevBuf qkey, ghash, ord1, ord2, rank, start, gid, idx, prank, pred, neg, taken, is_tail, tail_of, link, total, chain_of_tail, chain_id, score_key, o1, o2, o3, cls, big_list, n_big, rank_of, claim,
        tag_chain, tag_score, check_key, iota;
    uint64_t n_out = 0;
    uint32_t group_salt = 0; /* salt the last run's grouping passed its name check with (0 unless group keys collided) */
};
static ChainState &chain_state(paffy_hip_ctx *c);

static int chain_run(paffy_hip_ctx *c, const ChainOpts &o, paffy_error *err) {
    CovState &S = cov_state(c);
    ChainState &H = chain_state(c);
    memset(err, 0, sizeof(*err));
    H.n_out = 0;
    const uint32_t n = (uint32_t)S.n_rec;
    if (n == 0) return 0;
    DevInfo hinfo;
    if (cov_fetch(c, &hinfo, S.info.p, sizeof(hinfo))) return PAFFY_E_HIP;
    auto report = [&](unsigned long long key) -> int {
        err->code = (int32_t)(key & 0xff);
        err->stage = (int32_t)((key >> 8) & 0xff) - 1;
        err->record = (int64_t)(key >> 16);
        err->aux = 0;
        if (err->stage < 0) {
            RecMeta m;
            if (cov_fetch(c, &m, static_cast<RecMeta *>(S.meta.p) + err->record, sizeof(m))) return PAFFY_E_HIP;
            err->aux = m.err_aux;
        }
        return 0;
    };
    if (hinfo.first_err_key != ~0ull) return report(hinfo.first_err_key); /* read_pafs parses every line first */
    {
        std::vector<const uint8_t *> ptrs;
        for (const CovBatch &b : S.batches) ptrs.push_back(b.in);
        if (ensure(c, S.batch_ptrs, sizeof(void *) * ptrs.size())) return PAFFY_E_HIP;
        HIPCHK(c, hipMemcpyAsync(S.batch_ptrs.p, ptrs.data(), sizeof(void *) * ptrs.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    const size_t n1 = (size_t)n + 1;
    for (DevBuf &b : H.i64)
        if (ensure(c, b, sizeof(int64_t) * n1)) return PAFFY_E_HIP;
    DevBuf *u64s[] = {&H.qkey, &H.ghash, &H.score_key, &S.k64a, &S.k64b};
    for (DevBuf *b : u64s)
        if (ensure(c, *b, sizeof(uint64_t) * n1)) return PAFFY_E_HIP;
    DevBuf *u32s[] = {&H.ord1, &H.ord2, &H.rank, &H.start, &H.gid, &H.idx, &H.prank, &H.pred, &H.tail_of, &H.link, &H.chain_of_tail, &H.chain_id, &H.o1, &H.o2, &H.o3, &H.cls, &H.big_list, &H.rank_of, &H.claim,
                      &H.iota, &S.flags, &S.scan32, &S.v32a, &S.v32b, &S.order};
    for (DevBuf *b : u32s)
        if (ensure(c, *b, sizeof(uint32_t) * (n1 + 1))) return PAFFY_E_HIP;
    DevBuf *u8s[] = {&H.neg, &H.taken, &H.is_tail};
    for (DevBuf *b : u8s)
        if (ensure(c, *b, n1)) return PAFFY_E_HIP;
    if (ensure(c, H.total, sizeof(int64_t) * n1) || ensure(c, H.tag_chain, sizeof(int64_t) * n1) || ensure(c, H.tag_score, sizeof(int64_t) * n1) ||
        ensure(c, S.level, sizeof(int64_t) * n1) || ensure(c, H.check_key, sizeof(unsigned long long)) || ensure(c, H.n_big, sizeof(uint32_t)))
        return PAFFY_E_HIP;
    auto I64 = [&](int k) { return static_cast<int64_t *>(H.i64[k].p); };
    auto U32 = [&](DevBuf &b) { return static_cast<uint32_t *>(b.p); };
    auto U64 = [&](DevBuf &b) { return static_cast<uint64_t *>(b.p); };
    auto U8 = [&](DevBuf &b) { return static_cast<uint8_t *>(b.p); };
    const uint32_t grid = (n + PAFFY_NT - 1) / PAFFY_NT;
    RecMeta *meta = static_cast<RecMeta *>(S.meta.p);
    ChainRecs R{I64(0), I64(1), I64(2), I64(3), I64(4), U64(H.qkey), U64(H.ghash), static_cast<int64_t *>(S.level.p)};
    uint32_t n_groups = 0;
    for (uint32_t salt = 0;; salt++) {
        LAUNCH(c, "k_chain_keys", k_chain_keys, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const uint8_t *const *>(S.batch_ptrs.p), static_cast<const RecMeta *>(meta), n, o, R,
               static_cast<DevInfo *>(S.info.p), salt);
        if (cov_fetch(c, &hinfo, S.info.p, sizeof(hinfo))) return PAFFY_E_HIP;
        if (hinfo.first_err_key != ~0ull) return report(hinfo.first_err_key);
        /* processing order: query start, then input order (impl/chaining.c:14-21, 139) */
        LAUNCH(c, "k_iota32", k_iota32, dim3(grid), dim3(PAFFY_NT), 0, U32(H.iota), n);
        if (cov_sort_pairs(c, S, U64(H.qkey), U64(S.k64a), U32(H.iota), U32(H.ord1), n)) return PAFFY_E_HIP;
        LAUNCH(c, "k_chain_rank", k_chain_rank, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(U32(H.ord1)), n, U32(H.rank));
        /* groups: (query name, target name, strand), members in processing order */
        LAUNCH(c, "k_gather_u64", k_gather_u64, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const uint64_t *>(U64(H.ghash)), static_cast<const uint32_t *>(U32(H.ord1)), n, U64(S.k64b));
        if (cov_sort_pairs(c, S, U64(S.k64b), U64(S.k64a), U32(H.ord1), U32(H.ord2), n)) return PAFFY_E_HIP;
        LAUNCH(c, "k_cov_run_heads", k_cov_run_heads, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const uint64_t *>(U64(S.k64a)), n, U32(S.flags));
        if (cov_incl_scan32(c, S, U32(S.flags), U32(S.scan32), n)) return PAFFY_E_HIP;
        if (cov_fetch(c, &n_groups, U32(S.scan32) + (n - 1), sizeof(uint32_t))) return PAFFY_E_HIP;
        LAUNCH(c, "k_chain_group_starts", k_chain_group_starts, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(U32(S.flags)), static_cast<const uint32_t *>(U32(S.scan32)),
               n, U32(H.start), U32(H.gid));
        /* the names behind the hashes (impl/chaining.c:37-54 compares strings): a group holding two different keys is regrouped under the next salt */
        uint32_t *collide = U32(H.n_big), hit = 0;
        HIPCHK(c, hipMemsetAsync(collide, 0, sizeof(uint32_t), c->stream));
        LAUNCH(c, "k_chain_verify_groups", k_chain_verify_groups, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const uint8_t *const *>(S.batch_ptrs.p), static_cast<const RecMeta *>(meta),
               static_cast<const uint32_t *>(U32(H.ord2)), static_cast<const uint32_t *>(U32(H.start)), static_cast<const uint32_t *>(U32(H.gid)), n, collide);
        if (cov_fetch(c, &hit, collide, sizeof(uint32_t))) return PAFFY_E_HIP;
        H.group_salt = salt;
        if (!hit) break;
        if (salt == 7) {
            c->last_error = "chain groups keep colliding under eight differently salted 64-bit hashes";
            return PAFFY_E_UNSUPPORTED;
        }
    }
    ChainPos Q{I64(5), I64(6), I64(7), I64(8), I64(9), I64(10), U32(H.idx), U32(H.prank), I64(11), U32(H.pred), U8(H.neg)};
    LAUNCH(c, "k_chain_gather", k_chain_gather, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const RecMeta *>(meta), R, static_cast<const uint32_t *>(U32(H.ord2)),
           static_cast<const uint32_t *>(U32(H.rank)), n, Q);
    const uint32_t wgrid = (n_groups + PAFFY_NWAVE - 1) / PAFFY_NWAVE;
    LAUNCH(c, "k_chain_prefix_max", k_chain_prefix_max, dim3(wgrid), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(U32(H.start)), n_groups, Q);
    HIPCHK(c, hipMemsetAsync(H.n_big.p, 0, sizeof(uint32_t), c->stream));
    LAUNCH(c, "k_chain_big_list", k_chain_big_list, dim3((n_groups + PAFFY_NT - 1) / PAFFY_NT), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(U32(H.start)), n_groups,
           U32(H.big_list), U32(H.n_big));
    LAUNCH(c, "k_chain_dp", k_chain_dp, dim3(wgrid), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(U32(H.start)), n_groups, o, Q);
    LAUNCH(c, "k_chain_dp_big", k_chain_dp_big, dim3(256), dim3(CHAIN_BIG_NT), 0, static_cast<const uint32_t *>(U32(H.start)), static_cast<const uint32_t *>(U32(H.big_list)),
           static_cast<const uint32_t *>(U32(H.n_big)), o, Q);
    /* (chain score desc, processing index desc): least significant key first, stable sorts */
    LAUNCH(c, "k_chain_not", k_chain_not, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(U32(H.prank)), n, U32(S.v32a));
    if (cov_sort_pairs32(c, S, U32(S.v32a), U32(S.v32b), U32(H.iota), U32(H.o1), n)) return PAFFY_E_HIP;
    LAUNCH(c, "k_chain_desc_keys", k_chain_desc_keys, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const int64_t *>(I64(11)), static_cast<const uint32_t *>(U32(H.o1)), n, U64(S.k64b));
    if (cov_sort_pairs(c, S, U64(S.k64b), U64(S.k64a), U32(H.o1), U32(H.o2), n)) return PAFFY_E_HIP; /* o2: all positions by (score desc, index desc) */
    /* every chain end walks its chain (o2 gives the ranks) */
    HIPCHK(c, hipMemsetAsync(H.taken.p, 0, n1, c->stream));          /* has_child */
    HIPCHK(c, hipMemsetAsync(H.claim.p, 0xff, sizeof(uint32_t) * n1, c->stream));
    LAUNCH(c, "k_chain_rank_and_children", k_chain_rank_and_children, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(U32(H.o2)), Q, n, U32(H.rank_of), U8(H.taken));
    LAUNCH(c, "k_chain_claim", k_chain_claim, dim3(grid), dim3(PAFFY_NT), 0, Q, static_cast<const uint32_t *>(U32(H.rank_of)), static_cast<const uint8_t *>(U8(H.taken)), n,
           U32(H.claim));
    LAUNCH(c, "k_chain_own", k_chain_own, dim3(grid), dim3(PAFFY_NT), 0, o, Q, static_cast<const uint32_t *>(U32(H.rank_of)), static_cast<const uint8_t *>(U8(H.taken)),
           static_cast<const uint32_t *>(U32(H.claim)), n, U32(H.tail_of), U32(H.link), static_cast<int64_t *>(H.total.p), U8(H.is_tail));
    /* chain ids: the tails by (strand, score desc, index desc) */
    LAUNCH(c, "k_chain_class", k_chain_class, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(U32(H.o2)), static_cast<const uint8_t *>(U8(H.is_tail)),
           static_cast<const uint8_t *>(U8(H.neg)), n, U32(H.cls));
    if (cov_sort_pairs32(c, S, U32(H.cls), U32(S.v32b), U32(H.o2), U32(H.o1), n)) return PAFFY_E_HIP;
    LAUNCH(c, "k_chain_number", k_chain_number, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(U32(H.o1)), static_cast<const uint8_t *>(U8(H.is_tail)), n,
           U32(H.chain_of_tail));
    LAUNCH(c, "k_chain_out_keys", k_chain_out_keys, dim3(grid), dim3(PAFFY_NT), 0, Q, static_cast<const uint32_t *>(U32(H.tail_of)),
           static_cast<const uint32_t *>(U32(H.chain_of_tail)), n, U32(H.chain_id), U64(H.score_key));
    /* output order: own score desc; equal scores stay in the order the chains were written (chain, then link from the tail) */
    if (cov_sort_pairs32(c, S, U32(H.link), U32(S.v32b), U32(H.iota), U32(H.o1), n)) return PAFFY_E_HIP;
    LAUNCH(c, "k_gather_u32", k_gather_u32, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const uint32_t *>(U32(H.chain_id)), static_cast<const uint32_t *>(U32(H.o1)), n, U32(S.v32a));
    if (cov_sort_pairs32(c, S, U32(S.v32a), U32(S.v32b), U32(H.o1), U32(H.o2), n)) return PAFFY_E_HIP;
    LAUNCH(c, "k_gather_u64", k_gather_u64, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const uint64_t *>(U64(H.score_key)), static_cast<const uint32_t *>(U32(H.o2)), n, U64(S.k64b));
    if (cov_sort_pairs(c, S, U64(S.k64b), U64(S.k64a), U32(H.o2), U32(H.o3), n)) return PAFFY_E_HIP;
    HIPCHK(c, hipMemsetAsync(H.check_key.p, 0xff, sizeof(unsigned long long), c->stream));
    LAUNCH(c, "k_chain_finish", k_chain_finish, dim3(grid), dim3(PAFFY_NT), 0, meta, Q, static_cast<const uint32_t *>(U32(H.o3)), static_cast<const uint32_t *>(U32(H.tail_of)),
           static_cast<const uint32_t *>(U32(H.chain_id)), static_cast<const uint32_t *>(U32(H.link)), static_cast<const int64_t *>(H.total.p), n, U32(S.order),
           static_cast<int64_t *>(H.tag_chain.p), static_cast<int64_t *>(H.tag_score.p), static_cast<unsigned long long *>(H.check_key.p));
    LAUNCH(c, "k_chain_find_failed", k_chain_find_failed, dim3(grid), dim3(PAFFY_NT), 0, static_cast<const RecMeta *>(meta), Q, static_cast<const uint32_t *>(U32(H.chain_id)),
           static_cast<const uint32_t *>(U32(H.link)), n, static_cast<const unsigned long long *>(H.check_key.p), static_cast<DevInfo *>(S.info.p));
    if (cov_fetch(c, &hinfo, S.info.p, sizeof(hinfo))) return PAFFY_E_HIP;
    if (c->profile) prof_collect(c);
    if (hinfo.first_err_key != ~0ull) return report(hinfo.first_err_key);
    H.n_out = n;
    return 0;
}

#endif
