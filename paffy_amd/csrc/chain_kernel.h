/*
 * chain_kernel.h -- `paffy chain` (impl/paf_chain.c:123-127, impl/chaining.c:136-343) on the device.
 *
 * What the reference does, per strand: sort the records by (trimmed, for '-' mirrored) query start, then for each one look, in a
 * sorted set of the chains seen so far, for the best earlier record of the same query, target and strand that ends before it
 * starts on both sequences with gaps of at most max_gap (score = own + predecessor's chain score - gap cost, taken only when the
 * gap costs less than the record scores); then pull the chains out from the highest score down, cutting a chain where its next
 * link already belongs to a better one; number them; print all records by descending own score.
 *
 * The same result without a set: records that can chain share (query name, target name, strand) -- a group. Inside a group the
 * recurrence runs over the records in processing order, and the candidates of record i are the earlier records p with
 * target_end(p) <= target_start(i), query_end(p) <= query_start(i), both gaps <= max_gap; the set's descending scan takes the
 * first best one, i.e. the largest (target_end, query_end, address) among equal scores. Dropping chains from the set
 * (impl/chaining.c:176-179) changes nothing: what is dropped can never pass the query-gap test again, the query starts only grow.
 * Groups are independent: one wave per group for the recurrence (the 64 lanes share the candidates of a record; a workgroup of 1024
 * for groups of more than 4096 records), every chain end walks its own chain; sorting, grouping and numbering are device radix sorts
 * (chain_host.h).
 * Addresses (the last key of every comparator of the reference) are restated as creation order: the input index for records, the
 * processing index for chains (parity on exact ties is unpinned).
 */
#pragma once

#define CHAIN_NONE 0xffffffffu
#define CHAIN_BIG_GROUP 4096u /* groups above this size run their recurrence on a whole workgroup */
#define CHAIN_BIG_NT 1024

struct ChainOpts {
    int64_t gap_open, gap_extend, max_gap;
    float trim;
};

__device__ __forceinline__ int64_t chain_gap_cost(const ChainOpts &o, int64_t dq, int64_t dt) { /* impl/paf_chain.c:36-45 */
    return dq + dt == 0 ? 0 : o.gap_open + o.gap_extend * (dq + dt);
}

/* per record, input order: trimmed coordinates (impl/chaining.c:270-299), sort keys, group hash, the level the line writer keeps */
struct ChainRecs {
    int64_t *qs, *qe, *ts, *te, *sc;
    uint64_t *qkey;  /* query start, biased: ascending order */
    uint64_t *ghash; /* query name, target name, strand */
    int64_t *level;
};

__global__ __launch_bounds__(PAFFY_NT) void k_chain_keys(const uint8_t *const *batch_in, const RecMeta *meta, uint32_t n, ChainOpts o, ChainRecs R, DevInfo *info, uint32_t salt) {
    const uint32_t r = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (r >= n) return;
    const RecMeta m = meta[r];
    R.level[r] = m.tile_level;
    if (m.err) return; /* reported by k_cov_collect */
    int bad = 0;
    if (!(o.trim >= 0.0f && o.trim <= 1.0f)) bad = 1; /* assert, impl/chaining.c:275 */
    const float fq = (float)(m.qe - m.qs) * o.trim, ft = (float)(m.te - m.ts) * o.trim; /* int64 * float -> float, then truncated */
    const int64_t mq = (int64_t)fq, mt = (int64_t)ft;
    if (mq < 0 || mt < 0) bad = 1; /* asserts :278-279 */
    if (bad) {
        atomicMin(&info->first_err_key, ((unsigned long long)r << 16) | (1ull << 8) | (unsigned long long)PAFFY_ERR_CHAIN_ASSERT);
        return;
    }
    const int64_t t = (mq < mt ? mq : mt) / 2;
    int64_t qs = m.qs + t, qe = m.qe - t;
    if (!m.same_strand) { /* invert_query_strand, impl/chaining.c:255-259 */
        const int64_t k = qs;
        qs = -qe;
        qe = -k;
    }
    R.qs[r] = qs;
    R.qe[r] = qe;
    R.ts[r] = m.ts + t;
    R.te[r] = m.te - t;
    R.sc[r] = m.score;
    R.qkey[r] = (uint64_t)qs ^ 0x8000000000000000ull;
    const uint8_t *in = batch_in[m.pad1];
    /* the group of a record: a 64-bit hash first, the names behind it checked against the group's first record afterwards
       (k_chain_verify_groups; the reference compares the strings, impl/chaining.c:37-54); a collision regroups with the next salt */
    uint64_t h = 1469598103934665603ull ^ ((uint64_t)salt * 0x9E3779B97F4A7C15ull);
    for (uint32_t i = 0; i < m.qname_len; i++) h = (h ^ in[m.qname_off + i]) * 1099511628211ull;
    h = (h ^ 0xffu) * 1099511628211ull;
    for (uint32_t i = 0; i < m.tname_len; i++) h = (h ^ in[m.tname_off + i]) * 1099511628211ull;
    h = (h ^ (m.same_strand ? 0x2bu : 0x2du)) * 1099511628211ull;
    h ^= (uint64_t)m.qname_len << 32 | m.tname_len; /* names of different lengths never share a hash by accident of their bytes alone */
    R.ghash[r] = h ^ (h >> 31);
}

/* position j of the grouped order against the first position of its group: query name, target name and strand, byte for byte */
__global__ __launch_bounds__(PAFFY_NT) void k_chain_verify_groups(const uint8_t *const *batch_in, const RecMeta *meta, const uint32_t *ord, const uint32_t *start, const uint32_t *gid,
                                                                   uint32_t n, uint32_t *collide) {
    const uint32_t j = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (j >= n) return;
    const uint32_t h = start[gid[j]];
    if (h == j) return;
    const RecMeta &a = meta[ord[j]], &b = meta[ord[h]];
    const uint8_t *ia = batch_in[a.pad1], *ib = batch_in[b.pad1];
    if ((a.same_strand != 0) != (b.same_strand != 0) || !cov_same_bytes(ia + a.qname_off, a.qname_len, ib + b.qname_off, b.qname_len) ||
        !cov_same_bytes(ia + a.tname_off, a.tname_len, ib + b.tname_off, b.tname_len))
        atomicOr(collide, 1u);
}

__global__ __launch_bounds__(PAFFY_NT) void k_chain_rank(const uint32_t *ord, uint32_t n, uint32_t *rank) {
    const uint32_t k = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (k < n) rank[ord[k]] = k;
}
__global__ __launch_bounds__(PAFFY_NT) void k_chain_not(const uint32_t *in, uint32_t n, uint32_t *out) {
    const uint32_t k = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (k < n) out[k] = ~in[k]; /* ascending in this = descending in the processing index */
}
__global__ __launch_bounds__(PAFFY_NT) void k_chain_desc_keys(const int64_t *score, const uint32_t *pos, uint32_t n, uint64_t *key) {
    const uint32_t k = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (k < n) key[k] = cov_desc_key(score[pos[k]]);
}
__global__ __launch_bounds__(PAFFY_NT) void k_gather_u32(const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t *dst) {
    const uint32_t k = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (k < n) dst[k] = src[idx[k]];
}
__global__ __launch_bounds__(PAFFY_NT) void k_chain_group_starts(const uint32_t *flag, const uint32_t *scan, uint32_t n, uint32_t *start, uint32_t *gid) {
    const uint32_t j = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (j >= n) return;
    gid[j] = scan[j] - 1u;
    if (flag[j]) start[scan[j] - 1u] = j;
    if (j == n - 1) start[scan[j]] = n;
}

/* per position (records grouped, processing order inside a group) */
struct ChainPos {
    int64_t *qs, *qe, *ts, *te, *sc, *mq; /* mq: largest query end of the group up to here */
    uint32_t *idx, *rank;                 /* input index (the address stand-in), processing index */
    int64_t *best;                        /* chain->score */
    uint32_t *pred;                       /* chain->pChain as a position */
    uint8_t *neg;
};
__global__ __launch_bounds__(PAFFY_NT) void k_chain_gather(const RecMeta *meta, ChainRecs R, const uint32_t *ord, const uint32_t *rank, uint32_t n, ChainPos Q) {
    const uint32_t j = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (j >= n) return;
    const uint32_t r = ord[j];
    Q.qs[j] = R.qs[r];
    Q.qe[j] = R.qe[r];
    Q.ts[j] = R.ts[r];
    Q.te[j] = R.te[r];
    Q.sc[j] = R.sc[r];
    Q.idx[j] = r;
    Q.rank[j] = rank[r];
    Q.neg[j] = meta[r].same_strand ? 0 : 1;
}

__device__ __forceinline__ int64_t wave_shfl64(int64_t v, int src_lane) {
    const int lo = __shfl((int)(uint32_t)(uint64_t)v, src_lane), hi = __shfl((int)(uint32_t)((uint64_t)v >> 32), src_lane);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ int64_t wave_shfl_xor64(int64_t v, int m) {
    const int lo = __shfl_xor((int)(uint32_t)(uint64_t)v, m), hi = __shfl_xor((int)(uint32_t)((uint64_t)v >> 32), m);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

/* running maximum of the query ends inside every group: bounds the window of candidates from below */
__global__ __launch_bounds__(PAFFY_NT) void k_chain_prefix_max(const uint32_t *start, uint32_t n_groups, ChainPos Q) {
    const uint32_t g = blockIdx.x * PAFFY_NWAVE + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (g >= n_groups) return;
    const uint32_t g0 = start[g], g1 = start[g + 1];
    int64_t carry = INT64_MIN;
    for (uint32_t base = g0; base < g1; base += 64) {
        const uint32_t j = base + lane;
        int64_t v = j < g1 ? Q.qe[j] : INT64_MIN;
        for (int d = 1; d < 64; d <<= 1) {
            const int lo = __shfl_up((int)(uint32_t)(uint64_t)v, d), hi = __shfl_up((int)(uint32_t)((uint64_t)v >> 32), d);
            const int64_t o = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
            if ((int)lane >= d && o > v) v = o;
        }
        if (carry > v) v = carry;
        if (j < g1) Q.mq[j] = v;
        const int lo = __shfl((int)(uint32_t)(uint64_t)v, 63), hi = __shfl((int)(uint32_t)((uint64_t)v >> 32), 63);
        carry = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
    }
}

/*
 * The recurrence, impl/chaining.c:150-205: one wave per group. Lane l looks at the candidates p with (p - g0) % 64 == l and is the
 * one that writes best[] / pred[] of those positions: a lane only ever reads values it wrote itself, so the wave needs no fence.
 */
__global__ __launch_bounds__(PAFFY_NT) void k_chain_dp(const uint32_t *start, uint32_t n_groups, ChainOpts o, ChainPos Q) {
    const uint32_t g = blockIdx.x * PAFFY_NWAVE + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (g >= n_groups) return;
    const uint32_t g0 = start[g], g1 = start[g + 1];
    if (g1 - g0 > CHAIN_BIG_GROUP) return; /* k_chain_dp_big has it */
    uint32_t lo = g0; /* everything before lo ends more than max_gap before the current query start */
    for (uint32_t i = g0; i < g1; i++) {
        const int64_t qs_i = Q.qs[i], ts_i = Q.ts[i], sc_i = Q.sc[i];
        const uint32_t idx_i = Q.idx[i];
        while (lo < i && Q.mq[lo] < qs_i - o.max_gap) lo++;
        int64_t b_cs = INT64_MIN, b_te = 0, b_qe = 0;
        uint32_t b_idx = 0, b_p = CHAIN_NONE;
        uint32_t p = lo + ((lane + 64u - ((lo - g0) & 63u)) & 63u); /* first position >= lo that is mine */
        for (; p < i; p += 64) {
            const int64_t qe_p = Q.qe[p], te_p = Q.te[p];
            if (qs_i < qe_p || qs_i - qe_p > o.max_gap) continue;
            if (ts_i < te_p || ts_i - te_p > o.max_gap) continue;
            const uint32_t idx_p = Q.idx[p];
            if (te_p == ts_i && qe_p == qs_i && idx_p > idx_i) continue; /* sorts after the search key (impl/chaining.c:71-79) */
            const int64_t gc = chain_gap_cost(o, qs_i - qe_p, ts_i - te_p);
            if (!(gc < sc_i)) continue;
            const int64_t cs = sc_i + Q.best[p] - gc;
            /* the set is walked downwards and only a strictly better score replaces the choice: among equals the largest key wins */
            const bool better = cs > b_cs || (cs == b_cs && (te_p > b_te || (te_p == b_te && (qe_p > b_qe || (qe_p == b_qe && idx_p > b_idx)))));
            if (better) {
                b_cs = cs; b_te = te_p; b_qe = qe_p; b_idx = idx_p; b_p = p;
            }
        }
        for (int msk = 32; msk >= 1; msk >>= 1) {
            const int64_t o_cs = wave_shfl_xor64(b_cs, msk), o_te = wave_shfl_xor64(b_te, msk), o_qe = wave_shfl_xor64(b_qe, msk);
            const uint32_t o_idx = (uint32_t)__shfl_xor((int)b_idx, msk), o_p = (uint32_t)__shfl_xor((int)b_p, msk);
            const bool better = o_p != CHAIN_NONE && (b_p == CHAIN_NONE || o_cs > b_cs ||
                                                      (o_cs == b_cs && (o_te > b_te || (o_te == b_te && (o_qe > b_qe || (o_qe == b_qe && o_idx > b_idx))))));
            if (better) {
                b_cs = o_cs; b_te = o_te; b_qe = o_qe; b_idx = o_idx; b_p = o_p;
            }
        }
        if (lane == ((i - g0) & 63u)) {
            const bool take = b_p != CHAIN_NONE && b_cs > sc_i;
            Q.best[i] = take ? b_cs : sc_i;
            Q.pred[i] = take ? b_p : CHAIN_NONE;
        }
    }
}

/* the groups the one-wave kernel leaves alone */
__global__ __launch_bounds__(PAFFY_NT) void k_chain_big_list(const uint32_t *start, uint32_t n_groups, uint32_t *big_list, uint32_t *n_big) {
    const uint32_t g = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (g < n_groups && start[g + 1] - start[g] > CHAIN_BIG_GROUP) big_list[atomicAdd(n_big, 1u)] = g;
}
/*
 * The same recurrence for a large group on a 1024-thread workgroup, sixteen records at a time. A record's candidates are the records
 * before it in query-start order (thousands when the alignments are dense: everything that ends within max_gap in front of it). For a
 * tile of sixteen consecutive records the candidates in front of the tile are final, so sixteen waves scan them side by side, one
 * record each (phase 1: the lanes share the record's candidates, four loads in flight per lane, the best reduced with shuffles);
 * only the candidates inside the tile -- at most fifteen -- depend on scores of the same tile, and those are settled by wave 0 with one
 * lane per tile record, the records in order (phase 2). Rounds 1-2 ran one record at a time over all 1024 threads: two barriers and a
 * two-stage reduction per record, 5.6 us per record in a group of a million.
 */
struct ChainCand {
    int64_t cs, te, qe;
    uint32_t idx, p;
};
__device__ __forceinline__ bool chain_cand_better(const ChainCand &o, const ChainCand &b) { /* o replaces b: the set is walked downwards, among equal scores the largest key wins */
    return o.p != CHAIN_NONE && (b.p == CHAIN_NONE || o.cs > b.cs || (o.cs == b.cs && (o.te > b.te || (o.te == b.te && (o.qe > b.qe || (o.qe == b.qe && o.idx > b.idx))))));
}
__device__ __forceinline__ void chain_cand_reduce(ChainCand &b, int from_mask) { /* xor steps from_mask .. 1: every lane of the group gets the group's best */
    for (int msk = from_mask; msk >= 1; msk >>= 1) {
        ChainCand o;
        o.cs = wave_shfl_xor64(b.cs, msk);
        o.te = wave_shfl_xor64(b.te, msk);
        o.qe = wave_shfl_xor64(b.qe, msk);
        o.idx = (uint32_t)__shfl_xor((int)b.idx, msk);
        o.p = (uint32_t)__shfl_xor((int)b.p, msk);
        if (chain_cand_better(o, b)) b = o;
    }
}
/* candidate p (query / target end, input index, chain score) for the record (qs_i, ts_i, sc_i, idx_i): its chain score, or nothing */
__device__ __forceinline__ void chain_try(const ChainOpts &o, int64_t qs_i, int64_t ts_i, int64_t sc_i, uint32_t idx_i, int64_t qe_p, int64_t te_p, uint32_t idx_p,
                                          int64_t best_p, uint32_t p, ChainCand &b) {
    if (qs_i < qe_p || qs_i - qe_p > o.max_gap) return;
    if (ts_i < te_p || ts_i - te_p > o.max_gap) return;
    if (te_p == ts_i && qe_p == qs_i && idx_p > idx_i) return; /* sorts after the search key (impl/chaining.c:71-79) */
    const int64_t gc = chain_gap_cost(o, qs_i - qe_p, ts_i - te_p);
    if (!(gc < sc_i)) return;
    ChainCand c;
    c.cs = sc_i + best_p - gc;
    c.te = te_p; c.qe = qe_p; c.idx = idx_p; c.p = p;
    if (chain_cand_better(c, b)) b = c;
}
#define CHAIN_TILE (CHAIN_BIG_NT / 64)
#ifndef CHAIN_FLIGHT
#define CHAIN_FLIGHT 4
#endif
__global__ __launch_bounds__(CHAIN_BIG_NT) void k_chain_dp_big(const uint32_t *start, const uint32_t *big_list, const uint32_t *n_big, ChainOpts o, ChainPos Q) {
    __shared__ int64_t t_cs[CHAIN_TILE], t_te[CHAIN_TILE], t_qe[CHAIN_TILE];
    __shared__ uint32_t t_idx[CHAIN_TILE], t_p[CHAIN_TILE];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t count = *n_big;
    for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {
        const uint32_t g = big_list[k], g0 = start[g], g1 = start[g + 1];
        uint32_t lo = g0; /* per wave: everything before lo ends more than max_gap before this wave's record starts */
        for (uint32_t base = g0; base < g1; base += CHAIN_TILE) {
            const uint32_t i = base + wave;
            /* phase 1: the candidates in front of the tile, one record per wave */
            if (i < g1) {
                const int64_t qs_i = Q.qs[i], ts_i = Q.ts[i], sc_i = Q.sc[i];
                const uint32_t idx_i = Q.idx[i];
                while (lo < base && Q.mq[lo] < qs_i - o.max_gap) lo++;
                ChainCand b;
                b.cs = INT64_MIN; b.te = 0; b.qe = 0; b.idx = 0; b.p = CHAIN_NONE;
                for (uint32_t p0 = lo + lane; p0 < base; p0 += CHAIN_FLIGHT * 64) { /* CHAIN_FLIGHT candidates' loads in flight per lane */
                    int64_t qe4[CHAIN_FLIGHT], te4[CHAIN_FLIGHT], best4[CHAIN_FLIGHT];
                    uint32_t idx4[CHAIN_FLIGHT];
#pragma unroll
                    for (int u = 0; u < CHAIN_FLIGHT; u++) {
                        const uint32_t p = p0 + u * 64;
                        const bool in = p < base;
                        qe4[u] = in ? Q.qe[p] : INT64_MAX; /* fails the first test */
                        te4[u] = in ? Q.te[p] : 0;
                        idx4[u] = in ? Q.idx[p] : 0u;
                        best4[u] = in ? Q.best[p] : 0;
                    }
#pragma unroll
                    for (int u = 0; u < CHAIN_FLIGHT; u++) chain_try(o, qs_i, ts_i, sc_i, idx_i, qe4[u], te4[u], idx4[u], best4[u], p0 + u * 64, b);
                }
                chain_cand_reduce(b, 32);
                if (lane == 0) {
                    t_cs[wave] = b.cs; t_te[wave] = b.te; t_qe[wave] = b.qe; t_idx[wave] = b.idx; t_p[wave] = b.p;
                }
            }
            __syncthreads();
            /* phase 2: the candidates inside the tile; lane l of wave 0 is record base + l, the records are settled in order */
            if (wave == 0) {
                const uint32_t n_t = g1 - base < CHAIN_TILE ? g1 - base : CHAIN_TILE;
                const uint32_t me = base + (lane < n_t ? lane : 0u);
                const int64_t qs_m = Q.qs[me], ts_m = Q.ts[me], qe_m = Q.qe[me], te_m = Q.te[me], sc_m = Q.sc[me];
                const uint32_t idx_m = Q.idx[me];
                ChainCand mine; /* what phase 1 found for my record */
                mine.cs = INT64_MIN; mine.te = 0; mine.qe = 0; mine.idx = 0; mine.p = CHAIN_NONE;
                if (lane < n_t) {
                    mine.cs = t_cs[lane]; mine.te = t_te[lane]; mine.qe = t_qe[lane]; mine.idx = t_idx[lane]; mine.p = t_p[lane];
                }
                int64_t fin = sc_m; /* my record's chain score once it has been settled */
                for (uint32_t w = 0; w < n_t; w++) {
                    /* record w against the tile records in front of it: lane l < w offers itself */
                    const int64_t qs_w = wave_shfl64(qs_m, (int)w), ts_w = wave_shfl64(ts_m, (int)w), sc_w = wave_shfl64(sc_m, (int)w);
                    const uint32_t idx_w = (uint32_t)__shfl((int)idx_m, (int)w);
                    ChainCand b;
                    b.cs = INT64_MIN; b.te = 0; b.qe = 0; b.idx = 0; b.p = CHAIN_NONE;
                    if (lane < w) chain_try(o, qs_w, ts_w, sc_w, idx_w, qe_m, te_m, idx_m, fin, base + lane, b);
                    chain_cand_reduce(b, CHAIN_TILE / 2); /* lanes 0-15 hold the tile (lanes above offer nothing and only mix among themselves) */
                    if (lane == w) {
                        if (chain_cand_better(b, mine)) mine = b;
                        const bool take = mine.p != CHAIN_NONE && mine.cs > sc_m;
                        fin = take ? mine.cs : sc_m;
                        Q.best[me] = fin;
                        Q.pred[me] = take ? mine.p : CHAIN_NONE;
                    }
                }
                __threadfence_block();
            }
            __syncthreads(); /* the tile's scores are there for the records that follow; the LDS slots are free again */
        }
    }
}

/*
 * Chains from the highest score down, impl/chaining.c:213-230 + chain_to_pafs :118-135. The reference takes the best remaining chain
 * end, follows its links and cuts where a link already belongs to a better chain. A link always leads to a lower chain score (a record
 * is only chained when the gap costs less than it scores), so every record that something links to has been taken by the time its own
 * turn would come: the chain ends are exactly the records nothing links to, and a record goes to the best-ranked end that reaches it.
 * All ends walk at once: an end claims the records on its way with an atomic minimum of its rank (position in the order by score desc,
 * processing index desc) and stops where a better rank has been; a second walk over what it kept gives tail_of / link (distance from the
 * end) and total[end] = get_chain_score of the chain as cut (:93-116).
 */
__global__ __launch_bounds__(PAFFY_NT) void k_chain_rank_and_children(const uint32_t *by_score, ChainPos Q, uint32_t n, uint32_t *rank_of, uint8_t *has_child) {
    const uint32_t k = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (k >= n) return;
    rank_of[by_score[k]] = k;
    const uint32_t p = Q.pred[k];
    if (p != CHAIN_NONE) has_child[p] = 1;
}
__global__ __launch_bounds__(PAFFY_NT) void k_chain_claim(ChainPos Q, const uint32_t *rank_of, const uint8_t *has_child, uint32_t n, uint32_t *claim) {
    const uint32_t e = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (e >= n || has_child[e]) return;
    const uint32_t r = rank_of[e];
    uint32_t cur = e;
    for (;;) {
        if (atomicMin(&claim[cur], r) < r) break; /* a better end has been here: the rest of the way is its */
        cur = Q.pred[cur];
        if (cur == CHAIN_NONE) break;
    }
}
__global__ __launch_bounds__(PAFFY_NT) void k_chain_own(ChainOpts o, ChainPos Q, const uint32_t *rank_of, const uint8_t *has_child, const uint32_t *claim, uint32_t n,
                                                         uint32_t *tail_of, uint32_t *link, int64_t *total, uint8_t *is_tail) {
    const uint32_t e = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (e >= n) return;
    is_tail[e] = has_child[e] ? 0 : 1;
    if (has_child[e]) return;
    const uint32_t r = rank_of[e];
    tail_of[e] = e;
    link[e] = 0;
    int64_t sum = Q.sc[e];
    uint32_t cur = e, depth = 0;
    while (Q.pred[cur] != CHAIN_NONE) {
        const uint32_t p = Q.pred[cur];
        if (claim[p] != r) { /* the next link belongs to a better chain: this one ends here */
            Q.pred[cur] = CHAIN_NONE;
            break;
        }
        sum += Q.sc[p] - chain_gap_cost(o, Q.qs[cur] - Q.qe[p], Q.ts[cur] - Q.te[p]);
        tail_of[p] = e;
        link[p] = ++depth;
        cur = p;
    }
    total[e] = sum;
}

/* numbering: + strand chains first, each strand from the highest chain score down (the order they were pulled out in) */
__global__ __launch_bounds__(PAFFY_NT) void k_chain_class(const uint32_t *by_score, const uint8_t *is_tail, const uint8_t *neg, uint32_t n, uint32_t *cls) {
    const uint32_t k = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (k >= n) return;
    const uint32_t p = by_score[k];
    cls[k] = is_tail[p] ? (neg[p] ? 1u : 0u) : 2u;
}
__global__ __launch_bounds__(PAFFY_NT) void k_chain_number(const uint32_t *by_class, const uint8_t *is_tail, uint32_t n, uint32_t *chain_of_tail) {
    const uint32_t k = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (k >= n) return;
    const uint32_t p = by_class[k];
    if (is_tail[p]) chain_of_tail[p] = k; /* the tails come first */
}
/* per position: the keys of the output order (own score desc; among equals the order the chains were written in: chain, link) */
__global__ __launch_bounds__(PAFFY_NT) void k_chain_out_keys(ChainPos Q, const uint32_t *tail_of, const uint32_t *chain_of_tail, uint32_t n, uint32_t *chain_id, uint64_t *score_key) {
    const uint32_t p = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (p >= n) return;
    chain_id[p] = chain_of_tail[tail_of[p]];
    score_key[p] = cov_desc_key(Q.sc[p]);
}
/* output position k: the record, its tags; paf_check on the record as it was read (impl/chaining.c:323-335; no cigar was parsed) */
__global__ __launch_bounds__(PAFFY_NT) void k_chain_finish(RecMeta *meta, ChainPos Q, const uint32_t *out_pos, const uint32_t *tail_of, const uint32_t *chain_id,
                                                            const uint32_t *link, const int64_t *total, uint32_t n, uint32_t *order, int64_t *tag_chain, int64_t *tag_score,
                                                            unsigned long long *check_key) {
    const uint32_t k = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (k >= n) return;
    const uint32_t p = out_pos[k], r = Q.idx[p];
    order[k] = r;
    RecMeta &m = meta[r];
    m.chain_id = chain_id[p];
    m.chain_score = total[tail_of[p]];
    tag_chain[k] = m.chain_id;
    tag_score[k] = m.chain_score;
    int chk = 0;
    if (m.qs < 0 || m.qs >= m.qlen) chk = PAFFY_ERR_CHECK_QSTART;
    else if (m.qs > m.qe || m.qe > m.qlen) chk = PAFFY_ERR_CHECK_QEND;
    else if (m.ts < 0 || m.ts >= m.tlen) chk = PAFFY_ERR_CHECK_TSTART;
    else if (m.ts > m.te || m.te > m.tlen) chk = PAFFY_ERR_CHECK_TEND;
    /* the check runs chain by chain, link by link: the first failure in that order is the one the reference stops at */
    if (chk) atomicMin(check_key, ((unsigned long long)chain_id[p] << 32) | link[p]);
}
__global__ __launch_bounds__(PAFFY_NT) void k_chain_find_failed(const RecMeta *meta, ChainPos Q, const uint32_t *chain_id, const uint32_t *link, uint32_t n,
                                                                 const unsigned long long *check_key, DevInfo *info) {
    const uint32_t p = blockIdx.x * PAFFY_NT + threadIdx.x;
    if (p >= n || *check_key == ~0ull) return;
    if ((((unsigned long long)chain_id[p] << 32) | link[p]) != *check_key) return;
    const RecMeta &m = meta[Q.idx[p]];
    int chk = PAFFY_ERR_CHECK_TEND;
    if (m.qs < 0 || m.qs >= m.qlen) chk = PAFFY_ERR_CHECK_QSTART;
    else if (m.qs > m.qe || m.qe > m.qlen) chk = PAFFY_ERR_CHECK_QEND;
    else if (m.ts < 0 || m.ts >= m.tlen) chk = PAFFY_ERR_CHECK_TSTART;
    info->first_err_key = ((unsigned long long)Q.idx[p] << 16) | (1ull << 8) | (unsigned long long)chk;
}
