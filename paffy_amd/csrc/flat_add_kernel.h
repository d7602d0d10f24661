/*
 * flat_add_kernel.h -- `paffy add_mismatches` (paf_encode_mismatches, impl/paf.c:739-784; the command loop impl/paf_add_mismatches.c:108-136)
 * on the pieces of the flat pass (flat_kernel.h), for the pipe that consists of that command alone (BASELINE cfg4).
 *
 * The record kernels (record_kernel.h) encode a record inside the workgroup that holds its ops in LDS; a record whose ops fit no LDS
 * store goes through the arena class, one workgroup walking eight columns per lane and step -- on the heavy-tailed stream of SURVEY 8d the
 * thirty records per batch above 36 864 ops cost twice as much as all the others together (22.8 of 39.9 ms per step). An M op's runs do
 * not depend on any other op (every M op starts a new run, impl/paf.c:770-776), only on where the op stands on the two sequences -- and
 * the flat pass knows the bases in front of every PIECE of a cigar from its summaries. So the work item is the piece (about 330 ops of
 * one record, whatever the record's length), and the two walks of the LDS encoder (mismatch_count_wave / mismatch_fill_wave) run on it
 * unchanged:
 *
 *   k_flat_parse     ops into the mirror, a summary per piece (here with the 16-column chunks of its M ops: the items of walk 1)
 *   k_add_prep       one lane per record: what the flat pass keeps, paf_check on the parsed ops, per piece the ops and the query /
 *                    target bases in front of it, the words of scratch its items and passed-through ops take
 *   (scan)           places of the pieces' scratch
 *   k_add_count      one wave per piece: walk 1 -- item words (match masks of 16 columns) and the ops that pass through, the number of ops
 *                    the piece becomes
 *   (scan)           places of the pieces' new ops: the new cigars of all records stand back to back in the arena
 *   k_add_fill       one wave per piece: walk 2 -- the runs as ops, the bytes of their text
 *   k_add_final      one lane per record: the line's length, the plan for the line writers (k_emit_line reads the new ops where they are)
 *
 * A record anything of this does not take (what the flat pass leaves, a sequence that is missing or too short, sums beyond 2^30, more
 * than 62 passed-through ops in a row) is left to the record kernels, which report what there is to report.
 */
#ifndef PAFFY_FLAT_ADD_KERNEL_H_
#define PAFFY_FLAT_ADD_KERNEL_H_

struct AddPiece { /* per piece slot, written by k_add_prep */
    uint32_t op_base;       /* ops of the record in front of the piece */
    uint32_t q_base, t_base; /* query / target bases in front of it */
    uint32_t rec;           /* its record (FLAT_NO_CHUNK: the slot holds no piece of a record this pass encodes) */
};

struct AddParams {
    KParams P;
    PieceSum *sums;
    AddPiece *pieces;
    uint32_t n_piece_slots;
    uint32_t *scr_cnt;        /* per piece slot: words of scratch (items + passed-through ops) */
    const uint64_t *scr_off;  /* their exclusive prefix */
    uint32_t *scratch;
    uint64_t scr_cap;         /* words scratch[] holds */
    uint32_t *new_cnt;        /* per piece slot: ops the piece becomes */
    const uint64_t *new_off;  /* their exclusive prefix: where the piece's new ops stand in new_ops[] */
    uint32_t *new_ops;        /* 4-byte ops (len << 3 | op), every record's new cigar in one run */
    uint64_t new_cap;         /* words new_ops[] holds */
    uint32_t *text_cnt;       /* per piece slot: bytes of the new ops' text */
    uint32_t *rec_bad;        /* per record: set by k_add_count when a piece's bases lie outside a sequence (the record kernels report it) */
    uint8_t *flat_done;
};

/* exclusive prefix sums of n 32-bit counts into 64-bit offsets: per block of SCAN32_BLOCK, then the blocks' bases */
#define SCAN32_PER 8u
#define SCAN32_BLOCK (256u * SCAN32_PER)
__global__ __launch_bounds__(256) void k_scan32_part(const uint32_t *in, uint32_t n, uint64_t *out, uint64_t *part) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    BlockComm scratch{scratch_mem, 0};
    const uint32_t i0 = blockIdx.x * SCAN32_BLOCK + threadIdx.x * SCAN32_PER;
    uint32_t x[SCAN32_PER];
    int64_t v[1] = {0}, tot[1];
#pragma unroll
    for (uint32_t j = 0; j < SCAN32_PER; j++) {
        x[j] = i0 + j < n ? in[i0 + j] : 0u;
        v[0] += x[j];
    }
    block_excl_scan<1>(v, tot, scratch);
    uint64_t run = (uint64_t)v[0];
#pragma unroll
    for (uint32_t j = 0; j < SCAN32_PER; j++) {
        if (i0 + j < n) out[i0 + j] = run;
        run += x[j];
    }
    if (threadIdx.x == 0) part[blockIdx.x] = (uint64_t)tot[0];
}
__global__ __launch_bounds__(256) void k_scan32_fix(uint32_t n, uint32_t n_blocks, uint64_t *out, const uint64_t *part, uint64_t *total) {
    __shared__ int64_t scratch_mem[2 * PAFFY_NWAVE * 4];
    BlockComm scratch{scratch_mem, 0};
    int64_t v[2] = {0, 0};
    for (uint32_t b = threadIdx.x; b < n_blocks; b += 256u) {
        if (b < blockIdx.x) v[0] += (int64_t)part[b];
        v[1] += (int64_t)part[b];
    }
    block_sum<2>(v, scratch);
    const uint32_t i0 = blockIdx.x * SCAN32_BLOCK + threadIdx.x * SCAN32_PER;
#pragma unroll
    for (uint32_t j = 0; j < SCAN32_PER; j++)
        if (i0 + j < n) out[i0 + j] += (uint64_t)v[0];
    if (blockIdx.x == 0 && threadIdx.x == 0 && total) *total = (uint64_t)v[1];
}

/* one lane per record: is the record this pass's, where do its pieces stand */
__global__ __launch_bounds__(256) void k_add_prep(AddParams A) {
    const KParams &P = A.P;
    const uint32_t rec = blockIdx.x * 256u + threadIdx.x;
    if (rec >= P.n_rec) return;
    A.flat_done[rec] = 0;
    A.rec_bad[rec] = 0;
    const RecMeta &m = P.meta[rec];
    if (m.err || !m.has_cg || m.cg_len == 0) return;
    if (P.rec_qseq[rec] < 0 || P.rec_tseq[rec] < 0) return; /* impl/paf_add_mismatches.c:117-127: the record kernels report it */
    const uint32_t cg_off = m.cg_off, cg_end = cg_off + m.cg_len;
    const uint32_t np = ((cg_end - 1u) >> FLAT_TILE_SHIFT) - (cg_off >> FLAT_TILE_SHIFT) + 1u;
    const uint32_t slot0 = (cg_off >> FLAT_TILE_SHIFT) + rec;
    uint64_t n = 0, q = 0, t = 0;
    uint32_t flags = 0;
    for (uint32_t p = 0; p < np; p++) { /* first: may the record stay? */
        const FlatPre s = lane_piece(A.sums, slot0 + p);
        flags |= s.cnt >> 16;
        n += s.cnt & 0xffffu;
        q += (uint64_t)s.m + s.x - s.del;
        t += (uint64_t)s.m + s.x - s.ins;
    }
    if ((flags & FLAT_F_IRREG) || n == 0 || q >= (1ull << 30) || t >= (1ull << 30) || n >= (1ull << 31)) return;
    /* paf_check on the record as parsed (impl/paf_add_mismatches.c:134: the runs of an M op add up to it, so the encoded record passes or
       fails with this one); a failing record is the record kernels' to report */
    if (m.qs < 0 || m.qs >= m.qlen || m.qs > m.qe || m.qe > m.qlen || m.ts < 0 || m.ts >= m.tlen || m.ts > m.te || m.te > m.tlen || (int64_t)q != m.qe - m.qs ||
        (int64_t)t != m.te - m.ts)
        return;
    /* the sequences must hold the record's ranges (the reference reads outside them: undefined; PAFFY_ERR_SEQ_RANGE from the record kernels) */
    if (m.qe > P.seqs[P.rec_qseq[rec]].len || m.te > P.seqs[P.rec_tseq[rec]].len) return;
    uint32_t nb = 0, qb = 0, tb = 0;
    for (uint32_t p = 0; p < np; p++) {
        const FlatPre s = lane_piece(A.sums, slot0 + p);
        const uint32_t cnt = s.cnt & 0xffffu;
        AddPiece ap;
        ap.op_base = nb; ap.q_base = qb; ap.t_base = tb; ap.rec = rec;
        A.pieces[slot0 + p] = ap;
        A.scr_cnt[slot0 + p] = s.extra + (cnt - s.rows); /* items + ops other than M */
        nb += cnt;
        qb += s.m + s.x - s.del;
        tb += s.m + s.x - s.ins;
    }
    A.flat_done[rec] = 2; /* in progress: k_add_final decides */
}

#define ADD_WAVES 4u
#define ADD_SEG_OPS (PAFFY_ROWS_MAX_OPS / 2u) /* new ops of a segment of a long line (closed at the next piece boundary) */
/* walk 1, one wave per piece */
__global__ __launch_bounds__(64 * ADD_WAVES) void k_add_count(AddParams A) {
    const KParams &P = A.P;
    const uint32_t n_waves = gridDim.x * ADD_WAVES;
    for (uint32_t slot = uni(blockIdx.x * ADD_WAVES + (threadIdx.x >> 6)); slot < A.n_piece_slots; slot += n_waves) {
        const AddPiece ap = A.pieces[slot];
        const uint32_t rec = uni(ap.rec);
        if (rec == FLAT_NO_CHUNK) continue;
        const RecMeta &m = P.meta[rec];
        const uint32_t cnt = uni(A.sums[slot].cnt) & 0xffffu, n_items_sum = uni(A.sums[slot].extra);
        if (cnt == 0 || A.scr_off[slot] + uni(A.scr_cnt[slot]) > A.scr_cap) { /* no room: the host learns the total and encodes the batch again */
            if ((threadIdx.x & 63u) == 0) A.new_cnt[slot] = 0;
            continue;
        }
        RecState s;
        load_state(m, s);
        OpsGlobal ops{P.ops_mirror + (m.cg_off >> 1), true};
        View<OpsGlobal> v;
        v.reset(ops, uni(ap.op_base) + cnt); /* the window never reaches past the piece */
        const int32_t qi = P.rec_qseq[rec], ti = P.rec_tseq[rec];
        uint32_t *items = A.scratch + A.scr_off[slot], *nm_list = items + n_items_sum;
        int64_t bad = INT64_MAX;
        uint32_t n_items = 0, n_nm = 0;
        const uint32_t wcnt = mismatch_count_wave(P, s, v, P.seq_base + P.seqs[qi].off, P.seqs[qi].len, P.seq_base + P.seqs[ti].off, P.seqs[ti].len, uni(ap.op_base),
                                                  uni(ap.op_base) + cnt, (int64_t)uni(ap.q_base), (int64_t)uni(ap.t_base), items, nm_list, &n_items, &n_nm, &bad);
        bad = wave_min(bad);
        if ((threadIdx.x & 63u) == 0) {
            A.new_cnt[slot] = wcnt;
            if (bad != INT64_MAX || n_items != n_items_sum) A.rec_bad[rec] = 1; /* outside a sequence, or more than 62 passed-through ops in a row */
        }
    }
}

/* walk 2, one wave per piece */
__global__ __launch_bounds__(64 * ADD_WAVES) void k_add_fill(AddParams A) {
    __shared__ uint32_t s_slots[ADD_WAVES][PAFFY_FILL_SLOTS];
    const KParams &P = A.P;
    const uint32_t n_waves = gridDim.x * ADD_WAVES, wave = uni(threadIdx.x >> 6);
    for (uint32_t slot = uni(blockIdx.x * ADD_WAVES + (threadIdx.x >> 6)); slot < A.n_piece_slots; slot += n_waves) {
        const uint32_t rec = uni(A.pieces[slot].rec);
        if (rec == FLAT_NO_CHUNK) continue;
        const uint32_t cnt = uni(A.sums[slot].cnt) & 0xffffu, n_items = uni(A.sums[slot].extra), n_nm = cnt - uni(A.sums[slot].rows);
        uint32_t text = 0;
        const uint64_t at = A.new_off[slot];
        if (cnt && !uni(A.rec_bad[rec]) && at + uni(A.new_cnt[slot]) <= A.new_cap && A.scr_off[slot] + uni(A.scr_cnt[slot]) <= A.scr_cap) {
            const uint32_t *items = A.scratch + A.scr_off[slot];
            text = mismatch_fill_wave(items, n_items, items + n_items, n_nm, 0u, A.new_ops + at, s_slots[wave]);
        }
        if ((threadIdx.x & 63u) == 0) A.text_cnt[slot] = text;
        __builtin_amdgcn_wave_barrier();
    }
    (void)P;
}

/* one lane per record: the line's length and the plan for the line writers */
__global__ __launch_bounds__(256) void k_add_final(AddParams A) {
    const KParams &P = A.P;
    const uint32_t rec = blockIdx.x * 256u + threadIdx.x;
    if (rec >= P.n_rec) return;
    bool done = false;
    if (A.flat_done[rec] == 2 && !A.rec_bad[rec]) {
        const RecMeta &m = P.meta[rec];
        const uint32_t cg_off = m.cg_off, cg_end = cg_off + m.cg_len;
        const uint32_t np = ((cg_end - 1u) >> FLAT_TILE_SHIFT) - (cg_off >> FLAT_TILE_SHIFT) + 1u;
        const uint32_t slot0 = (cg_off >> FLAT_TILE_SHIFT) + rec;
        uint64_t n = 0, text = 0;
        for (uint32_t p = 0; p < np; p++) {
            n += A.new_cnt[slot0 + p];
            text += A.text_cnt[slot0 + p];
        }
        const uint64_t at = A.new_off[slot0];
        RecState s;
        load_state(m, s);
        const uint32_t lenH = header_len(s, false);
        /* the one-wave line writer takes the line whole up to PAFFY_ROWS_MAX_OPS ops, a longer one as segments cut at piece boundaries
           (EmitItem: the header goes with the first); headers of kilobytes stay with the record kernels */
        const bool whole = n <= PAFFY_ROWS_MAX_OPS;
        uint32_t n_seg = 0, item0 = 0;
        bool fits = n > 0 && n < (1ull << 31) && lenH + 8 <= PAFFY_TMPL_MAX && at + n <= A.new_cap;
        if (fits && !whole) {
            uint32_t seg_ops = 0;
            for (uint32_t p = 0; p < np; p++) { /* how many segments */
                const uint32_t c = A.new_cnt[slot0 + p];
                if (seg_ops && seg_ops + c > ADD_SEG_OPS) {
                    n_seg++;
                    seg_ops = 0;
                }
                seg_ops += c;
                if (seg_ops > 4u * PAFFY_ROWS_MAX_OPS) fits = false; /* one piece that becomes more ops than a wave should write */
            }
            n_seg += seg_ops ? 1u : 0u;
            if (fits) {
                item0 = atomicAdd(&P.info->n_items, n_seg);
                if (item0 + n_seg > P.items_cap) fits = false;
            }
        }
        if (fits) {
            RecPlan *plan = static_cast<RecPlan *>(P.rec_plan) + rec;
            P.status[rec] = (uint32_t)KLASS_LDS << 16;
            P.err_aux[rec] = 0;
            P.n_ops[rec] = 0;
            P.out_len[rec] = (int64_t)lenH + (int64_t)text + 1;
            P.out_rows[rec] = 1;
            P.arena_off[rec] = at; /* in 4-byte words of new_ops[] (flag bit 20) */
            plan->qs = s.qs; plan->qe = s.qe; plan->ts = s.ts; plan->te = s.te; plan->sub_lo = 0; plan->sub_hi = 0;
            plan->lo = 0; plan->n = (uint32_t)n;
            plan->flags = 8u | ((uint32_t)s.type << 8) | (whole ? 0x10000u : 0x80000u) | 0x20000u | 0x100000u;
            plan->chunk = (((uint32_t)n + 63u) / 64u) | 1u;
            for (int w = 0; w < 4; w++) plan->wq[w] = plan->wt[w] = plan->wo[w] = 0;
            if (!whole) {
                uint32_t seg_ops = 0, seg_b = 0, done_ops = 0, g = 0;
                uint64_t seg_text = 0, done_text = 0;
                for (uint32_t p = 0; p <= np; p++) {
                    const uint32_t c = p < np ? A.new_cnt[slot0 + p] : 0u;
                    if (seg_ops && (p == np || seg_ops + c > ADD_SEG_OPS)) {
                        EmitItem it;
                        it.rec = rec; it.wb = seg_b; it.we = seg_b + seg_ops; it.pad = 0; it.cq0 = 0; it.ct0 = 0;
                        it.wo = (int64_t)done_text;
                        P.items[item0 + g++] = it;
                        seg_b += seg_ops;
                        done_text += seg_text;
                        seg_ops = 0;
                        seg_text = 0;
                    }
                    if (p < np) {
                        seg_ops += c;
                        seg_text += A.text_cnt[slot0 + p];
                    }
                }
                (void)done_ops;
            }
            done = true;
        }
    }
    A.flat_done[rec] = done ? 1 : 0;
    if (!done) {
        P.out_len[rec] = 0;
        P.out_rows[rec] = 0;
        P.status[rec] = 0;
        atomicAdd(&P.info->flat_legacy, 1u);
    }
}

#endif
