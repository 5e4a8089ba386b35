/* ngw_oracle.c — CPU restatement of the reference's reset()/step() hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library; the shipped path (gym_novel_gridworlds_amd/csrc) never
 * links or calls it.
 *
 * Parity pinning: this restatement is checked against golden vectors captured from the imported,
 * unmodified reference (tests/golden/<cfg>.npz, generator tests/golden/gen_golden.py): seeded resets
 * incl. the MT19937 stream position, lock-step traces, injected-state single steps, solved episodes
 * and the random_action.py loop (tests/test_oracle_golden.py).
 *
 * Reference files followed (gtatiya/gym-novel-gridworlds v1.2):
 *   gym_novel_gridworlds/envs/pogostick_v1_env.py   reset :86-157, add_item_to_map :159-181,
 *       step :230-367, update_block_in_front :369-389, is_block_in_front_next_to :391-411,
 *       craft :413-474, grab_entities :538-554
 *   gym_novel_gridworlds/envs/bow_v1_env.py          Extract_string :293-304, craft :386-441
 *   gym_novel_gridworlds/novelty_wrappers.py         AxeEasy :9-114, AxeMedium :117-213, AddItem :991-1034
 * Third-party algorithm restated: numpy legacy RandomState (MT19937 init_genrand / genrand_int32 and the
 * masked-rejection bounded draw used by choice / randint / shuffle), numpy 2.2.6 as installed here.
 *
 * The same sampling code runs on two 32-bit word sources: the global MT19937 stream (bit-exact with the
 * reference) and a per-(env, episode) Philox4x32-10 counter stream (what the HIP reset kernel uses).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ngw.h"

/* ------------------------------------------------------------------ MT19937 (numpy legacy seeding) */
typedef struct { uint32_t mt[624]; int pos; } ngwo_mt;

void ngwo_mt_seed(ngwo_mt* s, uint32_t seed) {          /* init_genrand: np.random.seed(int) */
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->pos = 624;
}

uint32_t ngwo_mt_next(ngwo_mt* s) {                     /* genrand_int32 */
    if (s->pos >= 624) {
        uint32_t* mt = s->mt;
        for (int k = 0; k < 624; k++) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->pos = 0;
    }
    uint32_t y = s->mt[s->pos++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

/* ------------------------------------------------------------------ Philox4x32-10 counter stream */
typedef struct { uint32_t key[2]; uint32_t ctr[4]; uint32_t buf[4]; int have; } ngwo_philox;

static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void ngwo_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }

/* word i of the reset stream of (seed, global env index, episode): block i/4 -> ctr = (i/4, episode, env_lo, env_hi) */
static void philox_reset_init(ngwo_philox* p, uint64_t seed, uint64_t env, uint32_t episode) {
    p->key[0] = (uint32_t)seed; p->key[1] = (uint32_t)(seed >> 32);
    p->ctr[0] = 0; p->ctr[1] = episode; p->ctr[2] = (uint32_t)env; p->ctr[3] = (uint32_t)(env >> 32);
    p->have = 0;
}

static uint32_t philox_next(ngwo_philox* p) {
    if (!p->have) { philox4x32_10(p->ctr, p->key, p->buf); p->ctr[0]++; p->have = 4; }
    return p->buf[4 - p->have--];
}

/* uniform action of the fused rollout: a(t, env) = (w * A) >> 32 with w = word (t & 3) of
 * philox(key = seed ^ tag; ctr = (t >> 2 lo, t >> 2 hi, env_lo, env_hi)) - one Philox block serves four steps */
uint32_t ngwo_rollout_action(uint64_t action_seed, uint64_t env, uint64_t t, uint32_t n_actions) {
    uint32_t key[2] = {(uint32_t)action_seed, (uint32_t)(action_seed >> 32) ^ 0xA511E9B3u};
    uint64_t tb = t >> 2;
    uint32_t ctr[4] = {(uint32_t)tb, (uint32_t)(tb >> 32), (uint32_t)env, (uint32_t)(env >> 32)}, out[4];
    philox4x32_10(ctr, key, out);
    return (uint32_t)(((uint64_t)out[t & 3] * n_actions) >> 32);
}

/* ------------------------------------------------------------------ word source + numpy bounded draw */
typedef struct { ngwo_mt* mt; ngwo_philox* px; } ngwo_rng;

static uint32_t rng_next(ngwo_rng* r) { return r->mt ? ngwo_mt_next(r->mt) : philox_next(r->px); }

/* numpy legacy bounded integer in [0, max] (random_interval / buffered_bounded_masked_uint32):
 * max == 0 consumes NO word; otherwise draw 32-bit words, mask with the bit-smear of max, reject > max.
 * Used by np.random.choice(n, size=1) (max = n-1), randint(lo, hi, size=1) (max = hi-lo-1), shuffle. */
static uint32_t rng_bounded(ngwo_rng* r, uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max, v;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    while ((v = rng_next(r) & mask) > max) {}
    return v;
}

uint32_t ngwo_mt_bounded(ngwo_mt* s, uint32_t max) { ngwo_rng r = {s, 0}; return rng_bounded(&r, max); }

/* ------------------------------------------------------------------ reset */
static const int DR[4] = {-1, 1, 0, 0}, DC[4] = {0, 0, -1, 1};          /* NORTH SOUTH WEST EAST (:245-252) */
static const int TURN_LEFT[4] = {2, 3, 1, 0}, TURN_RIGHT[4] = {3, 2, 0, 1}; /* (:258-279) */

/* The reset passes of AddItem (novelty_wrappers.py:1017-1028), ReplaceItem (:1131-1144) and Fence (:871-884) share one
 * shape: np.where(<predicate>) in row-major order, np.random.shuffle of the index array (Fisher-Yates from the top),
 * percent = np.random.randint(lo, hi), then the first int(np.ceil(len * (percent / 100))) cells are edited. */
/* Philox mode of the passes NGW_PASS_SPARSE names (include/ngw.h, ngw_spec.n_passes): what np.random.shuffle + "the first
 * cnt" amounts to is a uniformly random cnt-subset of the matching cells (every chosen cell gets the same item, so the order
 * inside the subset never shows).  The device samples exactly that and this is its restatement, step for step:
 *   len     : cells with map == from (np.where over the whole map);  pct: the numpy bounded draw (a span of 1 draws nothing);
 *             cnt = ceil(len * pct / 100)
 *   draw    : min(cnt, len - cnt) distinct matching cells - the complement when that is the smaller set - by rejection:
 *             words come in whole Philox blocks from the next block boundary on; a candidate is a CELL INDEX of
 *             nb = bit_length(S*S - 1) bits, field j of word k of the block (fields from the low end, 32 / nb of them per word),
 *             in the order j = 0: words 0..3, j = 1: words 0..3, ...; a candidate beyond the map, not matching or already
 *             taken is skipped; what is left of the last block is discarded
 *   edit    : as the reference - the chosen cells get `item`, never the agent's cell (:1026 / :1142)
 * The MT mode keeps the numpy call sequence and stays pinned to the reference; the distribution of THIS mode is pinned by
 * the G6 fixtures (tests/golden/g6_*.npz: per-cell frequencies and count histograms of 10 000 reference resets). */
static void subset_pass_sparse(const ngw_spec* sp, ngwo_rng* rng, int8_t* map, int agent, int item, int from, int pct_lo, int pct_hi) {
    const int S = sp->map_size, S2 = S * S;
    int len = 0;
    for (int i = 0; i < S2; i++) len += map[i] == from;
    const int pct = pct_lo + (int)rng_bounded(rng, (uint32_t)(pct_hi - pct_lo - 1));
    const int cnt = (int)ceil((double)len * ((double)pct / 100.0));
    const int comp = 2 * cnt > len, need = comp ? len - cnt : cnt;
    ngwo_philox* px = rng->px;
    px->have = 0;                                                 /* next block boundary */
    int nb = 1;
    while ((1 << nb) < S2) nb++;                                  /* bit_length(S2 - 1) */
    const int F = 32 / nb;
    const uint32_t fm = (1u << nb) - 1u;
    for (int got = 0; got < need;) {
        uint32_t w[4];
        philox4x32_10(px->ctr, px->key, w);
        px->ctr[0]++;
        for (int j = 0; j < F; j++)
            for (int k = 0; k < 4 && got < need; k++) {
                const int cell = (int)((w[k] >> (j * nb)) & fm);
                if (cell >= S2 || map[cell] != from) continue;
                map[cell] = NGW_PASS_MARK;
                got++;
            }
    }
    for (int cell = 0; cell < S2; cell++) {
        const int marked = map[cell] == NGW_PASS_MARK;
        if (!marked && map[cell] != from) continue;
        map[cell] = (int8_t)((marked != comp && cell != agent) ? item : from);   /* chosen = marked (direct) / unmarked (complement) */
    }
}

static void subset_pass(const ngw_spec* sp, ngwo_rng* rng, int8_t* map, int agent, int kind, int item, int from, int pct_lo, int pct_hi) {
    const int S = sp->map_size;
    if (rng->px && NGW_PASS_SPARSE(kind, kind == NGW_PASS_ADDITEM ? 0 : from, sp->wall_item)) {
        subset_pass_sparse(sp, rng, map, agent, item, kind == NGW_PASS_ADDITEM ? 0 : from, pct_lo, pct_hi);
        return;
    }
    int n = 0;
    int16_t* cells = (int16_t*)malloc(sizeof(int16_t) * (size_t)(S * S));
    for (int i = 0; i < S * S; i++) {
        const int v = map[i];
        const int hit = kind == NGW_PASS_ADDITEM ? v == 0 : kind == NGW_PASS_REPLACE ? v == from : (v != 0 && v != sp->wall_item);
        if (hit) cells[n++] = (int16_t)i;
    }
    for (int i = n - 1; i >= 1; i--) {
        int j = (int)rng_bounded(rng, (uint32_t)i);
        int16_t t = cells[i]; cells[i] = cells[j]; cells[j] = t;
    }
    int pct = pct_lo + (int)rng_bounded(rng, (uint32_t)(pct_hi - pct_lo - 1));
    int cnt = (int)ceil((double)n * ((double)pct / 100.0));
    for (int i = 0; i < cnt; i++) {
        const int cell = cells[i];
        if (kind == NGW_PASS_FENCE) {                                 /* add_fence_around, pogostick_v1_env.py:524-536 */
            for (int rr = cell / S - 1; rr <= cell / S + 1; rr++)
                for (int cc = cell % S - 1; cc <= cell % S + 1; cc++)
                    if (map[rr * S + cc] == 0 && rr * S + cc != agent) map[rr * S + cc] = (int8_t)item;
        } else if (cell != agent) {                               /* :1027 / :1143 skip the agent cell */
            map[cell] = (int8_t)item;
        }
    }
    free(cells);
}

/* pogostick_v1_env.py:86-157 (+ AddItem.reset novelty_wrappers.py:1013-1034, AxeEasy.reset :29-35).
 * Returns 0, or NGW_E_PLACEMENT when the candidate list runs out ("Cannot place items, increase map size!"). */
static int reset_env(const ngw_spec* sp, ngwo_rng* rng, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv,
                     int32_t* selected, int32_t* step_count) {
    const int S = sp->map_size, K = sp->n_items;
    for (int i = 0; i < K; i++) inv[i] = 0;                       /* :119 */
    *selected = 0;                                                /* :120 selected_item = '' */
    *step_count = 0;                                              /* :124 */
    for (int r = 0; r < S; r++)                                   /* :129-130 wall ring around air */
        for (int c = 0; c < S; c++)
            map[r * S + c] = (r == 0 || c == 0 || r == S - 1 || c == S - 1) ? (int8_t)sp->wall_item : 0;
    int cap = (S > 4) ? (S - 4) * (S - 4) : 0, len = 0;
    int16_t* avail = (int16_t*)malloc(sizeof(int16_t) * (cap ? cap : 1));
    for (int r = 2; r < S - 2; r++)                               /* :136-138 row-major interior */
        for (int c = 2; c < S - 2; c++) avail[len++] = (int16_t)(r * S + c);
    if (len < 1) { free(avail); return NGW_E_PLACEMENT; }
    int agent = avail[rng_bounded(rng, (uint32_t)len - 1)];       /* :141-142 (agent cell stays in the list) */
    loc[0] = agent / S; loc[1] = agent % S;
    *facing = (int32_t)rng_bounded(rng, 3);                       /* :145 choice over 4 directions */
    for (int j = 0; j < sp->n_start; j++) {                       /* :147-148 insertion order */
        int item = sp->start_item[j], want = sp->start_qty[j], count = 0;
        while (count < want) {                                    /* add_item_to_map :159-181 */
            if (len < 1) { free(avail); return NGW_E_PLACEMENT; } /* :167 */
            int idx = (int)rng_bounded(rng, (uint32_t)len - 1);   /* :169 */
            int cell = avail[idx];
            int ok = 0;
            if (cell != agent) {                                  /* :172-174 agent cell: pop and retry */
                ok = map[cell] == 0 && map[cell - S] == 0 && map[cell + S] == 0 && map[cell - 1] == 0 &&
                     map[cell + 1] == 0;                          /* :177-178 */
                if (ok) { map[cell] = (int8_t)item; count++; }
            }
            memmove(avail + idx, avail + idx + 1, sizeof(int16_t) * (size_t)(len - idx - 1));   /* list.pop(idx) */
            len--;
        }
    }
    free(avail);
    if (sp->tap_item) {                                           /* Pogostick-v0, pogostick_v0_env.py:156-178 */
        int16_t* logs = (int16_t*)malloc(sizeof(int16_t) * (size_t)(S * S));
        int nl = 0;
        for (int i = 0; i < S * S; i++) if (map[i] == sp->tap_near) logs[nl++] = (int16_t)i;   /* np.where, row-major */
        if (nl <= 1) { free(logs); return NGW_E_PLACEMENT; }      /* assert len(result[0]) > 1 */
        for (;;) {
            const int d = (int)rng_bounded(rng, 3);               /* np.random.choice(4 directions) */
            const int cell = logs[rng_bounded(rng, (uint32_t)nl - 1)];
            const int rr = cell / S + DR[d], cc = cell % S + DC[d];
            if (rr >= 0 && rr <= S - 1 && cc >= 0 && cc <= S - 1 && map[rr * S + cc] == 0 && rr * S + cc != agent) {
                map[rr * S + cc] = (int8_t)sp->tap_item;
                break;                                            /* a tap is on the map now */
            }
        }
        free(logs);
    }
    /* stacked wrappers reset innermost first = injection order: AddItem.reset novelty_wrappers.py:1017-1028 (also Crate.reset :1071),
     * ReplaceItem.reset :1129-1148 (also FireWall.reset :1160), Fence.reset :867-889 (also FenceRestriction.reset :904) */
    for (int j = 0; j < sp->n_passes; j++)
        subset_pass(sp, rng, map, agent, sp->pass_kind[j], sp->pass_item[j], sp->pass_from[j], sp->pass_pct_lo[j], sp->pass_pct_hi[j]);
    for (int j = 0; j < sp->n_inv_start; j++) inv[sp->inv_start_item[j]] = sp->inv_start_qty[j];   /* AxeEasy.reset :33, AxetoBreakHard.reset :667-670 */
    return 0;
}

int ngwo_reset_mt(const ngw_spec* sp, ngwo_mt* mt, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv,
                  int32_t* selected, int32_t* step_count) {
    ngwo_rng r = {mt, 0};
    return reset_env(sp, &r, map, loc, facing, inv, selected, step_count);
}

int ngwo_reset_philox(const ngw_spec* sp, uint64_t seed, uint64_t env_index, uint32_t episode, int8_t* map,
                      int32_t* loc, int32_t* facing, int32_t* inv, int32_t* selected, int32_t* step_count) {
    ngwo_philox px;
    philox_reset_init(&px, seed, env_index, episode);
    ngwo_rng r = {0, &px};
    return reset_env(sp, &r, map, loc, facing, inv, selected, step_count);
}

/* ------------------------------------------------------------------ step */
/* is_block_in_front_next_to(item), :391-411: any in-bounds 4-neighbour of cell (r,c) holds `item` */
static int next_to(const ngw_spec* sp, const int8_t* map, int r, int c, int item) {
    const int S = sp->map_size;
    for (int d = 0; d < 4; d++) {
        int rr = r + DR[d], cc = c + DC[d];
        if (rr >= 0 && rr <= S - 1 && cc >= 0 && cc <= S - 1 && map[rr * S + cc] == item) return 1;
    }
    return 0;
}

/* One env, one action.  Caller guarantees 0 <= action < n_actions (the reference raises ValueError first, :236).
 * Outputs: reward, done, info word (result | done<<1 | cost<<2 | msg<<8 | arg<<16). */
void ngwo_step(const ngw_spec* sp, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv, int32_t* selected,
               int32_t* step_count, int32_t action, int32_t* reward_out, uint8_t* done_out, uint32_t* info_out) {
    const int S = sp->map_size;
    int r = loc[0], c = loc[1], f = *facing;
    int reward = sp->reward_step, result = 1, cost = 0, msg = NGW_MSG_NONE, arg = 0;   /* :239-242 */
    int fence_twice = 0;
    const int kind = sp->act_kind[action], aarg = sp->act_arg[action];
    /* block in front: cached by the reference but coherent at every read */
    const int fr = r + DR[f], fc = c + DC[f];
    const int front = map[fr * S + fc];

    switch (kind) {
    case NGW_ACT_FORWARD:                                         /* :244-257 */
        if (front == 0) { r = fr; c = fc; }
        else { result = 0; msg = NGW_MSG_BLOCK_IN_PATH; }
        cost = sp->cost_forward;
        break;
    case NGW_ACT_LEFT: f = TURN_LEFT[f]; cost = sp->cost_turn; break;      /* :258-268 */
    case NGW_ACT_RIGHT: f = TURN_RIGHT[f]; cost = sp->cost_turn; break;    /* :269-279 */
    case NGW_ACT_BREAK:                                           /* :280-294; axe: novelty_wrappers.py:144-183 */
        cost = sp->cost_break;
        if (sp->crate_item && front == sp->crate_item && !(sp->ext_flags & NGW_XF_CRATE_IN_FENCE))
            for (int i = 0; i < sp->n_items; i++) inv[i] += sp->crate_add[i];   /* Crate.step :1086-1089: the ingredients come first */
        if (sp->fence_mode && sp->breakable[front] && front != sp->fence_item) {   /* FenceRestriction.step :924-946 */
            int restricted = 0;
            if (sp->fence_mode == 1) {                            /* medium: fence beside the AGENT, across its facing */
                if (f <= 1) restricted = map[r * S + c - 1] == sp->fence_item || map[r * S + c + 1] == sp->fence_item;
                else restricted = map[(r - 1) * S + c] == sp->fence_item || map[(r + 1) * S + c] == sp->fence_item;
            } else {                                              /* hard: any fence in the 3x3 around the block in front */
                for (int rr = fr - 1; rr <= fr + 1; rr++)
                    for (int cc = fc - 1; cc <= fc + 1; cc++) restricted |= map[rr * S + cc] == sp->fence_item;
            }
            if (restricted) { result = 0; msg = NGW_MSG_FENCE_RESTRICTION; break; }
        }
        if (sp->crate_item && front == sp->crate_item && (sp->ext_flags & NGW_XF_CRATE_IN_FENCE))
            for (int i = 0; i < sp->n_items; i++) inv[i] += sp->crate_add[i];   /* Crate below FenceRestriction: only when delegated to */
        if (sp->fence_mode && sp->breakable[front]) fence_twice = 1;  /* the wrapper runs env.step() AND its own epilogue */
        if (sp->breakable[front]) {
            const int axe_ok = sp->axe_item && inv[sp->axe_item] >= 1 && *selected == sp->axe_item;
            if (axe_ok) {                                         /* axe held AND selected */
                map[fr * S + fc] = 0;
                inv[front] += sp->axe_qty;                        /* +2 with breakincrease */
                reward = sp->axe_reward;                          /* +10 for ANY block */
                cost = sp->axe_cost;                              /* 3600 * 0.5 / 0.25 */
            } else if (sp->axe_required) {                        /* AxetoBreak*: novelty_wrappers.py:589-591 */
                result = 0; msg = NGW_MSG_NEED_AXE; arg = sp->axe_item;
            } else {
                map[fr * S + fc] = 0;
                inv[front] += sp->break_qty[front];               /* 1, or 2 under BreakIncrease */
                if (!sp->axe_item) reward = sp->break_reward[front];   /* base: +10 iff tree_log; Axe env: stays -1 */
            }
        } else { result = 0; msg = NGW_MSG_CANNOT_BREAK; arg = front; }
        break;
    case NGW_ACT_CHOP:                                            /* AddChopAction.step, novelty_wrappers.py:1288-1308 */
        cost = sp->cost_chop;                                     /* 3600.0 * 1.2 */
        if (sp->breakable[front]) {
            map[fr * S + fc] = 0;
            inv[front] += 2;                                      /* 1 * 2 */
            reward = sp->chop_reward;                             /* reward_intermediate */
        } else { result = 0; msg = NGW_MSG_CANNOT_CHOP; arg = front; }
        break;
    case NGW_ACT_JUMP: {                                          /* AddJumpAction.step, :1362-1381: two cells ahead, the cell between is ignored */
        const int r2 = r + 2 * DR[f], c2 = c + 2 * DC[f];
        if (r2 >= 0 && r2 <= S - 1 && c2 >= 0 && c2 <= S - 1 && map[r2 * S + c2] == 0) { r = r2; c = c2; }
        else { result = 0; msg = NGW_MSG_BLOCK_IN_PATH; }
        cost = sp->cost_jump;                                     /* 27.906975 * 2 */
        break;
    }
    case NGW_ACT_PLACE:                                           /* :295-314 */
        if (inv[sp->place_item] >= 1) {
            if (front == 0) {
                map[fr * S + fc] = (int8_t)sp->place_item;
                inv[sp->place_item] -= 1;
                msg = NGW_MSG_PLACED; arg = sp->place_item;
                if (next_to(sp, map, fr, fc, sp->place_near)) reward = sp->place_reward;
            } else { result = 0; msg = NGW_MSG_ALREADY_EXISTS; arg = front; }
        } else { result = 0; msg = NGW_MSG_NOT_IN_INVENTORY; }
        cost = sp->cost_place;
        break;
    case NGW_ACT_EXTRACT:                                         /* :315-331 / bow_v1_env.py:293-304 */
        cost = sp->cost_extract;
        if (front == sp->ext_src) {
            if (!sp->ext_near || next_to(sp, map, fr, fc, sp->ext_near)) {
                inv[sp->ext_out] += sp->ext_qty;
                if (sp->ext_consume) map[fr * S + fc] = 0;
                reward = sp->ext_reward;
                cost = sp->ext_cost_ok;
            } else { result = 0; msg = NGW_MSG_EXTRACT_NOT_NEAR; }
        } else { result = 0; msg = NGW_MSG_EXTRACT_NO_SRC; }
        break;
    case NGW_ACT_CRAFT: {                                         /* :333-336 -> craft :413-474 */
        const int rx = aarg;
        int missing = 0;
        for (int j = 0; j < sp->recipe_n_in[rx]; j++) {           /* :422-427, dict order */
            int item = sp->recipe_in_item[rx][j];
            if (!(inv[item] >= sp->recipe_in[rx][item])) missing |= 1 << j;
        }
        if (missing) {                                            /* :430-440 */
            result = 0; msg = NGW_MSG_MISSING_ITEMS; arg = (rx << 8) | missing; cost = sp->cost_missing[rx];
        } else if (sp->recipe_needs_table[rx] && front != sp->table_item) {   /* :444-453 */
            result = 0; msg = NGW_MSG_NEED_TABLE; cost = sp->cost_no_table[rx];
        } else {                                                  /* :455-474 */
            reward = sp->recipe_reward[rx];
            for (int j = 0; j < sp->recipe_n_in[rx]; j++) {
                int item = sp->recipe_in_item[rx][j];
                inv[item] -= sp->recipe_in[rx][item];
            }
            inv[sp->recipe_out_item[rx]] += sp->recipe_out_qty[rx];
            cost = sp->cost_ok[rx];
            msg = NGW_MSG_CRAFTED; arg = sp->recipe_out_item[rx];
        }
        break;
    }
    case NGW_ACT_SELECT:                                          /* :338-347 */
        cost = sp->cost_select;
        if (inv[aarg] >= 1) *selected = aarg;
        else { result = 0; msg = NGW_MSG_NOT_IN_INVENTORY; }
        break;
    default: break;
    }
    /* grab_entities :538-554: 3x3 around the (new) agent cell incl. diagonals and own cell */
    if (sp->n_entities)
        for (int rr = r - 1; rr <= r + 1; rr++)
            for (int cc = c - 1; cc <= c + 1; cc++) {
                int id = map[rr * S + cc];
                if (id != 0 && sp->entity[id]) { map[rr * S + cc] = 0; inv[id] += 1; }
            }
    int done = 0;                                                 /* :354-357 sticky via the inventory */
    if (inv[sp->goal_item] >= 1) { reward = sp->reward_done; done = 1; }
    if (fence_twice) {                                            /* FenceRestriction.step :949-972 after env.step(): its own info */
        result = 1; cost = sp->cost_break; msg = NGW_MSG_NONE; arg = 0;   /* ... and a second step_count += 1 (:966) */
        *step_count += 1;
    }
    if (sp->fire_item && !((sp->ext_flags & NGW_XF_FIRE_SKIP_BREAK) && kind == NGW_ACT_BREAK) &&
        !(sp->fire_skip_recipe && kind == NGW_ACT_CRAFT && aarg + 1 == sp->fire_skip_recipe)) {   /* FireWall.step :1168-1187, after the wrapped step */
        if (map[(r - 1) * S + c] == sp->fire_item || map[(r + 1) * S + c] == sp->fire_item ||
            map[r * S + c - 1] == sp->fire_item || map[r * S + c + 1] == sp->fire_item) {
            reward = sp->fire_reward; done = 1; msg = NGW_MSG_FIRE_WALL; arg = 0;
        }
    }
    loc[0] = r; loc[1] = c; *facing = f;
    *step_count += 1;                                             /* :362 */
    *reward_out = reward;
    *done_out = (uint8_t)done;
    *info_out = (uint32_t)result | ((uint32_t)done << 1) | ((uint32_t)cost << 2) | ((uint32_t)msg << 8) | ((uint32_t)arg << 16);
}

/* ------------------------------------------------------------------ batched drivers (SoA as in ngw_get_state) */
/* Product autoreset convention (include/ngw.h ngw_set_autoreset, classic gym.vector "same-step" form): every call
 * steps every env; an env whose step ended with done, or whose step_count reached `horizon` (> 0), is then reset
 * (episode += 1, Philox stream of the new episode) and the call returns the NEW episode's first state together
 * with the terminal step's reward/info; done_out = 1 for both kinds of ending (info bit 1 tells goal-done). */
static uint32_t step_one(const ngw_spec* sp, int64_t i, int8_t* m, int32_t* loc, int32_t* facing, int32_t* iv,
                         int32_t* selected, int32_t* step_count, uint32_t* episode, int32_t a, int32_t* reward,
                         uint8_t* done, uint32_t* info, int autoreset, int horizon, uint64_t seed, int64_t env_index_base) {
    uint32_t flags = 0;
    if (a < 0 || a >= sp->n_actions) { *reward = 0; *done = 0; *info = 0; return NGW_F_INVALID_ACTION; }
    ngwo_step(sp, m, loc, facing, iv, selected, step_count, a, reward, done, info);
    if (autoreset && (*done || (horizon > 0 && *step_count >= horizon))) {
        *episode += 1;
        if (ngwo_reset_philox(sp, seed, (uint64_t)(env_index_base + i), *episode, m, loc, facing, iv, selected, step_count))
            flags |= NGW_F_PLACEMENT;
        *done = 1;
    }
    return flags;
}

/* actions[i] outside [0, A): env untouched, NGW_F_INVALID_ACTION raised. Returns flags. */
uint32_t ngwo_step_batch(const ngw_spec* sp, int64_t n, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv,
                         int32_t* selected, int32_t* step_count, uint32_t* episode, const int32_t* actions,
                         int32_t* reward, uint8_t* done, uint32_t* info, int autoreset, int horizon, uint64_t seed,
                         int64_t env_index_base) {
    const int S2 = sp->map_size * sp->map_size, K = sp->n_items;
    uint32_t flags = 0;
#pragma omp parallel for schedule(static) reduction(| : flags) if (n >= 512)
    for (int64_t i = 0; i < n; i++)
        flags |= step_one(sp, i, map + i * S2, loc + 2 * i, facing + i, inv + i * K, selected + i, step_count + i,
                          episode + i, actions[i], reward + i, done + i, info + i, autoreset, horizon, seed, env_index_base);
    return flags;
}

uint32_t ngwo_reset_batch(const ngw_spec* sp, int64_t n, const uint8_t* mask, int8_t* map, int32_t* loc, int32_t* facing,
                          int32_t* inv, int32_t* selected, int32_t* step_count, uint32_t* episode, uint64_t seed,
                          int64_t env_index_base) {
    const int S2 = sp->map_size * sp->map_size, K = sp->n_items;
    uint32_t flags = 0;
#pragma omp parallel for schedule(static) reduction(| : flags) if (n >= 512)
    for (int64_t i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        episode[i] += 1;
        if (ngwo_reset_philox(sp, seed, (uint64_t)(env_index_base + i), episode[i], map + i * S2, loc + 2 * i, facing + i,
                              inv + i * K, selected + i, step_count + i))
            flags |= NGW_F_PLACEMENT;
    }
    return flags;
}

/* T steps with the fused rollout's on-device uniform actions (ngw_rollout); outputs of the LAST step survive. */
uint32_t ngwo_rollout_batch(const ngw_spec* sp, int64_t n, int32_t n_steps, int64_t t0, uint64_t action_seed, int8_t* map,
                            int32_t* loc, int32_t* facing, int32_t* inv, int32_t* selected, int32_t* step_count,
                            uint32_t* episode, int32_t* reward, uint8_t* done, uint32_t* info, int autoreset, int horizon,
                            uint64_t seed, int64_t env_index_base) {
    const int S2 = sp->map_size * sp->map_size, K = sp->n_items;
    uint32_t flags = 0;
#pragma omp parallel for schedule(static) reduction(| : flags) if (n >= 512)
    for (int64_t i = 0; i < n; i++)
        for (int32_t t = 0; t < n_steps; t++) {
            int32_t a = (int32_t)ngwo_rollout_action(action_seed, (uint64_t)(env_index_base + i), (uint64_t)(t0 + t),
                                                      (uint32_t)sp->n_actions);
            flags |= step_one(sp, i, map + i * S2, loc + 2 * i, facing + i, inv + i * K, selected + i, step_count + i,
                              episode + i, a, reward + i, done + i, info + i, autoreset, horizon, seed, env_index_base);
        }
    return flags;
}

/* ------------------------------------------------------------------ LidarInFront observation */
/* observation_wrappers.py:32-80: per beam march range 1..max_range along the host-computed integer offsets; the
 * first non-air block stops the beam and, if it is a lidar item, stores the range in its channel; then the inventory. */
void ngwo_lidar(const ngw_lidar_cfg* cf, int S, int K, int64_t n, const int8_t* map, const int32_t* loc, const int32_t* facing,
                const int32_t* inv, int32_t* out) {
    const int L = cf->num_beams * cf->n_chan + cf->n_inv;
    for (int64_t i = 0; i < n; i++) {
        const int8_t* m = map + i * S * S;
        int32_t* o = out + i * L;
        const int r = loc[2 * i], c = loc[2 * i + 1], f = facing[i];
        for (int j = 0; j < L; j++) o[j] = 0;
        for (int b = 0; b < cf->num_beams; b++)
            for (int k = 1; k <= cf->max_range; k++) {                    /* :52 */
                const int rr = r + cf->dr[f][b][k - 1], cc = c + cf->dc[f][b][k - 1];
                const int id = m[rr * S + cc];                            /* :56 */
                if (id != 0) {                                            /* :59-64 */
                    if (cf->chan_of_item[id]) o[b * cf->n_chan + cf->chan_of_item[id] - 1] = k;
                    break;
                }
            }
        for (int j = 0; j < cf->n_inv; j++) o[cf->num_beams * cf->n_chan + j] = inv[i * K + cf->inv_item[j]];   /* :74-75 */
    }
}
int ngwo_lidar_cfg_size(void) { return (int)sizeof(ngw_lidar_cfg); }

#ifdef _OPENMP
#include <omp.h>
int ngwo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int ngwo_set_threads(int n) { (void)n; return 1; }
#endif
int ngwo_spec_size(void) { return (int)sizeof(ngw_spec); }
int ngwo_mt_size(void) { return (int)sizeof(ngwo_mt); }
