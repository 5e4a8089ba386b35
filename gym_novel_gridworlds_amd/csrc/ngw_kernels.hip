// ngw_kernels.hip — CDNA4 (gfx950) kernels of the batched step()/reset() hot path.
//
// One 64-lane wavefront = one workgroup = 64 environments, ONE LANE PER ENV.  Per launch each wave
//   1. stages its 64 tile maps (64*S*S contiguous bytes) HBM -> LDS with coalesced 16-B loads, and the
//      per-env scalars / inventory rows into registers / LDS,
//   2. runs the table-driven step (and, for envs whose episode ended, the reset) per lane on LDS,
//   3. writes the NEW state = the batched observation (map i8 [N,S,S], agent_location i32 [N,2],
//      agent_facing_id i32 [N], inventory i32 [N,K]) to the OTHER buffer of a ping-pong pair with
//      coalesced 16-B stores, plus reward / done / packed info.
// The observation buffers ARE the state, so a step moves 2*S*S + 8*K + ~60 bytes per env and nothing else
// (SURVEY.md §8(d) prices 2*S*S + 12*K + 45).  Integer/byte work only: no MFMA.
// Every global load of a phase is issued before its first consumer so a phase costs ONE memory round trip.
//
// Semantics follow the reference line by line (citations at each branch):
//   gym_novel_gridworlds/envs/pogostick_v1_env.py  reset :86-181, step :230-367, craft :413-474, grab :538-554
//   gym_novel_gridworlds/envs/bow_v1_env.py        Extract_string :293-304, craft :386-441
//   gym_novel_gridworlds/novelty_wrappers.py       AxeEasy :9-114, AxeMedium :117-213, AddItem :991-1034
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ngw.h"
#include "ngw_device.h"

static_assert(sizeof(NgwDevSpec) % 4 == 0, "the spec blob is copied to LDS by dwords");

namespace {

constexpr int EPB = NGW_EPB;   // envs per block = wavefront width

// ---------------------------------------------------------------- Philox4x32-10 (counter-based, per env & episode)
struct Philox {
    uint32_t k0, k1, c0, c1, c2, c3;
    uint32_t w0, w1, w2, w3;
    int have;
};

__device__ __forceinline__ void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                             uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

__device__ __forceinline__ void philox_init(Philox& p, uint64_t seed, uint64_t env, uint32_t episode) {
    p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32);
    p.c0 = 0; p.c1 = episode; p.c2 = (uint32_t)env; p.c3 = (uint32_t)(env >> 32);
    p.have = 0;
}

__device__ __forceinline__ uint32_t philox_next(Philox& p) {
    if (p.have == 0) {
        philox_block(p.c0, p.c1, p.c2, p.c3, p.k0, p.k1, p.w0, p.w1, p.w2, p.w3);
        p.c0++;
        p.have = 4;
    }
    uint32_t r = p.w0;
    p.w0 = p.w1; p.w1 = p.w2; p.w2 = p.w3;
    p.have--;
    return r;
}

// numpy legacy bounded draw in [0, max]: max == 0 consumes no word; else mask & reject (random_interval).
__device__ __forceinline__ uint32_t bounded(Philox& p, uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = 0xFFFFFFFFu >> __clz((int)max);
    uint32_t v;
    do { v = philox_next(p) & mask; } while (v > max);
    return v;
}

// index of the n-th (0-based) set bit of x; caller guarantees n < popc(x)
__device__ __forceinline__ int nth_set_bit(uint32_t x, int n) {
    int bit = 0, lo;
    lo = __popc(x & 0xFFFFu); if (n >= lo) { n -= lo; x >>= 16; bit += 16; }
    lo = __popc(x & 0xFFu);   if (n >= lo) { n -= lo; x >>= 8;  bit += 8; }
    lo = __popc(x & 0xFu);    if (n >= lo) { n -= lo; x >>= 4;  bit += 4; }
    lo = __popc(x & 0x3u);    if (n >= lo) { n -= lo; x >>= 2;  bit += 2; }
    lo = (int)(x & 1u);       if (n >= lo) { bit += 1; }
    return bit;
}

// ---------------------------------------------------------------- per-lane reset on the LDS map
// pogostick_v1_env.py:86-157 + add_item_to_map :159-181 (+ AddItem.reset, AxeEasy.reset).  `mp` = this lane's map
// in LDS, `inv` = this lane's inventory row, `cand` = candidate bitmask column (stride EPB).
__device__ __forceinline__ uint32_t reset_lane(const ngw_spec& sp, const double* addq, const NgwLaunch& a, int8_t* mp, int32_t* inv,
                                            uint32_t* cand, uint64_t env_global, int64_t env_local, uint32_t episode,
                                            int& r_out, int& c_out, int& f_out) {
    const int S = a.S, K = a.K, W = S - 4, ncand = W * W;
    Philox px;
    philox_init(px, a.seed, env_global, episode);
    for (int k = 0; k < K; k++) inv[k] = 0;                                        // :119
    for (int r = 0; r < S; r++)                                                    // :129-130 wall ring around air
        for (int c = 0; c < S; c++)
            mp[r * S + c] = (r == 0 || c == 0 || r == S - 1 || c == S - 1) ? (int8_t)sp.wall_item : (int8_t)0;
    for (int w = 0; w < a.CW; w++) {                                               // :136-138 all interior candidates
        int left = ncand - w * 32;
        cand[w * EPB] = left >= 32 ? 0xFFFFFFFFu : (left > 0 ? ((1u << left) - 1u) : 0u);
    }
    int len = ncand;
    uint32_t flags = 0;
    int apos = (int)bounded(px, (uint32_t)len - 1);                                // :141 (agent stays in the list)
    const int agent = (2 + apos / W) * S + 2 + apos % W;
    r_out = agent / S; c_out = agent % S;
    f_out = (int)bounded(px, 3);                                                   // :145
    for (int j = 0; j < sp.n_start; j++) {                                         // :147-148 insertion order
        const int item = sp.start_item[j], want = sp.start_qty[j];
        int count = 0;
        while (count < want) {                                                     // add_item_to_map
            if (len < 1) { flags |= NGW_F_PLACEMENT; break; }                     // :167
            int idx = (int)bounded(px, (uint32_t)len - 1);                         // :169
            int w = 0, pc;
            while (idx >= (pc = __popc(cand[w * EPB]))) { idx -= pc; w++; }         // idx-th remaining, row-major
            const int bit = nth_set_bit(cand[w * EPB], idx);
            cand[w * EPB] &= ~(1u << bit);                                         // list.pop(idx)
            len--;
            const int pos = w * 32 + bit;
            const int cell = (2 + pos / W) * S + 2 + pos % W;
            if (cell != agent &&                                                   // :172-174
                mp[cell] == 0 && mp[cell - S] == 0 && mp[cell + S] == 0 && mp[cell - 1] == 0 && mp[cell + 1] == 0) {
                mp[cell] = (int8_t)item;                                           // :177-180
                count++;
            }
        }
        if (flags) break;
    }
    if (sp.additem_item && !flags) {                                               // AddItem.reset novelty_wrappers.py:1017-1028
        uint16_t* perm = a.b.perm + env_local;                                     // [S2][n_pad] scratch column
        const int64_t ps = a.n_pad;
        int n_air = 0;
        for (int i = 0; i < a.S2; i++)
            if (mp[i] == 0) { perm[(int64_t)n_air * ps] = (uint16_t)i; n_air++; }   // np.where(map == 0)
        for (int i = n_air - 1; i >= 1; i--) {                                     // np.random.shuffle
            int j = (int)bounded(px, (uint32_t)i);
            uint16_t x = perm[(int64_t)i * ps], y = perm[(int64_t)j * ps];
            perm[(int64_t)i * ps] = y; perm[(int64_t)j * ps] = x;
        }
        const int pct = (int)bounded(px, (uint32_t)(sp.additem_pct_hi - sp.additem_pct_lo - 1));   // randint(lo, hi)
        const int cnt = (int)ceil((double)n_air * addq[pct]);                      // int(np.ceil(len * (pct / 100)))
        for (int i = 0; i < cnt; i++) {
            int cell = perm[(int64_t)i * ps];
            if (cell != agent) mp[cell] = (int8_t)sp.additem_item;                 // :1027
        }
    }
    if (sp.inv_start_item && !flags) inv[sp.inv_start_item] = sp.inv_start_qty;         // AxeEasy.reset :33
    return flags;
}

// ---------------------------------------------------------------- map staging HBM <-> LDS (coalesced 16-B pieces)
// The wave's 64 maps are one contiguous 64*S2-byte chunk in HBM = 4*S2 pieces of 16 B; lane l owns pieces
// l, l+64, ...  A round moves PB pieces per lane: ALL its global loads are issued before the first LDS write (and
// all LDS reads before the first global store), so a round costs one memory round trip, not PB.
// In LDS each env's map starts at e*MS bytes with MS/4 odd, so 64 lanes reading "their" cell hit distinct banks.
constexpr int PB = 8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));                       // native vector: stays in VGPRs

// Loads are UNCONDITIONAL on a clamped index (a duplicate in-bounds load is harmless and keeps the values in plain
// registers); only stores and LDS writes are predicated.
__device__ __forceinline__ void pieces_load(u32x4 (&buf)[PB], const u32x4* g4, int base, int npieces, int tid) {
#pragma unroll
    for (int j = 0; j < PB; j++) buf[j] = g4[min(base + tid + EPB * j, npieces - 1)];
}

__device__ __forceinline__ void pieces_store(const u32x4 (&buf)[PB], u32x4* g4, int base, int npieces, int tid) {
#pragma unroll
    for (int j = 0; j < PB; j++) {
        const int p = base + tid + EPB * j;
        if (p < npieces) g4[p] = buf[j];
    }
}

// LDS <-> register pieces.  TO_LDS: buf -> lds, else lds -> buf.
template <bool TO_LDS, int MAPMODE>
__device__ __forceinline__ void pieces_lds(u32x4 (&buf)[PB], const NgwLaunch& a, uint32_t* lds_map, int base, int npieces, int tid) {
    if (MAPMODE == NGW_MAP_STRAIGHT) {                                             // LDS image == HBM image
        u32x4* l4 = reinterpret_cast<u32x4*>(lds_map);
#pragma unroll
        for (int j = 0; j < PB; j++) {
            const int p = base + tid + EPB * j;
            if (TO_LDS) { if (p < npieces) l4[p] = buf[j]; } else buf[j] = l4[min(p, npieces - 1)];
        }
    } else if (MAPMODE == NGW_MAP_DWORD) {                                         // dword granularity, padded stride
        const uint32_t S2dw = (uint32_t)a.S2 >> 2, MSdw = (uint32_t)a.MS >> 2;
#pragma unroll
        for (int j = 0; j < PB; j++) {
            const int p0 = base + tid + EPB * j;
            const int p = TO_LDS ? p0 : min(p0, npieces - 1);
            if (!TO_LDS || p0 < npieces) {
                const uint32_t d = (uint32_t)p * 4u;                               // dword offset inside the chunk
                uint32_t e = __umulhi(d, a.magic);                                 // d / S2dw (exact, see ngw_abi.cpp)
                uint32_t o = d - e * S2dw;
                uint32_t v[4] = {buf[j].x, buf[j].y, buf[j].z, buf[j].w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (o >= S2dw) { o = 0; e++; }
                    uint32_t* cell = lds_map + e * MSdw + o;
                    if (TO_LDS) *cell = v[q]; else v[q] = *cell;
                    o++;
                }
                if (!TO_LDS) buf[j] = u32x4{v[0], v[1], v[2], v[3]};
            }
        }
    } else {                                                                       // odd S: byte granularity
        uint8_t* lb = reinterpret_cast<uint8_t*>(lds_map);
#pragma unroll
        for (int j = 0; j < PB; j++) {
            const int p0 = base + tid + EPB * j;
            const int p = TO_LDS ? p0 : min(p0, npieces - 1);
            if (!TO_LDS || p0 < npieces) {
                const uint32_t g = (uint32_t)p * 16u;
                uint32_t e = __umulhi(g, a.magic);                                 // g / S2
                uint32_t o = g - e * (uint32_t)a.S2;
                uint32_t v[4] = {buf[j].x, buf[j].y, buf[j].z, buf[j].w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t w = TO_LDS ? v[q] : 0u;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        if (o >= (uint32_t)a.S2) { o = 0; e++; }
                        uint8_t* cell = lb + e * (uint32_t)a.MS + o;
                        if (TO_LDS) *cell = (uint8_t)(w >> (8 * b)); else w |= (uint32_t)*cell << (8 * b);
                        o++;
                    }
                    v[q] = w;
                }
                if (!TO_LDS) buf[j] = u32x4{v[0], v[1], v[2], v[3]};
            }
        }
    }
}

// Inventory rows: the wave's 64 rows are one contiguous 64*K-dword chunk [e][K] = 16*K quads of 16 B; lane l owns
// quads l, l+64, ...  LDS layout [e][KP] with KP = K|1 (odd stride -> per-lane item reads are conflict-free).
constexpr int IQ = (NGW_MAX_ITEMS + 3) / 4;                                        // quads per lane, K <= 24

template <bool TO_LDS>
__device__ __forceinline__ void inv_lds(u32x4 (&q)[IQ], const NgwLaunch& a, int32_t* lds_inv, int tid) {
    const int nq = 16 * a.K;
    if (a.KP == a.K) {                                                             // K odd: LDS image == HBM image
        u32x4* l4 = reinterpret_cast<u32x4*>(lds_inv);
#pragma unroll
        for (int j = 0; j < IQ; j++) {
            const int p = tid + EPB * j;
            if (TO_LDS) { if (p < nq) l4[p] = q[j]; } else q[j] = l4[min(p, nq - 1)];
        }
    } else {                                                                       // K even: one pad dword per env
#pragma unroll
        for (int j = 0; j < IQ; j++) {
            const int p0 = tid + EPB * j;
            const int p = TO_LDS ? p0 : min(p0, nq - 1);
            if (!TO_LDS || p0 < nq) {
                const uint32_t d = (uint32_t)p * 4u;
                uint32_t e = __umulhi(d, a.magicK);                                // d / K
                uint32_t o = d - e * (uint32_t)a.K;
                uint32_t v[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (o >= (uint32_t)a.K) { o = 0; e++; }
                    int32_t* cell = lds_inv + e * (uint32_t)a.KP + o;
                    if (TO_LDS) *cell = (int32_t)v[i]; else v[i] = (uint32_t)*cell;
                    o++;
                }
                if (!TO_LDS) q[j] = u32x4{v[0], v[1], v[2], v[3]};
            }
        }
    }
}

// ---------------------------------------------------------------- the kernel
template <int MAPMODE>
__global__ void __launch_bounds__(NGW_EPB) ngw_kernel(const NgwDevSpec* __restrict__ dspec, const NgwLaunch a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (a.mode == NGW_MODE_DBG_NOP) return;
    const int tid = threadIdx.x;
    const int64_t env0 = (int64_t)blockIdx.x * EPB;
    const int64_t e = env0 + tid;                                                  // local env index of this lane
    const bool live = e < a.n;
    const int S = a.S, K = a.K;
    const int npieces = 4 * a.S2;                                                  // EPB * S2 / 16

    // LDS carve-up (dword offsets): maps | inventory [64][KP] | candidate masks [CW][64] | LUT copy of the spec
    uint32_t* lds_map = lds;
    int32_t* lds_inv = reinterpret_cast<int32_t*>(lds + a.off_inv);
    uint32_t* lds_cand = lds + a.off_cand;
    const NgwDevSpec& ds = *reinterpret_cast<const NgwDevSpec*>(lds + a.off_spec);
    const ngw_spec& sp = ds.sp;
    int8_t* mp = reinterpret_cast<int8_t*>(lds_map) + tid * a.MS;                 // this lane's map
    int32_t* inv = lds_inv + tid * a.KP;                                           // this lane's inventory row
    uint32_t* cand = lds_cand + tid;

    int cur = a.cur;
    // ---- issue EVERY global load of the prologue before touching LDS: LUT blob, first map round, scalars, inventory
    constexpr int NSPEC = (int)(sizeof(NgwDevSpec) / 4);
    static_assert(NSPEC <= 4 * EPB, "spec blob is loaded with 4 dwords per lane");
    uint32_t sv[4];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(dspec);
#pragma unroll
        for (int j = 0; j < 4; j++) sv[j] = src[min(tid + EPB * j, NSPEC - 1)];
    }
    u32x4 buf[PB];
    const u32x4* gin = reinterpret_cast<const u32x4*>(a.b.map[cur] + env0 * a.S2);
    pieces_load(buf, gin, 0, npieces, tid);
    int r = 1, c = 1, f = 0, sel = 0, steps = 0, action = 0;
    uint32_t episode = 0;
    if (live) {
        const int2 rc = reinterpret_cast<const int2*>(a.b.loc[cur])[e];
        r = rc.x; c = rc.y;
        f = a.b.facing[cur][e];
        sel = a.b.selected[e];
        steps = a.b.step_count[e];
        episode = a.b.episode[e];
        if (a.mode == NGW_MODE_STEP) action = a.actions[e];
        else if (a.mode == NGW_MODE_RESET) action = a.reset_mask ? (int)a.reset_mask[e] : 1;
    }
    u32x4 iq[IQ];
    {
        const u32x4* gi = reinterpret_cast<const u32x4*>(a.b.inv[cur] + env0 * K);
#pragma unroll
        for (int j = 0; j < IQ; j++) iq[j] = gi[min(tid + EPB * j, 16 * K - 1)];
    }
    // ---- land them in LDS
    {
        uint32_t* dst = lds + a.off_spec;
#pragma unroll
        for (int j = 0; j < 4; j++) { const int i = tid + EPB * j; if (i < NSPEC) dst[i] = sv[j]; }
    }
    pieces_lds<true, MAPMODE>(buf, a, lds_map, 0, npieces, tid);
    for (int base = EPB * PB; base < npieces; base += EPB * PB) {                  // big maps: further rounds
        pieces_load(buf, gin, base, npieces, tid);
        pieces_lds<true, MAPMODE>(buf, a, lds_map, base, npieces, tid);
    }
    inv_lds<true>(iq, a, lds_inv, tid);
    __syncthreads();

    uint32_t flags = 0;
    int reward = 0, ended = 0;
    uint32_t info = 0;
    uint32_t aw0 = 0, aw1 = 0, aw2 = 0, aw3 = 0;                                   // rollout: 4 actions per Philox block

    for (int t = 0; t < a.n_steps; t++) {
        if (live && a.mode != NGW_MODE_DBG_COPY) {
            bool do_reset = false;
            if (a.mode == NGW_MODE_RESET) {
                do_reset = action != 0;
            } else {
                if (a.mode == NGW_MODE_ROLLOUT) {
                    // action(t, env) = (w * A) >> 32 with w = word (t & 3) of philox(key = action_seed ^ tag; ctr = (t >> 2, env))
                    const uint64_t tt = (uint64_t)(a.t0 + t), eg = (uint64_t)(a.env_base + e);
                    if (t == 0 || (tt & 3) == 0) {
                        const uint64_t tb = tt >> 2;
                        philox_block((uint32_t)tb, (uint32_t)(tb >> 32), (uint32_t)eg, (uint32_t)(eg >> 32),
                                     (uint32_t)a.action_seed, (uint32_t)(a.action_seed >> 32) ^ 0xA511E9B3u, aw0, aw1, aw2, aw3);
                    }
                    const uint32_t q = (uint32_t)tt & 3u;
                    const uint32_t w = q == 0 ? aw0 : (q == 1 ? aw1 : (q == 2 ? aw2 : aw3));
                    action = (int)__umulhi(w, (uint32_t)sp.n_actions);
                }
                if (action < 0 || action >= sp.n_actions) {                        // reference: ValueError before any change (:236)
                    flags |= NGW_F_INVALID_ACTION;
                    reward = 0; ended = 0; info = 0;
                } else {
                    int rew = sp.reward_step, result = 1, cost = 0, msg = NGW_MSG_NONE, arg = 0;   // :239-242
                    const int kind = sp.act_kind[action], aarg = sp.act_arg[action];
                    const int dr = (f == 0) ? -1 : (f == 1 ? 1 : 0), dc = (f == 2) ? -1 : (f == 3 ? 1 : 0);
                    const int fr = r + dr, fc = c + dc, fcell = fr * S + fc;
                    const int front = mp[fcell];                                   // block in front (:369-389)
                    switch (kind) {
                    case NGW_ACT_FORWARD:                                          // :244-257
                        if (front == 0) { r = fr; c = fc; } else { result = 0; msg = NGW_MSG_BLOCK_IN_PATH; }
                        cost = sp.cost_forward;
                        break;
                    case NGW_ACT_LEFT:                                             // :258-268  N->W S->E W->S E->N
                        f = (0x0132 >> (f * 4)) & 3; cost = sp.cost_turn;
                        break;
                    case NGW_ACT_RIGHT:                                            // :269-279  N->E S->W W->N E->S
                        f = (0x1023 >> (f * 4)) & 3; cost = sp.cost_turn;
                        break;
                    case NGW_ACT_BREAK:                                            // :280-294, axe: novelty_wrappers.py:144-183
                        cost = sp.cost_break;
                        if (sp.breakable[front]) {
                            mp[fcell] = 0;
                            if (sp.axe_item && inv[sp.axe_item] >= 1 && sel == sp.axe_item) {
                                inv[front] += sp.axe_qty; rew = sp.axe_reward; cost = sp.axe_cost;
                            } else {
                                inv[front] += 1;
                                if (!sp.axe_item) rew = sp.break_reward[front];
                            }
                        } else { result = 0; msg = NGW_MSG_CANNOT_BREAK; arg = front; }
                        break;
                    case NGW_ACT_PLACE:                                            // :295-314
                        if (inv[sp.place_item] >= 1) {
                            if (front == 0) {
                                mp[fcell] = (int8_t)sp.place_item;
                                inv[sp.place_item] -= 1;
                                msg = NGW_MSG_PLACED; arg = sp.place_item;
                                // is_block_in_front_next_to(tree_log) :391-411, bounds-checked 4-neighbourhood
                                const int nr = sp.place_near;
                                bool near = (fr > 0 && mp[fcell - S] == nr) || (fr < S - 1 && mp[fcell + S] == nr) ||
                                            (fc > 0 && mp[fcell - 1] == nr) || (fc < S - 1 && mp[fcell + 1] == nr);
                                if (near) rew = sp.place_reward;
                            } else { result = 0; msg = NGW_MSG_ALREADY_EXISTS; arg = front; }
                        } else { result = 0; msg = NGW_MSG_NOT_IN_INVENTORY; }
                        cost = sp.cost_place;
                        break;
                    case NGW_ACT_EXTRACT:                                          // :315-331 / bow_v1_env.py:293-304
                        cost = sp.cost_extract;
                        if (front == sp.ext_src) {
                            const int nr = sp.ext_near;
                            bool near = !nr || (fr > 0 && mp[fcell - S] == nr) || (fr < S - 1 && mp[fcell + S] == nr) ||
                                        (fc > 0 && mp[fcell - 1] == nr) || (fc < S - 1 && mp[fcell + 1] == nr);
                            if (near) {
                                inv[sp.ext_out] += sp.ext_qty;
                                if (sp.ext_consume) mp[fcell] = 0;
                                rew = sp.ext_reward; cost = sp.ext_cost_ok;
                            } else { result = 0; msg = NGW_MSG_EXTRACT_NOT_NEAR; }
                        } else { result = 0; msg = NGW_MSG_EXTRACT_NO_SRC; }
                        break;
                    case NGW_ACT_CRAFT: {                                          // craft :413-474
                        const int rx = aarg, nin = sp.recipe_n_in[rx];
                        int missing = 0;
                        for (int j = 0; j < nin; j++) {                            // :422-427, dict order
                            const int item = sp.recipe_in_item[rx][j];
                            if (!(inv[item] >= (int)sp.recipe_in[rx][item])) missing |= 1 << j;
                        }
                        if (missing) {                                             // :430-440
                            result = 0; msg = NGW_MSG_MISSING_ITEMS; arg = (rx << 8) | missing; cost = sp.cost_missing[rx];
                        } else if (sp.recipe_needs_table[rx] && front != sp.table_item) {   // :444-453
                            result = 0; msg = NGW_MSG_NEED_TABLE; cost = sp.cost_no_table[rx];
                        } else {                                                   // :455-474
                            rew = sp.craft_reward;
                            for (int j = 0; j < nin; j++) {
                                const int item = sp.recipe_in_item[rx][j];
                                inv[item] -= (int)sp.recipe_in[rx][item];
                            }
                            inv[sp.recipe_out_item[rx]] += sp.recipe_out_qty[rx];
                            cost = sp.cost_ok[rx]; msg = NGW_MSG_CRAFTED; arg = sp.recipe_out_item[rx];
                        }
                        break;
                    }
                    case NGW_ACT_SELECT:                                           // :338-347
                        cost = sp.cost_select;
                        if (inv[aarg] >= 1) sel = aarg; else { result = 0; msg = NGW_MSG_NOT_IN_INVENTORY; }
                        break;
                    default: break;
                    }
                    if (sp.n_entities) {                                           // grab_entities :538-554 (3x3 incl. own cell)
                        for (int rr = r - 1; rr <= r + 1; rr++)
                            for (int cc = c - 1; cc <= c + 1; cc++) {
                                const int id = mp[rr * S + cc];
                                if (id != 0 && sp.entity[id]) { mp[rr * S + cc] = 0; inv[id] += 1; }
                            }
                    }
                    int done = 0;                                                  // :354-357
                    if (inv[sp.goal_item] >= 1) { rew = sp.reward_done; done = 1; }
                    steps += 1;                                                    // :362
                    reward = rew; ended = done;
                    info = (uint32_t)result | ((uint32_t)done << 1) | ((uint32_t)cost << 2) | ((uint32_t)msg << 8) |
                           ((uint32_t)arg << 16);
                    if (a.autoreset && (done || (a.horizon > 0 && steps >= a.horizon))) {   // same-step autoreset
                        do_reset = true; ended = 1;
                    }
                }
            }
            if (do_reset) {                                                        // single call site: the body is large
                episode++;
                flags |= reset_lane(sp, ds.addq, a, mp, inv, cand, (uint64_t)(a.env_base + e), e, episode, r, c, f);
                sel = 0; steps = 0;
            }
        }
        __syncthreads();
        // ---- write the new state == the observation into the other buffer: all LDS reads, then all stores
        const int nxt = cur ^ 1;
        u32x4* gout = reinterpret_cast<u32x4*>(a.b.map[nxt] + env0 * a.S2);
        for (int base = 0; base < npieces; base += EPB * PB) {
            pieces_lds<false, MAPMODE>(buf, a, lds_map, base, npieces, tid);
            pieces_store(buf, gout, base, npieces, tid);
        }
        {
            u32x4* go = reinterpret_cast<u32x4*>(a.b.inv[nxt] + env0 * K);
            inv_lds<false>(iq, a, lds_inv, tid);
#pragma unroll
            for (int j = 0; j < IQ; j++) { const int p = tid + EPB * j; if (p < 16 * K) go[p] = iq[j]; }
        }
        if (live) {
            reinterpret_cast<int2*>(a.b.loc[nxt])[e] = make_int2(r, c);
            a.b.facing[nxt][e] = f;
            if (a.mode != NGW_MODE_RESET) {
                a.b.reward[e] = reward;
                a.b.done[e] = (uint8_t)ended;
                a.b.info[e] = info;
            }
        }
        cur = nxt;
        __syncthreads();
    }
    if (live) {
        a.b.selected[e] = (uint8_t)sel;
        a.b.step_count[e] = steps;
        a.b.episode[e] = episode;
    }
    if (flags) atomicOr(a.b.flags, flags);
}

}  // namespace

template <int MAPMODE>
static hipError_t launch_mode(const NgwDevSpec* dspec, const NgwLaunch* a, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    // CDNA4 has 160 KiB of LDS per CU; anything above the 64 KiB default needs an explicit opt-in per device.
    static size_t lds_opt_in[64] = {0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (lds_bytes > 64 * 1024 && dev < 64 && lds_bytes > lds_opt_in[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(ngw_kernel<MAPMODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
        if (e != hipSuccess) return e;
        lds_opt_in[dev] = lds_bytes;
    }
    hipLaunchKernelGGL(ngw_kernel<MAPMODE>, dim3(grid), dim3(NGW_EPB), lds_bytes, stream, dspec, *a);
    return hipGetLastError();
}

extern "C" hipError_t ngw_launch(const NgwDevSpec* dspec, const NgwLaunch* a, int map_mode, unsigned grid, size_t lds_bytes,
                                 hipStream_t stream) {
    switch (map_mode) {
    case NGW_MAP_STRAIGHT: return launch_mode<NGW_MAP_STRAIGHT>(dspec, a, grid, lds_bytes, stream);
    case NGW_MAP_DWORD: return launch_mode<NGW_MAP_DWORD>(dspec, a, grid, lds_bytes, stream);
    default: return launch_mode<NGW_MAP_BYTE>(dspec, a, grid, lds_bytes, stream);
    }
}
