// ngw_kernels.hip — CDNA4 (gfx950) kernels of the batched step()/reset() hot path.
//
// One 64-lane wavefront = one workgroup = 64 environments, ONE LANE PER ENV.  Per launch each wave
//   1. stages its 64 tile maps (64*S*S contiguous bytes) HBM -> LDS with coalesced 16-B loads, and the
//      per-env scalars / inventory rows into registers / LDS,
//   2. runs the table-driven step (and, for envs whose episode ended, the reset) per lane on LDS,
//   3. writes THROUGH to HBM only what the step changed - the broken / placed map cell, the touched inventory
//      slots, the agent pose - plus reward / done / packed info; a reset rewrites the wave's whole chunk with
//      coalesced 16-B stores.
// The observation buffers (map i8 [N,S,S], agent_location i32 [N,2], agent_facing_id i32 [N], inventory i32 [N,K])
// ARE the state and are updated in place, so a step reads S*S + 4*K + ~30 bytes per env and writes ~30
// (SURVEY.md §8(d) prices 2*S*S + 12*K + 45 for a read-pack-write design).  Integer/byte work only: no MFMA.
// Every global load of a phase is issued before its first consumer so a phase costs ONE memory round trip.
//
// Semantics follow the reference line by line (citations at each branch):
//   gym_novel_gridworlds/envs/pogostick_v1_env.py  reset :86-181, step :230-367, craft :413-474, grab :538-554
//   gym_novel_gridworlds/envs/bow_v1_env.py        Extract_string :293-304, craft :386-441
//   gym_novel_gridworlds/novelty_wrappers.py       AxeEasy :9-114, AxeMedium :117-213, AddItem :991-1034,
//                                                  AxetoBreak :439-625, AddChop :1267-1337, AddJump :1340-1412,
//                                                  BreakIncrease :1415-1488, ExtractIncDec :1491-1581
//   gym_novel_gridworlds/envs/pogostick_v0_env.py  tree_tap reset pass :156-178
//   gym_novel_gridworlds/observation_wrappers.py   LidarInFront :10-80 (ngw_lidar_kernel and the fused epilogue)
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/ngw.h"
#include "ngw_device.h"

static_assert(NGW_MAX_PASSES == 4, "ResetArgs carries four pass words");
static_assert(sizeof(NgwDevSpec) % 4 == 0, "the spec blob is copied to LDS by dwords");

namespace {

// Keep a value in a register across the step loop: the empty asm makes it opaque, so the compiler can neither
// re-load it from the kernarg segment / HBM with s_load inside the loop nor recompute it (worst case it parks it in
// a VGPR lane, one v_readlane to bring it back - no memory wait).
#define PIN_S(x) asm volatile("" : "+s"(x))
#define PIN_V(x) asm volatile("" : "+v"(x))
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

constexpr int EPB = NGW_EPB;   // envs per block = wavefront width

// In-kernel timeline stamps (diagnostics build only: make stamps -> libngw_hip_stamps.so; tools/stamp_timeline.py).
// s_memrealtime = the chip-wide 100 MHz clock (aligns waves of different XCDs), s_memtime = shader cycles.
#ifdef NGW_STAMPS
#define STAMP_DECL uint64_t st_rt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_cy[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); st_rt[i] = __builtin_amdgcn_s_memrealtime(); st_cy[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_FLUSH(a) do { if ((a).stamps && threadIdx.x == 0) { for (int i_ = 0; i_ < 8; i_++) { (a).stamps[(size_t)blockIdx.x * 16 + i_] = st_rt[i_]; (a).stamps[(size_t)blockIdx.x * 16 + 8 + i_] = st_cy[i_]; } } } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH(a)
#endif

// ---------------------------------------------------------------- Philox4x32-10 (counter-based, per env & episode)
// Two word sources with one interface (same stream: block b = counter (b, episode, env_lo, env_hi) yields words 4b .. 4b+3).
//
// PhiloxRing: a per-lane ring of PHILOX_RING words in LDS, filled PHILOX_RING / 4 blocks at a time.  With a 4-word register
// buffer the 64 lanes run dry at different draws, so the wave executes the 10-round block for nearly EVERY draw; all lanes
// start together and a plain reset needs < 32 words, so with the ring the block code runs once per reset for most waves.
//
// PhiloxRegs: one block at a time in registers.  For resets with a shuffled-subset pass (hundreds of draws per lane - the
// lanes run dry at different draws whatever the buffer, and a 32-word refill per lane would execute the 8-block burst
// 64 times over: C5 50 -> 151 us per step), and wherever the ring's 8 KB of LDS would cost a resident wave per CU.
constexpr int PHILOX_RING = 32;

__device__ __forceinline__ void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                             uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_hi / mul_lo pair: integer multiplies are
        // quarter-rate, they are what a block costs
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

struct PhiloxRing {
    uint32_t k0, k1, c0, c1, c2, c3;
    LDS_AS uint32_t* ring;                                                          // word j of this lane at ring[j * EPB]
    int pos;
    uint32_t nxt;                                                                   // ring[pos], requested one draw ahead

    __device__ __forceinline__ void fill() {
        for (int b = 0; b < PHILOX_RING / 4; b++) {
            uint32_t w0, w1, w2, w3;
            philox_block(c0, c1, c2, c3, k0, k1, w0, w1, w2, w3);
            c0++;
            ring[(4 * b) * EPB] = w0; ring[(4 * b + 1) * EPB] = w1; ring[(4 * b + 2) * EPB] = w2; ring[(4 * b + 3) * EPB] = w3;
        }
        pos = 0;
        nxt = ring[0];
    }
    __device__ __forceinline__ void init(uint64_t seed, uint64_t env, uint32_t episode, LDS_AS uint32_t* ring_) {
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
        c0 = 0; c1 = episode; c2 = (uint32_t)env; c3 = (uint32_t)(env >> 32);
        ring = ring_;
        fill();
    }
    __device__ __forceinline__ uint32_t next() {
        const uint32_t r = nxt;
        pos++;
        if (pos == PHILOX_RING) fill();                                            // (eager: the stream itself is unchanged)
        else nxt = ring[pos * EPB];                                                // lands while the caller works on r
        return r;
    }
    // the sparse subset passes take whole blocks, from the next block boundary on
    __device__ __forceinline__ void align() {
        pos = (pos + 3) & ~3;
        if (pos == PHILOX_RING) fill(); else nxt = ring[pos * EPB];
    }
    __device__ __forceinline__ void block(uint32_t& w0, uint32_t& w1, uint32_t& w2, uint32_t& w3) {   // pos is a multiple of 4
        w0 = nxt; w1 = ring[(pos + 1) * EPB]; w2 = ring[(pos + 2) * EPB]; w3 = ring[(pos + 3) * EPB];
        pos += 4;
        if (pos == PHILOX_RING) fill(); else nxt = ring[pos * EPB];
    }
};

struct PhiloxRegs {
    uint32_t k0, k1, c0, c1, c2, c3;
    uint32_t w0, w1, w2, w3;
    int have;

    __device__ __forceinline__ void init(uint64_t seed, uint64_t env, uint32_t episode, LDS_AS uint32_t*) {
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
        c0 = 0; c1 = episode; c2 = (uint32_t)env; c3 = (uint32_t)(env >> 32);
        have = 0;
    }
    __device__ __forceinline__ uint32_t next() {
        if (have == 0) {
            philox_block(c0, c1, c2, c3, k0, k1, w0, w1, w2, w3);
            c0++;
            have = 4;
        }
        const uint32_t r = w0;
        w0 = w1; w1 = w2; w2 = w3;
        have--;
        return r;
    }
    __device__ __forceinline__ void align() { have = 0; }                          // what is left of the current block is dropped
    __device__ __forceinline__ void block(uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {   // have == 0
        philox_block(c0, c1, c2, c3, k0, k1, o0, o1, o2, o3);
        c0++;
    }
};

// numpy legacy bounded draw in [0, max]: max == 0 consumes no word; else mask & reject (random_interval).
template <class RNG>
__device__ __forceinline__ uint32_t bounded(RNG& p, uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = 0xFFFFFFFFu >> __clz((int)max);
    uint32_t v;
    do { v = p.next() & mask; } while (v > max);
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

// Handles of at most one wavefront (the single-env gym.Env adapter) have a sticky error word in GPU-addressable HOST memory
// beside the device one: lane 0 updates it with the wave's OR (one wave per launch, launches ordered by the stream: no
// atomic needed across PCIe).
__device__ __forceinline__ void raise_host_flags(uint32_t* flags_host, uint32_t flags) {
    if (flags_host) {
        uint32_t wf = flags;
        for (int o = 32; o >= 1; o >>= 1) wf |= (uint32_t)__shfl_xor((int)wf, o);
        if (threadIdx.x == 0 && wf) *flags_host |= wf;
    }
}

// The host mirror of such a handle (NgwMirror): once the step's (or reset's) own stores are out, the wave copies its rows
// from HBM - where the state lives - into the mirror arrays in host memory.  PCIe sees posted WRITES only; a kernel that kept
// its state in host memory instead spent 2.9 us of its 4.8 us waiting for reads across the bus.
__device__ __forceinline__ void mirror_wave(const NgwDevSpec* dspec, const NgwBufs& b, int S2, int K, int64_t n) {
    const GLOBAL_AS NgwMirror* mp = (const GLOBAL_AS NgwMirror*)&dspec->mir;
    NgwMirror m;
    m.map = mp->map; m.loc = mp->loc; m.facing = mp->facing; m.inv = mp->inv; m.selected = mp->selected; m.step_count = mp->step_count;
    m.reward = mp->reward; m.done = mp->done; m.info = mp->info;
    if (!m.map) return;                                                            // (uniform)
    __threadfence();                                                               // own stores are in L2, this wave's L1 lines are dropped
    const int tid = threadIdx.x;
    const int64_t env0 = (int64_t)blockIdx.x * NGW_EPB;
    const int nlive = (int)min((int64_t)NGW_EPB, n - env0);
    typedef uint32_t q4 __attribute__((ext_vector_type(4)));
    const GLOBAL_AS q4* sm = (const GLOBAL_AS q4*)(b.map + env0 * S2);             // (64 rows: 16-byte aligned on both sides; arrays are n_pad long)
    GLOBAL_AS q4* dm = (GLOBAL_AS q4*)(m.map + env0 * S2);
    for (int p = tid; p < (nlive * S2 + 15) >> 4; p += NGW_EPB) dm[p] = sm[p];
    const GLOBAL_AS q4* si = (const GLOBAL_AS q4*)(b.inv + env0 * K);
    GLOBAL_AS q4* di = (GLOBAL_AS q4*)(m.inv + env0 * K);
    for (int p = tid; p < (nlive * K * 4 + 15) >> 4; p += NGW_EPB) di[p] = si[p];
    if (tid < nlive) {
        const int64_t e = env0 + tid;
        const int pr = ((const GLOBAL_AS int32_t*)b.loc)[2 * e], pc = ((const GLOBAL_AS int32_t*)b.loc)[2 * e + 1];
        const int f = ((const GLOBAL_AS int32_t*)b.facing)[e], st = ((const GLOBAL_AS int32_t*)b.step_count)[e], rw = ((const GLOBAL_AS int32_t*)b.reward)[e];
        const uint8_t sel = ((const GLOBAL_AS uint8_t*)b.selected)[e], dn = ((const GLOBAL_AS uint8_t*)b.done)[e];
        const uint32_t info = ((const GLOBAL_AS uint32_t*)b.info)[e];
        ((GLOBAL_AS int32_t*)m.loc)[2 * e] = pr; ((GLOBAL_AS int32_t*)m.loc)[2 * e + 1] = pc;
        ((GLOBAL_AS int32_t*)m.facing)[e] = f; ((GLOBAL_AS int32_t*)m.step_count)[e] = st; ((GLOBAL_AS int32_t*)m.reward)[e] = rw;
        ((GLOBAL_AS uint8_t*)m.selected)[e] = sel; ((GLOBAL_AS uint8_t*)m.done)[e] = dn; ((GLOBAL_AS uint32_t*)m.info)[e] = info;
    }
}

// ... and when every store is out, lane 0 writes the launch's sequence number next to the flags word; the host polls that
// word instead of paying a stream synchronisation (ngw_step_host, ngw_reset_host).
__device__ __forceinline__ void signal_host_seq(uint32_t* flags_host, uint32_t seq) {
    if (flags_host && seq) {                                                       // (uniform)
        __builtin_amdgcn_s_waitcnt(0);                                             // the wave's stores have been accepted ...
        __threadfence_system();                                                    // ... and are ordered before the word below
        if (threadIdx.x == 0) *(volatile uint32_t*)(flags_host + NGW_SEQ_WORD) = seq;
    }
}

// ---------------------------------------------------------------- per-lane reset on the LDS map
// pogostick_v1_env.py:86-157 + add_item_to_map :159-181 (+ AddItem.reset, AxeEasy.reset).  `mp` = this lane's map
// in LDS, `inv` = this lane's inventory row, `cand` = candidate bitmask column (stride EPB).
// Out of line ON PURPOSE: this is the cold path (1 % of env-steps at H = 100); inlined, its register needs spill the
// scalars of the hot step loop.  Returns flags | r<<8 | c<<16 | facing<<24.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));                     // native vector type (VGPR quad)

struct ResetArgs {
    const NgwDevSpec* dspec;
    uint16_t* perm;                 // HBM scratch [S2][n_pad] (used when the shuffle array does not fit in LDS)
    int64_t n_pad;
    uint64_t seed;
    int S, S2, K, CW, perm_lds;
    uint32_t magicS;                // ceil(2^32 / S): cell / S for cell < S*S
    uint32_t rs0, rs1, rs2, rs3;             // NgwResetU's packed spec bytes: wall|tap|tap_near|n_place, n_passes|n_inv_start, inv_start_item[4], inv_start_qty[4]
    uint32_t pw0, pw1, pw2, pw3;              // shuffled-subset passes: kind | item << 8 | from << 16 | span << 24
};

// The subset passes NGW_PASS_SPARSE names (include/ngw.h, ngw_spec.n_passes; oracle: subset_pass_sparse) on the lane's byte
// map: AddItem / Crate over the air cells, ReplaceItem / FireWall over the wall cells.  No index array: the percent first,
// then min(cnt, len - cnt) distinct matching cells by rejection - the complement when that is the smaller set - candidates
// = nb-bit cell indices cut from whole Philox blocks (field j of words 0..3, then field j + 1; next block boundary on; the
// rest of the last block is dropped), a taken cell marked NGW_PASS_MARK until the closing sweep writes the items.  This is
// the cold form (resets inside a step when no prepared episode exists, stacks of passes, the fused lidar path);
// ngw_reset.inc runs the same draws on one bit per cell.
template <class RNG, typename MP>
__device__ __forceinline__ void sparse_pass(RNG& px, MP mp, int S2, int agent, int from, int item, int pct_span, const GLOBAL_AS double* pctq) {
    int len = 0;
    for (int i = 0; i < S2; i++) len += mp[i] == from;
    const int pct = (int)bounded(px, (uint32_t)(pct_span - 1));                    // randint(lo, hi) FIRST; a span of 1 draws nothing
    const int cnt = (int)ceil((double)len * pctq[pct]);                            // int(np.ceil(len * (pct / 100)))
    const bool comp = 2 * cnt > len;
    const int need = comp ? len - cnt : cnt;
    px.align();
    const int nb = 32 - __clz(S2 - 1), F = 32 / nb;                                // (uniform)
    const uint32_t fm = (1u << nb) - 1u;
    for (int got = 0; got < need;) {
        uint32_t w[4];
        px.block(w[0], w[1], w[2], w[3]);
        for (int j = 0; j < F; j++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int cell = (int)((w[k] >> (j * nb)) & fm);
                if (got < need && cell < S2 && mp[cell] == from) { mp[cell] = (int8_t)NGW_PASS_MARK; got++; }
            }
    }
    for (int cell = 0; cell < S2; cell++) {
        const int v = mp[cell];
        const bool marked = v == NGW_PASS_MARK;
        if (marked || v == from) mp[cell] = (int8_t)((marked != comp && cell != agent) ? item : from);   // chosen = marked (direct) / unmarked (complement)
    }
}

// The shuffled-subset reset passes - AddItem.reset (novelty_wrappers.py:1017-1028), ReplaceItem.reset (:1131-1144),
// Fence.reset (:871-884) - on a shuffle array `perm` with element stride `ps`: np.where(<predicate>) in row-major order,
// np.random.shuffle (Fisher-Yates from the top), percent = randint(lo, hi), edit the first ceil(len * (percent / 100)).
template <int KIND, typename P, class RNG, typename MP>
__device__ __forceinline__ void subset_pass(P perm, int64_t ps, RNG& px, MP mp, int S, int S2, int agent, int match,
                                            int item, int pct_span, const GLOBAL_AS double* pctq) {
    int n = 0;
    for (int i = 0; i < S2; i++) {
        const int v = mp[i];
        const bool hit = KIND == NGW_PASS_ADDITEM ? v == 0 : KIND == NGW_PASS_REPLACE ? v == match : (v != 0 && v != match);
        if (hit) { perm[(int64_t)n * ps] = (uint16_t)i; n++; }
    }
    for (int i = n - 1; i >= 1; i--) {
        const int j = (int)bounded(px, (uint32_t)i);
        const uint16_t x = perm[(int64_t)i * ps], y = perm[(int64_t)j * ps];
        perm[(int64_t)i * ps] = y; perm[(int64_t)j * ps] = x;
    }
    const int pct = (int)bounded(px, (uint32_t)(pct_span - 1));                    // randint(lo, hi); a span of 1 draws nothing
    const int cnt = (int)ceil((double)n * pctq[pct]);                              // int(np.ceil(len * (pct / 100)))
    for (int i = 0; i < cnt; i++) {
        const int cell = perm[(int64_t)i * ps];
        if (KIND == NGW_PASS_FENCE) {                                              // add_fence_around, pogostick_v1_env.py:524-536
            for (int dr = -S; dr <= S; dr += S)
                for (int dc = -1; dc <= 1; dc++) {
                    const int q = cell + dr + dc;
                    if (mp[q] == 0 && q != agent) mp[q] = (int8_t)item;
                }
        } else if (cell != agent) mp[cell] = (int8_t)item;                         // :1027 / :1143 skip the agent cell
    }
}

template <int KIND, class RNG, typename MP>
__device__ __forceinline__ void run_pass(const ResetArgs& a, int pass_index, LDS_AS uint16_t* perm_lds, int64_t env_local, RNG& px, MP mp,
                                         int agent, int match, int item, int pct_span) {
    const GLOBAL_AS double* pctq = (const GLOBAL_AS double*)a.dspec->pctq[pass_index];
    if (a.perm_lds) {
        // shuffle array in LDS, [i][32 lanes] u16: the two halves of the wave take turns (a wave executes divergent
        // halves one after the other and its LDS operations are in order, so they can share the region)
        const int lane = threadIdx.x;
        for (int half = 0; half < 2; half++)
            if ((lane >> 5) == half) subset_pass<KIND>(perm_lds + (lane & 31), 32, px, mp, a.S, a.S2, agent, match, item, pct_span, pctq);
    } else {
        subset_pass<KIND>((GLOBAL_AS uint16_t*)(a.perm + env_local), a.n_pad, px, mp, a.S, a.S2, agent, match, item, pct_span, pctq);
    }
}

// (LDS / global pointers carry their address space: across a real call generic pointers would turn every access
//  into a flat_* instruction)
// MP = the lane's map: LDS_AS int8_t* (staged kernels) or GLOBAL_AS int8_t* (the no-stage step kernel's fallback, which has
// no map in LDS and runs the same loop straight on the env's map row in HBM - slow, and only when no prepared episode exists).
template <class RNG, typename MP>
__device__ __forceinline__ uint32_t reset_lane(const ResetArgs a, MP mp, LDS_AS int32_t* inv, LDS_AS uint32_t* cand,
                                            const LDS_AS uint8_t* place_seq, LDS_AS uint16_t* perm_lds, LDS_AS uint32_t* rng_ring,
                                            uint64_t env_global, int64_t env_local, uint32_t episode) {
    // the spec bytes this path needs arrived with the call's other uniform arguments (no dependent spec loads in here)
    const int wall_item = a.rs0 & 255, tap_item = (a.rs0 >> 8) & 255, tap_near = (a.rs0 >> 16) & 255, n_place = a.rs0 >> 24;
    const int n_passes = a.rs1 & 255, n_inv_start = (a.rs1 >> 8) & 255;
    int r_out, c_out, f_out;
    const int S = a.S, K = a.K, W = S - 4, ncand = W * W;
    const uint32_t magicW = (uint32_t)((0x100000000ull + (uint32_t)W - 1) / (uint32_t)W);   // pos / W == umulhi(pos, magicW), pos < 2^12
    RNG px;
    px.init(a.seed, env_global, episode, rng_ring);
    for (int k = 0; k < K; k++) inv[k] = 0;                                        // :119
    for (int r = 0; r < S; r++)                                                    // :129-130 wall ring around air
        for (int c = 0; c < S; c++)
            mp[r * S + c] = (r == 0 || c == 0 || r == S - 1 || c == S - 1) ? (int8_t)wall_item : (int8_t)0;
    for (int w = 0; w < a.CW; w++) {                                               // :136-138 all interior candidates
        int left = ncand - w * 32;
        cand[w * EPB] = left >= 32 ? 0xFFFFFFFFu : (left > 0 ? ((1u << left) - 1u) : 0u);
    }
    int len = ncand;
    uint32_t flags = 0;
    int apos = (int)bounded(px, (uint32_t)len - 1);                                // :141 (agent stays in the list)
    const int arow = (int)__umulhi((uint32_t)apos, magicW), acol = apos - arow * W;
    const int agent = (2 + arow) * S + 2 + acol;
    r_out = 2 + arow; c_out = 2 + acol;
    f_out = (int)bounded(px, 3);                                                   // :145
    // :147-148 + add_item_to_map :159-181, FLATTENED: placement n takes item place_seq[n] (items_quantity in insertion
    // order), so a wave iterates max-over-lanes of the TOTAL number of tries, not the sum of per-item maxima.
    const int total = n_place;
    int n = 0;
    // Candidate bitmask: two words cover maps up to 10 x 10 (36 interior candidates) and then live in registers; larger
    // maps keep them in LDS.  Every LDS read of a try is issued before the first one is needed (the item to place, the
    // cell and its four neighbours - unconditionally: a short-circuit && would make them five dependent round trips).
    const bool cand_regs = a.CW <= 2;
    uint32_t cr0 = cand[0], cr1 = a.CW > 1 ? cand[EPB] : 0u;
    while (n < total) {
        if (len < 1) { flags |= NGW_F_PLACEMENT; break; }                         // :167
        const int item = place_seq[n];
        int idx = (int)bounded(px, (uint32_t)len - 1);                             // :169
        int pos;
        if (cand_regs) {
            const int pc0 = __popc(cr0);
            const bool lo = idx < pc0;
            const int bit = nth_set_bit(lo ? cr0 : cr1, lo ? idx : idx - pc0);     // idx-th remaining, row-major
            if (lo) cr0 &= ~(1u << bit); else cr1 &= ~(1u << bit);                 // list.pop(idx)
            pos = (lo ? 0 : 32) + bit;
        } else {
            int w = 0, pc;
            while (idx >= (pc = __popc(cand[w * EPB]))) { idx -= pc; w++; }
            const int bit = nth_set_bit(cand[w * EPB], idx);
            cand[w * EPB] &= ~(1u << bit);
            pos = w * 32 + bit;
        }
        len--;
        const int prow = (int)__umulhi((uint32_t)pos, magicW);
        const int cell = (2 + prow) * S + 2 + (pos - prow * W);
        const int m0 = mp[cell], mN = mp[cell - S], mS = mp[cell + S], mW = mp[cell - 1], mE = mp[cell + 1];
        if (cell != agent && (m0 | mN | mS | mW | mE) == 0) {                      // :172-178
            mp[cell] = (int8_t)item;                                               // :177-180
            n++;
        }
    }
    if (tap_item && !flags) {                                                      // Pogostick-v0, pogostick_v0_env.py:156-178
        const int near = tap_near;
        int nl = 0;
        for (int i = 0; i < a.S2; i++) nl += (mp[i] == near);                     // np.where(map == tree_log)
        if (nl <= 1) flags |= NGW_F_PLACEMENT;                                     // assert len(result[0]) > 1
        for (int tries = 0; !flags; tries++) {
            if (tries >= 4096) { flags |= NGW_F_PLACEMENT; break; }               // no log has a free neighbour: give up loudly
            const int d = (int)bounded(px, 3);                                     // np.random.choice(4 directions)
            int idx = (int)bounded(px, (uint32_t)nl - 1), cell = 0;
            for (int i = 0; i < a.S2; i++)                                         // idx-th log, row-major
                if (mp[i] == near) { if (idx == 0) { cell = i; break; } idx--; }
            const int lr = (int)__umulhi((uint32_t)cell, a.magicS);                // cell / S
            const int rr = lr + ((d == 0) ? -1 : (d == 1 ? 1 : 0)), cc = cell - lr * S + ((d == 2) ? -1 : (d == 3 ? 1 : 0));
            if (rr >= 0 && rr <= S - 1 && cc >= 0 && cc <= S - 1 && mp[rr * S + cc] == 0 && rr * S + cc != agent) {
                mp[rr * S + cc] = (int8_t)tap_item;
                break;
            }
        }
    }
    if (n_passes && !flags)
        for (int j = 0; j < n_passes; j++) {                                       // stacked wrappers reset innermost first = injection order
            const uint32_t w = j == 0 ? a.pw0 : (j == 1 ? a.pw1 : (j == 2 ? a.pw2 : a.pw3));
            const int kind = w & 255, item = (w >> 8) & 255, from = (w >> 16) & 255, span = w >> 24;
            const GLOBAL_AS double* pctq = (const GLOBAL_AS double*)a.dspec->pctq[j];
            if (kind == NGW_PASS_ADDITEM)                                          // AddItem / Crate: air cells
                sparse_pass(px, mp, a.S2, agent, 0, item, span, pctq);
            else if (kind == NGW_PASS_REPLACE && from == wall_item)                // ReplaceItem / FireWall of the wall ring
                sparse_pass(px, mp, a.S2, agent, from, item, span, pctq);
            else if (kind == NGW_PASS_REPLACE)                                     // ReplaceItem of an item of the interior
                run_pass<NGW_PASS_REPLACE>(a, j, perm_lds, env_local, px, mp, agent, from, item, span);
            else                                                                   // Fence / FenceRestriction
                run_pass<NGW_PASS_FENCE>(a, j, perm_lds, env_local, px, mp, agent, wall_item, item, span);
        }
    if (!flags)                                                                    // AxeEasy.reset :33, AxetoBreakHard.reset :667-670
        for (int j = 0; j < n_inv_start; j++) inv[(a.rs2 >> (8 * j)) & 255u] = (int)((a.rs3 >> (8 * j)) & 255u);
    return flags | ((uint32_t)r_out << 8) | ((uint32_t)c_out << 16) | ((uint32_t)f_out << 24);
}

// Prepared-next-episode fast path of a reset: if the shadow buffers hold the first state of `episode` for this env, copy
// it into the lane's LDS map / inventory row AND straight into the env's observation rows in HBM (so the wave does not
// have to store its whole 64-env chunk for this lane).  One HBM round trip instead of the placement loop's ~40
// dependent draws.  A lone lane runs this, so the INSTRUCTION COUNT is what costs: rows move as 16-byte chunks whose
// start is clamped to (row end - 16) - the last chunk overlaps its predecessor instead of being predicated per element.
// The rows were written by an earlier launch on the same stream (NGW_MODE_REFILL): plain visible global memory.
typedef GLOBAL_AS u32x4_t g_u32x4_t;
__device__ __forceinline__ uint32_t consume_lane(const NgwNx nx, LDS_AS int8_t* mp, LDS_AS int32_t* inv, GLOBAL_AS int8_t* gm,
                                                 GLOBAL_AS int32_t* gi, int64_t e /* row of the shadow arrays */, int S2, int K) {
    const GLOBAL_AS int8_t* src = (const GLOBAL_AS int8_t*)nx.map + e * S2;
    const GLOBAL_AS int32_t* sinv = (const GLOBAL_AS int32_t*)nx.inv + e * K;
    const int pr = ((const GLOBAL_AS int32_t*)nx.loc)[2 * e], pc = ((const GLOBAL_AS int32_t*)nx.loc)[2 * e + 1];
    const int f = ((const GLOBAL_AS int32_t*)nx.facing)[e];
    constexpr int R = 8;                                                           // 16-byte chunks per round trip
    u32x4_t q[(NGW_MAX_ITEMS + 3) / 4];
    const int nqi = (K + 3) >> 2;                                                  // K >= 4 always (air, wall, table, goal, ...)
#pragma unroll
    for (int j = 0; j < (NGW_MAX_ITEMS + 3) / 4; j++) q[j] = *(const g_u32x4_t*)(sinv + min(4 * j, K - 4));
    if ((S2 & 3) == 0) {                                                           // even S: rows are dword-aligned on both sides
        const GLOBAL_AS uint32_t* s4 = (const GLOBAL_AS uint32_t*)src;
        LDS_AS uint32_t* d4 = (LDS_AS uint32_t*)mp;
        GLOBAL_AS uint32_t* g4 = (GLOBAL_AS uint32_t*)gm;
        const int nd = S2 >> 2, nq = (nd + 3) >> 2;                                // nd >= 6 (S >= 5)
        for (int base = 0; base < nq; base += R) {
            u32x4_t v[R];
#pragma unroll
            for (int j = 0; j < R; j++) v[j] = *(const g_u32x4_t*)(s4 + min(4 * (base + j), nd - 4));
#pragma unroll
            for (int j = 0; j < R; j++)
                if (base + j < nq) {
                    const int o = min(4 * (base + j), nd - 4);
                    *(g_u32x4_t*)(g4 + o) = v[j];
                    d4[o] = v[j].x; d4[o + 1] = v[j].y; d4[o + 2] = v[j].z; d4[o + 3] = v[j].w;
                }
        }
    } else {                                                                       // odd S: byte rows
        for (int base = 0; base < S2; base += 16) {
            const int o = min(base, S2 - 16);
            int8_t v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = src[o + j];
#pragma unroll
            for (int j = 0; j < 16; j++) { mp[o + j] = v[j]; gm[o + j] = v[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < (NGW_MAX_ITEMS + 3) / 4; j++)
        if (j < nqi) {
            const int o = min(4 * j, K - 4);
            *(g_u32x4_t*)(gi + o) = q[j];
            inv[o] = (int)q[j].x; inv[o + 1] = (int)q[j].y; inv[o + 2] = (int)q[j].z; inv[o + 3] = (int)q[j].w;
        }
    return ((uint32_t)pr << 8) | ((uint32_t)pc << 16) | ((uint32_t)f << 24);
}

// The one out-of-line entry of the cold path: a prepared row if there is one, else the placement loop.  Bit
// NGW_F_ROWS_STORED of the result says the env's rows are already in HBM (the wave need not store its chunk for it).
constexpr uint32_t NGW_F_ROWS_STORED = 0x80u;
__device__ __noinline__ uint32_t new_episode(const NgwDevSpec* dspec, LDS_AS int8_t* mp, LDS_AS int32_t* inv, LDS_AS uint32_t* cand,
                                             const LDS_AS uint8_t* place_seq, LDS_AS uint16_t* perm_lds, uint64_t env_global,
                                             int64_t env_local, uint32_t episode, bool may_consume, bool count_miss = true) {
    const GLOBAL_AS NgwResetU* rp = (const GLOBAL_AS NgwResetU*)&dspec->ru;        // both blobs requested together
    const GLOBAL_AS NgwNx* np = (const GLOBAL_AS NgwNx*)&dspec->nx;
    NgwResetU ru; NgwNx nx;
    ru.perm = rp->perm; ru.map = rp->map; ru.inv = rp->inv; ru.n_pad = rp->n_pad; ru.seed = rp->seed; ru.S = rp->S;
    ru.S2 = rp->S2; ru.K = rp->K; ru.CW = rp->CW; ru.perm_lds = rp->perm_lds; ru.magicS = rp->magicS; ru.off_rng = rp->off_rng;
    const GLOBAL_AS uint32_t* rsw = (const GLOBAL_AS uint32_t*)&rp->wall_item;
    const uint32_t rs0 = rsw[0], rs1 = rsw[1], rs2 = rsw[2], rs3 = rsw[3], pw0 = rsw[4], pw1 = rsw[5], pw2 = rsw[6], pw3 = rsw[7];
    nx.map = np->map; nx.loc = np->loc; nx.facing = np->facing; nx.inv = np->inv; nx.episode = np->episode; nx.slow = np->slow;
    if (may_consume && nx.episode) {
        const int64_t row = (int64_t)(episode & (uint32_t)np->dmask) * np->stride + env_local;   // the slot of this episode
        if (((const GLOBAL_AS uint32_t*)nx.episode)[row] == episode)
            return consume_lane(nx, mp, inv, (GLOBAL_AS int8_t*)ru.map + env_local * ru.S2, (GLOBAL_AS int32_t*)ru.inv + env_local * ru.K,
                                row, ru.S2, ru.K) | NGW_F_ROWS_STORED;
        if (count_miss) atomicAdd(nx.slow, 1u);                                    // a stale row inside a step: the host shortens the refill cadence
    }
    const ResetArgs a = {dspec, ru.perm, ru.n_pad, ru.seed, ru.S, ru.S2, ru.K, ru.CW, ru.perm_lds, ru.magicS, rs0, rs1, rs2, rs3, pw0, pw1, pw2, pw3};
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_base[];            // the kernel's dynamic LDS (offset 0)
    if (ru.off_rng != 0xFFFFFFFFu)                                                 // which word source: decided with the LDS layout (ngw_abi.cpp)
        return reset_lane<PhiloxRing>(a, mp, inv, cand, place_seq, perm_lds, (LDS_AS uint32_t*)(lds_base + ru.off_rng + threadIdx.x),
                                      env_global, env_local, episode);
    return reset_lane<PhiloxRegs>(a, mp, inv, cand, place_seq, perm_lds, nullptr, env_global, env_local, episode);
}

// ---------------------------------------------------------------- map staging HBM <-> LDS (coalesced 16-B pieces)
// The wave's 64 maps are one contiguous 64*S2-byte chunk in HBM = 4*S2 pieces of 16 B; lane l owns pieces
// l, l+64, ...  A round moves PB pieces per lane: ALL its global loads are issued before the first LDS write (and
// all LDS reads before the first global store), so a round costs one memory round trip, not PB.
// In LDS each env's map starts at e*MS bytes with MS/4 odd, so 64 lanes reading "their" cell hit distinct banks.
constexpr int PB = 8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));                       // native vector: stays in VGPRs
typedef GLOBAL_AS u32x4 g_u32x4;
typedef GLOBAL_AS int2 g_int2;
typedef GLOBAL_AS int32_t g_i32;
typedef GLOBAL_AS uint32_t g_u32;
typedef GLOBAL_AS uint8_t g_u8;

// Loads are UNCONDITIONAL on a clamped index (a duplicate in-bounds load is harmless and keeps the values in plain
// registers); only stores and LDS writes are predicated.
__device__ __forceinline__ void pieces_load(u32x4 (&buf)[PB], const u32x4* g4, int base, int npieces, int tid) {
#pragma unroll
    for (int j = 0; j < PB; j++) buf[j] = g4[min(base + tid + EPB * j, npieces - 1)];
}

// LDS <-> register pieces.  TO_LDS: buf -> lds, else lds -> buf.
template <bool TO_LDS, int MAPMODE>
__device__ __forceinline__ void pieces_lds(u32x4 (&buf)[PB], const NgwLaunch& a, uint32_t* lds_map, int base, int npieces, int tid) {
    if (MAPMODE == NGW_MAP_STRAIGHT) {                                             // LDS image == HBM image
        u32x4* l4 = reinterpret_cast<u32x4*>(lds_map);
#pragma unroll
        for (int j = 0; j < PB; j++) {
            // (pieces_load clamps its addresses the same way: a slot beyond the chunk holds the LAST piece and stores it to the
            //  last piece's place again - no exec-mask bracket per piece)
            const int p = min(base + tid + EPB * j, npieces - 1);
            if (TO_LDS) l4[p] = buf[j]; else buf[j] = l4[p];
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
    const int rounds = (nq + EPB - 1) / EPB;                                       // uniform: quads per lane actually used
    if (a.KP == a.K) {                                                             // K odd: LDS image == HBM image
        u32x4* l4 = reinterpret_cast<u32x4*>(lds_inv);
#pragma unroll
        for (int j = 0; j < IQ; j++) {
            if (j >= rounds) break;
            const int p = tid + EPB * j;
            if (TO_LDS) { if (p < nq) l4[p] = q[j]; } else q[j] = l4[min(p, nq - 1)];
        }
    } else {                                                                       // K even: one pad dword per env
#pragma unroll
        for (int j = 0; j < IQ; j++) {
            if (j >= rounds) break;
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

// ---------------------------------------------------------------- LidarInFront rays (shared by both kernels)
// observation_wrappers.py:32-65 on the LDS map.  `agent` = this lane's agent cell; `toff` = flat int16 cell offsets
// [facing][beam][range-1]; 4 beams x 4 ranges per round: 4 table reads (4 offsets each), then 16 independent cell reads
// in flight - the march is an LDS latency chain and the beams are independent, so they share each latency period.
// A ray cannot leave the map before it hits the wall ring, so cells prefetched beyond the hit are simply ignored
// (the LDS layout keeps a guard on both sides of the maps for them).
__device__ __forceinline__ void lidar_march(const int8_t* agent, int f, int B, int R, int NC, const int16_t* toff,
                                            const uint8_t* chan_of_item, int32_t* row) {
    // 4 beams x 12 ranges per chunk: 12 table reads (4 int16 offsets each) and then 48 independent cell reads are in flight
    // together, so a chunk costs two LDS latencies, not 2 x 12.  The ids of 4 consecutive ranges are packed into one
    // dword; the first non-air block (:59-64) is its lowest non-zero byte (ffs).  No per-cell range test: beyond max_range
    // the table repeats the last in-range cell, so a padded entry can never be the FIRST non-zero one.
    const uint8_t* ag = reinterpret_cast<const uint8_t*>(agent);
    for (int b0 = 0; b0 < B; b0 += 4) {
        int hit_k[4] = {0, 0, 0, 0}, hit_id[4] = {0, 0, 0, 0};
        for (int k0 = 0; k0 < R; k0 += 12) {
            bool open = false;
#pragma unroll
            for (int bb = 0; bb < 4; bb++) open |= (b0 + bb < B) && !hit_k[bb];
            if (!open) break;
            uint2 o[4][3];
#pragma unroll
            for (int bb = 0; bb < 4; bb++) {
                const int16_t* t = toff + (f * NGW_LIDAR_MAX_BEAMS + min(b0 + bb, B - 1)) * NGW_LIDAR_MAX_RANGE + k0;
#pragma unroll
                for (int g = 0; g < 3; g++) o[bb][g] = *reinterpret_cast<const uint2*>(t + 4 * min(g, (NGW_LIDAR_MAX_RANGE - 1 - k0) / 4));
            }
            uint32_t w[4][3];
#pragma unroll
            for (int bb = 0; bb < 4; bb++)
#pragma unroll
                for (int g = 0; g < 3; g++) {
                    const uint32_t i0 = ag[(int16_t)(o[bb][g].x & 0xFFFFu)], i1 = ag[(int16_t)(o[bb][g].x >> 16)];
                    const uint32_t i2 = ag[(int16_t)(o[bb][g].y & 0xFFFFu)], i3 = ag[(int16_t)(o[bb][g].y >> 16)];
                    w[bb][g] = i0 | (i1 << 8) | (i2 << 16) | (i3 << 24);
                }
#pragma unroll
            for (int bb = 0; bb < 4; bb++)
                if (!hit_k[bb]) {
#pragma unroll
                    for (int g = 2; g >= 0; g--)
                        if (w[bb][g]) {
                            const int q = (__ffs((int)w[bb][g]) - 1) >> 3;                   // lowest non-zero byte
                            hit_k[bb] = k0 + 4 * g + q + 1;
                            hit_id[bb] = (int)((w[bb][g] >> (8 * q)) & 255u);
                        }
                }
        }
#pragma unroll
        for (int bb = 0; bb < 4; bb++)
            if (b0 + bb < B && hit_k[bb] && hit_k[bb] <= R) {
                const int ch = chan_of_item[hit_id[bb]];
                if (ch) row[(b0 + bb) * NC + ch - 1] = hit_k[bb];
            }
    }
}

// The wave's 64 observation rows leave the LDS tile as one contiguous block: 16 * L pieces of int32, or - with the int16
// output - 8 * L pieces, each packing eight consecutive values (they are >= 0; anything above 32767 saturates).
// (pointers carry their address space: a pinned generic pointer would turn every access into a flat_* instruction)
__device__ __forceinline__ void lidar_store(const LDS_AS uint32_t* tile, GLOBAL_AS uint32_t* out_chunk, int L, bool i16, int tid) {
    const LDS_AS u32x4* t4 = (const LDS_AS u32x4*)tile;
    GLOBAL_AS u32x4* g4 = (GLOBAL_AS u32x4*)out_chunk;
    if (!i16) {
        for (int p = tid; p < 16 * L; p += EPB) g4[p] = t4[p];
    } else {
        for (int p = tid; p < 8 * L; p += EPB) {
            const u32x4 a = t4[2 * p], b = t4[2 * p + 1];
            g4[p] = u32x4{min(a.x, 32767u) | (min(a.y, 32767u) << 16), min(a.z, 32767u) | (min(a.w, 32767u) << 16),
                          min(b.x, 32767u) | (min(b.y, 32767u) << 16), min(b.z, 32767u) | (min(b.w, 32767u) << 16)};
        }
    }
}

constexpr int LIDAR_TAB16 = 4 * NGW_LIDAR_MAX_BEAMS * NGW_LIDAR_MAX_RANGE * 2 / 16;     // ray table = 512 pieces of 16 B
static_assert(LIDAR_TAB16 == 8 * NGW_EPB, "ray table is 8 pieces per lane");
static_assert(offsetof(NgwLidarDev, chan_of_item) == 16 * LIDAR_TAB16 && offsetof(NgwLidarDev, inv_item) == 16 * LIDAR_TAB16 + NGW_MAX_ITEMS,
              "lidar tables are contiguous");

// ---------------------------------------------------------------- the kernel
// LDS reads of the step are issued in TWO parallel levels (L0: action descriptor, block in front and its four
// neighbours, the inventory slots whose item id is uniform; L1: the slots whose id comes out of L0) and the per-kind
// bodies then work on registers only - the dependency chain of a step is two LDS latencies plus ALU, whatever the kind.
template <int MAPMODE, int MODE, bool LIDAR, bool EXT>
__global__ void __launch_bounds__(NGW_EPB) ngw_kernel(const NgwDevSpec* __restrict__ dspec, const NgwLaunch a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (MODE == NGW_MODE_DBG_NOP) return;
    STAMP_DECL;
    STAMP(0);
    const int tid = threadIdx.x;
    const int64_t env0 = (int64_t)blockIdx.x * EPB;
    const int64_t e = env0 + tid;                                                  // local env index of this lane
    const bool live = e < a.n;
    const int S = a.S, K = a.K;
    const int npieces = 4 * a.S2;                                                  // EPB * S2 / 16

    // LDS carve-up (dword offsets): maps | inventory [64][KP] | candidate masks [CW][64] | action descriptors
    uint32_t* lds_map = lds + a.off_map;
    int32_t* lds_inv = reinterpret_cast<int32_t*>(lds + a.off_inv);
    uint32_t* lds_cand = lds + a.off_cand;
    const uint32_t* lds_act = lds + a.off_act;
    const NgwStepU U = dspec->u;                                                   // uniform: ONE scalar load, kept in SGPRs
    int8_t* mp = reinterpret_cast<int8_t*>(lds_map) + tid * a.MS;                 // this lane's map
    int32_t* inv = lds_inv + tid * a.KP;                                           // this lane's inventory row
    uint32_t* cand = lds_cand + tid;

    // ---- issue EVERY global load of the prologue before touching LDS: action table, first map round, scalars, inventory
    constexpr int NACT = NGW_MAX_ACTIONS * NGW_ACT_DW + NGW_MAX_PLACE / 4;       // action descriptors + placement sequence
    static_assert(NACT <= 4 * EPB, "the LUT block is loaded with 4 dwords per lane");
    static_assert(offsetof(NgwDevSpec, place_seq) == offsetof(NgwDevSpec, act_desc) + NGW_MAX_ACTIONS * NGW_ACT_DW * 4, "LUT block is contiguous");
    uint32_t sv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) sv[j] = dspec->act_desc[min(tid + EPB * j, NACT - 1)];
    u32x4 ltb[LIDAR ? 8 : 1];                                                       // fused lidar: ray table + item tables
    uint32_t lit = 0;
    if (LIDAR) {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.lcfg->off);
#pragma unroll
        for (int j = 0; j < 8; j++) ltb[j] = src[tid + EPB * j];
        if (tid < 2 * NGW_MAX_ITEMS / 4) lit = reinterpret_cast<const uint32_t*>(a.lcfg->chan_of_item)[tid];
    }
    int r = 1, c = 1, f = 0, sel = 0, steps = 0, action = 0;
    uint32_t episode = 0, nx_old = 0;
    if (MODE == NGW_MODE_REFILL) {
        // a.b is ONE SLOT of the shadow set (a.autoreset = its number, a.horizon = depth - 1); a.actions carries the main
        // episode[].  The slot belongs to the one episode E of (main, main + depth] with E & (depth - 1) == slot; its row is
        // stale unless it was prepared for exactly E.  Waves without a stale row leave before touching anything else.
        if (live) {
            nx_old = a.b.episode[e];
            const uint32_t main_ep = reinterpret_cast<const uint32_t*>(a.actions)[e];
            const uint32_t target = main_ep + 1u + (((uint32_t)a.autoreset - main_ep - 1u) & (uint32_t)a.horizon);
            if (nx_old != target) { action = 1; episode = target - 1u; } else episode = nx_old;
        }
        if (!__any(action)) return;
    }
    u32x4 buf[PB];
    const u32x4* gin = reinterpret_cast<const u32x4*>(a.b.map + env0 * a.S2);
    pieces_load(buf, gin, 0, npieces, tid);
    if (live) {
        const int2 rc = reinterpret_cast<const int2*>(a.b.loc)[e];
        r = rc.x; c = rc.y;
        f = a.b.facing[e];
        if (MODE != NGW_MODE_REFILL) {
            sel = a.b.selected[e];
            steps = a.b.step_count[e];
            episode = a.b.episode[e];
        }
        if (MODE == NGW_MODE_STEP || MODE == NGW_MODE_ROLLOUT_ACT) action = a.actions[e];
        if (MODE == NGW_MODE_STEP && a.use_action0) action = a.action0;            // (one-env handles: the action came with the arguments)
        else if (MODE == NGW_MODE_RESET) action = a.reset_mask ? (int)a.reset_mask[e] : 1;
    }
    u32x4 iq[IQ];
    {
        const u32x4* gi = reinterpret_cast<const u32x4*>(a.b.inv + env0 * K);
#pragma unroll
        for (int j = 0; j < IQ; j++) iq[j] = (j * EPB < 16 * K) ? gi[min(tid + EPB * j, 16 * K - 1)] : u32x4{0u, 0u, 0u, 0u};
    }
    STAMP(1);
    // ---- land them in LDS
    {
        uint32_t* dst = lds + a.off_act;
#pragma unroll
        for (int j = 0; j < 4; j++) { const int i = tid + EPB * j; if (i < NACT) dst[i] = sv[j]; }
    }
    pieces_lds<true, MAPMODE>(buf, a, lds_map, 0, npieces, tid);
    for (int base = EPB * PB; base < npieces; base += EPB * PB) {                  // big maps: further rounds
        pieces_load(buf, gin, base, npieces, tid);
        pieces_lds<true, MAPMODE>(buf, a, lds_map, base, npieces, tid);
    }
    inv_lds<true>(iq, a, lds_inv, tid);
    if (LIDAR) {
        u32x4* dst = reinterpret_cast<u32x4*>(lds + a.off_ltab);
#pragma unroll
        for (int j = 0; j < 8; j++) dst[tid + EPB * j] = ltb[j];
        if (tid < 2 * NGW_MAX_ITEMS / 4) lds[a.off_ltab + 4 * LIDAR_TAB16 + tid] = lit;
    }
    __syncthreads();
    STAMP(2);

    uint32_t flags = 0;
    int reward = 0, ended = 0;
    uint32_t info = 0;
    uint32_t aw0 = 0, aw1 = 0, aw2 = 0, aw3 = 0;                                   // rollout: 4 actions per Philox block
    int act_next = action;                                                         // rollout with the caller's actions: one step ahead

    // ---- everything the step loop needs, fetched once and pinned in registers
    // per-lane output addresses (VGPR pairs; 1 wave per SIMD leaves plenty)
    g_u32x4* gmap = (g_u32x4*)(reinterpret_cast<u32x4*>(a.b.map + env0 * a.S2) + tid);     // coalesced chunk (reset only)
    g_u32x4* ginv = (g_u32x4*)(reinterpret_cast<u32x4*>(a.b.inv + env0 * K) + tid);
    GLOBAL_AS int8_t* gm = (GLOBAL_AS int8_t*)(a.b.map + e * a.S2);                      // this env's map / inventory row
    g_i32* gi = (g_i32*)(a.b.inv + e * K);
    g_int2* gloc = (g_int2*)(reinterpret_cast<int2*>(a.b.loc) + e);
    g_i32* gfac = (g_i32*)(a.b.facing + e);
    g_i32* grew = (g_i32*)(a.b.reward + e);
    g_u8* gdone = (g_u8*)(a.b.done + e);
    g_u32* ginfo = (g_u32*)(a.b.info + e);
    PIN_V(gmap); PIN_V(ginv); PIN_V(gm); PIN_V(gi); PIN_V(gloc); PIN_V(gfac); PIN_V(grew); PIN_V(gdone); PIN_V(ginfo);
    // int16 output: the same buffer holds half as many bytes per row (chunk start = env0 * L * 2 bytes)
    GLOBAL_AS uint32_t* glid = LIDAR ? (GLOBAL_AS uint32_t*)(reinterpret_cast<uint32_t*>(a.lout) + ((env0 * a.lidar_len) >> (a.l_i16 ? 1 : 0))) : nullptr;
    int lB = 0, lR = 0, lNC = 0, lNI = 0;
    if (LIDAR) {
        lB = a.l_beams; lR = a.l_range; lNC = a.l_chan; lNI = a.l_inv;
        PIN_V(glid); PIN_S(lB); PIN_S(lR); PIN_S(lNC); PIN_S(lNI);
    }
    constexpr int mode = MODE;
    int n_steps = (MODE == NGW_MODE_ROLLOUT || MODE == NGW_MODE_ROLLOUT_ACT) ? a.n_steps : 1, autoreset = a.autoreset, horizon = a.horizon;
    PIN_S(n_steps); PIN_S(autoreset); PIN_S(horizon);
    const uint64_t env_global = (uint64_t)(a.env_base + e);
    uint32_t key0 = (uint32_t)a.action_seed, key1 = (uint32_t)(a.action_seed >> 32) ^ 0xA511E9B3u;
    uint64_t tt = (uint64_t)a.t0;
    PIN_S(key0); PIN_S(key1);
    // uniform step parameters, unpacked into scalars
    uint32_t brk_mask = U.brk_mask, ent_mask = U.ent_mask, rew_mask = U.rew_mask, brk2_mask = U.brk2_mask;
    int axe_required = U.axe_required, cost_chop = U.cost_chop, cost_jump = U.cost_jump, chop_reward = U.chop_reward;
    int n_actions = U.n_actions, reward_step = U.reward_step, reward_done = U.reward_done;
    int break_reward = U.break_reward;
    int cost_forward = U.cost_forward, cost_turn = U.cost_turn, cost_break = U.cost_break, cost_place = U.cost_place;
    int cost_extract = U.cost_extract, cost_select = U.cost_select, table_item = U.table_item, goal_item = U.goal_item;
    int place_item = U.place_item, place_near = U.place_near, n_entities = U.n_entities, ext_src = U.ext_src;
    int ext_near = U.ext_near, ext_out = U.ext_out, ext_qty = U.ext_qty, ext_consume = U.ext_consume;
    int ext_cost_ok = U.ext_cost_ok, axe_item = U.axe_item, axe_cost = U.axe_cost, axe_qty = U.axe_qty;
    int place_reward = U.place_reward, ext_reward = U.ext_reward, axe_reward = U.axe_reward;
    PIN_S(brk_mask); PIN_S(ent_mask); PIN_S(rew_mask); PIN_S(brk2_mask); PIN_S(axe_required); PIN_S(cost_chop); PIN_S(cost_jump); PIN_S(chop_reward); PIN_S(n_actions); PIN_S(reward_step); PIN_S(reward_done);
    PIN_S(break_reward); PIN_S(cost_forward); PIN_S(cost_turn); PIN_S(cost_break); PIN_S(cost_place);
    PIN_S(cost_extract); PIN_S(cost_select); PIN_S(table_item); PIN_S(goal_item); PIN_S(place_item); PIN_S(place_near);
    PIN_S(n_entities); PIN_S(ext_src); PIN_S(ext_near); PIN_S(ext_out); PIN_S(ext_qty); PIN_S(ext_consume);
    PIN_S(ext_cost_ok); PIN_S(axe_item); PIN_S(axe_cost); PIN_S(axe_qty); PIN_S(place_reward); PIN_S(ext_reward); PIN_S(axe_reward);
    // step-time novelty predicates (FireWall / FenceRestriction / Crate): only the EXT instantiation carries them
    int fire_item = 0, fire_reward = 0, fence_item = 0, fence_mode = 0, crate_item = 0;
    uint32_t crate_a0 = 0, crate_a1 = 0, crate_a2 = 0, nest = 0;
    if (EXT) {
        const NgwExtU& X = dspec->x;
        fire_item = X.fire_item; fire_reward = X.fire_reward; fence_item = X.fence_item; fence_mode = X.fence_mode;
        crate_item = X.crate_item; crate_a0 = X.crate_add[0]; crate_a1 = X.crate_add[1]; crate_a2 = X.crate_add[2]; nest = X.nest;
        PIN_S(nest); PIN_S(fire_item); PIN_S(fire_reward); PIN_S(fence_item); PIN_S(fence_mode); PIN_S(crate_item);
        PIN_S(crate_a0); PIN_S(crate_a1); PIN_S(crate_a2);
    }

    // fused rollouts: per-step rows and episode accumulators for whoever consumes the rollout (a learner, an evaluator)
    const bool rolling = MODE == NGW_MODE_ROLLOUT || MODE == NGW_MODE_ROLLOUT_ACT;
    int acc_ret = 0, acc_len = 0, acc_sum = 0, acc_eps = 0;
    if (rolling && a.acc && live) { acc_ret = a.acc[e]; acc_len = a.acc[a.n_pad + e]; acc_sum = a.acc[2 * a.n_pad + e]; acc_eps = a.acc[3 * a.n_pad + e]; }
    if (MODE == NGW_MODE_REFILL && blockIdx.x == 0 && tid == 0) {                  // what the host reads (without a sync) before the next refill
        uint32_t* const sh = dspec->nx.slow_host;                                 // (one report per refill: the launch of slot 0 makes it)
        if (sh && a.autoreset == 0) { sh[0] = dspec->nx.slow[0]; sh[1] = atomicAdd(dspec->nx.slow + 1, 1u) + 1u; }
    }
    STAMP(3);
    for (int t = 0; t < n_steps; t++, tt++) {
        bool do_reset = false;
        if (live && mode != NGW_MODE_DBG_COPY) {
            if (mode == NGW_MODE_RESET || mode == NGW_MODE_REFILL) {
                do_reset = action != 0;
            } else {
                if (mode == NGW_MODE_ROLLOUT) {
                    // action(t, env) = (w * A) >> 32 with w = word (t & 3) of philox(key = action_seed ^ tag; ctr = (t >> 2, env))
                    if (t == 0 || (tt & 3) == 0) {
                        const uint64_t tb = tt >> 2;
                        philox_block((uint32_t)tb, (uint32_t)(tb >> 32), (uint32_t)env_global, (uint32_t)(env_global >> 32),
                                     key0, key1, aw0, aw1, aw2, aw3);
                    }
                    const uint32_t q = (uint32_t)tt & 3u;
                    const uint32_t w = q == 0 ? aw0 : (q == 1 ? aw1 : (q == 2 ? aw2 : aw3));
                    action = (int)__umulhi(w, (uint32_t)n_actions);
                }
                if (mode == NGW_MODE_ROLLOUT_ACT) {
                    // the caller's action rows: this step's action was requested one step ago (or in the prologue), the next
                    // step's load is issued now and lands while this step runs
                    action = act_next;
                    if (t + 1 < n_steps) act_next = a.actions[(int64_t)(t + 1) * a.t0 + e];
                }
                if (action < 0 || action >= n_actions) {                         // reference: ValueError before any change (:236)
                    flags |= NGW_F_INVALID_ACTION;
                    reward = 0; ended = 0; info = 0;
                } else {
                    // ---------------- L0: independent LDS reads
                    const uint32_t* ad = lds_act + action * NGW_ACT_DW;
                    const uint32_t d0 = ad[0], d1 = ad[1], d2 = ad[2], d3 = ad[3], d4 = ad[4];
                    const int dr = (f == 0) ? -1 : (f == 1 ? 1 : 0), dc = (f == 2) ? -1 : (f == 3 ? 1 : 0);
                    const int fr = r + dr, fc = c + dc, fcell = fr * S + fc;
                    const int front = mp[fcell];                                   // block in front (:369-389)
                    // 4-neighbourhood of the front cell, only in-bounds cells count (is_block_in_front_next_to :391-411)
                    const bool okN = fr > 0, okS = fr < S - 1, okW = fc > 0, okE = fc < S - 1;
                    int nbN = mp[okN ? fcell - S : fcell], nbS = mp[okS ? fcell + S : fcell];
                    int nbW = mp[okW ? fcell - 1 : fcell], nbE = mp[okE ? fcell + 1 : fcell];
                                        const int fr2 = fr + dr, fc2 = fc + dc;                        // two cells ahead (Jump)
                    const bool ok2 = fr2 >= 0 && fr2 <= S - 1 && fc2 >= 0 && fc2 <= S - 1;
                    int front2 = mp[ok2 ? fr2 * S + fc2 : fcell];
                    int inv_place = inv[place_item], inv_ext = inv[ext_out], inv_axe = inv[axe_item];
                    // (LLVM sinks a load into the only branch that uses it, which would put one LDS latency back into
                    //  every divergent case; the empty asm makes each value "used" here, so the reads stay together)
                    { int p0 = (int)d0, p1 = (int)d1, p2 = (int)d2, p3 = (int)d3, p4 = (int)d4;
                      PIN_V(p0); PIN_V(p1); PIN_V(p2); PIN_V(p3); PIN_V(p4); }
                    // ---------------- L1: reads whose address came out of L0
                    const int kind = d0 & 255, aarg = (d0 >> 8) & 255, nin = (d0 >> 16) & 255;
                    const int in0 = d1 & 255, in1 = (d1 >> 8) & 255, in2 = (d1 >> 16) & 255, in3 = d1 >> 24;
                    const int out_item = d3 & 255;
                    int inv_front = inv[front];
                    int inv_arg = inv[min(aarg, K - 1)];
                    int iv0 = inv[in0], iv1 = inv[in1], iv2 = inv[in2], iv3 = inv[in3], inv_out = inv[out_item];
                    PIN_V(inv_front); PIN_V(inv_arg); PIN_V(iv0); PIN_V(iv1); PIN_V(iv2);
                    PIN_V(iv3); PIN_V(inv_out); PIN_V(front2); PIN_V(nbN); PIN_V(nbS);
                    PIN_V(nbW); PIN_V(nbE); PIN_V(inv_place); PIN_V(inv_ext); PIN_V(inv_axe);
                    // ---------------- register-only bodies
                    int rew = reward_step, result = 1, cost = 0, msg = NGW_MSG_NONE, arg = 0;   // :239-242
                    bool fence_twice = false;
                    switch (kind) {
                    case NGW_ACT_FORWARD:                                          // :244-257
                        if (front == 0) { r = fr; c = fc; } else { result = 0; msg = NGW_MSG_BLOCK_IN_PATH; }
                        cost = cost_forward;
                        break;
                    case NGW_ACT_LEFT:                                             // :258-268  N->W S->E W->S E->N
                        f = (0x0132 >> (f * 4)) & 3; cost = cost_turn;
                        break;
                    case NGW_ACT_RIGHT:                                            // :269-279  N->E S->W W->N E->S
                        f = (0x1023 >> (f * 4)) & 3; cost = cost_turn;
                        break;
                    case NGW_ACT_BREAK: {                                          // :280-294, axe: novelty_wrappers.py:144-183
                        cost = cost_break;
                        // Crate.step :1086-1089: the ingredients come first - unless the Crate wrapper sits BELOW FenceRestriction
                        // (NGW_XF_CRATE_IN_FENCE): then a restricted Break never reaches it
                        bool crate_now = EXT && crate_item && front == crate_item;
                        if (EXT && fence_mode && ((brk_mask >> front) & 1u)) {     // FenceRestriction.step :924-946
                            bool restricted = false;
                            if (front != fence_item) {
                                if (fence_mode == 1) {                             // medium: fence beside the AGENT, across its facing
                                    const int side = f <= 1 ? 1 : S;
                                    restricted = mp[r * S + c - side] == fence_item || mp[r * S + c + side] == fence_item;
                                } else {                                           // hard: any fence in the 3x3 around the block in front
                                    for (int dq = -S; dq <= S; dq += S)
                                        for (int dc2 = -1; dc2 <= 1; dc2++) restricted |= mp[fcell + dq + dc2] == fence_item;
                                }
                            }
                            if (restricted) {
                                result = 0; msg = NGW_MSG_FENCE_RESTRICTION;
                                if (nest & NGW_XF_CRATE_IN_FENCE) crate_now = false;
                            } else fence_twice = true;                             // the wrapper runs env.step() AND its own epilogue
                        }
                        if (crate_now)
                            for (int i = 1; i < K; i++) {
                                const uint32_t w = i < 8 ? crate_a0 : (i < 16 ? crate_a1 : crate_a2);
                                const int q = (int)((w >> (4 * (i & 7))) & 15u);
                                if (q) { const int nv = inv[i] + q; inv[i] = nv; gi[i] = nv; }
                            }
                        if (msg == NGW_MSG_FENCE_RESTRICTION) break;
                        if ((brk_mask >> front) & 1u) {
                            const bool axe_ok = axe_item && inv_axe >= 1 && sel == axe_item;
                            if (!axe_ok && axe_required) {                         // AxetoBreak*: novelty_wrappers.py:589-591
                                result = 0; msg = NGW_MSG_NEED_AXE; arg = axe_item;
                            } else {
                                mp[fcell] = 0; gm[fcell] = 0;
                                int nv = inv_front + 1 + (int)((brk2_mask >> front) & 1u);    // 2 under BreakIncrease
                                if (axe_ok) {
                                    nv = inv_front + axe_qty; rew = axe_reward; cost = axe_cost;
                                } else if (!axe_item && ((rew_mask >> front) & 1u)) rew = break_reward;
                                inv[front] = nv; gi[front] = nv;
                            }
                        } else { result = 0; msg = NGW_MSG_CANNOT_BREAK; arg = front; }
                        break;
                    }
                    case NGW_ACT_CHOP:                                             // AddChopAction.step, novelty_wrappers.py:1288-1308
                        cost = cost_chop;
                        if ((brk_mask >> front) & 1u) {
                            mp[fcell] = 0; gm[fcell] = 0;
                            inv[front] = inv_front + 2; gi[front] = inv_front + 2;
                            rew = chop_reward;
                        } else { result = 0; msg = NGW_MSG_CANNOT_CHOP; arg = front; }
                        break;
                    case NGW_ACT_JUMP:                                             // AddJumpAction.step :1362-1381 (cell between ignored)
                        if (ok2 && front2 == 0) { r = fr2; c = fc2; } else { result = 0; msg = NGW_MSG_BLOCK_IN_PATH; }
                        cost = cost_jump;
                        break;
                    case NGW_ACT_PLACE:                                            // :295-314
                        if (inv_place >= 1) {
                            if (front == 0) {
                                mp[fcell] = (int8_t)place_item; gm[fcell] = (int8_t)place_item;
                                inv[place_item] = inv_place - 1; gi[place_item] = inv_place - 1;
                                msg = NGW_MSG_PLACED; arg = place_item;
                                const int nr = place_near;
                                if ((okN && nbN == nr) || (okS && nbS == nr) || (okW && nbW == nr) || (okE && nbE == nr))
                                    rew = place_reward;
                            } else { result = 0; msg = NGW_MSG_ALREADY_EXISTS; arg = front; }
                        } else { result = 0; msg = NGW_MSG_NOT_IN_INVENTORY; }
                        cost = cost_place;
                        break;
                    case NGW_ACT_EXTRACT:                                          // :315-331 / bow_v1_env.py:293-304
                        cost = cost_extract;
                        if (front == ext_src) {
                            const int nr = ext_near;
                            if (!nr || (okN && nbN == nr) || (okS && nbS == nr) || (okW && nbW == nr) || (okE && nbE == nr)) {
                                inv[ext_out] = inv_ext + ext_qty; gi[ext_out] = inv_ext + ext_qty;
                                if (ext_consume) { mp[fcell] = 0; gm[fcell] = 0; }
                                rew = ext_reward; cost = ext_cost_ok;
                            } else { result = 0; msg = NGW_MSG_EXTRACT_NOT_NEAR; }
                        } else { result = 0; msg = NGW_MSG_EXTRACT_NO_SRC; }
                        break;
                    case NGW_ACT_CRAFT: {                                          // craft :413-474
                        const int nd0 = d2 & 255, nd1 = (d2 >> 8) & 255, nd2 = (d2 >> 16) & 255, nd3 = d2 >> 24;
                        const int missing = ((nin > 0 && iv0 < nd0) ? 1 : 0) | ((nin > 1 && iv1 < nd1) ? 2 : 0) |
                                            ((nin > 2 && iv2 < nd2) ? 4 : 0) | ((nin > 3 && iv3 < nd3) ? 8 : 0);   // :422-427
                        if (missing) {                                             // :430-440
                            result = 0; msg = NGW_MSG_MISSING_ITEMS; arg = (aarg << 8) | missing; cost = (d3 >> 16) & 255;
                        } else if (((d0 >> 24) & 1u) && front != table_item) {   // :444-453
                            result = 0; msg = NGW_MSG_NEED_TABLE; cost = d3 >> 24;
                        } else {                                                   // :455-474 (ids of a recipe are distinct)
                            rew = (int)(int8_t)(d4 >> 8);
                            if (nin > 0) { inv[in0] = iv0 - nd0; gi[in0] = iv0 - nd0; }
                            if (nin > 1) { inv[in1] = iv1 - nd1; gi[in1] = iv1 - nd1; }
                            if (nin > 2) { inv[in2] = iv2 - nd2; gi[in2] = iv2 - nd2; }
                            if (nin > 3) { inv[in3] = iv3 - nd3; gi[in3] = iv3 - nd3; }
                            const int nout = inv_out + (int)((d3 >> 8) & 255);
                            inv[out_item] = nout; gi[out_item] = nout;
                            cost = (int)(d4 & 255u); msg = NGW_MSG_CRAFTED; arg = out_item;
                        }
                        break;
                    }
                    case NGW_ACT_SELECT:                                           // :338-347
                        cost = cost_select;
                        if (inv_arg >= 1) sel = aarg; else { result = 0; msg = NGW_MSG_NOT_IN_INVENTORY; }
                        break;
                    default: break;
                    }
                    if (n_entities) {                                            // grab_entities :538-554 (3x3 incl. own cell)
                        for (int rr = r - 1; rr <= r + 1; rr++)
                            for (int cc = c - 1; cc <= c + 1; cc++) {
                                const int id = mp[rr * S + cc];
                                if (id != 0 && ((ent_mask >> id) & 1u)) {
                                    mp[rr * S + cc] = 0; gm[rr * S + cc] = 0;
                                    const int nv = inv[id] + 1;
                                    inv[id] = nv; gi[id] = nv;
                                }
                            }
                    }
                    int done = 0;                                                  // :354-357 (LDS ops of a wave are in order)
                    if (inv[goal_item] >= 1) { rew = reward_done; done = 1; }
                    if (EXT) {
                        if (fence_twice) {                                         // FenceRestriction.step :949-972: its own info + a
                            result = 1; cost = cost_break; msg = NGW_MSG_NONE; arg = 0;   // second step_count += 1 (:966)
                            steps += 1;
                        }
                        if (fire_item && !((nest & NGW_XF_FIRE_SKIP_BREAK) && kind == NGW_ACT_BREAK) &&
                            !((nest >> 8) && kind == NGW_ACT_CRAFT && (uint32_t)aarg + 1u == (nest >> 8))) {   // FireWall.step :1168-1189, after the wrapped step
                            const int ac = r * S + c;
                            if (mp[ac - S] == fire_item || mp[ac + S] == fire_item || mp[ac - 1] == fire_item || mp[ac + 1] == fire_item) {
                                rew = fire_reward; done = 1; msg = NGW_MSG_FIRE_WALL; arg = 0;
                            }
                        }
                    }
                    steps += 1;                                                    // :362
                    reward = rew; ended = done;
                    info = (uint32_t)result | ((uint32_t)done << 1) | ((uint32_t)cost << 2) | ((uint32_t)msg << 8) |
                           ((uint32_t)arg << 16);
                    if (autoreset && (done || (horizon > 0 && steps >= horizon))) {   // same-step autoreset
                        do_reset = true; ended = 1;
                    }
                }
            }
            if (do_reset) {                                                        // cold path, out of line
                episode++;
                uint32_t rr = new_episode(dspec, (LDS_AS int8_t*)mp, (LDS_AS int32_t*)inv, (LDS_AS uint32_t*)cand,
                                          (const LDS_AS uint8_t*)(lds_act + NGW_MAX_ACTIONS * NGW_ACT_DW),
                                          (LDS_AS uint16_t*)(lds + a.off_perm), env_global, e, episode, mode != NGW_MODE_REFILL,
                                          mode != NGW_MODE_RESET);   // (an explicit reset that finds nothing prepared is not a miss)
                do_reset = !(rr & NGW_F_ROWS_STORED);                              // from here on: "the wave must store its chunk"
                rr &= ~(uint32_t)NGW_F_ROWS_STORED;
                if (mode == NGW_MODE_REFILL && (rr & 0xFFu)) { rr &= ~0xFFu; episode = nx_old; }   // failed placement: leave the row
                                                                                   // stale, the real reset raises the flag
                flags |= rr & 0xFFu;
                r = (int)((rr >> 8) & 0xFFu); c = (int)((rr >> 16) & 0xFFu); f = (int)(rr >> 24);
                sel = 0; steps = 0;
            }
        }
        if (t == 0) STAMP(4);
        // ---- a reset rewrote whole maps / inventory rows in LDS: store the wave's chunk back with coalesced 16-B pieces
        //      (wave-uniform decision; lanes that only stepped have already written their few changed bytes through)
        if (mode == NGW_MODE_DBG_COPY || __any(do_reset)) {
            __syncthreads();
            for (int base = 0; base < npieces; base += EPB * PB) {
                pieces_lds<false, MAPMODE>(buf, a, lds_map, base, npieces, tid);
#pragma unroll
                for (int j = 0; j < PB; j++)
                    if (base + tid + EPB * j < npieces) gmap[base + EPB * j] = buf[j];
            }
            inv_lds<false>(iq, a, lds_inv, tid);
#pragma unroll
            for (int j = 0; j < IQ; j++) { if (j * EPB < 16 * K && tid + EPB * j < 16 * K) ginv[EPB * j] = iq[j]; }
            __syncthreads();
        }
        if (live) {
            gloc->x = r; gloc->y = c;
            *gfac = f;
            if (mode != NGW_MODE_RESET && mode != NGW_MODE_REFILL) {
                *grew = reward;
                *gdone = (uint8_t)ended;
                *ginfo = info;
            }
            if (rolling) {
                if (a.row_reward) a.row_reward[(int64_t)t * a.row_stride + e] = reward;
                if (a.row_done) a.row_done[(int64_t)t * a.row_stride + e] = (uint8_t)ended;
                acc_ret += reward; acc_len += 1;
                if (ended) { acc_sum += acc_ret; acc_eps += 1; acc_ret = 0; acc_len = 0; }
            }
        }
        if (LIDAR) {
            // ---- fused LidarInFront observation of the state this step produced (observation_wrappers.py:67-78)
            const int L = a.lidar_len, LB = lB, LR = lR, LNC = lNC, LNI = lNI;
            u32x4* t4 = reinterpret_cast<u32x4*>(lds + a.off_ltile);
            __syncthreads();
            for (int p = tid; p < 16 * L; p += EPB) t4[p] = u32x4{0u, 0u, 0u, 0u};
            __syncthreads();
            if (live) {
                const int16_t* toff = reinterpret_cast<const int16_t*>(lds + a.off_ltab);
                const uint8_t* chan_of_item = reinterpret_cast<const uint8_t*>(lds + a.off_ltab + 4 * LIDAR_TAB16);
                const uint8_t* inv_item = chan_of_item + NGW_MAX_ITEMS;
                int32_t* row = reinterpret_cast<int32_t*>(lds + a.off_ltile) + tid * L;
                lidar_march(mp + r * S + c, f, LB, LR, LNC, toff, chan_of_item, row);
                for (int j = 0; j < LNI; j++) row[LB * LNC + j] = inv[inv_item[j]];           // :74-75, inventory is in LDS
            }
            __syncthreads();
            lidar_store((const LDS_AS uint32_t*)(lds + a.off_ltile), glid, L, a.l_i16 != 0, tid);
        }
    }
    if (live) {
        if (MODE != NGW_MODE_REFILL) {
            a.b.selected[e] = (uint8_t)sel;
            a.b.step_count[e] = steps;
        }
        a.b.episode[e] = episode;
        if (rolling && a.acc) { a.acc[e] = acc_ret; a.acc[a.n_pad + e] = acc_len; a.acc[2 * a.n_pad + e] = acc_sum; a.acc[3 * a.n_pad + e] = acc_eps; }
    }
    if (flags) atomicOr(a.b.flags, flags);
    raise_host_flags(a.b.flags_host, flags);
    if (MODE == NGW_MODE_STEP || MODE == NGW_MODE_RESET) {
        if (a.seq) mirror_wave(dspec, a.b, a.S2, a.K, a.n);
        signal_host_seq(a.b.flags_host, a.seq);
    }
#ifdef NGW_STAMPS
    STAMP(5);
    __builtin_amdgcn_s_waitcnt(0);                                                 // every store acknowledged
    STAMP(6);
    STAMP_FLUSH(a);
#endif
}

#include "ngw_lean.inc"
#include "ngw_reset.inc"

// ---------------------------------------------------------------- LidarInFront observation kernel
// observation_wrappers.py:32-80.  Same wave = 64 envs decomposition and the same coalesced map staging as the step
// kernel; every lane marches its env's beams on the LDS map along the host-computed integer offsets (4 ranges per
// round: one 8-byte table read + 4 independent cell reads), builds its observation row in an LDS tile with an odd
// stride, and the wave then writes the tile out as one contiguous block of dwords.
template <int MAPMODE>
__global__ void __launch_bounds__(NGW_EPB) ngw_lidar_kernel(const NgwLidarDev* __restrict__ cfg, const NgwLaunch a,
                                                             int32_t* __restrict__ out, int L, uint32_t off_map, uint32_t off_tab,
                                                             uint32_t off_tile) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const int64_t env0 = (int64_t)blockIdx.x * EPB;
    const int64_t e = env0 + tid;
    const bool live = e < a.n;
    const int S = a.S, K = a.K, npieces = 4 * a.S2;
    uint32_t* lds_map = lds + off_map;
    // ray offset table (8 KiB = 512 pieces of 16 B, 8 per lane) + the two item tables right behind it
    constexpr int TAB16 = LIDAR_TAB16;
    u32x4 tb[8];
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(cfg->off);
#pragma unroll
        for (int j = 0; j < 8; j++) tb[j] = src[tid + EPB * j];
    }
    uint32_t it = 0;
    if (tid < 2 * NGW_MAX_ITEMS / 4) it = reinterpret_cast<const uint32_t*>(cfg->chan_of_item)[tid];
    u32x4 buf[PB];
    const u32x4* gin = reinterpret_cast<const u32x4*>(a.b.map + env0 * a.S2);
    pieces_load(buf, gin, 0, npieces, tid);
    int r = 1, c = 1, f = 0;
    if (live) {
        const int2 rc = reinterpret_cast<const int2*>(a.b.loc)[e];
        r = rc.x; c = rc.y;
        f = a.b.facing[e];
    }
    const int B = cfg->num_beams, R = cfg->max_range, NC = cfg->n_chan, NI = cfg->n_inv;
    {
        u32x4* dst = reinterpret_cast<u32x4*>(lds + off_tab);
#pragma unroll
        for (int j = 0; j < 8; j++) dst[tid + EPB * j] = tb[j];
        if (tid < 2 * NGW_MAX_ITEMS / 4) lds[off_tab + 4 * TAB16 + tid] = it;
        u32x4* t4 = reinterpret_cast<u32x4*>(lds + off_tile);                        // zero the observation tile
        for (int p = tid; p < 16 * L; p += EPB) t4[p] = u32x4{0u, 0u, 0u, 0u};
    }
    pieces_lds<true, MAPMODE>(buf, a, lds_map, 0, npieces, tid);
    for (int base = EPB * PB; base < npieces; base += EPB * PB) {
        pieces_load(buf, gin, base, npieces, tid);
        pieces_lds<true, MAPMODE>(buf, a, lds_map, base, npieces, tid);
    }
    __syncthreads();
    const int8_t* mp = reinterpret_cast<const int8_t*>(lds_map) + tid * a.MS;
    const int16_t* toff = reinterpret_cast<const int16_t*>(lds + off_tab);
    const uint8_t* chan_of_item = reinterpret_cast<const uint8_t*>(lds + off_tab + 4 * TAB16);
    const uint8_t* inv_item = chan_of_item + NGW_MAX_ITEMS;
    int32_t* row = reinterpret_cast<int32_t*>(lds + off_tile) + tid * L;
    // inventory tail (:74-75): issue every (scattered, per-lane) global load NOW, all at once - their latency hides under
    // the march; the values go into the tile afterwards.  A rolled loop here serialises NI dependent HBM round trips.
    int32_t ivals[NGW_MAX_ITEMS];
    if (live) {
        const int32_t* gi = a.b.inv + e * K;
#pragma unroll
        for (int j = 0; j < NGW_MAX_ITEMS; j++) ivals[j] = (j < NI) ? gi[inv_item[j]] : 0;
    }
    if (live) {
        lidar_march(mp + r * S + c, f, B, R, NC, toff, chan_of_item, row);
#pragma unroll
        for (int j = 0; j < NGW_MAX_ITEMS; j++)
            if (j < NI) row[B * NC + j] = ivals[j];
    }
    __syncthreads();
    // the wave's 64 rows are one contiguous block of 64 * L dwords in HBM = 16 * L pieces of 16 B
    lidar_store((const LDS_AS uint32_t*)(lds + off_tile), (GLOBAL_AS uint32_t*)(reinterpret_cast<uint32_t*>(out) + ((env0 * L) >> (a.l_i16 ? 1 : 0))), L,
                a.l_i16 != 0, tid);   // out is padded to n_pad rows
}

// Delta refresh of a host mirror (NgwDiff, ngw_step_host): region blockIdx.y is compared, 16 bytes at a time, with the shadow
// copy of what the host holds; only pieces that differ are stored - to the shadow, and straight into the host's page-locked
// mirror across PCIe (mapped memory).  A step changes a few bytes of an env's map / inventory, so this moves ~1 % of what a
// full copy moves.  Regions are 16-byte aligned on all three sides; a tail shorter than 16 bytes goes by bytes.
__global__ __launch_bounds__(256) void ngw_diff_kernel(const NgwDiff p) {
    const int r = blockIdx.y;
    const uint64_t nb = p.nbytes[r];
    const uint8_t* c = p.cur[r];
    uint8_t* s = p.shadow[r];
    uint8_t* h = p.host[r];
    const uint64_t stride = (uint64_t)gridDim.x * 256u, t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t n16 = nb >> 4;
    for (uint64_t i = t; i < n16; i += stride) {
        const u32x4 a = reinterpret_cast<const u32x4*>(c)[i], b = reinterpret_cast<const u32x4*>(s)[i];
        if (a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w) {
            reinterpret_cast<u32x4*>(s)[i] = a;
            reinterpret_cast<u32x4*>(h)[i] = a;
        }
    }
    for (uint64_t i = (n16 << 4) + t; i < nb; i += stride)
        if (c[i] != s[i]) { s[i] = c[i]; h[i] = c[i]; }
}

// Region copies (NgwPack): region blockIdx.y, grid-stride over 16-byte pieces; tails and unaligned regions go by bytes.
// The destination may be host memory mapped into the GPU's address space (the stores then travel over PCIe).
__global__ __launch_bounds__(256) void ngw_pack_kernel(const NgwPack p) {
    const int r = blockIdx.y;
    const uint64_t nb = p.nbytes[r];
    const uint8_t* s = p.src[r];
    uint8_t* d = p.dst[r];
    const uint64_t stride = (uint64_t)gridDim.x * 256u, t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if ((((uintptr_t)s | (uintptr_t)d) & 15u) == 0) {
        const uint64_t n16 = nb >> 4;
        for (uint64_t i = t; i < n16; i += stride) reinterpret_cast<u32x4*>(d)[i] = reinterpret_cast<const u32x4*>(s)[i];
        for (uint64_t i = (n16 << 4) + t; i < nb; i += stride) d[i] = s[i];
    } else {
        for (uint64_t i = t; i < nb; i += stride) d[i] = s[i];
    }
}

// AgentMap (reference observation_wrappers.py:104-121): the (2V+1) x (2V+1) window of the map centred on the agent, 0 outside
// the map.  HBM-bound byte gather: one lane produces 4 consecutive output bytes (one coalesced dword store); the map reads
// hit each env's 100-B row image, which one wave covers with a handful of cache lines.
__global__ __launch_bounds__(256) void ngw_agent_view_kernel(const int8_t* __restrict__ map, const int32_t* __restrict__ loc,
                                                             uint32_t* __restrict__ out, uint32_t n_dwords, int S, int V,
                                                             uint32_t magicW) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_dwords) return;
    const uint32_t W = 2u * (uint32_t)V + 1u, WW = W * W;
    uint32_t idx = t * 4u;
    uint32_t e = idx / WW;                           // one full division per lane; the rest are small-operand magics
    uint32_t rem = idx - e * WW;
    uint32_t word = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t r = __umulhi(rem, magicW), c = rem - r * W;
        const int mr = loc[2 * (size_t)e] + (int)r - V, mc = loc[2 * (size_t)e + 1] + (int)c - V;
        uint32_t v = 0;
        if ((unsigned)mr < (unsigned)S && (unsigned)mc < (unsigned)S) v = (uint8_t)map[(size_t)e * (S * S) + mr * S + mc];
        word |= v << (8 * j);
        if (++rem == WW) { rem = 0; ++e; }
    }
    out[t] = word;
}

}  // namespace

extern "C" hipError_t ngw_pack_launch(const NgwPack* p, hipStream_t stream) {
    if (p->n_regions < 1) return hipSuccess;
    uint64_t most = 0;
    for (int r = 0; r < p->n_regions; r++) most = p->nbytes[r] > most ? p->nbytes[r] : most;
    uint64_t blocks = (most / 16u + 255u) / 256u;                          // one 16-byte piece per thread, up to 2048 blocks per region
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ngw_pack_kernel, dim3((unsigned)blocks, (unsigned)p->n_regions), dim3(256), 0, stream, *p);
    return hipGetLastError();
}

extern "C" hipError_t ngw_diff_launch(const NgwDiff* p, hipStream_t stream) {
    if (p->n_regions < 1) return hipSuccess;
    uint64_t most = 0;
    for (int r = 0; r < p->n_regions; r++) most = p->nbytes[r] > most ? p->nbytes[r] : most;
    uint64_t blocks = (most / 16u + 255u) / 256u;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ngw_diff_kernel, dim3((unsigned)blocks, (unsigned)p->n_regions), dim3(256), 0, stream, *p);
    return hipGetLastError();
}

extern "C" hipError_t ngw_agent_view_launch(const int8_t* map, const int32_t* loc, uint32_t* out, uint32_t n_dwords, int S, int V,
                                            hipStream_t stream) {
    const uint32_t W = 2u * (uint32_t)V + 1u;
    const uint32_t magicW = (uint32_t)((0x100000000ull + W - 1) / W);     // exact for operands < W * W
    hipLaunchKernelGGL(ngw_agent_view_kernel, dim3((n_dwords + 255u) / 256u), dim3(256), 0, stream, map, loc, out, n_dwords, S, V,
                       magicW);
    return hipGetLastError();
}

extern "C" hipError_t ngw_lidar_launch(const NgwLidarDev* cfg, const NgwLaunch* a, int map_mode, int32_t* out, int L,
                                       uint32_t off_map, uint32_t off_tab, uint32_t off_tile, unsigned grid, size_t lds_bytes,
                                       hipStream_t stream) {
    const void* fn = map_mode == NGW_MAP_STRAIGHT ? reinterpret_cast<const void*>(ngw_lidar_kernel<NGW_MAP_STRAIGHT>)
                     : map_mode == NGW_MAP_DWORD  ? reinterpret_cast<const void*>(ngw_lidar_kernel<NGW_MAP_DWORD>)
                                                  : reinterpret_cast<const void*>(ngw_lidar_kernel<NGW_MAP_BYTE>);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    switch (map_mode) {
    case NGW_MAP_STRAIGHT:
        hipLaunchKernelGGL(ngw_lidar_kernel<NGW_MAP_STRAIGHT>, dim3(grid), dim3(NGW_EPB), lds_bytes, stream, cfg, *a, out, L, off_map, off_tab, off_tile);
        break;
    case NGW_MAP_DWORD:
        hipLaunchKernelGGL(ngw_lidar_kernel<NGW_MAP_DWORD>, dim3(grid), dim3(NGW_EPB), lds_bytes, stream, cfg, *a, out, L, off_map, off_tab, off_tile);
        break;
    default:
        hipLaunchKernelGGL(ngw_lidar_kernel<NGW_MAP_BYTE>, dim3(grid), dim3(NGW_EPB), lds_bytes, stream, cfg, *a, out, L, off_map, off_tab, off_tile);
    }
    return hipGetLastError();
}

namespace {

template <int MAPMODE, int MODE, bool LIDAR, bool EXT>
hipError_t launch_one(const NgwDevSpec* dspec, const NgwLaunch* a, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    // CDNA4 has 160 KiB of LDS per CU; anything above the 64 KiB default needs an explicit opt-in per device.
    static size_t lds_opt_in[64] = {0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (lds_bytes > 64 * 1024 && dev < 64 && lds_bytes > lds_opt_in[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(ngw_kernel<MAPMODE, MODE, LIDAR, EXT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
        if (e != hipSuccess) return e;
        lds_opt_in[dev] = lds_bytes;
    }
    hipLaunchKernelGGL((ngw_kernel<MAPMODE, MODE, LIDAR, EXT>), dim3(grid), dim3(NGW_EPB), lds_bytes, stream, dspec, *a);
    return hipGetLastError();
}

template <int MAPMODE, bool LIDAR, bool EXT>
static hipError_t launch_mode(const NgwDevSpec* dspec, const NgwLaunch* a, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    switch (a->mode) {
    case NGW_MODE_STEP: return launch_one<MAPMODE, NGW_MODE_STEP, LIDAR, EXT>(dspec, a, grid, lds_bytes, stream);
    case NGW_MODE_RESET: return launch_one<MAPMODE, NGW_MODE_RESET, LIDAR, EXT>(dspec, a, grid, lds_bytes, stream);
    case NGW_MODE_ROLLOUT: return launch_one<MAPMODE, NGW_MODE_ROLLOUT, LIDAR, EXT>(dspec, a, grid, lds_bytes, stream);
    case NGW_MODE_ROLLOUT_ACT: return launch_one<MAPMODE, NGW_MODE_ROLLOUT_ACT, LIDAR, EXT>(dspec, a, grid, lds_bytes, stream);
    default: break;
    }
    if (a->mode == NGW_MODE_REFILL) return launch_one<MAPMODE, NGW_MODE_REFILL, false, false>(dspec, a, grid, lds_bytes, stream);
    if (LIDAR || EXT) return hipErrorInvalidValue;
    if (a->mode == NGW_MODE_DBG_COPY) return launch_one<MAPMODE, NGW_MODE_DBG_COPY, false, false>(dspec, a, grid, lds_bytes, stream);
    return launch_one<MAPMODE, NGW_MODE_DBG_NOP, false, false>(dspec, a, grid, lds_bytes, stream);
}

template <int MAPMODE>
static hipError_t launch_feat(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    switch (feat & 3) {
    case 0: return launch_mode<MAPMODE, false, false>(dspec, a, grid, lds_bytes, stream);
    case 1: return launch_mode<MAPMODE, true, false>(dspec, a, grid, lds_bytes, stream);
    case 2: return launch_mode<MAPMODE, false, true>(dspec, a, grid, lds_bytes, stream);
    default: return launch_mode<MAPMODE, true, true>(dspec, a, grid, lds_bytes, stream);
    }
}

__global__ void ngw_nop_kernel(const NgwDevSpec* dspec, const NgwLaunch a) {}

}  // namespace

extern "C" hipError_t ngw_reset_fast_launch(const NgwDevSpec* dspec, const NgwResetFast* a, int nw, int subset, unsigned grid, size_t lds_bytes,
                                            hipStream_t stream) {
    const void* fn = nullptr;
#define NGW_RF(NWV, SV) if (nw == NWV && (subset != 0) == SV) fn = reinterpret_cast<const void*>(ngw_reset_fast<NWV, SV>)
    NGW_RF(2, false); NGW_RF(2, true); NGW_RF(8, false); NGW_RF(8, true); NGW_RF(0, false); NGW_RF(0, true);
#undef NGW_RF
    if (!fn) return hipErrorInvalidValue;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    void* args[] = {const_cast<NgwDevSpec**>(&dspec), const_cast<NgwResetFast*>(a)};
    return hipLaunchKernel(fn, dim3(grid), dim3(NGW_EPB), args, lds_bytes, stream);
}

extern "C" hipError_t ngw_launch(const NgwDevSpec* dspec, const NgwLaunch* a, int map_mode, int feat, unsigned grid,
                                 size_t lds_bytes, hipStream_t stream) {
    if (a->mode >= 10 && a->mode <= 12) {       // diagnostics: empty kernels with other workgroup shapes over the same lanes
        const unsigned tpb = a->mode == 10 ? 256 : (a->mode == 11 ? 1024 : 128);
        hipLaunchKernelGGL(ngw_nop_kernel, dim3(grid * NGW_EPB / tpb), dim3(tpb), a->mode == 12 ? lds_bytes * 2 : 0, stream, dspec, *a);
        return hipGetLastError();
    }
    // Everything but the fused LidarInFront epilogue steps through the lean kernels; the wrapper predicates (feat & 2) are a
    // template flag of the same body.
#define NGW_LEAN_EXT(CALL_F, CALL_T) ((feat & 2) ? CALL_T : CALL_F)
    if ((feat & 4) && a->mode == NGW_MODE_STEP && (!(feat & 1) || !(feat & 8))) {   // one step (the lidar epilogue needs the staged form)
        if (feat & 8)                                               // no-stage (big maps)
            return NGW_LEAN_EXT((launch_lean<NGW_MAP_STRAIGHT, false, false, false>(dspec, a, grid, lds_bytes, stream)),
                                (launch_lean<NGW_MAP_STRAIGHT, false, true, false>(dspec, a, grid, lds_bytes, stream)));
#define NGW_LEAN_STEP(MM) ((feat & 1) ? NGW_LEAN_EXT((launch_lean<MM, true, false, true>(dspec, a, grid, lds_bytes, stream)),   \
                                                     (launch_lean<MM, true, true, true>(dspec, a, grid, lds_bytes, stream)))    \
                                      : NGW_LEAN_EXT((launch_lean<MM, true, false, false>(dspec, a, grid, lds_bytes, stream)),  \
                                                     (launch_lean<MM, true, true, false>(dspec, a, grid, lds_bytes, stream))))
        switch (map_mode) {
        case NGW_MAP_STRAIGHT: return NGW_LEAN_STEP(NGW_MAP_STRAIGHT);
        case NGW_MAP_DWORD: return NGW_LEAN_STEP(NGW_MAP_DWORD);
        default: return NGW_LEAN_STEP(NGW_MAP_BYTE);
        }
#undef NGW_LEAN_STEP
    }
    if ((feat & 4) && !(feat & 1) && (a->mode == NGW_MODE_ROLLOUT || a->mode == NGW_MODE_ROLLOUT_ACT)) {   // fused rollout
        const bool sup = a->mode == NGW_MODE_ROLLOUT_ACT;
#define NGW_LEAN_RO(MM) (sup ? NGW_LEAN_EXT((launch_rollout_lean<MM, true, false>(dspec, a, grid, lds_bytes, stream)),   \
                                            (launch_rollout_lean<MM, true, true>(dspec, a, grid, lds_bytes, stream)))    \
                             : NGW_LEAN_EXT((launch_rollout_lean<MM, false, false>(dspec, a, grid, lds_bytes, stream)),  \
                                            (launch_rollout_lean<MM, false, true>(dspec, a, grid, lds_bytes, stream))))
        switch (map_mode) {
        case NGW_MAP_STRAIGHT: return NGW_LEAN_RO(NGW_MAP_STRAIGHT);
        case NGW_MAP_DWORD: return NGW_LEAN_RO(NGW_MAP_DWORD);
        default: return NGW_LEAN_RO(NGW_MAP_BYTE);
        }
#undef NGW_LEAN_RO
    }
#undef NGW_LEAN_EXT
    switch (map_mode) {
    case NGW_MAP_STRAIGHT: return launch_feat<NGW_MAP_STRAIGHT>(dspec, a, feat, grid, lds_bytes, stream);
    case NGW_MAP_DWORD: return launch_feat<NGW_MAP_DWORD>(dspec, a, feat, grid, lds_bytes, stream);
    default: return launch_feat<NGW_MAP_BYTE>(dspec, a, feat, grid, lds_bytes, stream);
    }
}
