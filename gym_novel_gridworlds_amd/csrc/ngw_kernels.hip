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

// The library is built from this file ELEVEN times, in parallel (Makefile): -DNGW_PART=n keeps the launchers - and with them the
// kernel instantiations - of one part; without NGW_PART (make asm) everything is in one unit.
//   0: ngw_launch + the general kernel   1 / 6 / 7: step kernels per map addressing mode (1 also holds the in-place ones)
//   2 / 3 / 4: rollout kernels per map addressing mode   5: new-episode (reset_fast), lidar, diff / wire / pack / agent-view kernels
//   8: the bit-row (boards) lidar: in-place step kernels with the O(1) observation, ngw_boards_kernel, ngw_lidar_boards_kernel
#ifdef NGW_PART
#define NGW_HAS(p) (NGW_PART == (p))
#else
#define NGW_HAS(p) 1
#endif

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
#define STAMP_STRIDE 32        /* u64 per workgroup: [0, 8) chip clock, [8, 16) shader cycles of the kernel's stamps, [16, 32) shader cycles inside helpers (STAMP_SUB) */
#define STAMP_FLUSH(a) do { if ((a).stamps && threadIdx.x == 0) { for (int i_ = 0; i_ < 8; i_++) { (a).stamps[(size_t)blockIdx.x * STAMP_STRIDE + i_] = st_rt[i_]; (a).stamps[(size_t)blockIdx.x * STAMP_STRIDE + 8 + i_] = st_cy[i_]; } } } while (0)
#define STAMP_SUB(a, i) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F) /* lgkmcnt(0): LDS work up to here is done */; const uint64_t c_ = __builtin_amdgcn_s_memtime(); if ((a).stamps && threadIdx.x == 0) (a).stamps[(size_t)blockIdx.x * STAMP_STRIDE + 16 + (i)] = c_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_SUBV(a, i) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0) /* every counter: global loads have landed too */; const uint64_t c_ = __builtin_amdgcn_s_memtime(); if ((a).stamps && threadIdx.x == 0) (a).stamps[(size_t)blockIdx.x * STAMP_STRIDE + 16 + (i)] = c_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH(a)
#define STAMP_SUB(a, i)
#define STAMP_SUBV(a, i)
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
constexpr int PHILOX_RING = 32;                                                   // (16: a C2 reset of every env 23.5 -> 30 us - too many lanes need a second fill)

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

// End of a step launch, the parts that almost never run, with every argument read from the HBM blob's copy of the launch block
// INSIDE the uniform branches (the hot path keeps none of these pointers in registers): sticky error flags (an invalid action id, a
// placement that cannot succeed), and - single-wavefront handles stepped by ngw_step_host only (seq != 0) - the host mirror and the
// sequence word the host polls.
__device__ __forceinline__ void step_signals(const NgwDevSpec* dspec, uint32_t flags, uint32_t seq) {
    if (__any(flags != 0)) {
        const GLOBAL_AS NgwLaunch* lp = (const GLOBAL_AS NgwLaunch*)&dspec->lp;
        uint32_t* const fl = lp->b.flags; uint32_t* const fh = lp->b.flags_host;
        if (flags) atomicOr(fl, flags);
        raise_host_flags(fh, flags);
    }
    if (seq) {                                                                     // (uniform)
        const GLOBAL_AS NgwLaunch* lp = (const GLOBAL_AS NgwLaunch*)&dspec->lp;
        NgwBufs b;
        b.map = lp->b.map; b.loc = lp->b.loc; b.facing = lp->b.facing; b.inv = lp->b.inv; b.selected = lp->b.selected; b.step_count = lp->b.step_count;
        b.reward = lp->b.reward; b.done = lp->b.done; b.info = lp->b.info; b.flags_host = lp->b.flags_host;
        const int S2 = lp->S2, K = lp->K;
        const int64_t n = lp->n;
        mirror_wave(dspec, b, S2, K, n);
        signal_host_seq(b.flags_host, seq);
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
    {   // :129-130 wall ring around air: the interior's air row by row, then the ring's 4 (S - 1) cells - no comparison per cell (a C2 reset of
        // every env 23.5 -> 21.6 us, one lane per wave 30.3 -> 27 us; FenceRestriction 53.6 -> 51.3 us: profiles/r05_ab.md)
        for (int r = 1; r < S - 1; r++)
            for (int c = 1; c < S - 1; c++) mp[r * S + c] = (int8_t)0;
        for (int c = 0; c < S; c++) { mp[c] = (int8_t)wall_item; mp[(S - 1) * S + c] = (int8_t)wall_item; }
        for (int r = 1; r < S - 1; r++) { mp[r * S] = (int8_t)wall_item; mp[r * S + S - 1] = (int8_t)wall_item; }
    }
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

// The same choice INLINED into its one call site in a step or rollout kernel, Philox words from register blocks (one
// instantiation of the placement loop).  A real call costs those kernels a stack frame (scratch memory enabled for every dispatch),
// the callee's 242 VGPRs in their own allocation and, around the call site, the ABI's save / restore of the live scalars -
// v_writelane / v_readlane traffic in kernels whose hot path is bound by instruction issue.  The general new-episode kernel
// (explicit resets, refills), where the placement loop IS the work, keeps the out-of-line form with the LDS word ring.
__device__ __forceinline__ uint32_t new_episode_inline(const NgwDevSpec* dspec, LDS_AS int8_t* mp, LDS_AS int32_t* inv, LDS_AS uint32_t* cand,
                                                       const LDS_AS uint8_t* place_seq, LDS_AS uint16_t* perm_lds, uint64_t env_global,
                                                       int64_t env_local, uint32_t episode, bool may_consume) {
    const GLOBAL_AS NgwResetU* rp = (const GLOBAL_AS NgwResetU*)&dspec->ru;        // both blobs requested together
    const GLOBAL_AS NgwNx* np = (const GLOBAL_AS NgwNx*)&dspec->nx;
    NgwResetU ru; NgwNx nx;
    ru.perm = rp->perm; ru.map = rp->map; ru.inv = rp->inv; ru.n_pad = rp->n_pad; ru.seed = rp->seed; ru.S = rp->S;
    ru.S2 = rp->S2; ru.K = rp->K; ru.CW = rp->CW; ru.perm_lds = rp->perm_lds; ru.magicS = rp->magicS;
    const GLOBAL_AS uint32_t* rsw = (const GLOBAL_AS uint32_t*)&rp->wall_item;
    const uint32_t rs0 = rsw[0], rs1 = rsw[1], rs2 = rsw[2], rs3 = rsw[3], pw0 = rsw[4], pw1 = rsw[5], pw2 = rsw[6], pw3 = rsw[7];
    nx.map = np->map; nx.loc = np->loc; nx.facing = np->facing; nx.inv = np->inv; nx.episode = np->episode; nx.slow = np->slow;
    if (may_consume && nx.episode) {
        const int64_t row = (int64_t)(episode & (uint32_t)np->dmask) * np->stride + env_local;   // the slot of this episode
        if (((const GLOBAL_AS uint32_t*)nx.episode)[row] == episode)
            return consume_lane(nx, mp, inv, (GLOBAL_AS int8_t*)ru.map + env_local * ru.S2, (GLOBAL_AS int32_t*)ru.inv + env_local * ru.K,
                                row, ru.S2, ru.K) | NGW_F_ROWS_STORED;
        atomicAdd(nx.slow, 1u);                                                    // a stale row inside a step: the host shortens the refill cadence
    }
    const ResetArgs a = {dspec, ru.perm, ru.n_pad, ru.seed, ru.S, ru.S2, ru.K, ru.CW, ru.perm_lds, ru.magicS, rs0, rs1, rs2, rs3, pw0, pw1, pw2, pw3};
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

// ---------------------------------------------------------------- LidarInFront observation (shared by every kernel that produces it)
// observation_wrappers.py:32-78 on the LDS map.  The wave's 64 rows are built in an LDS tile that is the exact HBM image of
// those rows in the chosen format (NGW_LFMT_*: int32, int16, or uint8 beam entries + an int16 inventory tail) and leave as one
// contiguous run of 16-byte pieces.  Two marches:
//   * WORLD FRAME (num_beams a multiple of 4 - the reference's default 8, and 12, 16; ngw_lidar_configure verifies it entry by
//     entry): the four facings shoot the same rays, numbered from a different start, so the cell offsets of a ray are the same
//     for EVERY lane of the wave - they arrive through scalar loads (no per-lane table reads, no unpacking) and a cell costs
//     one address add and one byte read; all eight rays advance four ranges per round (32 reads in flight) and the wave stops
//     when no lane has an open ray left;
//   * per-lane table (any other beam count): flat int16 offsets [facing][beam][range-1] staged into LDS by the epilogue itself.
// A ray cannot leave the map before it hits the wall ring; cells read beyond the hit are ignored (the LDS layout keeps a
// guard on both sides of the maps for them), and beyond max_range a table repeats the last in-range cell, so a padded entry
// can never be the FIRST non-zero one.
#define CONST_AS __attribute__((address_space(4)))

// A wave alone on its SIMD issues one instruction every ~6 cycles whatever the instruction is (stamped: 1 150 instructions of march in
// 7 100 cycles, with either march), so what the march costs is its INSTRUCTION COUNT.  Per cell: an address add, a byte read and
// three quarters of a pack; per ray and round, three instructions that remember the first non-zero word; the hit itself is resolved
// once per ray, after the rounds.
//
// Four cells of one ray as one dword, bytes in range order: four byte reads and three shift-ors.  (The d16 forms of the byte read -
// two reads filling the halves of one register, one shift-or per four cells - are no use on this part: with SRAM ECC on, a d16 load
// ZEROES the other half of its destination instead of preserving it, which is why the compiler never emits them here; tried through
// inline asm, every observation came out wrong.)
__device__ __forceinline__ uint32_t lidar_cells4(const LDS_AS uint8_t* ag, int o0, int o1, int o2, int o3) {
    const uint32_t c0 = ag[o0], c1 = ag[o1], c2 = ag[o2], c3 = ag[o3];
    return (c0 | (c1 << 8)) | ((c2 | (c3 << 8)) << 16);
}

// Eight rays, one round of four ranges each (k0 + 1 .. k0 + 4): `first` keeps a ray's first non-zero word, `kc` the round it came
// from.  Returns whether any of the eight is still open in this lane.
__device__ __forceinline__ bool lidar_round(const uint32_t (&word)[8], int k0, uint32_t (&first)[8], int (&kc)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const bool take = first[i] == 0;
        first[i] = take ? word[i] : first[i];
        kc[i] = take ? k0 : kc[i];
    }
    const uint32_t m = min(min(min(first[0], first[1]), min(first[2], first[3])), min(min(first[4], first[5]), min(first[6], first[7])));
    return m == 0;
}

// The hits of up to eight rays (observation_wrappers.py:59-64: the first non-air block = the lowest non-zero byte of the ray's first
// non-zero word) go into the row: the channel look-ups go out together (a ray without a hit looks up item 0 = air = no channel), a
// beam entry is a range <= 64, so ONE byte store at the entry's place serves every row format (the tile was zeroed, rows are
// little-endian; `sh` = log2 of the entry size), and a ray that reports nothing stores into the lane's dump byte behind the tile -
// no branch per ray.
__device__ __forceinline__ void lidar_hits(const uint32_t (&first)[8], const int (&kc)[8], int pos0 /* row entry of ray slot 0's beam: beam * NC */,
                                           int nb, int R, int NC, int wrap /* num_beams * NC */, int sh, const LDS_AS uint8_t* chan_of_item,
                                           LDS_AS uint8_t* rowp, int dump) {
    int hk[8], ch[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int q = (__ffs((int)(first[i] | 0x80000000u)) - 1) >> 3;
        hk[i] = kc[i] + q + 1;
        ch[i] = chan_of_item[(first[i] >> (8 * q)) & 255u];                        // (first == 0: item 0)
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int p = pos0 + i * NC;                                                     // consecutive ray slots are consecutive beams (mod num_beams)
        p -= p >= wrap ? wrap : 0;
        const bool hit = i < nb && ch[i] != 0 && hk[i] <= R;
        rowp[hit ? (p + ch[i] - 1) << sh : dump] = (uint8_t)hk[i];
    }
}

// The constant-offset march (NgwLaunch::l_world == 2): the reference's default 8 rays on an S x S map with S a compile-time
// constant.  What bounds a march is the LDS itself: every ray cell is a byte read at a per-lane address (the agents stand
// anywhere, so the 64 addresses of a read fall on the banks at random: ~3.5-way conflicts), and the four waves of a CU share one LDS
// pipe - stamped: ~25 cycles per read instruction, 104 of them in the table-driven marches = 2 600 of their 4 100 cycles, whatever
// the arithmetic around them looked like.  So this form issues FEWER reads:
//   * a diagonal ray advances by round(0.71 k) cells: ranges 1..11 visit the 8 diagonal cells d = 1..8, some twice; the first block
//     is the same either way, so rays are walked in GEOMETRIC steps d = 1..8 (8 cells per ray, not 11) and the range reported is the
//     first k that reaches d (a 4-bit-per-entry constant: 1 3 4 5 7 8 10 11); an axis ray cannot run further than S - 2 = 8 cells;
//   * every address is the agent's cell + an instruction immediate: no address arithmetic, no offset table, no loop.
// 64 byte reads per lane instead of 104.  (Tried: one 8-byte read for each of the two rays along the agent's row and one 4- / 8-byte
// read per neighbouring row for the three rays that cross it - 38 reads; the LDS takes unaligned wide reads, but at ~65 cycles
// apiece: 3 000 cycles for the 38.)  The reads are inline asm (the compiler folds the bias of the negative offsets back into adds and
// packs bytes through v_and / v_perm thickets); ngw_lidar_configure checks the host's ray table against ngw_lidar8_dr / _dc.
template <int OFF>
__device__ __forceinline__ uint32_t lds_u8_imm(const LDS_AS uint8_t* base) {
    static_assert(OFF >= 0 && OFF < 65536, "DS immediates are unsigned 16-bit");
    uint32_t v;
    asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF));
    return v;                                                                      // (zero-extended)
}
// byte 0 of a | byte 0 of b << 8 | byte 0 of c << 16 | byte 0 of d << 24 (the values are zero-extended bytes: three shift-ors)
__device__ __forceinline__ uint32_t pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return (a | (b << 8)) | ((c | (d << 8)) << 16); }

template <int S, int DR, int DC>
__device__ __forceinline__ void lidar_ray8(const LDS_AS uint8_t* agb, uint32_t (&c)[8]) {
    constexpr int BIAS = 8 * (S + 1);
#define NGW_O(d) (BIAS + (d) * (DR * S + DC))
    c[0] = lds_u8_imm<NGW_O(1)>(agb); c[1] = lds_u8_imm<NGW_O(2)>(agb); c[2] = lds_u8_imm<NGW_O(3)>(agb); c[3] = lds_u8_imm<NGW_O(4)>(agb);
    c[4] = lds_u8_imm<NGW_O(5)>(agb); c[5] = lds_u8_imm<NGW_O(6)>(agb); c[6] = lds_u8_imm<NGW_O(7)>(agb); c[7] = lds_u8_imm<NGW_O(8)>(agb);
#undef NGW_O
}

template <int S>
__device__ __forceinline__ void lidar_march_const8(const NgwLaunch& a, const LDS_AS uint8_t* ag, int f, int NC, int sh, const LDS_AS uint8_t* chan_of_item,
                                                   LDS_AS uint8_t* rowp, int dump) {
    static_assert(S - 2 == 8, "eight geometric steps per ray: two words");
    static_assert(8 * (S + 1) <= 11 * (S + 1), "inside the LDS guard (11 * (S + 1) bytes on both sides of the maps)");
    const LDS_AS uint8_t* agb = ag - 8 * (S + 1);                                  // (immediates are unsigned)
    // world rays in table order: 0 (+d, 0)  1 (+d, +d)  2 (0, +d)  3 (-d, +d)  4 (-d, 0)  5 (-d, -d)  6 (0, -d)  7 (+d, -d)
    uint32_t c[8][8];
    lidar_ray8<S, 1, 0>(agb, c[0]); lidar_ray8<S, 1, 1>(agb, c[1]); lidar_ray8<S, 0, 1>(agb, c[2]); lidar_ray8<S, -1, 1>(agb, c[3]);
    lidar_ray8<S, -1, 0>(agb, c[4]); lidar_ray8<S, -1, -1>(agb, c[5]); lidar_ray8<S, 0, -1>(agb, c[6]); lidar_ray8<S, 1, -1>(agb, c[7]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             // one wait for all 64 reads ...
#pragma unroll
    for (int w = 0; w < 8; w++)                                                    // ... and every value pinned behind it (no instruction; the compiler does not count asm reads)
        asm volatile("" : "+v"(c[w][0]), "+v"(c[w][1]), "+v"(c[w][2]), "+v"(c[w][3]), "+v"(c[w][4]), "+v"(c[w][5]), "+v"(c[w][6]), "+v"(c[w][7]));
    STAMP_SUB(a, 2);
    const int uf = f == 0 ? 4 : (f == 1 ? 0 : (f == 2 ? 6 : 2));                   // NORTH pi, SOUTH 0, WEST 3 pi / 2, EAST pi / 2 in units of pi / 4 (:38)
    int pos = ((4 - uf) & 7) * NC;                                                 // beam b = (world ray + rot) mod 8; its row entries start at b * NC
    const int wrap = 8 * NC;
    int hk[8], ch[8];
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const uint32_t lo = pack4(c[w][0], c[w][1], c[w][2], c[w][3]), hi = pack4(c[w][4], c[w][5], c[w][6], c[w][7]);
        const uint32_t t = lo ? lo : hi;                                           // (the wall ring stops every ray within 8 cells: t != 0)
        const int q = (__ffs((int)(t | 0x80000000u)) - 1) >> 3;
        const int d = (lo ? 0 : 4) + q;                                            // geometric distance - 1
        hk[w] = (w & 1) ? (int)((0xBA875431u >> (4 * d)) & 15u) : d + 1;          // a diagonal reports the first range k with round(0.71 k) = d + 1
        ch[w] = chan_of_item[(t >> (8 * q)) & 255u];
    }
    STAMP_SUB(a, 6);
#pragma unroll
    for (int w = 0; w < 8; w++) {
        rowp[ch[w] ? (pos + ch[w] - 1) << sh : dump] = (uint8_t)hk[w];
        pos += NC;
        pos -= pos >= wrap ? wrap : 0;
    }
}

// One workgroup IS one wavefront here (NGW_EPB == 64), and a wave's LDS operations execute in program order: a lane reading what
// another lane of the same wave wrote earlier needs no hardware wait, only the compiler must not reorder the accesses.
// __syncthreads() would also drain every outstanding GLOBAL store of the wave (s_waitcnt vmcnt(0)): ~500 cycles after the step's
// output stores, three times per lidar epilogue.
__device__ __forceinline__ void wave_lds_sync() {
    static_assert(NGW_EPB == 64, "one wavefront per workgroup");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void lidar_march_world(const LDS_AS uint8_t* ag, int f, int B, int R, int NC, int sh, const NgwLidarDev* cfg,
                                                  const LDS_AS uint8_t* chan_of_item, LDS_AS uint8_t* rowp, int dump) {
    const CONST_AS int32_t* woff = (const CONST_AS int32_t*)(const void*)&cfg->woff[0][0];   // uniform addresses: scalar loads
    const int half = B >> 1, quarter = B >> 2;
    const int uf = f == 0 ? half : (f == 1 ? 0 : (f == 2 ? half + quarter : quarter));     // NORTH pi, SOUTH 0, WEST 3 pi / 2, EAST pi / 2 (:38)
    int rot = half - uf;                                                                  // beam b = (world ray + rot) mod B
    rot += rot < 0 ? B : 0;
    for (int w0 = 0; w0 < B; w0 += 8) {
        uint32_t first[8];
        int kc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { first[i] = 0; kc[i] = 0; }
        // Scalar loads and LDS reads share one counter (lgkmcnt) and scalar loads return out of order, so a wait for offsets drains
        // the cell reads too.  Hence the order below: the cell reads of this round go out, THEN the offsets of the next round are
        // requested, and one wait covers both (left alone, the scheduler interleaved them: four full drains per round).
        int o[8][4];
#pragma unroll
        for (int i = 0; i < 8; i++) {                                                       // (a ray slot beyond B repeats ray B - 1)
            const CONST_AS int32_t* t = woff + min(w0 + i, B - 1) * NGW_LIDAR_MAX_RANGE;
            o[i][0] = t[0]; o[i][1] = t[1]; o[i][2] = t[2]; o[i][3] = t[3];
        }
        for (int k0 = 0; k0 < R; k0 += 4) {
            uint32_t c[8][4];
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) c[i][j] = ag[o[i][j]];
            __builtin_amdgcn_sched_barrier(0);
            const int kn = min(k0 + 4, NGW_LIDAR_MAX_RANGE - 4);                            // (the last round re-requests in-table entries nobody uses)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const CONST_AS int32_t* t = woff + min(w0 + i, B - 1) * NGW_LIDAR_MAX_RANGE + kn;
                o[i][0] = t[0]; o[i][1] = t[1]; o[i][2] = t[2]; o[i][3] = t[3];
            }
            __builtin_amdgcn_sched_barrier(0);
            uint32_t word[8];
#pragma unroll
            for (int i = 0; i < 8; i++) word[i] = (c[i][0] | (c[i][1] << 8)) | ((c[i][2] | (c[i][3] << 8)) << 16);
            if (!__any(lidar_round(word, k0, first, kc))) break;
        }
        int b0 = w0 + rot;
        b0 -= b0 >= B ? B : 0;
        lidar_hits(first, kc, b0 * NC, B - w0, R, NC, B * NC, sh, chan_of_item, rowp, dump);
    }
}

// Per-lane table form: flat int16 offsets [facing][beam][range-1] in LDS; eight rays advance four ranges per round (8 table
// reads of 4 offsets each, then 32 cell reads in flight).
__device__ __forceinline__ void lidar_march_table(const LDS_AS uint8_t* ag, int f, int B, int R, int NC, int sh, const LDS_AS int16_t* toff,
                                                  const LDS_AS uint8_t* chan_of_item, LDS_AS uint8_t* rowp, int dump) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    for (int b0 = 0; b0 < B; b0 += 8) {
        uint32_t first[8];
        int kc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { first[i] = 0; kc[i] = 0; }
        for (int k0 = 0; k0 < R; k0 += 4) {
            u32x2 o[8];
#pragma unroll
            for (int i = 0; i < 8; i++)
                o[i] = *reinterpret_cast<const LDS_AS u32x2*>(toff + (f * NGW_LIDAR_MAX_BEAMS + min(b0 + i, B - 1)) * NGW_LIDAR_MAX_RANGE + k0);
            uint32_t word[8];
#pragma unroll
            for (int i = 0; i < 8; i++)
                word[i] = lidar_cells4(ag, (int16_t)(o[i].x & 0xFFFFu), (int16_t)(o[i].x >> 16), (int16_t)(o[i].y & 0xFFFFu), (int16_t)(o[i].y >> 16));
            if (!__any(lidar_round(word, k0, first, kc))) break;
        }
        lidar_hits(first, kc, b0 * NC, B - b0, R, NC, B * NC, sh, chan_of_item, rowp, dump);
    }
}

constexpr int LIDAR_TAB16 = 4 * NGW_LIDAR_MAX_BEAMS * NGW_LIDAR_MAX_RANGE * 2 / 16;     // per-lane ray table = 512 pieces of 16 B
constexpr int LIDAR_ITEM_DW = 2 * NGW_MAX_ITEMS / 4;                                    // chan_of_item | inv_item
static_assert(LIDAR_TAB16 == 8 * NGW_EPB, "ray table is 8 pieces per lane");
static_assert(offsetof(NgwLidarDev, chan_of_item) == 16 * LIDAR_TAB16 && offsetof(NgwLidarDev, inv_item) == 16 * LIDAR_TAB16 + NGW_MAX_ITEMS,
              "lidar tables are contiguous");

// The observation of the wave's 64 envs: `agent` = the lane's agent cell in its LDS map, `inv` = its inventory row in LDS; the two
// item tables are in LDS at a.off_litem (the caller's prologue put them there).  Whole-wave call (barriers inside).
// The observation tile starts as zeros.  A step kernel does this in its PROLOGUE, while its global loads are in flight (the wave has
// nothing else to do there for ~1 100 cycles); only a launch whose cold path used the tile's region for something else (the reset
// path's Philox ring shares it) zeroes it again in the epilogue.
__device__ __forceinline__ void lidar_zero_tile(const NgwLaunch& a, uint32_t* lds, int tid) {
    const int npc = 4 * a.l_rb;
    LDS_AS u32x4* t4 = (LDS_AS u32x4*)(lds + a.off_ltile);
    for (int base = 0; base < npc; base += EPB * 4) {                              // (a piece index beyond the tile zeroes the last piece again)
#pragma unroll
        for (int j = 0; j < 4; j++) t4[min(base + tid + EPB * j, npc - 1)] = u32x4{0u, 0u, 0u, 0u};
    }
}

__device__ __forceinline__ void lidar_epilogue(const NgwLaunch& a, uint32_t* lds, int tid, bool live, const int8_t* agent, int f, const int32_t* inv,
                                               bool zeroed = false) {
    const int B = a.l_beams, R = a.l_range, NC = a.l_chan, NI = a.l_inv, rb = a.l_rb, npc = 4 * rb;   // 64 rows = 4 * rb pieces of 16 B
    const int sh = a.l_fmt == NGW_LFMT_I32 ? 2 : (a.l_fmt == NGW_LFMT_I16 ? 1 : 0);                   // beam entry = 1 << sh bytes
    LDS_AS u32x4* t4 = (LDS_AS u32x4*)(lds + a.off_ltile);
    wave_lds_sync();
    STAMP_SUB(a, 0);
    if (!zeroed) lidar_zero_tile(a, lds, tid);
    if (!a.l_world) {                                                              // (uniform) the per-lane ray table, staged now: 8 KiB
        const u32x4* src = reinterpret_cast<const u32x4*>(a.lcfg->off);
        LDS_AS u32x4* dst = (LDS_AS u32x4*)(lds + a.off_ltab);
        u32x4 tb[8];
#pragma unroll
        for (int j = 0; j < 8; j++) tb[j] = src[tid + EPB * j];
#pragma unroll
        for (int j = 0; j < 8; j++) dst[tid + EPB * j] = tb[j];
    }
    wave_lds_sync();
    STAMP_SUB(a, 1);
    const LDS_AS uint8_t* chan_of_item = (const LDS_AS uint8_t*)(lds + a.off_litem);
    LDS_AS uint8_t* rowp = (LDS_AS uint8_t*)(lds + a.off_ltile) + tid * rb;
    if (live) {
        const LDS_AS uint8_t* ag = (const LDS_AS uint8_t*)agent;
        const int dump = EPB * rb - tid * rb + tid;                                // the 64 bytes behind the tile: one per lane
        if (a.l_world == 2) lidar_march_const8<NGW_LIDAR_CONST_S>(a, ag, f, NC, sh, chan_of_item, rowp, dump);     // (the host checked table and range: R = 11)
        else if (a.l_world) lidar_march_world(ag, f, B, R, NC, sh, a.lcfg, chan_of_item, rowp, dump);
        else lidar_march_table(ag, f, B, R, NC, sh, (const LDS_AS int16_t*)(lds + a.off_ltab), chan_of_item, rowp, dump);
    }
    STAMP_SUB(a, 3);
    if (live) {
        // the inventory tail (:74-75): the item ids sit in LDS behind the channel table (every lane reads the same six dwords: broadcast
        // reads), an entry is then one read of the lane's inventory row and one store; eight entries in flight
        const LDS_AS uint32_t* idw = (const LDS_AS uint32_t*)(chan_of_item + NGW_MAX_ITEMS);
        uint32_t iw[NGW_MAX_ITEMS / 4];
#pragma unroll
        for (int i = 0; i < NGW_MAX_ITEMS / 4; i++) iw[i] = idw[i];
        const LDS_AS int32_t* iv = (const LDS_AS int32_t*)inv;
        LDS_AS uint8_t* tail = rowp + a.l_invoff;
        const bool wide = a.l_fmt == NGW_LFMT_I32;
#pragma unroll
        for (int j0 = 0; j0 < NGW_MAX_ITEMS; j0 += 8) {
            if (j0 < NI) {                                                         // (uniform)
                int cnt[8];
#pragma unroll
                for (int i = 0; i < 8; i++) cnt[i] = iv[(iw[(j0 + i) >> 2] >> (8 * ((j0 + i) & 3))) & 255u];   // (entries beyond NI name item 0)
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (j0 + i < NI) {
                        if (wide) *(LDS_AS int32_t*)(tail + 4 * (j0 + i)) = cnt[i];
                        else *(LDS_AS uint16_t*)(tail + 2 * (j0 + i)) = (uint16_t)min(cnt[i], 32767);
                    }
            }
        }
    }
    wave_lds_sync();
    STAMP_SUB(a, 4);
    GLOBAL_AS u32x4* g4 = (GLOBAL_AS u32x4*)(reinterpret_cast<char*>(a.lout) + (uint64_t)blockIdx.x * (uint32_t)(EPB * rb));
    for (int base = 0; base < npc; base += EPB * 4) {                              // four pieces per lane in flight
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = t4[min(base + tid + EPB * j, npc - 1)];
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (base + tid + EPB * j < npc) g4[base + tid + EPB * j] = v[j];
    }
    STAMP_SUB(a, 5);
}

// ---------------------------------------------------------------- the general new-episode kernel
// Explicit resets (NGW_MODE_RESET) and refills of the prepared next episodes (NGW_MODE_REFILL) of every configuration the dedicated
// new-episode kernel (ngw_reset.inc) does not take: reset passes that read the map (Fence, ReplaceItem of an interior item), stacks
// of passes, the v0 tree tap, 10 x 10 plain maps, and resets that refresh the fused LidarInFront observation.  The wave stages its
// 64 maps and inventory rows into LDS, the lanes that reset run new_episode there (a prepared row if there is one, else the
// placement loop), and the chunk goes back with coalesced 16-byte pieces.  Steps and fused rollouts live in ngw_lean.inc - up to
// round 3 this kernel also carried a switch-dispatched step; there is ONE step implementation now (lean_body).
template <int MAPMODE, int MODE, bool LIDAR>
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

    // LDS carve-up (dword offsets): maps | inventory [64][KP] | candidate masks [CW][64] | placement sequence
    uint32_t* lds_map = lds + a.off_map;
    int32_t* lds_inv = reinterpret_cast<int32_t*>(lds + a.off_inv);
    int8_t* mp = reinterpret_cast<int8_t*>(lds_map) + tid * a.MS;                 // this lane's map
    int32_t* inv = lds_inv + tid * a.KP;                                           // this lane's inventory row
    uint32_t* cand = lds + a.off_cand + tid;

    // ---- issue EVERY global load of the prologue before touching LDS: placement sequence, first map round, scalars, inventory
    const uint32_t psv = reinterpret_cast<const uint32_t*>(dspec->place_seq)[min(tid, NGW_MAX_PLACE / 4 - 1)];
    uint32_t lit = 0;
    if (LIDAR && tid < LIDAR_ITEM_DW) lit = reinterpret_cast<const uint32_t*>(a.lcfg->chan_of_item)[tid];
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
    // A whole wavefront of envs that ALL get a new episode (an unmasked reset; a refill behind a batch that ended its episodes together): nothing of
    // the old rows survives, so they are not staged - the prologue's round trip and S*S + 4 K bytes per env less (a C2 reset of every env 21.6 -> 20.x us).
    const bool all_new = (int64_t)EPB <= a.n - env0 &&
                         ((MODE == NGW_MODE_RESET && !a.reset_mask) || (MODE == NGW_MODE_REFILL && __all(action != 0)));
    u32x4 buf[PB];
    const u32x4* gin = reinterpret_cast<const u32x4*>(a.b.map + env0 * a.S2);
    if (!all_new) pieces_load(buf, gin, 0, npieces, tid);
    if (live) {
        if (!all_new) {
            const int2 rc = reinterpret_cast<const int2*>(a.b.loc)[e];
            r = rc.x; c = rc.y;
            f = a.b.facing[e];
        }
        if (MODE != NGW_MODE_REFILL) {
            if (!all_new) {
                sel = a.b.selected[e];
                steps = a.b.step_count[e];
            }
            episode = a.b.episode[e];
        }
        if (MODE == NGW_MODE_RESET) action = a.reset_mask ? (int)a.reset_mask[e] : 1;
    }
    u32x4 iq[IQ];
    if (!all_new) {
        const u32x4* gi = reinterpret_cast<const u32x4*>(a.b.inv + env0 * K);
#pragma unroll
        for (int j = 0; j < IQ; j++) iq[j] = (j * EPB < 16 * K) ? gi[min(tid + EPB * j, 16 * K - 1)] : u32x4{0u, 0u, 0u, 0u};
    }
    STAMP(1);
    // ---- land them in LDS
    if (tid < NGW_MAX_PLACE / 4) lds[a.off_act + tid] = psv;
    if (!all_new) {
        pieces_lds<true, MAPMODE>(buf, a, lds_map, 0, npieces, tid);
        for (int base = EPB * PB; base < npieces; base += EPB * PB) {              // big maps: further rounds
            pieces_load(buf, gin, base, npieces, tid);
            pieces_lds<true, MAPMODE>(buf, a, lds_map, base, npieces, tid);
        }
        inv_lds<true>(iq, a, lds_inv, tid);
    }
    if (LIDAR && tid < LIDAR_ITEM_DW) lds[a.off_litem + tid] = lit;
    __syncthreads();
    STAMP(2);

    uint32_t flags = 0;
    g_u32x4* gmap = (g_u32x4*)(reinterpret_cast<u32x4*>(a.b.map + env0 * a.S2) + tid);     // the wave's coalesced chunk
    g_u32x4* ginv = (g_u32x4*)(reinterpret_cast<u32x4*>(a.b.inv + env0 * K) + tid);
    const uint64_t env_global = (uint64_t)(a.env_base + e);
    if (MODE == NGW_MODE_REFILL && blockIdx.x == 0 && tid == 0) {                  // what the host reads (without a sync) before the next refill
        uint32_t* const sh = dspec->nx.slow_host;                                 // (one report per refill: the launch of slot 0 makes it)
        if (sh && a.autoreset == 0) { sh[0] = dspec->nx.slow[0]; sh[1] = atomicAdd(dspec->nx.slow + 1, 1u) + 1u; }
    }
    STAMP(3);
    bool do_reset = live && (MODE == NGW_MODE_RESET || MODE == NGW_MODE_REFILL) && action != 0;
    bool renewed = false;                                                          // this lane's map in LDS is a new episode's
    if (do_reset) {                                                                // out of line: new_episode
        episode++;
        uint32_t rr = new_episode(dspec, (LDS_AS int8_t*)mp, (LDS_AS int32_t*)inv, (LDS_AS uint32_t*)cand, (const LDS_AS uint8_t*)(lds + a.off_act),
                                  (LDS_AS uint16_t*)(lds + a.off_perm), env_global, e, episode, MODE != NGW_MODE_REFILL,
                                  MODE != NGW_MODE_RESET);   // (an explicit reset that finds nothing prepared is not a miss)
        do_reset = !(rr & NGW_F_ROWS_STORED);                                      // from here on: "the wave must store its chunk"
        rr &= ~(uint32_t)NGW_F_ROWS_STORED;
        renewed = !(MODE == NGW_MODE_REFILL && (rr & 0xFFu));
        if (MODE == NGW_MODE_REFILL && (rr & 0xFFu)) { rr &= ~0xFFu; episode = nx_old; }   // failed placement: leave the row
                                                                                   // stale, the real reset raises the flag
        flags |= rr & 0xFFu;
        r = (int)((rr >> 8) & 0xFFu); c = (int)((rr >> 16) & 0xFFu); f = (int)(rr >> 24);
        sel = 0; steps = 0;
    }
    STAMP(4);
    // ---- a reset rewrote whole maps / inventory rows in LDS: store the wave's chunk back with coalesced 16-B pieces
    if (MODE == NGW_MODE_DBG_COPY || __any(do_reset)) {
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
        reinterpret_cast<int2*>(a.b.loc)[e] = int2{r, c};
        a.b.facing[e] = f;
        if (MODE != NGW_MODE_REFILL) {
            a.b.selected[e] = (uint8_t)sel;
            a.b.step_count[e] = steps;
        }
        a.b.episode[e] = episode;
    }
    // Boards mode (a.b.brd: the bit-row lidar is on): the occupancy bit rows of the maps this launch made, from the lane's map in LDS - one word per
    // row, bit c = cell (r, c) holds a block.  (Behind this kernel the host used to rebuild the bit rows of EVERY map with another launch: with
    // FireWall's refill every 18 steps over four prepared slots that was the larger part of a 35 us lidar step.)
    if ((MODE == NGW_MODE_RESET || MODE == NGW_MODE_REFILL) && a.b.brd && renewed) {
        GLOBAL_AS uint32_t* br = (GLOBAL_AS uint32_t*)a.b.brd + e * a.BS;
        const LDS_AS uint8_t* m8 = (const LDS_AS uint8_t*)mp;
        for (int rr_ = 0; rr_ < S; rr_++) {
            uint32_t w = 0;
            for (int cc = 0; cc < S; cc++) w |= (m8[rr_ * S + cc] != 0 ? 1u : 0u) << cc;
            br[rr_] = w;
        }
        for (int rr_ = S; rr_ < a.BS; rr_++) br[rr_] = 0u;
    }
    if (LIDAR) lidar_epilogue(a, lds, tid, live, mp + r * S + c, f, inv);          // the observation of the state this reset produced
    if (flags) atomicOr(a.b.flags, flags);
    raise_host_flags(a.b.flags_host, flags);
    if (MODE == NGW_MODE_RESET) {
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

// a store into the HOST's memory (the write-through targets, NgwWT): system scope - written through every cache level at once (global_store ... sc0 sc1),
// so that "the wave's stores are acknowledged" (s_waitcnt) means "the host can see them".  A plain store may sit dirty in this XCD's L2 until the
// kernel ends: wire_done's counter would then announce data that has not left the chip (seen with 2 MB of observation rows per launch).
template <class T> __device__ __forceinline__ void stgs(void* base, uint32_t off, T v) {
    __hip_atomic_store((GLOBAL_AS T*)((GLOBAL_AS char*)base + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... 16 bytes of it (no 128-bit atomic store in the language: the instruction with the same cache bits)
__device__ __forceinline__ void stgs16(void* base, uint32_t off, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" : : "v"(off), "v"(v), "s"(base) : "memory");
}

#include "ngw_boards.inc"
#include "ngw_lean.inc"
#include "ngw_solo.inc"
#include "ngw_reset.inc"

// ---------------------------------------------------------------- LidarInFront observation kernel (stand-alone launch)
// observation_wrappers.py:32-80 of the CURRENT state.  Same wave = 64 envs decomposition and the same coalesced staging of the
// maps and inventory rows as the other kernels, then the shared row builder (lidar_epilogue).  `a` carries the launch's own LDS
// layout (ngw_lidar_configure: off_litem, off_ltab, off_ltile, off_map, off_inv).
template <int MAPMODE>
__global__ void __launch_bounds__(NGW_EPB) ngw_lidar_kernel(const NgwLaunch a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const int64_t env0 = (int64_t)blockIdx.x * EPB;
    const int64_t e = env0 + tid;
    const bool live = e < a.n;
    const int S = a.S, K = a.K, npieces = 4 * a.S2;
    uint32_t* lds_map = lds + a.off_map;
    int32_t* lds_inv = reinterpret_cast<int32_t*>(lds + a.off_inv);
    uint32_t it = 0;
    if (tid < LIDAR_ITEM_DW) it = reinterpret_cast<const uint32_t*>(a.lcfg->chan_of_item)[tid];
    u32x4 buf[PB];
    const u32x4* gin = reinterpret_cast<const u32x4*>(a.b.map + env0 * a.S2);
    pieces_load(buf, gin, 0, npieces, tid);
    int r = 1, c = 1, f = 0;
    if (live) {
        const int2 rc = reinterpret_cast<const int2*>(a.b.loc)[e];
        r = rc.x; c = rc.y;
        f = a.b.facing[e];
    }
    u32x4 iq[IQ];
    {
        const u32x4* gi = reinterpret_cast<const u32x4*>(a.b.inv + env0 * K);
#pragma unroll
        for (int j = 0; j < IQ; j++) iq[j] = (j * EPB < 16 * K) ? gi[min(tid + EPB * j, 16 * K - 1)] : u32x4{0u, 0u, 0u, 0u};
    }
    if (tid < LIDAR_ITEM_DW) lds[a.off_litem + tid] = it;
    pieces_lds<true, MAPMODE>(buf, a, lds_map, 0, npieces, tid);
    for (int base = EPB * PB; base < npieces; base += EPB * PB) {
        pieces_load(buf, gin, base, npieces, tid);
        pieces_lds<true, MAPMODE>(buf, a, lds_map, base, npieces, tid);
    }
    inv_lds<true>(iq, a, lds_inv, tid);
    const int8_t* mp = reinterpret_cast<const int8_t*>(lds_map) + tid * a.MS;
    lidar_epilogue(a, lds, tid, live, mp + r * S + c, f, lds_inv + tid * a.KP);     // (its first barrier makes the staging visible)
}

#if NGW_HAS(5)
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

// Narrow wire format of the host step (NgwWire, ngw_step_host_packed): one lane per env narrows pose / reward / done / info into
// four dense arrays of a staging payload (reads 22 B, writes 13 B per env; coalesced both ways).
__global__ __launch_bounds__(256) void ngw_wire_kernel(const NgwWire p) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e == 0) *p.flags_out = *p.flags;
    if (e >= p.n) return;
    const int r = p.loc[2 * e], c = p.loc[2 * e + 1], f = p.facing[e];
    const uint32_t sel = p.selected[e];
    p.pose[e] = (uint32_t)(r & 255) | ((uint32_t)(c & 255) << 8) | ((uint32_t)(f & 255) << 16) | (sel << 24);
    p.reward32[e] = p.reward[e];
    p.done8[e] = p.done[e];
    p.info32[e] = p.info[e];
}

// Delta refresh and narrowing in ONE launch (ngw_step_host_packed's steady state: the block is a mirror and takes direct stores): slices
// y < n_regions are ngw_diff_kernel's regions, the last slice narrows pose / reward / done / info with a grid-stride loop over the envs.
__global__ __launch_bounds__(256) void ngw_diff_wire_kernel(const NgwDiff p, const NgwWire w) {
    const int r = blockIdx.y;
    const uint64_t stride = (uint64_t)gridDim.x * 256u, t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r == p.n_regions) {
        if (t == 0) *w.flags_out = *w.flags;
        for (uint64_t e = t; e < (uint64_t)w.n; e += stride) {
            const int rr = w.loc[2 * e], c = w.loc[2 * e + 1], f = w.facing[e];
            const uint32_t sel = w.selected[e];
            w.pose[e] = (uint32_t)(rr & 255) | ((uint32_t)(c & 255) << 8) | ((uint32_t)(f & 255) << 16) | (sel << 24);
            w.reward32[e] = w.reward[e];
            w.done8[e] = w.done[e];
            w.info32[e] = w.info[e];
        }
        return;
    }
    const uint64_t nb = p.nbytes[r];
    const uint8_t* c = p.cur[r];
    uint8_t* s = p.shadow[r];
    uint8_t* h = p.host[r];
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
#endif  // NGW_HAS(5)

}  // namespace

#if NGW_HAS(5)
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

extern "C" hipError_t ngw_wire_launch(const NgwWire* p, hipStream_t stream) {
    hipLaunchKernelGGL(ngw_wire_kernel, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0, stream, *p);
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

extern "C" hipError_t ngw_diff_wire_launch(const NgwDiff* p, const NgwWire* w, hipStream_t stream) {
    uint64_t most = ((uint64_t)w->n + 15u) / 16u * 16u;                            // (the narrowing slice: one env per lane and round)
    for (int r = 0; r < p->n_regions; r++) most = p->nbytes[r] > most ? p->nbytes[r] : most;
    uint64_t blocks = (most / 16u + 255u) / 256u;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ngw_diff_wire_kernel, dim3((unsigned)blocks, (unsigned)p->n_regions + 1u), dim3(256), 0, stream, *p, *w);
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

extern "C" hipError_t ngw_lidar_launch(const NgwLaunch* a, int map_mode, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    const void* fn = map_mode == NGW_MAP_STRAIGHT ? reinterpret_cast<const void*>(ngw_lidar_kernel<NGW_MAP_STRAIGHT>)
                     : map_mode == NGW_MAP_DWORD  ? reinterpret_cast<const void*>(ngw_lidar_kernel<NGW_MAP_DWORD>)
                                                  : reinterpret_cast<const void*>(ngw_lidar_kernel<NGW_MAP_BYTE>);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    switch (map_mode) {
    case NGW_MAP_STRAIGHT: hipLaunchKernelGGL(ngw_lidar_kernel<NGW_MAP_STRAIGHT>, dim3(grid), dim3(NGW_EPB), lds_bytes, stream, *a); break;
    case NGW_MAP_DWORD: hipLaunchKernelGGL(ngw_lidar_kernel<NGW_MAP_DWORD>, dim3(grid), dim3(NGW_EPB), lds_bytes, stream, *a); break;
    default: hipLaunchKernelGGL(ngw_lidar_kernel<NGW_MAP_BYTE>, dim3(grid), dim3(NGW_EPB), lds_bytes, stream, *a);
    }
    return hipGetLastError();
}
#endif  // NGW_HAS(5)

namespace {

#if NGW_HAS(0)
template <int MAPMODE, int MODE, bool LIDAR>
hipError_t launch_one(const NgwDevSpec* dspec, const NgwLaunch* a, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    // CDNA4 has 160 KiB of LDS per CU; anything above the 64 KiB default needs an explicit opt-in per device.
    static size_t lds_opt_in[64] = {0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (lds_bytes > 64 * 1024 && dev < 64 && lds_bytes > lds_opt_in[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(ngw_kernel<MAPMODE, MODE, LIDAR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
        if (e != hipSuccess) return e;
        lds_opt_in[dev] = lds_bytes;
    }
    hipLaunchKernelGGL((ngw_kernel<MAPMODE, MODE, LIDAR>), dim3(grid), dim3(NGW_EPB), lds_bytes, stream, dspec, *a);
    return hipGetLastError();
}

// the general new-episode kernel: explicit resets (with or without the fused lidar observation), refills, diagnostics
template <int MAPMODE>
static hipError_t launch_general(const NgwDevSpec* dspec, const NgwLaunch* a, bool lidar, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    switch (a->mode) {
    case NGW_MODE_RESET:
        return lidar ? launch_one<MAPMODE, NGW_MODE_RESET, true>(dspec, a, grid, lds_bytes, stream)
                     : launch_one<MAPMODE, NGW_MODE_RESET, false>(dspec, a, grid, lds_bytes, stream);
    case NGW_MODE_REFILL: return launch_one<MAPMODE, NGW_MODE_REFILL, false>(dspec, a, grid, lds_bytes, stream);
    case NGW_MODE_DBG_COPY: return launch_one<MAPMODE, NGW_MODE_DBG_COPY, false>(dspec, a, grid, lds_bytes, stream);
    case NGW_MODE_DBG_NOP: return launch_one<MAPMODE, NGW_MODE_DBG_NOP, false>(dspec, a, grid, lds_bytes, stream);
    default: return hipErrorInvalidValue;
    }
}

__global__ void ngw_nop_kernel(const NgwDevSpec* dspec, const NgwLaunch a) {}
#endif  // NGW_HAS(0)

}  // namespace

#if NGW_HAS(5)
extern "C" hipError_t ngw_solo_launch(const NgwDevSpec* dspec, const NgwSolo* p, int ext, size_t lds_bytes, hipStream_t stream) {
    const void* fn = ext ? reinterpret_cast<const void*>(ngw_solo_kernel<true>) : reinterpret_cast<const void*>(ngw_solo_kernel<false>);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    void* args[] = {const_cast<NgwDevSpec**>(&dspec), const_cast<NgwSolo*>(p)};
    return hipLaunchKernel(fn, dim3(1), dim3(NGW_EPB), args, lds_bytes, stream);
}
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
#endif  // NGW_HAS(5)

// feat: 1 = fused LidarInFront epilogue, 2 = wrapper predicates (EXT), 8 = no-stage step (maps read in place), 16 = host write-through (with 8)
extern "C" hipError_t ngw_part_step(const NgwDevSpec* dspec, const NgwLaunch* a, int map_mode, int feat, unsigned grid, size_t lds_bytes,
                                    hipStream_t stream);
extern "C" hipError_t ngw_part_rollout_straight(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream);
extern "C" hipError_t ngw_part_rollout_dword(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream);
extern "C" hipError_t ngw_part_rollout_byte(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream);

// ONE batched step(): ngw_step_lean.  The staged kernels of one map addressing mode (four each) are a unit of their own.
#define NGW_STEP_PART(NAME, MM)                                                                                                         \
    extern "C" hipError_t NAME(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream) { \
        const bool lidar = (feat & 1) != 0, ext = (feat & 2) != 0;                                                                      \
        return lidar ? (ext ? launch_lean<MM, true, true, true>(dspec, a, grid, lds_bytes, stream)                                      \
                            : launch_lean<MM, true, false, true>(dspec, a, grid, lds_bytes, stream))                                    \
                     : (ext ? launch_lean<MM, true, true, false>(dspec, a, grid, lds_bytes, stream)                                     \
                            : launch_lean<MM, true, false, false>(dspec, a, grid, lds_bytes, stream));                                  \
    }
extern "C" hipError_t ngw_part_step_straight(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream);
extern "C" hipError_t ngw_part_step_dword(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream);
extern "C" hipError_t ngw_part_step_byte(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream);
extern "C" hipError_t ngw_part_step_boards(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream);
extern "C" hipError_t ngw_part_step_wire(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream);
extern "C" hipError_t ngw_part_step_wire_boards(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream);
#if NGW_HAS(9)
// in-place step with the host write-through (NgwWT: ngw_step_host_packed's steady state)
extern "C" hipError_t ngw_part_step_wire(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    if (feat & 1) return ngw_part_step_wire_boards(dspec, a, feat, grid, lds_bytes, stream);
    return (feat & 2) ? launch_lean<NGW_MAP_STRAIGHT, false, true, false, 0, true>(dspec, a, grid, lds_bytes, stream)
                      : launch_lean<NGW_MAP_STRAIGHT, false, false, false, 0, true>(dspec, a, grid, lds_bytes, stream);
}
#endif  // NGW_HAS(9)
#if NGW_HAS(10)
extern "C" hipError_t ngw_part_step_wire_boards(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    const bool ext = (feat & 2) != 0;
    if (!a->l_boards || a->BS < 4 || a->BS > 32) return hipErrorInvalidValue;
    if (a->BS <= 12) return ext ? launch_lean<NGW_MAP_STRAIGHT, false, true, true, 12, true>(dspec, a, grid, lds_bytes, stream)
                                : launch_lean<NGW_MAP_STRAIGHT, false, false, true, 12, true>(dspec, a, grid, lds_bytes, stream);
    if (a->BS <= 20) return ext ? launch_lean<NGW_MAP_STRAIGHT, false, true, true, 20, true>(dspec, a, grid, lds_bytes, stream)
                                : launch_lean<NGW_MAP_STRAIGHT, false, false, true, 20, true>(dspec, a, grid, lds_bytes, stream);
    return ext ? launch_lean<NGW_MAP_STRAIGHT, false, true, true, 32, true>(dspec, a, grid, lds_bytes, stream)
               : launch_lean<NGW_MAP_STRAIGHT, false, false, true, 32, true>(dspec, a, grid, lds_bytes, stream);
}
#endif  // NGW_HAS(10)
#if NGW_HAS(8)
// in-place step + the LidarInFront observation from the occupancy bit rows (ngw_boards.inc): NR = 12 / 20 / 32 register rows
extern "C" hipError_t ngw_part_step_boards(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    const bool ext = (feat & 2) != 0;
    if (!a->l_boards || a->BS < 4 || a->BS > 32) return hipErrorInvalidValue;
    if (a->BS <= 12) return ext ? launch_lean<NGW_MAP_STRAIGHT, false, true, true, 12>(dspec, a, grid, lds_bytes, stream)
                                : launch_lean<NGW_MAP_STRAIGHT, false, false, true, 12>(dspec, a, grid, lds_bytes, stream);
    if (a->BS <= 20) return ext ? launch_lean<NGW_MAP_STRAIGHT, false, true, true, 20>(dspec, a, grid, lds_bytes, stream)
                                : launch_lean<NGW_MAP_STRAIGHT, false, false, true, 20>(dspec, a, grid, lds_bytes, stream);
    return ext ? launch_lean<NGW_MAP_STRAIGHT, false, true, true, 32>(dspec, a, grid, lds_bytes, stream)
               : launch_lean<NGW_MAP_STRAIGHT, false, false, true, 32>(dspec, a, grid, lds_bytes, stream);
}
// bit rows of `rows` (a multiple of 64) maps at `map` -> `brd`; a = the launch's own LDS layout (maps at off_map, word tile at off_ltile, magicK = ceil(2^32 / BS))
extern "C" hipError_t ngw_boards_launch(const NgwLaunch* a, int map_mode, const int8_t* map, uint32_t* brd, int64_t rows, size_t lds_bytes, hipStream_t stream) {
    const void* fn = map_mode == NGW_MAP_STRAIGHT ? reinterpret_cast<const void*>(ngw_boards_kernel<NGW_MAP_STRAIGHT>)
                   : (map_mode == NGW_MAP_DWORD ? reinterpret_cast<const void*>(ngw_boards_kernel<NGW_MAP_DWORD>) : reinterpret_cast<const void*>(ngw_boards_kernel<NGW_MAP_BYTE>));
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    void* args[] = {const_cast<NgwLaunch*>(a), &map, &brd};
    return hipLaunchKernel(fn, dim3((unsigned)(rows / NGW_EPB)), dim3(NGW_EPB), args, lds_bytes, stream);
}
extern "C" hipError_t ngw_lidar_boards_launch(const NgwLaunch* a, unsigned grid, size_t lds_bytes, hipStream_t stream) {
    if (a->BS <= 12) hipLaunchKernelGGL(ngw_lidar_boards_kernel<12>, dim3(grid), dim3(NGW_EPB), lds_bytes, stream, *a);
    else if (a->BS <= 20) hipLaunchKernelGGL(ngw_lidar_boards_kernel<20>, dim3(grid), dim3(NGW_EPB), lds_bytes, stream, *a);
    else hipLaunchKernelGGL(ngw_lidar_boards_kernel<32>, dim3(grid), dim3(NGW_EPB), lds_bytes, stream, *a);
    return hipGetLastError();
}
#endif  // NGW_HAS(8)
#if NGW_HAS(1)
NGW_STEP_PART(ngw_part_step_straight, NGW_MAP_STRAIGHT)
extern "C" hipError_t ngw_part_step(const NgwDevSpec* dspec, const NgwLaunch* a, int map_mode, int feat, unsigned grid, size_t lds_bytes,
                                    hipStream_t stream) {
    if (feat & 8) {                                                     // no-stage: with the lidar observation, the one on the occupancy bit rows
        if (feat & 16) return ngw_part_step_wire(dspec, a, feat, grid, lds_bytes, stream);   // ... with the host write-through
        if (feat & 1) return ngw_part_step_boards(dspec, a, feat, grid, lds_bytes, stream);
        return (feat & 2) ? launch_lean<NGW_MAP_STRAIGHT, false, true, false>(dspec, a, grid, lds_bytes, stream)
                          : launch_lean<NGW_MAP_STRAIGHT, false, false, false>(dspec, a, grid, lds_bytes, stream);
    }
    switch (map_mode) {
    case NGW_MAP_STRAIGHT: return ngw_part_step_straight(dspec, a, feat, grid, lds_bytes, stream);
    case NGW_MAP_DWORD: return ngw_part_step_dword(dspec, a, feat, grid, lds_bytes, stream);
    default: return ngw_part_step_byte(dspec, a, feat, grid, lds_bytes, stream);
    }
}
#endif  // NGW_HAS(1)
#if NGW_HAS(6)
NGW_STEP_PART(ngw_part_step_dword, NGW_MAP_DWORD)
#endif
#if NGW_HAS(7)
NGW_STEP_PART(ngw_part_step_byte, NGW_MAP_BYTE)
#endif
#undef NGW_STEP_PART

// fused rollout: ngw_rollout_lean, one part per map addressing mode (eight kernels each: the heaviest to compile)
#define NGW_ROLLOUT_PART(NAME, MM)                                                                                                     \
    extern "C" hipError_t NAME(const NgwDevSpec* dspec, const NgwLaunch* a, int feat, unsigned grid, size_t lds_bytes, hipStream_t stream) { \
        const bool lidar = (feat & 1) != 0, ext = (feat & 2) != 0;                                                                     \
        if (a->mode == NGW_MODE_ROLLOUT_ACT)                                                                                           \
            return lidar ? (ext ? launch_rollout_lean<MM, true, true, true>(dspec, a, grid, lds_bytes, stream)                         \
                                : launch_rollout_lean<MM, true, false, true>(dspec, a, grid, lds_bytes, stream))                       \
                         : (ext ? launch_rollout_lean<MM, true, true, false>(dspec, a, grid, lds_bytes, stream)                        \
                                : launch_rollout_lean<MM, true, false, false>(dspec, a, grid, lds_bytes, stream));                     \
        return lidar ? (ext ? launch_rollout_lean<MM, false, true, true>(dspec, a, grid, lds_bytes, stream)                            \
                            : launch_rollout_lean<MM, false, false, true>(dspec, a, grid, lds_bytes, stream))                          \
                     : (ext ? launch_rollout_lean<MM, false, true, false>(dspec, a, grid, lds_bytes, stream)                           \
                            : launch_rollout_lean<MM, false, false, false>(dspec, a, grid, lds_bytes, stream));                        \
    }
#if NGW_HAS(2)
NGW_ROLLOUT_PART(ngw_part_rollout_straight, NGW_MAP_STRAIGHT)
#endif
#if NGW_HAS(3)
NGW_ROLLOUT_PART(ngw_part_rollout_dword, NGW_MAP_DWORD)
#endif
#if NGW_HAS(4)
NGW_ROLLOUT_PART(ngw_part_rollout_byte, NGW_MAP_BYTE)
#endif
#undef NGW_ROLLOUT_PART

#if NGW_HAS(0)
extern "C" hipError_t ngw_launch(const NgwDevSpec* dspec, const NgwLaunch* a, int map_mode, int feat, unsigned grid,
                                 size_t lds_bytes, hipStream_t stream) {
    if (a->mode >= 10 && a->mode <= 12) {       // diagnostics: empty kernels with other workgroup shapes over the same lanes
        const unsigned tpb = a->mode == 10 ? 256 : (a->mode == 11 ? 1024 : 128);
        hipLaunchKernelGGL(ngw_nop_kernel, dim3(grid * NGW_EPB / tpb), dim3(tpb), a->mode == 12 ? lds_bytes * 2 : 0, stream, dspec, *a);
        return hipGetLastError();
    }
    if (a->mode == 13) {                        // the launch floor: an empty kernel in the STEP kernel's launch shape (ngw_debug_launch_floor)
        hipLaunchKernelGGL(ngw_nop_kernel, dim3(grid), dim3(NGW_EPB), lds_bytes, stream, dspec, *a);
        return hipGetLastError();
    }
    if (a->mode == NGW_MODE_STEP) return ngw_part_step(dspec, a, map_mode, feat, grid, lds_bytes, stream);
    if (a->mode == NGW_MODE_ROLLOUT || a->mode == NGW_MODE_ROLLOUT_ACT) {
        switch (map_mode) {
        case NGW_MAP_STRAIGHT: return ngw_part_rollout_straight(dspec, a, feat, grid, lds_bytes, stream);
        case NGW_MAP_DWORD: return ngw_part_rollout_dword(dspec, a, feat, grid, lds_bytes, stream);
        default: return ngw_part_rollout_byte(dspec, a, feat, grid, lds_bytes, stream);
        }
    }
    const bool lidar = (feat & 1) != 0;
    switch (map_mode) {
    case NGW_MAP_STRAIGHT: return launch_general<NGW_MAP_STRAIGHT>(dspec, a, lidar, grid, lds_bytes, stream);
    case NGW_MAP_DWORD: return launch_general<NGW_MAP_DWORD>(dspec, a, lidar, grid, lds_bytes, stream);
    default: return launch_general<NGW_MAP_BYTE>(dspec, a, lidar, grid, lds_bytes, stream);
    }
}
#endif  // NGW_HAS(0)
