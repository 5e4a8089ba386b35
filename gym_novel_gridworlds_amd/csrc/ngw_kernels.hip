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

#include "ngw_newepisode.inc"

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

#include "ngw_lidar_march.inc"

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
