// ngw_device.h — launch descriptor shared by the kernel TU (ngw_kernels.hip) and the C-ABI TU (ngw_abi.cpp).
#ifndef NGW_DEVICE_H
#define NGW_DEVICE_H
#include <stdint.h>

#include "../../include/ngw.h"

#define NGW_EPB 64            /* envs per workgroup = one CDNA wavefront, one lane per env */

enum { NGW_MODE_STEP = 0, NGW_MODE_RESET = 1, NGW_MODE_ROLLOUT = 2,
       NGW_MODE_REFILL = 3 /* prepare next episodes in the shadow buffers: a.b = shadow set, a.actions = the main episode[] */,
       NGW_MODE_ROLLOUT_ACT = 4 /* fused rollout with the caller's actions: step t of env e takes a.actions[t * a.t0 + e] */,
       NGW_MODE_DBG_NOP = 8 /* exit at once: launch floor */, NGW_MODE_DBG_COPY = 9 /* stage in/out, no step logic */ };
/* how a wave's map chunk is laid out in LDS: same image as HBM / odd-dword-padded rows / byte-granular (odd S) */
enum { NGW_MAP_STRAIGHT = 0, NGW_MAP_DWORD = 1, NGW_MAP_BYTE = 2 };

/* Device buffers of one handle.  map/loc/facing/inv are the batched observation AND the state, updated IN PLACE:
 * a step writes through only the bytes it changes (a map cell, a few inventory slots, the agent pose); a reset
 * rewrites the wave's whole chunk with coalesced stores.  Each env is owned by exactly one lane, so there is no
 * intra-launch hazard; launches are ordered by the stream. */
struct NgwBufs {
    int8_t* map;          /* [n_pad][S*S]  */
    int32_t* loc;         /* [n_pad][2]    */
    int32_t* facing;      /* [n_pad]       */
    int32_t* inv;         /* [n_pad][K]    */
    uint8_t* selected;    /* [n_pad] item id, 0 = ''  */
    int32_t* step_count;  /* [n_pad] */
    uint32_t* episode;    /* [n_pad] reset counter, keys the Philox stream */
    int32_t* reward;      /* [n_pad] */
    uint8_t* done;        /* [n_pad] */
    uint32_t* info;       /* [n_pad] packed, see NGW_INFO_* */
    uint32_t* flags;      /* [1] sticky NGW_F_* */
    uint32_t* flags_host; /* single-wavefront handles (host mirror, NgwMirror): the same word in GPU-addressable host memory, or nullptr */
    uint16_t* perm;       /* [S*S][n_pad] shuffle scratch of the subset reset passes, or nullptr */
    uint32_t* brd;        /* [n_pad][BS] occupancy bit rows of the maps (NGW_BOARD_*), part of the state slab; nullptr for maps beyond 32 x 32 */
};

/* Occupancy bit rows ("boards") of a map: word r of an env = bit c set iff map[r][c] != 0, BS = S rounded up to a multiple of 4 words per
 * env (rows >= S are zero), for maps up to 32 x 32.  Auxiliary state of the O(1) LidarInFront observation (ngw_lean.inc, lidar_boards_rows):
 * the in-place step kernel reads a lane's BS words with its prologue loads, cuts the agent's row, column and two diagonals out of them in
 * registers and finds a ray's first block with one find-first-bit - no map in LDS, no march.  Kept coherent by whoever changes a map while
 * the mode is on (NgwLaunch::l_boards): the step kernel's own cell writes, the prepared-episode copies of its cold path, ngw_boards_kernel
 * after every launch that rewrites maps wholesale (resets, refills, rollouts, ngw_set_state). */
#define NGW_BOARD_MAX_S 32
#define NGW_BOARD_STRIDE(S) (((S) + 3) & ~3)

struct NgwLaunch {
    NgwBufs b;
    int64_t n, n_pad, env_base, t0;
    uint64_t seed, action_seed;
    const int32_t* actions;      /* device, NGW_MODE_STEP */
    const uint8_t* reset_mask;   /* device or nullptr, NGW_MODE_RESET */
    int32_t mode, n_steps, autoreset, horizon;
    int32_t S, S2, MS, K, KP, CW; /* MS = LDS bytes per env map (MS/4 odd), KP = K|1 LDS inventory stride, CW = candidate words */
    uint32_t off_map;            /* LDS dword offset of the maps (0, or a guard when the lidar epilogue is fused) */
    /* fused LidarInFront epilogue (kernel template flag LIDAR) */
    const struct NgwLidarDev* lcfg;
    int32_t* lout;               /* [n_pad][lidar_len] */
    int32_t lidar_len, l_beams, l_range, l_chan, l_inv;
    int32_t l_fmt;               /* row format NGW_LFMT_*: int32 / int16 (values saturate at 32767) / packed (uint8 beam entries + int16 inventory tail) */
    int32_t l_world;             /* 1: the ray table is one world-frame table rotated by the facing (NgwLidarDev::woff): wave-uniform ray offsets;
                                  * 2: and it is the reference's default 8-beam table on a NGW_LIDAR_CONST_S map: compile-time offsets */
    int32_t l_rb, l_invoff;      /* bytes per observation row in this format; byte offset of its inventory tail */
    int32_t l_boards;            /* 1: the observation is built from the occupancy bit rows (NgwBufs::brd) by the in-place step kernel / ngw_lidar_boards_kernel */
    int32_t BS;                  /* words per env in NgwBufs::brd (NGW_BOARD_STRIDE(S)), 0 = no boards for this map size */
    uint32_t off_litem;          /* LDS dword offset of the two item tables (chan_of_item | inv_item: 12 dwords) */
    uint32_t off_ltab, off_ltile; /* ... of the per-lane ray table (8 KiB; only when !l_world) and of the observation tile (64 rows, l_rb bytes each) */
    int32_t perm_lds;            /* AddItem shuffle array: 1 = LDS at off_perm ([S2][32] u16, two half-wave batches), 0 = HBM scratch */
    uint32_t off_perm;
    uint32_t magic;              /* ceil(2^32 / (S2/4)) (or / S2 for odd S): exact division of chunk offsets */
    uint32_t magicK;             /* ceil(2^32 / K): exact division of inventory chunk offsets (< 64*K) */
    uint32_t magicS;             /* ceil(2^32 / S): cell / S */
    uint32_t off_inv, off_cand, off_act;    /* LDS dword offsets */
    uint64_t* stamps;            /* diagnostics builds (-DNGW_STAMPS): [grid][16] in-kernel clock stamps, or nullptr */
    uint32_t seq;                /* single-wavefront handles: the step refreshes the host mirror (NgwMirror) and then writes this number to
                                  * flags_host[NGW_SEQ_WORD] (0 = neither) */
    int32_t action0;             /* one-env handles: the step's action travels in the argument block ... */
    int32_t use_action0;         /* ... when this is 1 (`actions` still points at valid device memory); 2: `actions` holds one BYTE per env (ngw_step_host_packed) */
    /* fused rollouts: per-step output rows and per-env episode accumulators (ngw_rollout_outputs), any of them nullptr */
    int32_t* row_reward;         /* [n_steps][row_stride]: reward of step t of env e at [t * row_stride + e] */
    uint8_t* row_done;           /* [n_steps][row_stride]: 1 where the step ended an episode (done, or the horizon under autoreset) */
    int64_t row_stride;
    int32_t* acc;                /* [4][n_pad]: return / length of the running episode, sum of returns / count of the finished ones */
    uint32_t bid0;               /* a per-launch step over a SLICE of the batch: the slice's first block (first env / 64); b.*, actions and the lidar rows
                                  * point at the slice's first env, n is the slice's env count, the cold path adds bid0 to index the blob's unshifted arrays */
};

/* Uniform step parameters: every lane uses the same value, so the kernel reads them with SCALAR loads straight
 * from the HBM blob (no LDS latency in the step's dependency chain). */
struct NgwStepU {
    uint32_t brk_mask, ent_mask, rew_mask;  /* bit i: item i is breakable / an entity / gives break_reward when broken */
    uint32_t brk2_mask;                     /* bit i: breaking item i without an axe yields 2 (BreakIncrease) */
    int32_t n_actions, reward_step, reward_done, break_reward;
    uint8_t cost_forward, cost_turn, cost_break, cost_place, cost_extract, cost_select, table_item, goal_item;
    uint8_t place_item, place_near, n_entities, ext_src, ext_near, ext_out, ext_qty, ext_consume;
    uint8_t ext_cost_ok, axe_item, axe_cost, axe_qty;
    uint8_t cost_chop, cost_jump;
    int8_t chop_reward;
    uint8_t feat;                           /* NGW_FEAT_*: which optional action kinds the spec has (uniform skip of their reads) */
    int8_t place_reward, ext_reward, axe_reward;
    uint8_t axe_required;                   /* AxetoBreak*: Break fails without the selected axe */
    uint32_t _pad64;                        /* 64 bytes: the kernels fetch the struct as 16 dwords with ONE scalar load */
};

enum { NGW_FEAT_JUMP = 1, NGW_FEAT_CHOP = 2 };

/* Lean step kernel (ngw_lean.inc): per-action micro-op table, NGW_LEAN_DW dwords per action, held one action per lane and
 * fetched with ds_bpermute.  A step evaluates a per-lane vector of condition bits NGW_CB_* once; an action names the two
 * conditions that make it fail (outcome s = 1 if A holds, else 2 if B holds, else 0 = success) and carries, per outcome,
 * the message / message argument / cost code, plus what a success does (move, turn, front-cell write, one inventory slot
 * delta, select, reward).  No per-kind code is left in the kernel.
 *   e0 = kind | n_inputs<<4 | needs_table<<7 | arg<<8 | argconst<<16 | is_break<<24
 *   e1 = recipe input item ids (4 bytes, dict order)         e2 = recipe input quantities (4 bytes, 0 beyond n_inputs)
 *   e3 = const slot | cell value<<8 | (int16) slot delta<<16
 *   e4 = A bit | B bit<<4 | msg[0..2]<<8 (4 bits each) | argsel[0..2]<<20 (2 bits each: 0 none, 1 block in front, 2 argconst,
 *        3 recipe<<8|missing mask) | move<<26 (1 front, 2 two ahead) | turn<<28 (1 left, 2 right) | cell write<<30 | select<<31
 *   e5 = reward const (int8) | reward condition bit<<8 | slot select<<12 (0 none, 1 block in front, 2 const slot) | cost[0..2]<<14 (6 bits each)
 * An all-zero entry is a no-op (what an out-of-range action id fetches). */
#define NGW_LEAN_DW 6
enum { NGW_CB_FALSE = 0, NGW_CB_FRONT_NZ = 1, NGW_CB_JUMP_BLOCKED = 2, NGW_CB_NOT_BRK = 3, NGW_CB_NO_PLACE_ITEM = 4, NGW_CB_NOT_SRC = 5,
       NGW_CB_NOT_NEAR = 6, NGW_CB_MISSING = 7, NGW_CB_NEED_TABLE = 8, NGW_CB_NO_ARG_ITEM = 9, NGW_CB_NEED_AXE = 10,
       NGW_CB_NEAR_PLACE = 11, NGW_CB_BRK_REWARD = 12, NGW_CB_TRUE = 13 };

/* Prepared next episodes (ngw_set_reset_prefetch): shadow buffers holding the first states of the next `depth` episodes of
 * every env (depth = dmask + 1, a power of two).  The row of episode E of env e is row (E & dmask) * stride + e of every
 * array, and it is valid iff episode[row] == E; a reset whose new episode number matches copies it instead of running the
 * placement loop.  All null when the feature is off.  Read only on the cold reset path, with scalar loads from the HBM blob. */
#define NGW_MAX_DEPTH 8          /* prepared episodes per env, at most (ngw_set_reset_prefetch_depth) */
#define NGW_SEQ_WORD 8           /* flags_host[8]: sequence number of the last finished step launch (the host polls it instead of a stream sync) */
struct NgwNx {
    int8_t* map;          /* [depth][n_pad][S*S] */
    int32_t* loc;         /* [depth][n_pad][2]   */
    int32_t* facing;      /* [depth][n_pad]      */
    int32_t* inv;         /* [depth][n_pad][K]   */
    uint32_t* episode;    /* [depth][n_pad] episode the row was prepared for (0 = nothing prepared) */
    uint32_t* brd;        /* [depth][n_pad][BS] occupancy bit rows of the prepared maps (valid while the boards mode is on), or nullptr */
    uint32_t* slow;       /* [0] resets that found their row stale and ran the placement loop inside a step (cumulative); [1] refills run so far */
    uint32_t* slow_host;  /* [2] GPU-addressable host words every refill copies `slow` to: the host adapts depth and cadence */
    int64_t stride;       /* rows per slot = n_pad */
    int32_t dmask;        /* depth - 1 */
    int32_t _pad;
};

/* What the cold reset path needs besides the spec: static per handle, read there with scalar loads from the HBM blob
 * (passing them as arguments of the out-of-line reset would spill the call's argument list to scratch). */
struct NgwResetU {
    uint16_t* perm;       /* = NgwBufs.perm */
    int8_t* map;          /* = NgwBufs.map / .inv of the MAIN buffer set: rows a consumed prepared episode is stored to */
    int32_t* inv;
    int64_t n_pad;
    uint64_t seed;
    int32_t S, S2, K, CW, perm_lds;
    uint32_t magicS;
    uint32_t off_rng;     /* LDS dword offset of the Philox word ring [32][64] of the reset path */
    /* the bytes of ngw_spec the reset reads, packed (8 dwords): fetched together with the rest of this struct, so the
     * reset never waits on one more dependent load of the spec for each optional pass */
    uint8_t wall_item, tap_item, tap_near, n_place;
    uint8_t n_passes, n_inv_start, _pad[2];
    uint8_t inv_start_item[NGW_MAX_INV_START], inv_start_qty[NGW_MAX_INV_START];
    uint32_t pass[NGW_MAX_PASSES];         /* kind | item << 8 | from << 16 | percent span << 24 of each shuffled-subset pass */
};

/* Uniform parameters of the step-time novelty predicates (kernel template flag EXT): FireWall, FenceRestriction, Crate */
struct NgwExtU {
    int32_t fire_item, fire_reward, fence_item, fence_mode, crate_item;
    uint32_t crate_add[3];                  /* 4 bits per item id: how many of it a crate holds */
    uint32_t nest;                          /* ngw_spec.ext_flags | fire_skip_recipe << 8 (wrapper nesting of a stack) */
};

/* Blob kept in HBM (one per handle). */
#define NGW_MAX_PLACE 64            /* items placed by one reset (sum of items_quantity); reference: 6-7 */
/* Host mirror of a single-wavefront handle (the gym.Env adapter: n = 1): the same rows in page-locked host memory the GPU
 * addresses directly.  The state itself lives in HBM like any other handle's; a step (or explicit reset) launched by
 * ngw_step_host / ngw_reset_host copies the wave's rows here before it signals completion, so the host reads its results
 * in place - no copy call, no stream synchronisation - and the kernel never READS across PCIe.  All nullptr: no mirror. */
struct NgwMirror {
    int8_t* map; int32_t* loc; int32_t* facing; int32_t* inv; uint8_t* selected; int32_t* step_count;
    int32_t* reward; uint8_t* done; uint32_t* info;
};

/* Terminal observations under same-step autoreset (ngw_set_terminal_capture): before a resetting env's rows are overwritten by its
 * next episode, the step kernel's cold path copies them here - map row, inventory row, pose of the state the episode ENDED in
 * (reference loops look at that observation: tests/test.py:30-41, enjoy.py:107-116).  Row e is valid for the envs whose `done` the
 * same launch set; all null = off (one scalar load and a uniform branch on the cold path, nothing on the hot path). */
struct NgwTerm {
    int8_t* map;          /* [n_pad][S*S] */
    int32_t* loc;         /* [n_pad][2]   */
    int32_t* facing;      /* [n_pad]      */
    int32_t* inv;         /* [n_pad][K]   */
};

/* Host write-through of the step kernel (ngw_step_host_packed's steady state on the in-place kernel): the caller's page-locked block is a
 * mirror of the state and is mapped into the GPU's address space - the step kernel itself stores what the step changes into it across
 * PCIe (the map cells and inventory slots it writes, whole rows of the envs that start a new episode, pose / reward / done / info of every
 * env), counts its finished blocks and the last one publishes the error flags and the launch's sequence number, which the host polls:
 * one launch per host step, no delta kernel behind it, no stream synchronisation.  All null = off. */
struct NgwWT {
    int8_t* map;          /* [n][S*S]  section 0 of the block (device address of the mapped host memory) */
    int32_t* inv;         /* [n][K]    section 1 */
    uint32_t* pose;       /* [n]       section 2: r | c << 8 | facing << 16 | selected << 24 */
    int32_t* reward;      /* [n]       section 3 */
    uint8_t* done;        /* [n]       section 4 */
    uint32_t* info;       /* [n]       section 5 */
    uint32_t* flags;      /* section 6: [0] error flags, [1] sequence number of the last finished step */
    uint32_t* count;      /* device memory: blocks of the current launch that have finished */
    uint8_t* rows;        /* fused LidarInFront observation: the caller's row buffer ([n_pad] rows, ngw_lidar_host_rows), or null */
};

struct NgwDevSpec {
    NgwStepU u;
    uint8_t place_seq[NGW_MAX_PLACE];   /* item id of the n-th placement of a reset (items_quantity flattened in order): the reset paths copy it to LDS */
    int32_t n_place;
    uint32_t act_lean[NGW_MAX_ACTIONS * NGW_LEAN_DW];   /* lean step kernel: micro-op table, see NGW_LEAN_DW */
    NgwLaunch lp;                /* launch prototype (layout + buffer pointers): the lean kernel's cold reset path reads it from here */
    NgwLaunch lp_ns;             /* the same for the no-stage lean kernel (its own, small LDS layout: inventory rows | candidate masks | placement sequence) */
    ngw_spec sp;                 /* full spec: the (cold) reset path reads it with scalar loads */
    NgwExtU x;
    NgwNx nx;
    NgwResetU ru;
    NgwMirror mir;
    NgwTerm term;
    NgwWT wt;
    double pctq[NGW_MAX_PASSES][64];   /* per reset pass: pct / 100.0 for pct in [pct_lo, pct_hi) as the host's IEEE double */
};

#ifdef __cplusplus
extern "C"
#endif
hipError_t ngw_launch(const NgwDevSpec* dspec, const NgwLaunch* a, int map_mode, int feat /* 1 = fused lidar, 2 = EXT, 8 = no-stage step */, unsigned grid, size_t lds_bytes,
                      hipStream_t stream);
/* Arguments of the dedicated new-episode kernel (ngw_reset.inc: explicit resets and prepared next episodes of the plain
 * configurations and of those with ONE subset pass over the air of the interior (AddItem / Crate) or the wall ring (ReplaceItem /
 * FireWall of the wall item)). */
struct NgwResetFast {            // kernel arguments (by value)
    NgwBufs main;                // the handle's state (episode counters; RESET destination)
    NgwNx nx;                    // shadow rows (REFILL destination)
    const uint8_t* reset_mask;   // RESET: device mask or nullptr (all)
    const double* pctq;          // AddItem: pct / 100.0 table of the host
    int64_t n, env_base;
    uint64_t seed;
    uint32_t* flags;
    int32_t mode;                // NGW_MODE_RESET / NGW_MODE_REFILL
    int32_t S, S2, K, CW, n_place, wall_item;
    int32_t additem_item, additem_span;   // the subset pass: item written, width of its percent range
    uint32_t seq;                // single-wavefront handles: refresh the host mirror, then write this to flags_host[NGW_SEQ_WORD] (0 = neither)
    int32_t pass_wall;           // the subset pass replaces WALL cells (ReplaceItem / FireWall of the ring) instead of filling air cells
    int32_t n_inv_start;
    uint32_t inv_start_items, inv_start_qtys;   // 4 bytes each
    uint32_t magicW;             // ceil(2^32 / (S-4))
    uint32_t magicS;             // ceil(2^32 / S)
    int32_t sub_nb, sub_fields;  // subset pass candidates: bits of a cell index (bit_length(S*S - 1)), fields per 32-bit word
    uint32_t magic_tail;         // ceil(2^32 / store units of the last (short) 128-byte window of a row), 0 if S*S % 128 == 0
    uint32_t magicS2;            // ceil(2^32 / (S*S)): byte offset inside the wave's chunk -> env
    int32_t img;                 // rows up to 512 bytes: the LDS tile is the exact image of the wave's 64 rows, stored as one coalesced run
    uint32_t off_ring, off_masks, off_placed, off_tmpl, off_dom, off_mcol, off_tile;    // LDS dword offsets
    uint32_t off_ctab;           // [64] destination rows of a pass (u32) + [64 * NGW_MAX_DEPTH] stale (env, slot) pairs of a compacting refill (u16)
    int32_t boards, BS;          // boards mode: also write the occupancy bit rows (NgwBufs::brd / NgwNx::brd, BS words per env) of every map made or copied
    uint64_t* stamps;            // diagnostics builds (-DNGW_STAMPS), or nullptr
};
#ifdef __cplusplus
extern "C"
#endif
hipError_t ngw_reset_fast_launch(const NgwDevSpec* dspec, const struct NgwResetFast* a, int nw, int subset, unsigned grid, size_t lds_bytes,
                                 hipStream_t stream);

/* Row formats of the LidarInFront observation (ngw_lidar_set_output): what one env's row looks like in the device buffer. */
enum { NGW_LFMT_I32 = 0,      /* int32 [B * NC + NI] */
       NGW_LFMT_I16 = 1,      /* int16 [B * NC + NI], values saturate at 32767 */
       NGW_LFMT_PACKED = 2 }; /* uint8 [B * NC] beam entries (a range is <= 64), padded to an even count, then int16 [NI] inventory (saturating) */

/* The reference's DEFAULT LidarInFront rays - 8 beams (observation_wrappers.py:16), i.e. the four axes and the four diagonals, the
 * diagonals advancing by round(0.71 k) cells (np.round(np.cos(pi / 4), 2) = 0.71, :49-55) - as compile-time constants: world ray w
 * (0 .. 7 = +r, +r+c, +c, -r+c, -r, -r-c, -c, +r-c), range k >= 1.  With the map size a compile-time constant too, every cell
 * offset of the march is an instruction immediate.  ngw_lidar_configure compares this table with the host's entry by entry and only
 * an exact match (NgwLaunch::l_world == 2) selects the kernels' constant-offset march. */
#define NGW_LIDAR_CONST_S 10                 /* the map size the constant march is instantiated for: the reference's default */
static constexpr int ngw_lidar8_diag(int k) { return (71 * k + 50) / 100; }        /* round(0.71 k): no tie for k < 50 */
static constexpr int ngw_lidar8_dr(int w, int k) {
    return w == 0 ? k : (w == 4 ? -k : ((w == 1 || w == 7) ? ngw_lidar8_diag(k) : ((w == 3 || w == 5) ? -ngw_lidar8_diag(k) : 0)));
}
static constexpr int ngw_lidar8_dc(int w, int k) {
    return w == 2 ? k : (w == 6 ? -k : ((w == 1 || w == 3) ? ngw_lidar8_diag(k) : ((w == 5 || w == 7) ? -ngw_lidar8_diag(k) : 0)));
}

/* Device-side lidar tables, built by ngw_lidar_configure from ngw_lidar_cfg: flat cell offsets dr * S + dc. */
struct NgwLidarDev {
    int16_t off[4][NGW_LIDAR_MAX_BEAMS][NGW_LIDAR_MAX_RANGE];   /* [facing][beam][range-1], 8 KiB */
    uint8_t chan_of_item[NGW_MAX_ITEMS];
    uint8_t inv_item[NGW_MAX_ITEMS];
    int32_t num_beams, max_range, n_chan, n_inv;
    /* World-frame form of the same table, valid when `world` is set: with num_beams a multiple of 4 the four facings shoot the
     * SAME num_beams directions, only numbered from a different start (observation_wrappers.py:38-43: the angles are
     * direction_radian[facing] - pi + b * 2 pi / B and the four direction_radian values are multiples of 2 pi / 4), so ray b of
     * facing f is world ray (u_f + b - B / 2) mod B with u_f = B / 2, 0, 3 B / 4, B / 4 for NORTH, SOUTH, WEST, EAST.  ngw_lidar_configure
     * checks that property on the host's table entry by entry; the offsets of a ray are then the same for every lane of a wave
     * (scalar loads, no table in LDS).  Ranges beyond max_range repeat the last in-range cell (rows padded to a multiple of 4). */
    int32_t woff[NGW_LIDAR_MAX_BEAMS][NGW_LIDAR_MAX_RANGE];
    int32_t world;
};

#ifdef __cplusplus
extern "C"
#endif
/* Region copies in ONE launch (the "pack" kernel): region r = nbytes[r] bytes from src[r] to dst[r], coalesced 16-byte pieces
 * when both ends are 16-byte aligned.  Three users: the small-batch host step (device regions -> one page-locked host buffer
 * the GPU addresses directly, replacing seven hipMemcpyAsync calls), ngw_pack_obs (the SoA observation / output arrays ->
 * one contiguous payload for the multi-GPU gather) and ngw_unpack_obs (the gathered payloads -> global arrays on the root). */
#define NGW_PACK_MAX 64
struct NgwPack {
    const uint8_t* src[NGW_PACK_MAX];
    uint8_t* dst[NGW_PACK_MAX];
    uint64_t nbytes[NGW_PACK_MAX];
    int32_t n_regions;
};
#ifdef __cplusplus
extern "C"
#endif
hipError_t ngw_pack_launch(const struct NgwPack* p, hipStream_t stream);
/* Delta refresh of the host mirrors (ngw_step_host on big batches): per region the live array, the device-side shadow of what
 * the host holds, and the host mirror itself as a mapped pointer; pieces of 16 bytes that differ are written to both. */
#define NGW_DIFF_MAX 4
struct NgwDiff {
    const uint8_t* cur[NGW_DIFF_MAX];
    uint8_t* shadow[NGW_DIFF_MAX];
    uint8_t* host[NGW_DIFF_MAX];
    uint64_t nbytes[NGW_DIFF_MAX];
    int32_t n_regions;
};
#ifdef __cplusplus
extern "C"
#endif
hipError_t ngw_diff_launch(const struct NgwDiff* p, hipStream_t stream);
/* Narrow wire format of the host step (ngw_step_host_packed): per env the pose as four bytes (r, c, facing, selected), the reward as
 * int16, done as a byte and the packed info word, written as four dense arrays into one staging payload that a single copy brings
 * across PCIe (11 B per env instead of the 26 B of the int32 SoA arrays). */
struct NgwWire {
    const int32_t* loc; const int32_t* facing; const uint8_t* selected; const int32_t* reward; const uint8_t* done; const uint32_t* info;
    const uint32_t* flags;
    uint32_t* pose; int32_t* reward32; uint8_t* done8; uint32_t* info32; uint32_t* flags_out;
    int64_t n;
};
#ifdef __cplusplus
extern "C"
#endif
hipError_t ngw_wire_launch(const struct NgwWire* p, hipStream_t stream);
#ifdef __cplusplus
extern "C"
#endif
/* delta refresh (NgwDiff) and narrowing (NgwWire) as ONE launch */
hipError_t ngw_diff_wire_launch(const struct NgwDiff* p, const struct NgwWire* w, hipStream_t stream);
#ifdef __cplusplus
extern "C"
#endif
/* AgentMap window gather: out = [n][2V+1][2V+1] int8 packed as n_dwords dwords */
hipError_t ngw_agent_view_launch(const int8_t* map, const int32_t* loc, uint32_t* out, uint32_t n_dwords, int S, int V,
                                 hipStream_t stream);
#ifdef __cplusplus
extern "C"
#endif
hipError_t ngw_lidar_launch(const NgwLaunch* a /* with the stand-alone launch's own LDS offsets */, int map_mode, unsigned grid, size_t lds_bytes,
                            hipStream_t stream);
/* Occupancy bit rows (NGW_BOARD_*): rebuild them for `rows` maps (ngw_boards_kernel), and the LidarInFront observation built from them as
 * its own launch (ngw_lidar_boards_kernel) */
#ifdef __cplusplus
extern "C"
#endif
hipError_t ngw_boards_launch(const NgwLaunch* a, int map_mode, const int8_t* map, uint32_t* brd, int64_t rows, size_t lds_bytes, hipStream_t stream);
#ifdef __cplusplus
extern "C"
#endif
hipError_t ngw_lidar_boards_launch(const NgwLaunch* a, unsigned grid, size_t lds_bytes, hipStream_t stream);

/* The one-env handle's resident step loop (ngw_solo.inc): kernel arguments.  The kernel stays resident between the steps of a Python loop,
 * speculates the outcome of every action from the committed state into host-visible records, and commits the action the host posts. */
struct NgwSolo {
    NgwBufs b;                   // env 0's rows (the handle's ordinary state arrays)
    volatile uint32_t* mbox;     // host -> device, page-locked host memory: [0] command sequence number, [1] action, [2] quit
    uint32_t* out;               // device -> host, page-locked host memory: [0] sequence of the speculated state, [1] exited, [NGW_SOLO_REC0 ..) records
    int32_t S, S2, K, A, MSp;    // MSp = bytes per private map copy (S2 rounded up to 16)
    uint32_t k0;                 // sequence number of the state the launch starts from
    int32_t commit0;             // an action the host posted while the previous launch was exiting: committed first (-1 = none)
    uint32_t timeout_ticks;      // idle time after which the loop ends (100 MHz ticks)
    uint32_t off_map, off_master, off_inv, off_invb;   // LDS dword offsets: private maps [64][MSp] | master map | inventory rows [64][KP] | their backup
    int32_t KP, rec_dw;          // LDS inventory row stride; dwords per record (8 + K, rounded up to 4)
};
#define NGW_SOLO_REC0 16         // records start at this dword of `out`
// record dwords: [0] reward  [1] info  [2] done | r << 8 | c << 16 | f << 24  [3] selected | wcell << 8 | cell value << 16  [4] step_count
//                [5] cell index of the write  [6] 3 x 3 pick-up mask around the new position  [7] -  [8 ..) the inventory row
#ifdef __cplusplus
extern "C"
#endif
hipError_t ngw_solo_launch(const NgwDevSpec* dspec, const struct NgwSolo* p, int ext, size_t lds_bytes, hipStream_t stream);

#endif
