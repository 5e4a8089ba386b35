/* ngw.h — C-ABI of the MI355X-native batched step()/reset() hot path of gym-novel-gridworlds.
 *
 * The reference has no FFI of its own: its hot path sits behind the OpenAI-gym `gym.Env`
 * Python API (reference: gym_novel_gridworlds/envs/pogostick_v1_env.py:86 reset, :230 step,
 * :214 get_observation; envs/bow_v1_env.py same lines; novelty_wrappers.py:117-213 AxeMedium,
 * :991-1034 AddItem, :1586 inject_novelty).  This header is the boundary a maintainer would bind
 * from Python with ctypes (stub shown in INTEGRATION.md): plain pointers and sizes only.
 *
 * One handle = N independent environments resident on ONE GPU (one process per GPU; shard by
 * creating one handle per rank with `env_index_base = rank * N`).  All state lives in HBM as
 * structure-of-arrays; the observation buffers ARE the state, updated in place, see DESIGN.md.
 *
 * Every function returns 0 on success or a negative NGW_E_* code; ngw_last_error() gives the
 * thread-local message.  A handle is not thread-safe (one host thread per handle), matching the
 * single-threaded reference.
 */
#ifndef NGW_H
#define NGW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGW_ABI_VERSION 3    /* 3: the narrow wire format carries int32 rewards (ngw_host_step_layout_packed: section 3); the lidar rows default to int32 */

#define NGW_MAX_ITEMS 24        /* reference asserts len(items) <= max_items = 20 (pogostick_v1_env.py:75,220) */
#define NGW_MAX_ACTIONS 48
#define NGW_MAX_RECIPES 8
#define NGW_MAX_RECIPE_INPUTS 4  /* the reference's recipes have <= 3 inputs (pogostick_v1_env.py:56-59) */
#define NGW_MAX_START_ITEMS 8
#define NGW_MAX_INV_START 4
#define NGW_MAX_PASSES 4          /* shuffled-subset reset passes of one stack of wrappers */
#define NGW_MAX_MAP_SIZE 64     /* S; the LDS-resident kernel supports S*S <= 4096 */

/* error codes */
#define NGW_OK 0
#define NGW_E_INVALID_ARG (-1)
#define NGW_E_HIP (-2)           /* a HIP runtime call failed; no CPU fallback exists */
#define NGW_E_INVALID_ACTION (-3)/* reference: ValueError "<a> is not in list" (pogostick_v1_env.py:236) */
#define NGW_E_PLACEMENT (-4)     /* reference: AssertionError "Cannot place items, increase map size!" (:167) */
#define NGW_E_NO_DEVICE (-5)

/* device-side sticky error flags (ngw_error_flags) */
#define NGW_F_INVALID_ACTION 1u
#define NGW_F_PLACEMENT 2u

/* action kinds (act_kind[]); act_arg[] = recipe index (CRAFT) or item id (SELECT) */
enum { NGW_ACT_FORWARD = 0, NGW_ACT_LEFT = 1, NGW_ACT_RIGHT = 2, NGW_ACT_BREAK = 3, NGW_ACT_PLACE = 4,
       NGW_ACT_EXTRACT = 5, NGW_ACT_CRAFT = 6, NGW_ACT_SELECT = 7,
       NGW_ACT_CHOP = 8 /* AddChopAction, novelty_wrappers.py:1267 */, NGW_ACT_JUMP = 9 /* AddJumpAction, :1340 */ };

/* info['message'] codes; the host formats the string (reference strings cited in spec.py) */
enum { NGW_XF_FIRE_SKIP_BREAK = 1, NGW_XF_CRATE_IN_FENCE = 2 };
enum { NGW_PASS_ADDITEM = 1, NGW_PASS_REPLACE = 2, NGW_PASS_FENCE = 3 };   /* ngw_spec.pass_kind */
/* passes sampled without an index array (see ngw_spec.n_passes): AddItem / Crate (air of the interior), ReplaceItem / FireWall of the wall ring */
#define NGW_PASS_SPARSE(kind, from, wall_item) ((kind) == NGW_PASS_ADDITEM || ((kind) == NGW_PASS_REPLACE && (from) == (wall_item)))
#define NGW_PASS_MARK 0x7F       /* transient cell value while such a pass runs (item ids are < NGW_MAX_ITEMS) */
enum { NGW_MSG_NONE = 0, NGW_MSG_BLOCK_IN_PATH = 1, NGW_MSG_CANNOT_BREAK = 2 /* arg = item */,
       NGW_MSG_PLACED = 3 /* arg = item */, NGW_MSG_ALREADY_EXISTS = 4 /* arg = front item */,
       NGW_MSG_NOT_IN_INVENTORY = 5, NGW_MSG_EXTRACT_NO_SRC = 6, NGW_MSG_EXTRACT_NOT_NEAR = 7,
       NGW_MSG_MISSING_ITEMS = 8 /* arg = recipe<<8 | mask over the recipe's inputs in dict order */,
       NGW_MSG_NEED_TABLE = 9, NGW_MSG_CRAFTED = 10 /* arg = crafted item */,
       NGW_MSG_NEED_AXE = 11 /* arg = axe item: "Cannot break without <axe> selected" */,
       NGW_MSG_CANNOT_CHOP = 12 /* arg = item */,
       NGW_MSG_FENCE_RESTRICTION = 13 /* "Cannot break due to fence restriction" */,
       NGW_MSG_FIRE_WALL = 14 /* "You died due to fire_wall" */ };

/* packed per-env info word produced by the step kernel:
 *   bit 0 result | bit 1 done | bits 2..7 cost code | bits 8..15 message code | bits 16..31 message arg */
#define NGW_INFO_RESULT(w) ((w) & 1u)
#define NGW_INFO_DONE(w) (((w) >> 1) & 1u)
#define NGW_INFO_COST(w) (((w) >> 2) & 63u)
#define NGW_INFO_MSG(w) (((w) >> 8) & 255u)
#define NGW_INFO_ARG(w) ((w) >> 16)

/* Immutable environment specification, compiled on the host from (env id, map_size, novelty args)
 * into flat integer look-up tables (SURVEY.md §8(a) a2 "LUT set").  `cost_*` fields are CODES into
 * the host-side step_cost table (the reference mixes Python float and int step costs,
 * pogostick_v1_env.py:257-470; 27.906975 is not float32-representable, so values never enter the GPU). */
typedef struct ngw_spec {
    int32_t abi_version;                 /* = NGW_ABI_VERSION */
    int32_t map_size;                    /* S  (pogostick_v1_env.py:30) */
    int32_t n_items;                     /* K  incl. air(0) and wall (set_items_id :200-212) */
    int32_t n_actions;                   /* A = len(actions_id) (:52-68); may exceed action_space.n with novelties */
    int32_t n_recipes;                   /* R */
    int32_t reward_step;                 /* -1  (:239) */
    int32_t reward_done;                 /* 50  (:82, forced while inv[goal] >= 1, :354-357) */
    uint8_t act_kind[NGW_MAX_ACTIONS];
    uint8_t act_arg[NGW_MAX_ACTIONS];
    uint8_t breakable[NGW_MAX_ITEMS];    /* item not in unbreakable_items (:41,:283) */
    uint8_t entity[NGW_MAX_ITEMS];       /* item in entities, picked up by grab_entities (:538-554) */
    int8_t break_reward[NGW_MAX_ITEMS];  /* +10 for tree_log else -1 (:288-289); BreakIncrease: +10 for every block */
    uint8_t break_qty[NGW_MAX_ITEMS];    /* blocks gained by Break without an axe: 1, or 2 (BreakIncrease, novelty_wrappers.py:1449-1454) */
    uint8_t wall_item, table_item, goal_item, n_entities;
    /* recipes (:56-59, craft :413-474) */
    uint8_t recipe_in[NGW_MAX_RECIPES][NGW_MAX_ITEMS];        /* required quantity per item id */
    uint8_t recipe_n_in[NGW_MAX_RECIPES];
    uint8_t recipe_in_item[NGW_MAX_RECIPES][NGW_MAX_RECIPE_INPUTS]; /* inputs in dict order (message order) */
    uint8_t recipe_out_item[NGW_MAX_RECIPES];
    uint8_t recipe_out_qty[NGW_MAX_RECIPES];
    uint8_t recipe_needs_table[NGW_MAX_RECIPES];             /* len(input) > 1 (:444) */
    uint8_t cost_missing[NGW_MAX_RECIPES], cost_no_table[NGW_MAX_RECIPES], cost_ok[NGW_MAX_RECIPES];
    /* reward of a successful craft: 10 Pogostick-v1 (:455) / 50 Bow-v1 (bow_v1_env.py:424); the craftable axe of
     * AxeHard / AxetoBreakHard always gives 10 (the wrappers' own craft(), novelty_wrappers.py:331) */
    int8_t recipe_reward[NGW_MAX_RECIPES];
    /* fixed-action cost codes */
    uint8_t cost_forward, cost_turn, cost_break, cost_place, cost_extract, cost_select;
    uint8_t cost_chop, cost_jump;        /* 3600.0 * 1.2 (novelty_wrappers.py:1294), 27.906975 * 2 (:1381) */
    int8_t chop_reward;                  /* reward_intermediate for any breakable block (:1303) */
    uint8_t _pad3;
    /* Place_<item> (:295-314): place `place_item` in front; +place_reward iff a 4-neighbour of the front cell is place_near */
    uint8_t place_item, place_near;
    int8_t place_reward;
    /* Extract_* (Pogostick :315-331, Bow bow_v1_env.py:293-304) */
    uint8_t ext_src, ext_near /* 0 = no adjacency requirement */, ext_out, ext_qty, ext_consume, ext_cost_ok;
    int8_t ext_reward;
    /* Break override of the axe novelties (novelty_wrappers.py:144-183); axe_item = 0 -> base Break */
    uint8_t axe_item, axe_cost, axe_qty;
    int8_t axe_reward;
    uint8_t axe_required;                /* AxetoBreak*: Break fails without the selected axe (novelty_wrappers.py:589-591) */
    uint8_t _pad2[3];
    /* reset (:86-157): items placed in insertion order of items_quantity */
    uint8_t n_start;
    uint8_t start_item[NGW_MAX_START_ITEMS];
    uint8_t start_qty[NGW_MAX_START_ITEMS];
    /* Pogostick-v0 reset pass (pogostick_v0_env.py:156-178): put one `tap_item` on a free 4-neighbour (random direction)
     * of a random `tap_near` block; tap_item = 0 -> disabled */
    uint8_t tap_item, tap_near;
    /* items present in the inventory after every reset: AxeEasy / AxetoBreakEasy (novelty_wrappers.py:29-35, :456-462: the
     * axe), AxetoBreakHard (:663-672: the axe's ingredients) */
    uint8_t n_inv_start;
    uint8_t inv_start_item[NGW_MAX_INV_START], inv_start_qty[NGW_MAX_INV_START];
    /* Shuffled-subset reset passes, in the order they run = the order their wrappers were injected (a wrapper's reset() calls
     * the wrapped env's first): np.where(<predicate>) row-major, np.random.shuffle, randint(pct_lo, pct_hi), then the first
     * ceil(len * pct / 100) cells are edited (never the agent cell).  pass_kind: NGW_PASS_ADDITEM - AddItem / Crate
     * (novelty_wrappers.py:1013-1034, :1071): air cells become pass_item; NGW_PASS_REPLACE - ReplaceItem / FireWall
     * (:1129-1148): cells holding pass_from become pass_item; NGW_PASS_FENCE - Fence / FenceRestriction (:867-889): every free
     * 8-neighbour of a chosen non-air, non-wall cell gets pass_item (add_fence_around, pogostick_v1_env.py:524-536).  Any
     * number of passes of the same kind may be stacked (additem + crate, fence + fencerestriction, replaceitem + firewall).
     *
     * How the subset is DRAWN on the device (per-(env, episode) Philox stream; the CPU oracle's Philox mode runs the same
     * steps, its MT19937 mode follows numpy call for call): passes whose source cells are the air of the interior or the
     * wall of the ring (NGW_PASS_SPARSE) never build the index array.  The result of shuffle + "first cnt" is a uniformly
     * random cnt-subset of the matching cells - the order inside it is irrelevant, every chosen cell gets the same item -
     * and that is what is sampled: percent first (the numpy bounded draw, as before), cnt = ceil(len * pct / 100), then
     * min(cnt, len - cnt) distinct matching cells by rejection (the complement when that is the smaller set).  A candidate
     * is a cell index of nb = bit_length(S*S - 1) bits: field j of word k of a Philox block (32 / nb fields per word, from
     * the low end), taken in the order j = 0: words 0..3, j = 1: words 0..3, ...; blocks start at the next block boundary
     * and the rest of the last one is discarded; a candidate beyond the map, not matching or already taken is skipped.
     * Same distribution of maps as the reference's, pinned by the distribution fixtures tests/golden/g6_*.npz; every other
     * pass keeps shuffle-then-prefix. */
    uint8_t n_passes;
    uint8_t pass_kind[NGW_MAX_PASSES], pass_item[NGW_MAX_PASSES], pass_from[NGW_MAX_PASSES];
    uint8_t pass_pct_lo[NGW_MAX_PASSES], pass_pct_hi[NGW_MAX_PASSES];
    /* FenceRestriction Break predicate (:906-988) on fence_item: fence_mode 0 none (fence, fencerestriction easy), 1 medium (no
     * fence beside the AGENT, across its facing), 2 hard (no fence in the 3x3 around the block in front) */
    uint8_t fence_item, fence_mode;
    /* FireWall.step (:1164-1200): after the step, a fire_item 4-neighbour of the agent -> reward fire_reward, done,
     * message 'You died due to fire_wall'; fire_item = 0 -> disabled */
    uint8_t fire_item;
    int8_t fire_reward;
    /* Crate.step (:1078-1092): Break with crate_item in front first adds crate_add[item] of every item to the
     * inventory (the ingredient multiset drawn at injection, :1055-1068); crate_item = 0 -> disabled */
    uint8_t crate_item;
    uint8_t crate_add[NGW_MAX_ITEMS];
    /* wrapper nesting of a stack, as far as the step can tell: NGW_XF_FIRE_SKIP_BREAK - the FireWall wrapper sits BELOW a
     * Break-overriding one (axe / axetobreak / breakincrease handle Break without calling the env they wrap), so its check
     * does not run on Break steps; fire_skip_recipe = 1 + recipe of a craftable axe whose wrapper sits above FireWall (its
     * Craft action is handled the same way); NGW_XF_CRATE_IN_FENCE - the Crate wrapper sits below FenceRestriction, so a
     * restricted Break never reaches it */
    uint8_t ext_flags, fire_skip_recipe;
} ngw_spec;

/* LidarInFront observation (reference gym_novel_gridworlds/observation_wrappers.py:10-80): `num_beams` rays at equally
 * spaced angles around the agent; per ray, the distance to the first non-air block, reported in the channel of that
 * block's item (0 if the block is not a lidar item or nothing is hit within max_range), followed by the inventory of
 * the breakable items in alphabetical order.  The host precomputes the integer ray offsets with the reference's own
 * float arithmetic (np.round(np.cos(angle), 2), np.round(range * ratio)), so the GPU only marches integers. */
#define NGW_LIDAR_MAX_BEAMS 16
#define NGW_LIDAR_MAX_RANGE 64
typedef struct ngw_lidar_cfg {
    int32_t num_beams;                   /* LidarInFront(env, num_beams) */
    int32_t max_range;                   /* int(sqrt(2 * (map_size - 2)^2)) at wrap time (observation_wrappers.py:25) */
    int32_t n_chan;                      /* len(lidar_items) = items without air and the goal item (:21-24) */
    int32_t n_inv;                       /* inventory entries appended (:74-75) */
    uint8_t chan_of_item[NGW_MAX_ITEMS]; /* 1-based lidar channel of an item id, 0 = not a lidar item */
    uint8_t inv_item[NGW_MAX_ITEMS];     /* item ids of the appended inventory, in sorted-name order */
    int8_t dr[4][NGW_LIDAR_MAX_BEAMS][NGW_LIDAR_MAX_RANGE];   /* [facing][beam][range-1] row offset */
    int8_t dc[4][NGW_LIDAR_MAX_BEAMS][NGW_LIDAR_MAX_RANGE];   /* column offset */
} ngw_lidar_cfg;

typedef struct ngw_handle ngw_handle;

int ngw_abi_version(void);
int ngw_spec_size(void);            /* sizeof(ngw_spec), checked by the host binding */
const char* ngw_last_error(void);
/* Number of visible GPUs (hipGetDeviceCount); 0 when there is none. */
int ngw_device_count(void);

/* Creates n_envs environments on GPU `device`.  `seed` keys the per-env counter-based reset streams
 * (Philox4x32-10, counter = (word block, episode, global env index)); `env_index_base` is the global index
 * of local env 0 so results do not depend on how envs are sharded over GPUs.  Replaces
 * `gym.make(id)` + attribute edits + `inject_novelty` (gym_novel_gridworlds/__init__.py:57-60). State is
 * undefined until ngw_reset / ngw_set_state. */
int ngw_create(const ngw_spec* spec, int64_t n_envs, int device, uint64_t seed, int64_t env_index_base,
               ngw_handle** out);
int ngw_destroy(ngw_handle* h);

/* autoreset = 0 reproduces the reference (sticky done, no time limit).  autoreset = 1 (classic gym.vector
 * "same-step" form): every call steps every env; an env whose step ended with done, or whose step_count
 * reached `horizon` (> 0), is reset in the same call and the observation returned is the new episode's first
 * one; reward/info are the terminal step's, done = 1 for both endings (info bit 1 set only for goal-done). */
int ngw_set_autoreset(ngw_handle* h, int autoreset, int horizon);
/* Run the handle's kernels on an external hipStream_t (e.g. torch's current stream); NULL = own stream. */
/* Prepared next episodes.  A reset (explicit, or the same-step autoreset) normally runs the reference's placement loop
 * (~40 dependent random draws per env) inside the step launch; when only a few envs of a batch end in a given step, the
 * whole launch waits for them.  With every_n_steps > 0 the library keeps, per env, the first state of its NEXT episode
 * in shadow buffers: a reset then copies that row (one memory round trip), and one extra launch every `every_n_steps`
 * batched steps (and after every ngw_reset) re-prepares the rows consumed since.  Results are bit-identical with the
 * feature on or off (the shadow row is the output of the same per-(env, episode) Philox stream).  0 = off.
 * ngw_set_autoreset(h, 1, horizon) switches it ON by itself (a refill every 3/4 horizon, 32 to 128 steps, 32 without a horizon; not for horizons under 64 steps, where
 * rows would go stale before they are needed) unless the caller has chosen a cadence with this call, before or after.
 * Fused rollouts run as launches of at most every_n_steps steps with the refills between them. */
int ngw_set_reset_prefetch(ngw_handle* h, int32_t every_n_steps);
/* The refill cadence in effect (0 = prepared episodes off): what ngw_set_reset_prefetch set, or the default ngw_set_autoreset chose. */
int ngw_get_reset_prefetch(ngw_handle* h, int32_t* every_n_steps);
/* How many episodes ahead are prepared per env: depth 1, 2, 4 or 8 (a power of two; shadow memory grows with it), 0 = automatic
 * (the default: 1, growing to 2 and 4 by itself when envs end episodes faster than a refill comes round - FireWall kills within
 * a few steps - so that a reset still finds a prepared row; see adapt_cadence in ngw_abi.cpp).  Results do not depend on it. */
int ngw_set_reset_prefetch_depth(ngw_handle* h, int32_t depth);
int ngw_get_reset_prefetch_depth(ngw_handle* h, int32_t* depth);
int ngw_set_stream(ngw_handle* h, void* hip_stream);
/* Order the handle's stream and another hipStream_t of the same device behind each other WITHOUT a host wait (one event
 * record + one stream wait): handle_waits != 0 - work submitted to the handle after this call runs after everything
 * `other_stream` holds now; handle_waits == 0 - the other way round.  What dist.py uses around the one collective (pack /
 * unpack launches on the handle's stream, RCCL on torch's current stream); NULL = the default stream. */
int ngw_stream_order(ngw_handle* h, void* other_stream, int handle_waits);

/* reset(): pogostick_v1_env.py:86-157 (+ AddItem.reset).  mask = NULL resets all envs, else mask[i] != 0. */
int ngw_reset(ngw_handle* h, const uint8_t* mask_host);
/* ngw_reset + the state it produced, in ONE call (any output may be NULL): what reset() of the host API returns.  For handles of at
 * most one wavefront (the single-env gym.Env adapter) it waits for the reset kernel alone - not for the refill launch that
 * re-prepares the consumed episode behind it - by polling a word the kernel writes when its stores (the state in HBM and its
 * host mirror, see ngw_obs_device_ptrs) are out. */
int ngw_reset_host(ngw_handle* h, const uint8_t* mask_host, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv, uint8_t* selected,
                   int32_t* step_count, uint32_t* error_flags);
/* step(action_id): pogostick_v1_env.py:230-367 / AxeMedium.step.  Host actions are validated first
 * (NGW_E_INVALID_ACTION, nothing stepped - the reference raises before touching state). */
int ngw_step(ngw_handle* h, const int32_t* actions_host);
/* Same with actions already in HBM; an out-of-range id sets NGW_F_INVALID_ACTION and leaves that env untouched. */
int ngw_step_device(ngw_handle* h, const int32_t* actions_dev);
/* n_steps consecutive ngw_step_device launches from ONE call: step i reads actions_dev + i * step_stride (int32 elements).  For
 * short open-loop stretches where a host loop's per-call overhead (an interpreter, a binding) would outweigh the 4-5 us a step
 * takes on the device; semantically identical to n_steps calls of ngw_step_device. */
int ngw_step_device_many(ngw_handle* h, const int32_t* actions_dev, int64_t step_stride, int32_t n_steps);
/* ngw_step + ngw_get_obs + ngw_get_step_out as ONE call with one stream synchronisation: what a host-driven loop pays per
 * step() is launch and PCIe latency, so the three round trips of the separate calls matter at small batch sizes.  Any
 * output pointer may be NULL; batches whose outputs fit in 1 MiB travel through host memory the GPU addresses directly
 * (no copy calls at all); page-locked buffers (ngw_host_alloc) make the copies truly asynchronous. */
int ngw_step_host(ngw_handle* h, const int32_t* actions_host, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv,
                  int32_t* reward, uint8_t* done, uint8_t* result, uint8_t* cost_code, uint16_t* msg_code, uint16_t* msg_arg,
                  uint32_t* error_flags /* the sticky NGW_F_* word, as ngw_error_flags */,
                  uint8_t* selected /* item id, 0 = '' */, int32_t* step_count);
/* Big batches: when the output arrays handed to ngw_step_host are the sections of ONE page-locked block laid out as this call
 * says (offsets11[0..9] = byte offsets of map | agent_location | agent_facing_id | inventory | reward | done | info words
 * (internal) | error flags | selected | step_count from the block's start, each section padded to 256 bytes - in memory the
 * sections lie in the order map, inventory, selected, then the rest; offsets11[10] = block size; allocate it with
 * ngw_host_alloc) AND the caller hands in the same block call after call (it is his mirror of the observation), a step moves
 * only what changed: the pose / reward / done / info / step_count sections with ONE copy (26 B per env), and of the map,
 * inventory and selected rows only the 16-byte pieces that differ from what the block already holds - the device keeps a
 * shadow of the block's content and writes the differing pieces straight into it across PCIe.  The first call on a block, and
 * the first one after anything else touched the state (ngw_reset, ngw_set_state, device steps, rollouts, graph replays) or
 * after ngw_host_mirror_invalidate, copies the whole observation; NGW_HOST_DELTA=0 in the environment makes every call do so.
 * The block must not be written by the caller between calls. */
int ngw_host_step_layout(ngw_handle* h, uint64_t* offsets11);
/* The next ngw_step_host copies the whole observation again (e.g. after the caller wrote into his block). */
int ngw_host_mirror_invalidate(ngw_handle* h);
/* Fused bench mode: T steps in one launch with on-device uniform actions
 * a(t, env) = (word (t & 3) of philox(action_seed; t >> 2, env) * A) >> 32; state stays in LDS/registers between
 * steps and every step's changes are written through to the observation buffers. */
int ngw_rollout(ngw_handle* h, int32_t n_steps, uint64_t action_seed, int64_t t0);
/* The same fused launch driven by the CALLER's actions (open-loop sequences, action repeat / frame skip, evaluating plans):
 * batched step t takes int32 actions_dev[t * step_stride + e] (device memory, step_stride >= n_envs).  An action outside
 * [0, n_actions) leaves that env untouched for that step and raises NGW_F_INVALID_ACTION, as in ngw_step_device. */
int ngw_rollout_actions(ngw_handle* h, const int32_t* actions_dev, int64_t step_stride, int32_t n_steps);

/* Per-step outputs of the fused rollouts, for whoever consumes them (the reference's loop sees (obs, reward, done, info) after
 * every step: tests/random_action.py:51-64, tests/train.py:122-135).  With rows set, step t of a ngw_rollout /
 * ngw_rollout_actions call also stores reward_rows_dev[t * row_stride + e] (int32) and done_rows_dev[t * row_stride + e]
 * (uint8: 1 where the step ended an episode - done, or the horizon under autoreset); either pointer may be NULL, both NULL
 * switches the rows off.  accumulate = 1 additionally keeps four per-env int32 counters across rollout calls: return and
 * length of the running episode, sum of returns and number of the episodes finished so far (ngw_episode_stats reads them
 * into host arrays, any of which may be NULL; clear = 1 zeroes them afterwards).  Device memory, row_stride >= n_envs. */
int ngw_rollout_outputs(ngw_handle* h, int32_t* reward_rows_dev, uint8_t* done_rows_dev, int64_t row_stride, int accumulate);
int ngw_episode_stats(ngw_handle* h, int32_t* run_return, int32_t* run_length, int32_t* sum_return, int32_t* n_episodes, int clear);

/* get_observation(): pogostick_v1_env.py:214-228, batched: map i8 [N,S,S], agent_location i32 [N,2] (r,c),
 * agent_facing_id i32 [N], inventory_items_quantity i32 [N,K] in items_id order.  Any pointer may be NULL. */
int ngw_get_obs(ngw_handle* h, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv);
/* (reward, done, info) of the last step: pogostick_v1_env.py:354-367. */
int ngw_get_step_out(ngw_handle* h, int32_t* reward, uint8_t* done, uint8_t* result, uint8_t* cost_code,
                     uint16_t* msg_code, uint16_t* msg_arg);
/* Full state of envs [first, first+count): the observation arrays plus selected item id (0 = ''),
 * step_count and episode counter.  Checkpoint/restore and oracle-state injection
 * (reference: direct attribute mutation, tests/keyboard_interface.py:93-100). */
int ngw_get_state(ngw_handle* h, int64_t first, int64_t count, int8_t* map, int32_t* loc, int32_t* facing,
                  int32_t* inv, int32_t* selected, int32_t* step_count, uint32_t* episode);
int ngw_set_state(ngw_handle* h, int64_t first, int64_t count, const int8_t* map, const int32_t* loc,
                  const int32_t* facing, const int32_t* inv, const int32_t* selected,
                  const int32_t* step_count, const uint32_t* episode);

/* Page-locked host memory for the arrays handed to ngw_step / ngw_get_obs / ngw_get_step_out: copies to and from
 * pinned buffers run at full PCIe rate without a staging pass (API mode).  Plain malloc'ed arrays work too. */
void* ngw_host_alloc(uint64_t bytes);
int ngw_host_free(void* p);

/* Device pointers of the observation / output buffers (fixed for the handle's lifetime; contents are the state
 * after the last enqueued step and are updated in place by the next one): HBM for every handle.  Handles of at most one
 * wavefront (<= 64 envs: the single-env gym.Env adapter) additionally keep a MIRROR of these rows in page-locked host memory
 * the GPU addresses directly: a step issued by ngw_step_host (an explicit reset by ngw_reset_host) ends by copying the wave's
 * rows there and writing a sequence word the host polls, so such a call is one launch with no copy call and no stream
 * synchronisation. */
int ngw_obs_device_ptrs(ngw_handle* h, void** map, void** loc, void** facing, void** inv);
int ngw_out_device_ptrs(ngw_handle* h, void** reward, void** done, void** info);
int ngw_sync(ngw_handle* h);
/* Reads and clears the sticky device error flags (NGW_F_*). */
int ngw_error_flags(ngw_handle* h, uint32_t* flags);
/* Device time of everything enqueued on the handle's stream between the two calls, measured with a HIP event pair
 * recorded on that stream (bench.py roofline leg: elapsed / launches = average launch duration incl. gaps). */
int ngw_timing_begin(ngw_handle* h);
int ngw_timing_end(ngw_handle* h, double* elapsed_ms);
/* Records the closing event now, without waiting: a later ngw_timing_end only waits for it and reads the pair (a caller that
 * synchronises anyway - a benchmark's closing fence - keeps the event wait out of its wall-clock region). */
int ngw_timing_mark(ngw_handle* h);

/* hipGraph stepping for launch-bound loops: captures n_steps consecutive ngw_step_device launches whose
 * actions are read from actions_dev + i * step_stride (int32 elements) into one executable graph, then replays it.
 * Semantically identical to calling ngw_step_device n_steps * reps times with those action rows.  Prepared next episodes: a graph
 * that is at least half the refill cadence long carries its refills (every replay ends with one, so that it leaves the cadence where
 * it found it); a shorter one is captured without any and ngw_graph_launch issues them between the replays, whenever the next replay
 * would overrun the cadence. */
int ngw_graph_build(ngw_handle* h, const int32_t* actions_dev, int64_t step_stride, int32_t n_steps);
int ngw_graph_launch(ngw_handle* h, int32_t reps);

/* Multi-GPU observation stack (the only collective of the path, outside step()).  One process per GPU,
 * each with its own handle; per step (or whenever the host wants the whole batch) every rank packs its observation and
 * step outputs into ONE contiguous device payload, the ranks gather the payloads on a root (RCCL gather over xGMI through
 * torch.distributed) and the root scatters them into global arrays.
 *   ngw_pack_layout  byte offsets of the seven sections (map i8 [n][S*S] | agent_location i32 [n][2] | agent_facing_id i32 [n] |
 *                    inventory i32 [n][K] | reward i32 [n] | done u8 [n] | info u32 [n]; each padded to 16 bytes) and, in
 *                    offsets8[7], the payload size.  Equal for every rank that holds the same number of envs.
 *   ngw_pack_obs     one kernel launch on the handle's stream: the seven SoA arrays -> payload_dev (16-byte aligned).
 *   ngw_unpack_obs   root side: `world` payloads back to back at payloads_dev -> global arrays (rank r's envs at [r*n, (r+1)*n));
 *                    any destination may be NULL.  One launch on the handle's stream. */
int ngw_pack_layout(ngw_handle* h, uint64_t* offsets8);
int ngw_pack_obs(ngw_handle* h, void* payload_dev);
int ngw_unpack_obs(ngw_handle* h, const void* payloads_dev, int32_t world, int8_t* map, int32_t* loc, int32_t* facing,
                   int32_t* inv, int32_t* reward, uint8_t* done, uint32_t* info);

/* The host step in its NARROW WIRE FORMAT (big batches; what VecNovelGridworld.step() uses from a few thousand envs on).  One
 * page-locked block holds everything a step returns; ngw_host_step_layout_packed gives its section offsets (index: 0 map int8
 * [n][S*S], 1 inventory int32 [n][K], 2 pose uint32 [n] = r | c << 8 | facing << 16 | selected << 24, 3 reward int32 [n] (ABI 3: the
 * type ngw_step_host returns whatever the batch size; int16 before), 4 done uint8 [n], 5 info uint32 [n] (NGW_INFO_*), 6 error flags
 * uint32 followed by one uint32 the library uses itself (the sequence number of the last finished step: the call polls it instead of
 * synchronising the stream); sections padded to 256 bytes, offsets8[7] = the block's size).
 * ngw_step_host_packed(h, actions, block, with_map): int32 actions from host memory are validated and narrowed to bytes on the way
 * into a buffer the step kernel reads in place (no copy call); map and inventory are refreshed by deltas as in ngw_step_host (the block
 * is a mirror the caller hands in call after call; with_map = 0 skips the map's delta for this call); the dense sections 2-6 - 13 B per
 * env instead of the 26 B of the int32 arrays - are stored straight into the block by the device once the block is a mirror (mapped
 * into the GPU's address space; the first call on a block, and a block that cannot be mapped, bring them across with one copy).
 * In that steady state the call is ONE launch: the step kernel itself stores what the step changes into the block (system-scope
 * stores) - cells, inventory slots, the rows of envs that start an episode, the dense sections, and the fused lidar rows when
 * ngw_lidar_host_rows registered a buffer and n_envs is a multiple of 64 - and the call returns when the kernel's last block has
 * published its sequence number; a refill launch behind the step may still be running then (it touches nothing the caller sees).
 * Widening (pose bytes -> int32 arrays) is the caller's, when he needs it. */
int ngw_host_step_layout_packed(ngw_handle* h, uint64_t* offsets8);
int ngw_step_host_packed(ngw_handle* h, const int32_t* actions_host, void* block, int with_map);

/* Terminal observations under same-step autoreset.  A step that ends an env's episode (done, or the horizon) returns the NEXT
 * episode's first observation (ngw_set_autoreset); the reference's loops look at the last observation of the old one before they
 * reset (tests/test.py:30-41, enjoy.py:107-116), and a learner bootstraps from it at a horizon cut.  enable = 1: the envs that reset in
 * a step launch first copy the state their episode ended in - map row, pose, inventory row - into a side set, which
 * ngw_get_terminal_obs copies out whole ([n_envs] rows; row e is meaningful for the envs whose `done` the last step set, and keeps its
 * value until env e ends an episode again) and ngw_terminal_device_ptrs exposes in place.  Off by default; when off the step kernels'
 * hot path is untouched (the copy sits behind the "some lane resets" branch).  Fused rollouts capture too: the lane that resets inside
 * the T-loop stores its map and inventory row from LDS into the side set first; after a rollout row e holds the state env e's LAST
 * finished episode ended in (the per-step done rows of ngw_rollout_outputs say which steps ended one). */
int ngw_set_terminal_capture(ngw_handle* h, int enable);
int ngw_get_terminal_obs(ngw_handle* h, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv);
int ngw_terminal_device_ptrs(ngw_handle* h, void** map, void** loc, void** facing, void** inv);

/* Which per-launch step kernel this handle's ngw_step* calls run right now: *map_in_place = 1 - the one that reads the <= 14 map
 * cells a step needs straight from HBM (the default at every map size), 0 - the one that stages the wave's 64 maps through
 * LDS (any size while the fused lidar epilogue is on, or where NGW_NOSTAGE says so).  What bench.py names its kernel from. */
int ngw_step_kernel_info(ngw_handle* h, int32_t* map_in_place);

/* LidarInFront: configure once, then ngw_lidar() computes the observation of the CURRENT state of every env into a device
 * buffer of [N] rows of num_beams * n_chan beam entries + n_inv inventory entries (enqueued on the handle's stream, after the
 * steps before it). */
int ngw_lidar_configure(ngw_handle* h, const ngw_lidar_cfg* cfg);
int ngw_lidar(ngw_handle* h);
/* enable = 1: every following reset / step / rollout launch also refreshes the lidar observation in its epilogue (the maps are
 * already in LDS there), so ngw_lidar() is not needed; 0 restores the plain kernels.  A fused rollout refreshes it ONCE, for
 * the state the launch ends in (the buffer holds one row per env). */
int ngw_lidar_fuse(ngw_handle* h, int enable);
/* Row format of the observation in the device buffer (and of what ngw_get_lidar copies out):
 *   32 (default)  int32 [len]: what a caller that never calls this function reads (ngw_get_lidar / ngw_lidar_device_ptr);
 *   16            int16 [len]: half the bytes; values saturate at 32767 - a beam entry is a range <= 64, the inventory tail is the
 *                 only part that could ever exceed it;
 *    8            packed: uint8 [num_beams * n_chan] beam entries, padded to an even count, then int16 [n_inv] inventory
 *                 (saturating) - 70 B per env for the reference's 8 beams on Pogostick-v1 against 252 B as int32.
 * ngw_lidar_row_layout reports bytes per row, bytes per beam entry, the byte offset of the inventory tail and bytes per
 * inventory entry of the current format (any pointer may be NULL). */
int ngw_lidar_set_output(ngw_handle* h, int bits);
int ngw_lidar_row_layout(ngw_handle* h, int32_t* row_bytes, int32_t* beam_bytes, int32_t* inv_offset, int32_t* inv_bytes);
int ngw_get_lidar(ngw_handle* h, void* out_host /* [n_envs] rows of the current format */);
/* With the observation fused (ngw_lidar_fuse): rows_host != NULL - a page-locked buffer (ngw_host_alloc) of [n_envs] rows of the current
 * format - makes every following ngw_step_host_packed deliver the rows of the state it produced into that buffer as part of the call
 * (pipelined with the step's slices on big batches: what LidarInFront(VecNovelGridworld).step() returns needs no second call and no
 * second synchronisation); NULL switches it off.  ngw_lidar_set_output / ngw_lidar_configure switch it off too (the row size changed). */
int ngw_lidar_host_rows(ngw_handle* h, void* rows_host);
int ngw_lidar_device_ptr(ngw_handle* h, void** out);

/* AgentMap (observation_wrappers.py:83-129): ngw_agent_view() gathers, for every env, the (2*view_size+1)^2 window of the
 * CURRENT map centred on the agent (0 outside the map; get_agentView :104-121) into an internal int8
 * [n_envs][2*view_size+1][2*view_size+1] buffer.  The reference fixes view_size = 5 (:96).  The other two entries of that
 * wrapper's observation (agent_facing_id, inventory_items_quantity) are the ngw_get_obs buffers. */
int ngw_agent_view(ngw_handle* h, int view_size);
int ngw_get_agent_view(ngw_handle* h, int8_t* out_host);
int ngw_agent_view_device_ptr(ngw_handle* h, void** out);

#ifdef __cplusplus
}
#endif
#endif /* NGW_H */
