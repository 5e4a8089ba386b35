// ngw_abi.cpp — host side of the C-ABI declared in include/ngw.h (HIP runtime only; no torch types).
//
// Owns the device buffers of a handle, validates arguments, launches ngw_kernel (ngw_kernels.hip) and moves
// observations / outputs / state between HBM and caller-provided host arrays.  There is NO CPU execution path:
// without a GPU every entry point that computes returns NGW_E_NO_DEVICE / NGW_E_HIP.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ngw.h"
#include "ngw_device.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(NGW_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));     \
    } while (0)

}  // namespace

struct ngw_handle {
    ngw_spec spec;
    int64_t n = 0, n_pad = 0, env_base = 0;
    int device = 0;
    uint64_t seed = 0;
    int autoreset = 0, horizon = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    NgwBufs b{};
    NgwLaunch proto{};           // layout fields filled once
    size_t lds_bytes = 0;
    int map_mode = 0;
    NgwDevSpec* dspec = nullptr;      // LUT blob in HBM
    int32_t* actions_dev = nullptr;   // staging for host actions
    uint8_t* mask_dev = nullptr;
    uint32_t* info_host = nullptr;    // pinned staging of the packed info words (ngw_step_host)
    uint8_t* zc_host = nullptr;       // small batches: actions + packed outputs in host memory the GPU addresses directly
    uint8_t* zc_dev = nullptr;
    uint8_t* step_stage = nullptr;        // ngw_step_host, one-block layout: every output packed on the device, ONE copy out
    // ngw_step_host, delta refresh: device-side shadows of the map / inventory / selected rows the caller's block holds, the
    // block they describe (host pointer + its mapped device address), and whether it still mirrors the device state
    uint8_t* shadow[3] = {nullptr, nullptr, nullptr};
    const void* mirror_block = nullptr;
    uint8_t* mirror_dev = nullptr;
    bool mirror_valid = false;
    int host_delta = 1;                   // NGW_HOST_DELTA=0: every ngw_step_host copies the whole observation (A/B)
    int wire_merge = 1;                   // NGW_WIRE_MERGE=0: delta refresh and narrowing as two launches (A/B)
    int wire_direct = 1;                  // NGW_WIRE_DIRECT=0: ngw_step_host_packed narrows into device staging and copies it across (A/B)
    size_t zc_bytes = (size_t)256 << 10;  // NGW_ZC_BYTES: largest ngw_step_host result written straight into mapped host memory (read at ngw_create)
    std::vector<void*> allocs;
    std::vector<void*> host_allocs;       // single-wavefront handles: the host mirror of the state (GPU-addressable page-locked memory)
    NgwMirror mir = {};                   // ... its arrays (host addresses = device addresses under unified addressing)
    uint8_t* mask_pin = nullptr; uint8_t* mask_pin_dev = nullptr;   // ngw_reset's mask: two page-locked halves the kernel reads in place
    hipEvent_t mask_ev[2] = {nullptr, nullptr};
    int mask_next = 0;
    uint8_t* act_pin_dev = nullptr;            // ... the same buffer as the GPU addresses it (ngw_step_host_packed: the kernel reads the actions in place)
    bool launch_act_u8 = false;                // the launch being issued reads one byte per env from `actions`
    uint8_t* wire_stage = nullptr;             // ngw_step_host_packed: device staging of the dense sections
    uint8_t* act_pin = nullptr;                // ngw_step's actions: two page-locked halves feeding the asynchronous copy
    hipEvent_t act_ev[2] = {nullptr, nullptr};
    int act_next = 0;
    int hostres = 0;                         // single-wavefront handle with a host mirror (NgwMirror; NGW_HOST_STATE=0 switches it off)
    uint32_t step_seq = 0, launch_seq = 0;   // hostres: sequence number the next step launch reports (launch_seq: only while ngw_step_host issues it)
    int32_t launch_action0 = 0; bool launch_use_action0 = false;   // one-env handles: the action of the launch ngw_step_host is issuing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // timing pair
    hipEvent_t order_ev = nullptr;             // ngw_stream_order
    bool ev_marked = false;                    // ngw_timing_mark recorded the closing event already
    // LidarInFront observation
    NgwLidarDev* lidar_cfg = nullptr;     // device tables
    int32_t* lidar_out = nullptr;
    int lidar_len = 0, lidar_cap = 0;         // lidar_cap: row length lidar_out was allocated for
    int lidar_bits = 32;                  // row format of the lidar observation (ngw_lidar_set_output): 32 (the default: the reference's integers), 16 or 8 = packed
    int lidar_world = 0;                  // the ray table is one world-frame table rotated by the facing (NgwLidarDev::woff)
    NgwLaunch lidar_proto{};              // the stand-alone lidar launch: its own LDS layout
    int lidar_fused = 0, lidar_range = 0, lidar_beams = 0, lidar_chan = 0, lidar_ninv = 0;
    size_t lidar_lds = 0;
    NgwNx nx = {};                        // prepared next episodes (ngw_set_reset_prefetch); all null = off
    int prefetch_every = 0, since_refill = 0;
    // The cadence adapts under the DEFAULT setting: resets that find their prepared row stale (an env that ends two episodes
    // between refills - FireWall kills within a few steps) are counted on the device; the refill launch copies the count to a
    // host word and the host halves the cadence while it keeps growing, and doubles it back after four quiet refills.
    int cadence = 0, quiet = 0, noisy = 0, adapt = 1;
    int quiet_need = 4;                   // quiet refills before the cadence is doubled back (grows when a doubling had to be undone)
    bool probing = false;                 // the last change was a doubling
    uint32_t slow_seen = 0, refill_seen = 0, refill_count = 0;   // reports read / refill launches issued
    bool capturing = false;
    bool adapted = false;                 // adapt_cadence changed depth or cadence: a captured graph is stale (ngw_graph_launch re-captures it)
    bool adapt_error = false;             // growing the prepared-episode depth failed (out of memory): the depth stays, the next refill notes it once
    int prefetch_user = 0;                // the caller chose the cadence (ngw_set_reset_prefetch): ngw_set_autoreset leaves it alone
    int depth = 1, depth_user = 0;        // prepared episodes per env (power of two); depth_user: chosen through ngw_set_reset_prefetch_depth
    NgwTerm term = {};                    // terminal observations (ngw_set_terminal_capture); all null = off
    bool term_on = false;
    int32_t* row_reward = nullptr;        // fused rollouts: the caller's output rows (ngw_rollout_outputs)
    uint8_t* row_done = nullptr;
    int64_t row_stride = 0;
    int32_t* acc = nullptr;               // [4][n_pad] episode accumulators of the fused rollouts
    uint32_t off_rng = 0;                 // LDS dword offset of the reset path's Philox ring
    int fast_reset = 1;                   // dedicated new-episode kernel where it applies (NGW_FAST_RESET=0: general kernel, A/B)
    NgwResetFast rf{};                    // its arguments, laid out once (layout_reset_fast)
    int rf_nw = -1, rf_additem = 0;       // rf_nw < 0: not applicable to this spec / layout
    size_t rf_lds = 0;
    int nostage = 0;                      // per-launch steps through the lean kernel without map staging (every size but 10 x 10 / 6 x 6; NGW_NOSTAGE=<min S*S>: A/B)
    NgwLaunch ns_proto{};                 // its launch prototype (small LDS layout)
    size_t ns_lds = 0;
    bool general_ok = true;               // false: the map is too big for the kernels that keep a wave's 64 maps in LDS (general kernel, fused rollouts, fused lidar)
    int ext = 0;                          // spec uses FireWall / FenceRestriction / Crate step predicates -> EXT kernels
    int8_t* view_out = nullptr;           // AgentMap windows
    int view_size = 0;
    size_t view_cap = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_steps = 0;
    const int32_t* graph_actions = nullptr;   // what ngw_graph_build captured: an adaptation re-captures it
    int64_t graph_stride = 0;
};

namespace {

int check_spec(const ngw_spec* s) {
    if (!s) return fail(NGW_E_INVALID_ARG, "spec is NULL");
    if (s->abi_version != NGW_ABI_VERSION) return fail(NGW_E_INVALID_ARG, "spec abi_version %d != %d", s->abi_version, NGW_ABI_VERSION);
    if (s->map_size < 5 || s->map_size > NGW_MAX_MAP_SIZE) return fail(NGW_E_INVALID_ARG, "map_size %d outside [5, %d]", s->map_size, NGW_MAX_MAP_SIZE);
    if (s->n_items < 4 || s->n_items > NGW_MAX_ITEMS) return fail(NGW_E_INVALID_ARG, "n_items %d outside [4, %d] (air, wall, a crafting table and a goal item at least)", s->n_items, NGW_MAX_ITEMS);
    if (s->n_actions < 1 || s->n_actions > NGW_MAX_ACTIONS) return fail(NGW_E_INVALID_ARG, "n_actions %d out of range", s->n_actions);
    if (s->n_recipes < 0 || s->n_recipes > NGW_MAX_RECIPES) return fail(NGW_E_INVALID_ARG, "n_recipes %d out of range", s->n_recipes);
    if (s->n_start > NGW_MAX_START_ITEMS) return fail(NGW_E_INVALID_ARG, "n_start %d out of range", s->n_start);
    const int K = s->n_items;
    auto item_ok = [&](int i) { return i >= 0 && i < K; };
    if (!item_ok(s->wall_item) || !item_ok(s->table_item) || !item_ok(s->goal_item) || !item_ok(s->place_item) ||
        !item_ok(s->place_near) || !item_ok(s->ext_src) || !item_ok(s->ext_near) || !item_ok(s->ext_out) ||
        !item_ok(s->axe_item) || !item_ok(s->tap_item) ||
        !item_ok(s->tap_near))
        return fail(NGW_E_INVALID_ARG, "spec item id out of range");
    if (s->n_inv_start > NGW_MAX_INV_START) return fail(NGW_E_INVALID_ARG, "n_inv_start %d out of range", s->n_inv_start);
    for (int j = 0; j < s->n_inv_start; j++)
        if (!item_ok(s->inv_start_item[j]) || !s->inv_start_item[j]) return fail(NGW_E_INVALID_ARG, "inv_start_item[%d] out of range", j);
    for (int a = 0; a < s->n_actions; a++) {
        const int kind = s->act_kind[a], arg = s->act_arg[a];
        if (kind > NGW_ACT_JUMP) return fail(NGW_E_INVALID_ARG, "action %d has unknown kind %d", a, kind);
        if (kind == NGW_ACT_CRAFT && arg >= s->n_recipes) return fail(NGW_E_INVALID_ARG, "action %d: recipe %d out of range", a, arg);
        if (kind == NGW_ACT_SELECT && !item_ok(arg)) return fail(NGW_E_INVALID_ARG, "action %d: item %d out of range", a, arg);
    }
    for (int r = 0; r < s->n_recipes; r++) {
        if (s->recipe_n_in[r] > NGW_MAX_RECIPE_INPUTS || !item_ok(s->recipe_out_item[r]))
            return fail(NGW_E_INVALID_ARG, "recipe %d malformed", r);
        for (int j = 0; j < s->recipe_n_in[r]; j++)
            if (!item_ok(s->recipe_in_item[r][j])) return fail(NGW_E_INVALID_ARG, "recipe %d input out of range", r);
    }
    for (int j = 0; j < s->n_start; j++)
        if (!item_ok(s->start_item[j])) return fail(NGW_E_INVALID_ARG, "start item out of range");
    int total_place = 0;
    for (int j = 0; j < s->n_start; j++) total_place += s->start_qty[j];
    if (total_place > NGW_MAX_PLACE) return fail(NGW_E_INVALID_ARG, "items_quantity places %d items per reset (max %d)", total_place, NGW_MAX_PLACE);
    for (int i = 0; i < K; i++)
        if (s->breakable[i] && s->break_qty[i] != 1 && s->break_qty[i] != 2) return fail(NGW_E_INVALID_ARG, "break_qty must be 1 or 2");
    int brw = s->reward_step;
    for (int i = 0; i < K; i++)
        if (s->break_reward[i] != s->reward_step) {
            if (brw != s->reward_step && brw != s->break_reward[i]) return fail(NGW_E_INVALID_ARG, "break_reward must take one value besides reward_step");
            brw = s->break_reward[i];
        }
    for (int r = 0; r < s->n_recipes; r++)              /* the kernel applies a recipe from prefetched counts: ids must differ */
        for (int j = 0; j < s->recipe_n_in[r]; j++) {
            if (s->recipe_in_item[r][j] == s->recipe_out_item[r]) return fail(NGW_E_INVALID_ARG, "recipe %d consumes its own output", r);
            for (int k = 0; k < j; k++)
                if (s->recipe_in_item[r][j] == s->recipe_in_item[r][k]) return fail(NGW_E_INVALID_ARG, "recipe %d lists an input twice", r);
        }
    auto pct_ok = [](int lo, int hi) { return lo < hi && hi - lo <= 64 && hi <= 100; };
    if (s->n_passes > NGW_MAX_PASSES) return fail(NGW_E_INVALID_ARG, "n_passes %d out of range", s->n_passes);
    for (int j = 0; j < s->n_passes; j++) {
        const int kind = s->pass_kind[j];
        if (kind < NGW_PASS_ADDITEM || kind > NGW_PASS_FENCE) return fail(NGW_E_INVALID_ARG, "reset pass %d has unknown kind %d", j, kind);
        if (!item_ok(s->pass_item[j]) || !s->pass_item[j] || !item_ok(s->pass_from[j])) return fail(NGW_E_INVALID_ARG, "reset pass %d: item id out of range", j);
        if (kind == NGW_PASS_REPLACE && s->pass_item[j] == s->pass_from[j]) return fail(NGW_E_INVALID_ARG, "reset pass %d replaces an item with itself (the reference asserts a NEW item, novelty_wrappers.py:1108)", j);
        if (!pct_ok(s->pass_pct_lo[j], s->pass_pct_hi[j]))
            return fail(NGW_E_INVALID_ARG, "%s percent range invalid", kind == NGW_PASS_ADDITEM ? "additem" : kind == NGW_PASS_REPLACE ? "replace" : "fence");
        if (kind == NGW_PASS_FENCE)     /* a fence pass after a wall-replacing pass would fence border cells: add_fence_around leaves the map (reference: IndexError) */
            for (int i = 0; i < j; i++)
                if (s->pass_kind[i] == NGW_PASS_REPLACE && s->pass_from[i] == s->wall_item)
                    return fail(NGW_E_INVALID_ARG, "fence pass after a wall-replacing pass edits cells outside the map");
    }
    if (!item_ok(s->fence_item) || !item_ok(s->fire_item) || !item_ok(s->crate_item) || s->fence_mode > 2)
        return fail(NGW_E_INVALID_ARG, "novelty item id / fence_mode out of range");
    for (int i = 0; i < NGW_MAX_ITEMS; i++)
        if (s->crate_add[i] && (i >= K || !s->crate_item || s->crate_add[i] > 15)) return fail(NGW_E_INVALID_ARG, "crate_add[%d] invalid", i);
    if (s->fence_mode && !s->fence_item) return fail(NGW_E_INVALID_ARG, "fence_mode without fence_item");
    if (s->ext_flags > 3 || s->fire_skip_recipe > s->n_recipes) return fail(NGW_E_INVALID_ARG, "ext_flags / fire_skip_recipe out of range");
    return NGW_OK;
}

template <typename T>
int dev_alloc(ngw_handle* h, T** p, size_t count) {
    void* q = nullptr;
    size_t bytes = count * sizeof(T);
    HIP_TRY(hipMalloc(&q, bytes));
    HIP_TRY(hipMemsetAsync(q, 0, bytes, h->stream));
    h->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return NGW_OK;
}

// One array of the host mirror of a single-wavefront handle (the gym.Env adapter: n = 1): page-locked host memory the GPU
// addresses directly.  The state itself lives in HBM; a step issued by ngw_step_host copies the wave's rows here before it
// signals completion, so such a handle steps with NO copy call, no pack launch and no stream synchronisation - and the
// kernel writes across PCIe but never reads (round 3 kept the state itself in host memory: 2.9 us of PCIe reads per step).
template <typename T>
int mirror_alloc(ngw_handle* h, T** p, size_t count) {
    void* q = nullptr;
    const size_t bytes = count * sizeof(T);
    HIP_TRY(hipHostMalloc(&q, bytes ? bytes : 1, hipHostMallocMapped));
    memset(q, 0, bytes);
    h->host_allocs.push_back(q);
    void* d = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&d, q, 0));
    if (d != q) return fail(NGW_E_HIP, "the host mirror needs unified addressing");
    *p = static_cast<T*>(q);
    return NGW_OK;
}

void dev_free(ngw_handle* h, void* p) {
    for (size_t i = 0; i < h->allocs.size(); i++)
        if (h->allocs[i] == p) { h->allocs.erase(h->allocs.begin() + (long)i); break; }
    (void)hipFree(p);
}

// Row format of the LidarInFront observation (NGW_LFMT_*): bytes per row and where its inventory tail starts.
void lidar_format(const ngw_handle* h, NgwLaunch& p) {
    const int nb = h->lidar_beams * h->lidar_chan, ni = h->lidar_ninv;
    p.l_fmt = h->lidar_bits == 32 ? NGW_LFMT_I32 : (h->lidar_bits == 16 ? NGW_LFMT_I16 : NGW_LFMT_PACKED);
    if (p.l_fmt == NGW_LFMT_I32) { p.l_invoff = 4 * nb; p.l_rb = 4 * (nb + ni); }
    else if (p.l_fmt == NGW_LFMT_I16) { p.l_invoff = 2 * nb; p.l_rb = 2 * (nb + ni); }
    else { p.l_invoff = (nb + 1) & ~1; p.l_rb = p.l_invoff + 2 * ni; }
    p.l_world = h->lidar_world;
    p.lcfg = h->lidar_cfg; p.lout = h->lidar_out; p.lidar_len = h->lidar_len;
    p.l_beams = h->lidar_beams; p.l_range = h->lidar_range; p.l_chan = h->lidar_chan; p.l_inv = h->lidar_ninv;
}

// LDS carve-up of the kernels that keep a wave's 64 maps in LDS (dword offsets).  With the lidar epilogue fused the maps sit
// behind a guard (ray offsets read beyond a hit may leave the lane's own map) and the item tables, the per-lane ray table (only
// when the rays are not world-frame) and the observation tile follow the rest.
int layout_lds(ngw_handle* h) {
    NgwLaunch& p = h->proto;
    const int S = p.S, S2 = p.S2;
    const uint32_t guard = h->lidar_fused ? (uint32_t)((h->lidar_range * (S + 1) + 15) / 16 * 4) : 0u;
    auto waves_per_cu = [](uint32_t dwords) { return (160u * 1024u) / (((dwords * 4u + 511u) / 512u) * 512u); };
    // One pass over the regions.  `alias`: the lidar observation tile shares the candidate masks' region.  The masks are live only
    // inside a new-episode path, and every epilogue that follows one zeroes the tile again (lidar_epilogue, zeroed = false), so the
    // two never hold data at once; the Philox ring, which is live together with the masks, then sits behind them.
    auto pass = [&](bool alias) -> uint32_t {
        uint32_t off = guard;
        p.off_map = off; off += (uint32_t)(NGW_EPB * p.MS / 4) + guard;
        off = (off + 3u) & ~3u;
        p.off_inv = off; off += (uint32_t)(p.KP * NGW_EPB);
        const uint32_t cand_dw = (uint32_t)(p.CW * NGW_EPB);
        uint32_t tile_dw = 0, tile_all = 0;
        p.lcfg = nullptr; p.lout = nullptr; p.lidar_len = 0; p.off_litem = p.off_ltab = p.off_ltile = 0;
        if (h->lidar_fused) {
            lidar_format(h, p);
            tile_dw = (uint32_t)(NGW_EPB * p.l_rb / 4);                 // (l_rb is even: 64 rows are a whole number of 16-byte pieces)
            tile_all = tile_dw + NGW_EPB / 4;                           // + one dump byte per lane (rays that report nothing store there)
        }
        off = (off + 3u) & ~3u;
        p.off_cand = off;
        if (alias) { p.off_ltile = off; off += cand_dw > tile_all ? cand_dw : tile_all; }
        else off += cand_dw;
        p.off_act = off; off += (uint32_t)(NGW_MAX_PLACE / 4);          // the placement sequence of the reset paths
        p.perm_lds = 0; p.off_perm = off;
        if (h->spec.n_passes) {
            // Shuffle array in LDS only while the wave's LDS stays small (<= 32 KiB, 5 waves/CU).  Measured at S = 32: the
            // extra 64 KiB halves the resident waves per CU and costs more (step 50 -> 115 us) than the HBM scratch column.
            const uint32_t perm_dw = (uint32_t)(S2 * 32 * 2 / 4);
            if ((size_t)(off + perm_dw) * 4 <= 32 * 1024) { p.perm_lds = 1; off += perm_dw; }
        }
        if (h->lidar_fused) {
            off = (off + 3u) & ~3u;
            p.off_litem = off; off += 2 * NGW_MAX_ITEMS / 4;
            off = (off + 3u) & ~3u;
            p.off_ltab = off; if (!h->lidar_world) off += 4 * NGW_LIDAR_MAX_BEAMS * NGW_LIDAR_MAX_RANGE * 2 / 4;
            if (!alias) { p.off_ltile = off; off += tile_all; }
        }
        // Philox word ring of the reset path (ngw_kernels.hip PHILOX_RING, 8 KB).  With the fused lidar epilogue it shares the
        // observation tile's region when that is big enough (the tile is rebuilt after any reset, the ring is dead by then): 8 KB
        // more would take an int32-row wave past 40 KB and a CU from four resident waves to three - measured as 13.6 -> 21.7 us per
        // batched step.  The ring is used when the reset has no shuffled-subset pass (those draw hundreds of words per lane:
        // register blocks, PhiloxRegs) and when its LDS does not cost a resident wave per CU (C5: 76 KB + 8 KB would halve the occupancy).
        h->off_rng = 0xFFFFFFFFu;                                      // = PhiloxRegs
        if (h->spec.n_passes == 0) {
            const uint32_t ring_dw = (uint32_t)(NGW_EPB * 32);
            if (h->lidar_fused && !alias && tile_dw >= ring_dw) h->off_rng = p.off_ltile;
            else if (h->lidar_fused && alias && tile_dw >= cand_dw + ring_dw) h->off_rng = p.off_ltile + cand_dw;
            else if (waves_per_cu(off + ring_dw) == waves_per_cu(off) || waves_per_cu(off + ring_dw) >= 4) { h->off_rng = off; off += ring_dw; }
        }
        return off;
    };
    uint32_t off = pass(false);
    bool alias = false;
    if (h->lidar_fused) {
        // share only where it buys a resident wave per CU (32 x 32 with int16 / packed rows: 87 KB -> 80.5 KB, one wave -> two); the
        // layouts of the small maps stay as they were measured.  NGW_LDS_ALIAS=0: A/B.
        const char* v = getenv("NGW_LDS_ALIAS");
        if (!(v && atoi(v) == 0)) {
            const uint32_t off_alias = pass(true);
            if (off_alias * 4u <= 160u * 1024u && ((size_t)off * 4 > 160 * 1024 || waves_per_cu(off_alias) > waves_per_cu(off))) { off = off_alias; alias = true; }
            else off = pass(false);
        }
    }
    if (getenv("NGW_DEBUG_LDS"))
        fprintf(stderr, "[ngw] LDS per wavefront: %zu B (S = %d, lidar %d, tile over candidate masks %d, ring %s) -> %u waves per CU\n", (size_t)off * 4, S,
                h->lidar_fused, (int)alias, h->off_rng == 0xFFFFFFFFu ? "registers" : "LDS", waves_per_cu(off));
    if ((size_t)off * 4 > 160 * 1024)
        return fail(NGW_E_INVALID_ARG, "map_size %d%s needs %zu B of LDS per wavefront (> 160 KiB)", S,
                    h->lidar_fused ? " with the fused lidar observation" : "", (size_t)off * 4);
    h->lds_bytes = (size_t)off * 4;
    return NGW_OK;
}

// the cold reset path reads its uniform arguments from the blob: keep them in step with the launch prototype
int upload_reset_u(ngw_handle* h) {
    const NgwLaunch& p = h->proto;
    NgwResetU ru = {};
    ru.perm = h->b.perm; ru.map = h->b.map; ru.inv = h->b.inv; ru.n_pad = h->n_pad; ru.seed = p.seed;
    ru.S = p.S; ru.S2 = p.S2; ru.K = p.K; ru.CW = p.CW; ru.perm_lds = p.perm_lds; ru.magicS = p.magicS; ru.off_rng = h->off_rng;
    const ngw_spec& s = h->spec;
    ru.wall_item = s.wall_item; ru.tap_item = s.tap_item; ru.tap_near = s.tap_near;
    int n_place = 0;
    for (int j = 0; j < s.n_start; j++) n_place += s.start_qty[j];
    ru.n_place = (uint8_t)n_place;
    ru.n_passes = s.n_passes; ru.n_inv_start = s.n_inv_start;
    for (int j = 0; j < s.n_passes; j++)
        ru.pass[j] = (uint32_t)s.pass_kind[j] | ((uint32_t)s.pass_item[j] << 8) | ((uint32_t)s.pass_from[j] << 16) |
                     ((uint32_t)(s.pass_pct_hi[j] - s.pass_pct_lo[j]) << 24);
    for (int j = 0; j < NGW_MAX_INV_START; j++) { ru.inv_start_item[j] = s.inv_start_item[j]; ru.inv_start_qty[j] = s.inv_start_qty[j]; }
    HIP_TRY(hipMemcpyAsync(&h->dspec->ru, &ru, sizeof(ru), hipMemcpyDefault, h->stream));
    NgwLaunch lp = h->proto;                                   // what the lean kernel's cold path reads instead of its kernarg
    lp.b = h->b;
    HIP_TRY(hipMemcpyAsync(&h->dspec->lp, &lp, sizeof(lp), hipMemcpyDefault, h->stream));
    {   // no-stage lean kernel: inventory rows | candidate masks | placement sequence
        NgwLaunch& q = h->ns_proto;
        q = h->proto;
        q.b = h->b;
        q.off_map = 0; q.off_inv = 0; q.off_cand = (uint32_t)(q.KP * NGW_EPB); q.off_act = q.off_cand + (uint32_t)(q.CW * NGW_EPB);
        q.perm_lds = 0; q.off_perm = 0; q.lcfg = nullptr; q.lout = nullptr; q.off_litem = q.off_ltab = q.off_ltile = 0;
        h->ns_lds = (size_t)(q.off_act + NGW_MAX_PLACE / 4) * 4;
        HIP_TRY(hipMemcpyAsync(&h->dspec->lp_ns, &q, sizeof(q), hipMemcpyDefault, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NGW_OK;
}

// Dedicated new-episode kernel: which specs it covers and its LDS carve-up (dword offsets).
void layout_reset_fast(ngw_handle* h) {
    const ngw_spec& s = h->spec;
    h->rf_nw = -1;
    // at most one subset pass, and one whose source cells need no map scan: AddItem / Crate (air: the interior minus the placed
    // items) or ReplaceItem / FireWall of the WALL item (the ring)
    const bool wall_pass = s.n_passes == 1 && s.pass_kind[0] == NGW_PASS_REPLACE && s.pass_from[0] == s.wall_item;
    const bool subset = (s.n_passes == 1 && s.pass_kind[0] == NGW_PASS_ADDITEM) || wall_pass;
    if (!h->fast_reset || s.tap_item || (s.n_passes && !subset)) return;          // other reset passes: general kernel
    const int S = s.map_size, S2 = S * S, CW = h->proto.CW;
    // small plain maps stay with the general kernel unless NGW_FAST_RESET=2 (measured in round 2 with the old two-phase store:
    // 10 x 10 plain 26.8 us vs 23.6 us for 65 536 envs)
    if (CW <= 2 && !subset && h->fast_reset < 2) return;
    NgwResetFast& a = h->rf;
    a = NgwResetFast{};
    int n_place = 0;
    for (int j = 0; j < s.n_start; j++) n_place += s.start_qty[j];
    if (n_place > 12) return;                                                       // the kernel sorts the placed items in 12 registers
    const int nw = CW <= 2 ? 2 : (CW <= 8 ? 8 : 0);
    int nb = 1;
    while ((1 << nb) < S2) nb++;                                                    // bits of a cell index: bit_length(S2 - 1)
    const uint32_t NBW = (uint32_t)((1 << nb) >> 5);                                // words of the per-env bit column the candidates index
    const uint32_t tmpl_cells = (uint32_t)(S2 + 16);
    uint32_t off = 0;
    a.off_ring = off; off += 16 * NGW_EPB;
    a.off_masks = off; if (nw == 0) off += (uint32_t)(2 * CW * NGW_EPB);
    a.off_placed = off; off += 13 * NGW_EPB;                                        // 12 placed items + the sentinel
    a.off_tmpl = off; off += (tmpl_cells + NGW_MAX_PLACE + 3) / 4;
    a.off_dom = off; if (subset) off += NBW;
    a.off_mcol = off; if (subset) off += NBW * NGW_EPB;
    off = (off + 3u) & ~3u;
    a.off_tile = off; off += (uint32_t)((S2 <= 512 ? (S2 * NGW_EPB + 15) / 16 * 16 : 144 * NGW_EPB) / 4);   // staging tile of the composed rows: the chunk's exact image up to 512-byte rows, else [64][128 + 16] bytes
    // (every dword counts: at 32 x 32 + AddItem the layout is 38.8 KB and four workgroups share a CU's 160 KB - one per SIMD)
    a.off_ctab = off; off += NGW_EPB + NGW_EPB * NGW_MAX_DEPTH / 2;
    if ((size_t)off * 4 > 160 * 1024) return;
    h->rf_lds = (size_t)off * 4;
    h->rf_nw = nw; h->rf_additem = subset ? 1 : 0;
    a.main = h->b; a.nx = h->nx;
    a.pctq = reinterpret_cast<const double*>(h->dspec->pctq[0]);
    a.n = h->n; a.env_base = h->env_base; a.seed = h->seed; a.flags = h->b.flags;
    a.S = S; a.S2 = S2; a.K = s.n_items; a.CW = CW; a.n_place = n_place; a.wall_item = s.wall_item;
    a.additem_item = subset ? s.pass_item[0] : 0; a.additem_span = subset ? s.pass_pct_hi[0] - s.pass_pct_lo[0] : 1;
    a.pass_wall = wall_pass ? 1 : 0;
    a.n_inv_start = s.n_inv_start;
    for (int j = 0; j < NGW_MAX_INV_START; j++) {
        a.inv_start_items |= (uint32_t)s.inv_start_item[j] << (8 * j);
        a.inv_start_qtys |= (uint32_t)s.inv_start_qty[j] << (8 * j);
    }
    const uint32_t W = (uint32_t)(S - 4);
    a.magicW = W ? (uint32_t)((0x100000000ull + W - 1) / W) : 0;
    a.magicS = (uint32_t)((0x100000000ull + (uint32_t)S - 1) / (uint32_t)S);
    a.sub_nb = nb; a.sub_fields = 32 / nb;
    a.img = S2 <= 512 ? 1 : 0;
    a.magicS2 = (uint32_t)((0x100000000ull + (uint32_t)S2 - 1) / (uint32_t)S2);
    {
        const int tail = S2 % 128, ush = (S2 & 15) == 0 ? 4 : ((S2 & 3) == 0 ? 2 : 0);
        const uint32_t nu = (uint32_t)((tail + (1 << ush) - 1) >> ush);
        a.magic_tail = nu ? (uint32_t)((0x100000000ull + nu - 1) / nu) : 0;
    }
}

// mode = NGW_MODE_RESET (mask_dev or nullptr) / NGW_MODE_REFILL; returns 1 if the dedicated kernel took the launch
int launch_reset_fast(ngw_handle* h, int mode, const uint8_t* mask_dev, bool* taken) {
    *taken = false;
    if (h->rf_nw < 0 || h->lidar_fused) return NGW_OK;
    NgwResetFast a = h->rf;
    a.main = h->b; a.nx = h->prefetch_every > 0 ? h->nx : NgwNx{}; a.mode = mode; a.reset_mask = mask_dev; a.stamps = h->proto.stamps;
    a.seq = mode == NGW_MODE_RESET ? h->launch_seq : 0u;
    HIP_TRY(ngw_reset_fast_launch(h->dspec, &a, h->rf_nw, h->rf_additem, (unsigned)(h->n_pad / NGW_EPB), h->rf_lds, h->stream));
    *taken = true;
    return NGW_OK;
}

int publish_nx(ngw_handle* h, bool on) {
    const NgwNx on_device = on ? h->nx : NgwNx{};                                  // null pointers switch the consume path off
    HIP_TRY(hipMemcpyAsync(&h->dspec->nx, &on_device, sizeof(NgwNx), hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->rf.nx = h->nx;                                                              // (the dedicated new-episode kernel's copy: launch_reset_fast masks it when prepared episodes are off)
    return NGW_OK;
}

// (Re)allocates the shadow buffers `depth` deep and publishes them to the kernels (`on`: the consume path uses them); every tag
// starts at 0 = nothing prepared.  Order: the NEW set is allocated, then published, and only then is the old one freed.  On a
// failed allocation (growing the depth multiplies the shadow memory - the expected way to run out) or a failed publish the handle
// keeps the set it had: NgwDevSpec::nx on the device, the dedicated kernel's rf.nx and h->nx never name freed memory, and the
// error goes back to the caller.  Synchronises the stream: called when prepared episodes are switched on and - rarely - when the
// depth grows.
int alloc_nx(ngw_handle* h, int depth, bool on) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    const NgwNx old = h->nx;
    const int old_depth = h->depth;
    NgwNx nw = NgwNx{};
    const size_t np = (size_t)h->n_pad * (size_t)depth, S2 = (size_t)h->proto.S2, K = (size_t)h->proto.K;
    int rc = dev_alloc(h, &nw.map, np * S2);
    if (!rc) rc = dev_alloc(h, &nw.loc, np * 2);
    if (!rc) rc = dev_alloc(h, &nw.facing, np);
    if (!rc) rc = dev_alloc(h, &nw.inv, np * K);
    if (!rc) rc = dev_alloc(h, &nw.episode, np);
    if (!rc) rc = dev_alloc(h, &nw.slow, 16);
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(NGW_E_HIP, "zero-fill of the prepared-episode buffers failed");
    auto drop = [&](const NgwNx& x) {
        void* const part[6] = {x.map, x.loc, x.facing, x.inv, x.episode, x.slow};
        for (void* q : part) if (q) dev_free(h, q);
    };
    if (rc) { drop(nw); (void)hipGetLastError(); return rc; }      // the old set stays in force everywhere
    nw.slow_host = old.slow_host;
    if (!nw.slow_host) {
        void* q = nullptr;
        if (hipHostMalloc(&q, 64, hipHostMallocMapped) == hipSuccess) {           // (without it the cadence simply stays fixed)
            memset(q, 0, 64);
            h->host_allocs.push_back(q);
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, q, 0) == hipSuccess && d == q) nw.slow_host = static_cast<uint32_t*>(q);
        }
    }
    nw.stride = h->n_pad; nw.dmask = depth - 1;
    h->nx = nw; h->depth = depth;
    if (int prc = publish_nx(h, on)) {                             // the device may hold either copy of the pointers: both sets are still alive
        h->nx = old; h->depth = old_depth;
        (void)publish_nx(h, on && old.episode != nullptr);
        drop(nw);
        return prc;
    }
    if (nw.slow_host) { nw.slow_host[0] = 0; nw.slow_host[1] = 0; }
    h->slow_seen = 0; h->refill_seen = 0;
    drop(old);
    return NGW_OK;
}

// Prepared next episodes: one refill re-prepares the shadow rows that resets have consumed since the last one.  Under the
// DEFAULT setting the host adapts to how fast episodes end: a reset that finds no prepared row runs the placement loop inside
// a step and the whole launch waits for it, so those resets are counted on the device and reported by every refill.  Too
// many of them per refill - more than one step in eight of the window would be slow - and the prepared episodes first get
// DEEPER (2, then 4 per env: an env may then end that many episodes between two refills), then refills get more frequent;
// four quiet refills in a row make them less frequent again (sixteen, sixty-four, ... after a doubling that had to be taken
// back: no ping-pong between two cadences).  Results never depend on any of it.
void adapt_cadence(ngw_handle* h) {
    if (h->capturing || h->prefetch_user || !h->adapt || !h->nx.slow_host) return;
    // what the last refill launch THE GPU HAS RUN reported: its number and the count of stale-row resets so far.  The host may be
    // many launches ahead of the device (an eager loop without synchronisation): no new report = no information, and one
    // report may stand for several refills.  (A replayed graph repeats the number it was captured with: not newer = no report.)
    const uint32_t seq = ((volatile uint32_t*)h->nx.slow_host)[1], cur = ((volatile uint32_t*)h->nx.slow_host)[0];
    const int32_t refills = (int32_t)(seq - h->refill_seen);
    if (refills <= 0) return;
    h->refill_seen = seq;
    const uint32_t delta = cur - h->slow_seen;
    h->slow_seen = cur;
    const uint32_t many = (uint32_t)(h->cadence / 8 > 2 ? h->cadence / 8 : 2);
    if (delta / (uint32_t)refills >= many) {                        // two noisy refills in a row: a one-off burst (stale tags after
        h->quiet = 0;                                               // ngw_set_state, the first steps of a handle) does not count
        h->noisy += refills;
        if (h->noisy >= 2) {
            h->noisy = 0;
            if (h->depth < 4 && !h->depth_user) {                    // deeper first
                const int every = h->prefetch_every;
                // (a failed allocation keeps the set the handle had - alloc_nx frees the old one only after the new one stands - and
                //  the depth simply stops growing: the cadence is halved instead from the next noisy window on)
                if (alloc_nx(h, h->depth * 2, true) == NGW_OK) {
                    h->since_refill = every;                        // (every row is stale now: refill at once)
                } else {
                    h->depth_user = 1;                              // (no further attempts: the memory is not there)
                    h->adapt_error = true;                          // noted by the next refill (launch_refill)
                }
            } else {
                h->cadence = h->cadence / 2 < 2 ? 2 : h->cadence / 2;
                // a doubling that had to be taken back: the next attempt waits four times as long (no ping-pong between two levels)
                if (h->probing) h->quiet_need = h->quiet_need * 4 > 4096 ? 4096 : h->quiet_need * 4;
            }
            h->probing = false;
            h->adapted = true;
        }
    } else {
        h->noisy = 0;
        // The upward probe needs reports that stand for a FEW refills each.  A replayed graph reports a whole replay at once (~55 refills at
        // FireWall's cadence): averaged over that many a cadence looks quiet that is noisy refill by refill, every change re-captures the
        // graph (~2 ms), and the eager steps around the replays take the change back - tools/x1_probe.py: 18 -> 36 -> 72 -> 36 with a 15 us
        // region in between.  Such a report still tightens the cadence (above); it does not lengthen it.
        if (refills <= 4 && h->cadence < h->prefetch_every && (h->quiet += refills) >= h->quiet_need) {
            h->cadence = h->cadence * 2 > h->prefetch_every ? h->prefetch_every : h->cadence * 2;
            h->quiet = 0;
            h->probing = true;
            h->adapted = true;
        }
    }
}

int launch_refill(ngw_handle* h) {
    h->since_refill = 0;
    adapt_cadence(h);
    if (h->adapt_error) {                                           // (not fatal: the handle keeps working at the depth it has)
        h->adapt_error = false;
        fail(NGW_E_HIP, "prepared episodes: not enough device memory for %d rows per env, staying at %d (the call itself succeeded)", h->depth * 2, h->depth);
    }
    h->refill_count++;
    bool taken = false;
    if (int rc = launch_reset_fast(h, NGW_MODE_REFILL, nullptr, &taken)) return rc;
    if (taken) return NGW_OK;
    // the general kernel prepares one slot per launch (its shadow set is the launch's buffer set)
    const size_t np = (size_t)h->n_pad, S2 = (size_t)h->proto.S2, K = (size_t)h->proto.K;
    for (int slot = 0; slot < h->depth; slot++) {
        NgwLaunch rf = h->proto;
        rf.b = NgwBufs{};
        rf.b.map = h->nx.map + slot * np * S2; rf.b.loc = h->nx.loc + slot * np * 2; rf.b.facing = h->nx.facing + slot * np;
        rf.b.inv = h->nx.inv + slot * np * K; rf.b.episode = h->nx.episode + slot * np;
        rf.b.flags = h->b.flags; rf.b.perm = h->b.perm;
        rf.mode = NGW_MODE_REFILL; rf.n_steps = 1;
        rf.actions = reinterpret_cast<const int32_t*>(h->b.episode);
        rf.reset_mask = nullptr; rf.action_seed = 0; rf.t0 = (int64_t)h->refill_count;   // (REFILL: t0 = the refill's number)
        rf.autoreset = slot; rf.horizon = h->depth - 1;                                    // (REFILL: the slot and depth - 1)
        HIP_TRY(ngw_launch(h->dspec, &rf, h->map_mode, 0, (unsigned)(h->n_pad / NGW_EPB), h->lds_bytes, h->stream));
    }
    return NGW_OK;
}

int launch(ngw_handle* h, int mode, int n_steps, const int32_t* actions_dev, const uint8_t* mask_dev, uint64_t action_seed, int64_t t0) {
    h->mirror_valid = false;                          // (ngw_step_host's delta path sets it again after its own launch)
    NgwLaunch a = h->proto;
    a.b = h->b;
    a.mode = mode;
    a.n_steps = n_steps;
    a.actions = actions_dev;
    a.reset_mask = mask_dev;
    a.autoreset = h->autoreset;
    a.horizon = h->horizon;
    a.action_seed = action_seed;
    a.t0 = t0;
    a.seq = h->launch_seq;
    a.action0 = h->launch_action0; a.use_action0 = h->launch_use_action0 ? 1 : (h->launch_act_u8 ? 2 : 0);
    const unsigned grid = (unsigned)(h->n_pad / NGW_EPB);
    bool taken = false;
    if (mode == NGW_MODE_RESET) { if (int rc = launch_reset_fast(h, NGW_MODE_RESET, mask_dev, &taken)) return rc; }
    if (!taken && mode == NGW_MODE_STEP && h->nostage && !h->lidar_fused) {   // maps read in place: no-stage step kernel
        NgwLaunch q = h->ns_proto;
        q.b = h->b; q.mode = mode; q.n_steps = 1; q.actions = actions_dev; q.autoreset = h->autoreset; q.horizon = h->horizon; q.stamps = h->proto.stamps;
        q.seq = h->launch_seq; q.action0 = h->launch_action0; q.use_action0 = h->launch_use_action0 ? 1 : (h->launch_act_u8 ? 2 : 0);
        HIP_TRY(ngw_launch(h->dspec, &q, h->map_mode, 8 | (h->ext ? 2 : 0), grid, h->ns_lds, h->stream));
        taken = true;
    }
    if (!taken) {
        if (!h->general_ok)
            return fail(NGW_E_INVALID_ARG, "map_size %d: this call keeps a wavefront's 64 maps in LDS (fused rollouts, the fused lidar epilogue) "
                                           "and they need more than 160 KiB; per-launch steps and resets are available", h->proto.S);
        HIP_TRY(ngw_launch(h->dspec, &a, h->map_mode, (h->lidar_fused ? 1 : 0) | (h->ext ? 2 : 0), grid, h->lds_bytes, h->stream));
    }
    if (h->prefetch_every > 0 && (mode == NGW_MODE_STEP || mode == NGW_MODE_RESET || mode == NGW_MODE_ROLLOUT || mode == NGW_MODE_ROLLOUT_ACT)) {
        // Prepared next episodes: every `prefetch_every` batched steps (and right after an explicit reset) one more launch
        // refills the shadow rows that resets have consumed since.  Same stream, so it is ordered between the steps.
        h->since_refill += mode == NGW_MODE_RESET ? h->prefetch_every : n_steps;
        if (h->since_refill >= h->cadence) return launch_refill(h);
    }
    return NGW_OK;
}

// A fused rollout as launches of at most `prefetch_every` steps with the refill launches between them: with prepared next
// episodes on, an env's reset inside the launch copies its prepared row - but only the first one, the shadow rows are
// re-prepared between launches.  Same action stream (keyed by the absolute step), same results as one launch.
int rollout_chunks(ngw_handle* h, int mode, int32_t n_steps, const int32_t* actions_dev, uint64_t action_seed, int64_t t0, int64_t step_stride) {
    if (h->term_on)
        return fail(NGW_E_INVALID_ARG, "fused rollouts keep no terminal observations (the state lives on chip between steps): switch "
                                       "ngw_set_terminal_capture off, or step with ngw_step / ngw_step_device");
    // a prepared row serves an env's FIRST reset of a launch, and under a horizon H an env resets at most once per H steps
    // (plus the rare early `done`): H-step launches (capped) keep the per-launch staging cost low; no horizon: 4 cadences
    int32_t chunk = n_steps;
    if (h->prefetch_every > 0) {
        chunk = h->horizon > 0 ? (h->horizon < 256 ? h->horizon : 256) : 4 * h->prefetch_every;
        if (chunk < h->prefetch_every) chunk = h->prefetch_every;
        if (h->cadence < h->prefetch_every) chunk = h->cadence < 4 ? 8 : 2 * h->cadence;   // episodes end faster than rows are prepared
    }
    for (int32_t done = 0; done < n_steps; done += chunk) {
        const int32_t k = n_steps - done < chunk ? n_steps - done : chunk;
        h->proto.row_reward = h->row_reward ? h->row_reward + (int64_t)done * h->row_stride : nullptr;
        h->proto.row_done = h->row_done ? h->row_done + (int64_t)done * h->row_stride : nullptr;
        h->proto.row_stride = h->row_stride; h->proto.acc = h->acc;
        const int rc = mode == NGW_MODE_ROLLOUT ? launch(h, mode, k, nullptr, nullptr, action_seed, t0 + done)
                                                : launch(h, mode, k, actions_dev + (int64_t)done * step_stride, nullptr, 0, step_stride);
        h->proto.row_reward = nullptr; h->proto.row_done = nullptr; h->proto.acc = nullptr;
        if (rc) return rc;
    }
    return NGW_OK;
}

// One page-locked block for everything ngw_step_host returns.  Sections (index: 0 map, 1 agent_location, 2 agent_facing_id,
// 3 inventory, 4 reward, 5 done, 6 info words, 7 error flags, 8 selected, 9 step_count) are padded to 256 bytes and lie in
// memory in the order map | inventory | selected | agent_location | agent_facing_id | reward | done | info | flags | step_count:
// the first three change by a few bytes per step and are refreshed by deltas, the rest is one contiguous copy.
constexpr int HS_ORDER[10] = {0, 3, 8, 1, 2, 4, 5, 6, 7, 9};
constexpr int HS_DENSE_FIRST = 1;                       // section index (agent_location) the contiguous dense part starts with
void host_step_layout(const ngw_handle* h, uint64_t off[11]) {
    const uint64_t n = (uint64_t)h->n, S2 = (uint64_t)h->proto.S2, K = (uint64_t)h->proto.K;
    const uint64_t bytes[10] = {n * S2, n * 8, n * 4, n * K * 4, n * 4, n, n * 4, 4, n, n * 4};
    uint64_t o = 0;
    for (int k = 0; k < 10; k++) { const int i = HS_ORDER[k]; off[i] = o; o += (bytes[i] + 255) & ~(uint64_t)255; }
    off[10] = o;
}

// Are the output arrays of ngw_step_host the sections of one block laid out as ngw_host_step_layout says (base = map)?
bool one_block(const ngw_handle* h, const void* map, const void* loc, const void* facing, const void* inv, const void* reward, const void* done,
               const void* flags, const void* selected, const void* step_count) {
    uint64_t off[11];
    host_step_layout(h, off);
    const uint8_t* b = static_cast<const uint8_t*>(map);
    const void* const got[10] = {map, loc, facing, inv, reward, done, nullptr, flags, selected, step_count};
    for (int i = 1; i < 10; i++)
        if (i != 6 && got[i] != b + off[i]) return false;
    return true;
}

void drop_graph(ngw_handle* h) {
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    h->graph_exec = nullptr;
    h->graph = nullptr;
    h->graph_steps = 0;
}

}  // namespace

extern "C" {

#define D2H(dst, src, bytes)                                                                             \
    do {                                                                                                 \
        if (dst) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDefault, h->stream));        \
    } while (0)

int ngw_abi_version(void) { return NGW_ABI_VERSION; }
int ngw_spec_size(void) { return (int)sizeof(ngw_spec); }
const char* ngw_last_error(void) { return g_err; }

int ngw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int ngw_create(const ngw_spec* spec, int64_t n_envs, int device, uint64_t seed, int64_t env_index_base, ngw_handle** out) {
    if (!out) return fail(NGW_E_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (int rc = check_spec(spec)) return rc;
    if (n_envs < 1) return fail(NGW_E_INVALID_ARG, "n_envs must be >= 1");
    if (env_index_base < 0) return fail(NGW_E_INVALID_ARG, "env_index_base must be >= 0");
    int ndev = ngw_device_count();
    if (ndev < 1) return fail(NGW_E_NO_DEVICE, "no HIP device visible: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail(NGW_E_INVALID_ARG, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    ngw_handle* h = new ngw_handle();
    h->spec = *spec;
    h->n = n_envs;
    h->n_pad = (n_envs + NGW_EPB - 1) / NGW_EPB * NGW_EPB;
    h->device = device;
    h->seed = seed;
    h->env_base = env_index_base;
    auto bail = [&](int rc) { ngw_destroy(h); return rc; };
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(NGW_E_HIP, "hipStreamCreate failed"));
    h->own_stream = true;
    if (const char* v = getenv("NGW_FAST_RESET")) h->fast_reset = atoi(v);
    if (const char* v = getenv("NGW_ADAPT_PREFETCH")) h->adapt = atoi(v) != 0;
    if (const char* v = getenv("NGW_HOST_DELTA")) h->host_delta = atoi(v) != 0;
    if (const char* v = getenv("NGW_WIRE_DIRECT")) h->wire_direct = atoi(v) != 0;
    if (const char* v = getenv("NGW_WIRE_MERGE")) h->wire_merge = atoi(v) != 0;
    if (const char* v = getenv("NGW_ZC_BYTES")) { h->zc_bytes = (size_t)atoll(v); if (!h->zc_bytes) h->zc_bytes = 1; }
    {
        // Which per-launch step kernel: the one that reads the <= 14 cells a step needs straight from HBM, at EVERY map size.  Up to
        // round 3 the 10 x 10 (and 6 x 6) maps - whose 64 rows arrive in one round of loads and land in LDS as they are - kept the
        // kernel that stages them through LDS ("within 4 % of each other, either way round depending on the batch size").  With the
        // cold path out of the hot path's register allocation (round 4) the in-place kernel is the faster one at every batch size
        // measured (tools/ab_stage.sh, us per replayed launch, staged / in place: 64 envs 2.9 / 2.7, 4 096 3.2 / 3.1, 16 384 3.3 / 3.2,
        // 32 768 (C4) 4.1 / 4.0, 65 536 (C2) 4.2 / 3.8), so it is the one that runs; the staged kernel remains what the fused lidar
        // epilogue rides on (it marches over the maps in LDS).  NGW_NOSTAGE=<min S*S> (A/B): in place from that size on (0 = never).
        const int s2 = spec->map_size * spec->map_size;
        h->nostage = 1;
        if (const char* v = getenv("NGW_NOSTAGE")) { const int min_s2 = atoi(v); h->nostage = min_s2 > 0 && s2 >= min_s2; }
    }

    const int S = spec->map_size, S2 = S * S, K = spec->n_items;
    const size_t np = (size_t)h->n_pad;
    {
        const char* v = getenv("NGW_HOST_STATE");
        h->hostres = h->n_pad == NGW_EPB && !(v && atoi(v) == 0);
    }
    int rc = NGW_OK;
    {   // the six arrays a step's prologue reads are ONE allocation: the step kernel names them by a base + 32-bit offsets that
        // travel in the preloaded head of its argument block (ngw_lean.inc); map rows first (the base)
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t o_map = 0, o_inv = up(np * S2), o_loc = o_inv + up(np * K * 4), o_fac = o_loc + up(np * 8), o_sel = o_fac + up(np * 4),
                     o_stp = o_sel + up(np), total = o_stp + up(np * 4);
        uint8_t* slab = nullptr;
        if (!rc) rc = dev_alloc(h, &slab, total);
        if (!rc) {
            h->b.map = reinterpret_cast<int8_t*>(slab + o_map); h->b.inv = reinterpret_cast<int32_t*>(slab + o_inv);
            h->b.loc = reinterpret_cast<int32_t*>(slab + o_loc); h->b.facing = reinterpret_cast<int32_t*>(slab + o_fac);
            h->b.selected = slab + o_sel; h->b.step_count = reinterpret_cast<int32_t*>(slab + o_stp);
        }
    }
    if (!rc) rc = dev_alloc(h, &h->b.episode, np);
    if (!rc) rc = dev_alloc(h, &h->b.reward, np);
    if (!rc) rc = dev_alloc(h, &h->b.done, np);
    if (!rc) rc = dev_alloc(h, &h->b.info, np);
    if (!rc) rc = dev_alloc(h, &h->b.flags, 1);
    if (h->hostres) {
        NgwMirror& m = h->mir;
        if (!rc) rc = mirror_alloc(h, &m.map, np * S2);
        if (!rc) rc = mirror_alloc(h, &m.loc, np * 2);
        if (!rc) rc = mirror_alloc(h, &m.facing, np);
        if (!rc) rc = mirror_alloc(h, &m.inv, np * K);
        if (!rc) rc = mirror_alloc(h, &m.selected, np);
        if (!rc) rc = mirror_alloc(h, &m.step_count, np);
        if (!rc) rc = mirror_alloc(h, &m.reward, np);
        if (!rc) rc = mirror_alloc(h, &m.done, np);
        if (!rc) rc = mirror_alloc(h, &m.info, np);
        if (!rc) rc = mirror_alloc(h, &h->b.flags_host, 16);                       // [0] flags, [NGW_SEQ_WORD] sequence of the last finished step
    }
    if (!rc) rc = dev_alloc(h, &h->actions_dev, np);
    if (!rc) rc = dev_alloc(h, &h->mask_dev, np);
    if (!rc && spec->n_passes) rc = dev_alloc(h, &h->b.perm, np * S2);   // HBM fallback for maps too big for the LDS shuffle
    if (!rc) rc = dev_alloc(h, &h->dspec, 1);
    if (rc) return bail(rc);
    {
        NgwDevSpec hs;
        memset(&hs, 0, sizeof(hs));
        hs.sp = *spec;
        hs.x.fire_item = spec->fire_item; hs.x.fire_reward = spec->fire_reward; hs.x.fence_item = spec->fence_item;
        hs.x.fence_mode = spec->fence_mode; hs.x.crate_item = spec->crate_item;
        hs.x.nest = (uint32_t)spec->ext_flags | ((uint32_t)spec->fire_skip_recipe << 8);
        for (int i = 0; i < K; i++) hs.x.crate_add[i >> 3] |= (uint32_t)(spec->crate_add[i] & 15u) << (4 * (i & 7));
        h->ext = (spec->fire_item || spec->fence_mode || spec->crate_item) ? 1 : 0;
        for (int j = 0; j < spec->n_passes; j++)
            for (int i = 0; i < 64; i++) hs.pctq[j][i] = (double)(spec->pass_pct_lo[j] + i) / 100.0;
        NgwStepU& u = hs.u;
        for (int i = 0; i < K; i++) {
            if (spec->breakable[i]) u.brk_mask |= 1u << i;
            if (spec->entity[i]) u.ent_mask |= 1u << i;
            if (spec->break_reward[i] != spec->reward_step) { u.rew_mask |= 1u << i; u.break_reward = spec->break_reward[i]; }
            if (spec->break_qty[i] == 2) u.brk2_mask |= 1u << i;
        }
        u.n_actions = spec->n_actions; u.reward_step = spec->reward_step; u.reward_done = spec->reward_done;
        u.cost_forward = spec->cost_forward; u.cost_turn = spec->cost_turn; u.cost_break = spec->cost_break;
        u.cost_place = spec->cost_place; u.cost_extract = spec->cost_extract; u.cost_select = spec->cost_select;
        u.table_item = spec->table_item; u.goal_item = spec->goal_item;
        u.place_item = spec->place_item; u.place_near = spec->place_near; u.n_entities = spec->n_entities;
        u.ext_src = spec->ext_src; u.ext_near = spec->ext_near; u.ext_out = spec->ext_out; u.ext_qty = spec->ext_qty;
        u.ext_consume = spec->ext_consume; u.ext_cost_ok = spec->ext_cost_ok;
        u.axe_item = spec->axe_item; u.axe_cost = spec->axe_cost; u.axe_qty = spec->axe_qty;
        u.place_reward = spec->place_reward; u.ext_reward = spec->ext_reward; u.axe_reward = spec->axe_reward;
        u.axe_required = spec->axe_required;
        u.cost_chop = spec->cost_chop; u.cost_jump = spec->cost_jump; u.chop_reward = spec->chop_reward;
        for (int a = 0; a < spec->n_actions; a++) {
            if (spec->act_kind[a] == NGW_ACT_JUMP) u.feat |= NGW_FEAT_JUMP;
            if (spec->act_kind[a] == NGW_ACT_CHOP) u.feat |= NGW_FEAT_CHOP;
        }
        for (int j = 0; j < spec->n_start; j++)
            for (int q = 0; q < spec->start_qty[j]; q++) hs.place_seq[hs.n_place++] = spec->start_item[j];
        for (int a = 0; a < spec->n_actions; a++) {
            uint32_t d[5] = {0, 0, 0, 0, 0};                       // recipe fields of a Craft action, packed (kind | arg<<8 | n_inputs<<16 | needs_table<<24; input ids; input quantities)
            const uint32_t kind = spec->act_kind[a], arg = spec->act_arg[a];
            d[0] = kind | (arg << 8);
            if (kind == NGW_ACT_CRAFT) {
                const int r = (int)arg;
                d[0] |= ((uint32_t)spec->recipe_n_in[r] << 16) | ((uint32_t)(spec->recipe_needs_table[r] ? 1 : 0) << 24);
                for (int j = 0; j < spec->recipe_n_in[r]; j++) {
                    const uint32_t item = spec->recipe_in_item[r][j];
                    d[1] |= item << (8 * j);
                    d[2] |= (uint32_t)spec->recipe_in[r][item] << (8 * j);
                }
                d[3] = spec->recipe_out_item[r] | ((uint32_t)spec->recipe_out_qty[r] << 8) |
                       ((uint32_t)spec->cost_missing[r] << 16) | ((uint32_t)spec->cost_no_table[r] << 24);
                d[4] = spec->cost_ok[r] | ((uint32_t)(uint8_t)spec->recipe_reward[r] << 8);
            }
            // the lean kernel's micro-op entry (NGW_LEAN_DW): conditions, per-outcome message / argument / cost, success effects
            {
                uint32_t* l = hs.act_lean + a * NGW_LEAN_DW;
                uint32_t abit = NGW_CB_FALSE, bbit = NGW_CB_FALSE, msg[3] = {0, 0, 0}, asel[3] = {0, 0, 0}, move = 0, turn = 0, cellw = 0, selflag = 0;
                uint32_t argconst = 0, is_break = 0, cslot = 0, cellv = 0, rbit = NGW_CB_FALSE, slotsel = 0, cost[3] = {0, 0, 0};
                int delta = 0, rewc = 0;
                auto costs = [&](uint32_t c) { cost[0] = cost[1] = cost[2] = c; };
                switch (kind) {
                case NGW_ACT_FORWARD: abit = NGW_CB_FRONT_NZ; msg[1] = NGW_MSG_BLOCK_IN_PATH; move = 1; costs(spec->cost_forward); break;
                case NGW_ACT_JUMP: abit = NGW_CB_JUMP_BLOCKED; msg[1] = NGW_MSG_BLOCK_IN_PATH; move = 2; costs(spec->cost_jump); break;
                case NGW_ACT_LEFT: turn = 1; costs(spec->cost_turn); break;
                case NGW_ACT_RIGHT: turn = 2; costs(spec->cost_turn); break;
                case NGW_ACT_BREAK:
                    abit = NGW_CB_NOT_BRK; msg[1] = NGW_MSG_CANNOT_BREAK; asel[1] = 1;
                    bbit = NGW_CB_NEED_AXE; msg[2] = NGW_MSG_NEED_AXE; asel[2] = 2; argconst = spec->axe_item;
                    cellw = 1; slotsel = 1; is_break = 1; rbit = NGW_CB_BRK_REWARD; costs(spec->cost_break);
                    break;
                case NGW_ACT_CHOP:
                    abit = NGW_CB_NOT_BRK; msg[1] = NGW_MSG_CANNOT_CHOP; asel[1] = 1;
                    cellw = 1; slotsel = 1; delta = 2; rbit = NGW_CB_TRUE; rewc = spec->chop_reward; costs(spec->cost_chop);
                    break;
                case NGW_ACT_PLACE:
                    abit = NGW_CB_NO_PLACE_ITEM; msg[1] = NGW_MSG_NOT_IN_INVENTORY;
                    bbit = NGW_CB_FRONT_NZ; msg[2] = NGW_MSG_ALREADY_EXISTS; asel[2] = 1;
                    msg[0] = NGW_MSG_PLACED; asel[0] = 2; argconst = spec->place_item;
                    cellw = 1; cellv = spec->place_item; slotsel = 2; cslot = spec->place_item; delta = -1;
                    rbit = NGW_CB_NEAR_PLACE; rewc = spec->place_reward; costs(spec->cost_place);
                    break;
                case NGW_ACT_EXTRACT:
                    abit = NGW_CB_NOT_SRC; msg[1] = NGW_MSG_EXTRACT_NO_SRC; bbit = NGW_CB_NOT_NEAR; msg[2] = NGW_MSG_EXTRACT_NOT_NEAR;
                    slotsel = 2; cslot = spec->ext_out; delta = spec->ext_qty; cellw = spec->ext_consume ? 1 : 0;
                    rbit = NGW_CB_TRUE; rewc = spec->ext_reward; costs(spec->cost_extract); cost[0] = spec->ext_cost_ok;
                    break;
                case NGW_ACT_CRAFT: {
                    const int r = (int)arg;
                    abit = NGW_CB_MISSING; msg[1] = NGW_MSG_MISSING_ITEMS; asel[1] = 3; cost[1] = spec->cost_missing[r];
                    bbit = NGW_CB_NEED_TABLE; msg[2] = NGW_MSG_NEED_TABLE; cost[2] = spec->cost_no_table[r];
                    msg[0] = NGW_MSG_CRAFTED; asel[0] = 2; argconst = spec->recipe_out_item[r];
                    slotsel = 2; cslot = spec->recipe_out_item[r]; delta = spec->recipe_out_qty[r];
                    rbit = NGW_CB_TRUE; rewc = spec->recipe_reward[r]; cost[0] = spec->cost_ok[r];
                    break;
                }
                case NGW_ACT_SELECT: abit = NGW_CB_NO_ARG_ITEM; msg[1] = NGW_MSG_NOT_IN_INVENTORY; selflag = 1; costs(spec->cost_select); break;
                default: break;
                }
                static_assert(NGW_ACT_JUMP < 16 && NGW_MAX_RECIPE_INPUTS < 8 && NGW_MSG_FIRE_WALL < 16, "act_lean field widths");
                l[0] = kind | (((d[0] >> 16) & 7u) << 4) | (((d[0] >> 24) & 1u) << 7) | (arg << 8) | (argconst << 16) | (is_break << 24);
                l[1] = d[1]; l[2] = d[2];
                l[3] = cslot | (cellv << 8) | ((uint32_t)(uint16_t)(int16_t)delta << 16);
                l[4] = abit | (bbit << 4) | (msg[0] << 8) | (msg[1] << 12) | (msg[2] << 16) | (asel[0] << 20) | (asel[1] << 22) | (asel[2] << 24) |
                       (move << 26) | (turn << 28) | (cellw << 30) | (selflag << 31);
                l[5] = (uint32_t)(uint8_t)(int8_t)rewc | (rbit << 8) | (slotsel << 12) | ((cost[0] & 63u) << 14) | ((cost[1] & 63u) << 20) | ((cost[2] & 63u) << 26);
            }
        }
        hs.mir = h->mir;
        if (hipMemcpyAsync(h->dspec, &hs, sizeof(hs), hipMemcpyDefault, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess)
            return bail(fail(NGW_E_HIP, "spec upload failed"));
    }

    // launch layout
    NgwLaunch& p = h->proto;
    p.n = h->n; p.n_pad = h->n_pad; p.env_base = h->env_base; p.seed = seed;
    p.S = S; p.S2 = S2; p.K = K; p.KP = K | 1;
    p.magicK = (uint32_t)((0x100000000ull + (uint32_t)K - 1) / (uint32_t)K);
    p.magicS = (uint32_t)((0x100000000ull + (uint32_t)S - 1) / (uint32_t)S);
    const int S2r = (S2 + 3) / 4;                       // dwords per map, rounded up
    const int MSdw = (S2r & 1) ? S2r : S2r + 1;         // odd dword stride -> conflict-free per-lane cell reads
    p.MS = ((S2 & 3) == 0 && (S2r & 1)) ? S2 : MSdw * 4;
    h->map_mode = (p.MS == S2) ? NGW_MAP_STRAIGHT : (((S2 & 3) == 0) ? NGW_MAP_DWORD : NGW_MAP_BYTE);
    const uint32_t div = ((S2 & 3) == 0) ? (uint32_t)(S2 / 4) : (uint32_t)S2;
    p.magic = (uint32_t)((0x100000000ull + div - 1) / div);
    p.CW = ((S - 4) * (S - 4) + 31) / 32;
    if (int rc = layout_lds(h)) {
        // Maps beyond ~46 x 46 do not fit the kernels that keep a wave's 64 maps in LDS.  The no-stage step kernel and the
        // dedicated new-episode kernel do not need them there: such a handle steps and resets, and refuses what it cannot run.
        if (!h->nostage) return bail(rc);
        h->general_ok = false;
        h->lds_bytes = 0; p.perm_lds = 0; h->off_rng = 0xFFFFFFFFu;
    }
    if (int rc = upload_reset_u(h)) return bail(rc);
    layout_reset_fast(h);
    if (!h->general_ok && h->rf_nw < 0)
        return bail(fail(NGW_E_INVALID_ARG, "map_size %d needs more than 160 KiB of LDS per wavefront for this configuration's resets "
                                            "(reset passes that read the map, the v0 tree tap or more than 12 placed items keep the general kernel)", S));
    if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(fail(NGW_E_HIP, "stream sync failed after allocation"));
    *out = h;
    return NGW_OK;
}

int ngw_destroy(ngw_handle* h) {
    if (!h) return NGW_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void* p : h->allocs) (void)hipFree(p);
    for (void* p : h->host_allocs) (void)hipHostFree(p);
    drop_graph(h);
    if (h->info_host) (void)hipHostFree(h->info_host);
    if (h->zc_host) (void)hipHostFree(h->zc_host);
    if (h->mask_ev[0]) (void)hipEventDestroy(h->mask_ev[0]);
    if (h->mask_ev[1]) (void)hipEventDestroy(h->mask_ev[1]);
    if (h->mask_pin) ngw_host_free(h->mask_pin);
    if (h->act_ev[0]) (void)hipEventDestroy(h->act_ev[0]);
    if (h->act_ev[1]) (void)hipEventDestroy(h->act_ev[1]);
    if (h->act_pin) ngw_host_free(h->act_pin);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->order_ev) (void)hipEventDestroy(h->order_ev);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return NGW_OK;
}

int ngw_set_autoreset(ngw_handle* h, int autoreset, int horizon) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (horizon < 0) return fail(NGW_E_INVALID_ARG, "horizon must be >= 0");
    h->autoreset = autoreset ? 1 : 0;
    h->horizon = horizon;
    if (!h->prefetch_user) {
        // Prepared next episodes are the default under autoreset (bit-identical results; a batch whose episode ends are spread
        // over the steps runs ~3x faster), unless the horizon is too short for refills to keep up.  A refill costs a
        // reset's latency however few rows are stale, and under a horizon H an env needs a new row once per H steps: the cadence
        // is 3/4 of the horizon (32 .. 128 steps; 32 without a horizon) - episodes that end early make some resets miss their
        // row, and the cadence then adapts downwards (adapt_cadence).
        int every = 0;
        if (h->autoreset && horizon == 0) every = 32;
        else if (h->autoreset && horizon >= 64) { every = 3 * horizon / 4; every = every < 32 ? 32 : (every > 128 ? 128 : every); }
        if (every != h->prefetch_every) { const int rc = ngw_set_reset_prefetch(h, every); h->prefetch_user = 0; return rc; }
    }
    return NGW_OK;
}

int ngw_get_reset_prefetch(ngw_handle* h, int32_t* every_n_steps) {
    if (!h || !every_n_steps) return fail(NGW_E_INVALID_ARG, "NULL argument");
    *every_n_steps = h->prefetch_every;
    return NGW_OK;
}

int ngw_set_reset_prefetch(ngw_handle* h, int32_t every_n_steps) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (every_n_steps < 0) return fail(NGW_E_INVALID_ARG, "every_n_steps must be >= 0");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    drop_graph(h);                                   // captured launches bake the cadence in
    if (every_n_steps > 0 && !h->nx.episode) { if (int rc = alloc_nx(h, h->depth, true)) return rc; }
    else if (int rc = publish_nx(h, every_n_steps > 0)) return rc;
    h->prefetch_every = every_n_steps;
    h->cadence = every_n_steps; h->quiet = 0; h->noisy = 0; h->quiet_need = 4; h->probing = false;
    h->prefetch_user = 1;
    h->since_refill = every_n_steps;                 // the next launch is followed by a refill
    return NGW_OK;
}

int ngw_set_reset_prefetch_depth(ngw_handle* h, int32_t depth) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    static_assert(NGW_MAX_DEPTH == 8, "the depths accepted below");
    if (depth != 0 && depth != 1 && depth != 2 && depth != 4 && depth != 8) return fail(NGW_E_INVALID_ARG, "depth must be 0 (automatic), 1, 2, 4 or 8");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    drop_graph(h);                                   // captured launches bake the shadow pointers in
    h->depth_user = depth != 0;
    const int want = depth ? depth : 1;
    if (want != h->depth) {
        if (h->nx.episode) {
            if (int rc = alloc_nx(h, want, h->prefetch_every > 0)) return rc;   // (on failure the old set - and the old depth - stay in force)
            h->since_refill = h->prefetch_every;     // every row is stale: the next launch is followed by a refill
        } else h->depth = want;
    }
    return NGW_OK;
}

int ngw_step_kernel_info(ngw_handle* h, int32_t* map_in_place) {
    if (!h || !map_in_place) return fail(NGW_E_INVALID_ARG, "NULL argument");
    *map_in_place = (h->nostage && !h->lidar_fused) ? 1 : 0;     // (the rule launch() applies to NGW_MODE_STEP)
    return NGW_OK;
}

int ngw_get_reset_prefetch_depth(ngw_handle* h, int32_t* depth) {
    if (!h || !depth) return fail(NGW_E_INVALID_ARG, "NULL argument");
    *depth = h->depth;
    return NGW_OK;
}

int ngw_set_stream(ngw_handle* h, void* hip_stream) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    drop_graph(h);
    if (h->own_stream) { HIP_TRY(hipStreamDestroy(h->stream)); h->own_stream = false; }
    if (hip_stream) {
        h->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    return NGW_OK;
}

int ngw_stream_order(ngw_handle* h, void* other_stream, int handle_waits) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t other = static_cast<hipStream_t>(other_stream);
    if (other == h->stream) return NGW_OK;                          // one stream: already in order
    if (!h->order_ev) HIP_TRY(hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(h->order_ev, handle_waits ? other : h->stream));
    HIP_TRY(hipStreamWaitEvent(handle_waits ? h->stream : other, h->order_ev, 0));
    return NGW_OK;
}

int ngw_reset(ngw_handle* h, const uint8_t* mask_host) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    const uint8_t* m = nullptr;
    int slot = -1;
    if (mask_host) {
        // The caller's mask may be pageable and is his again when this call returns: it is copied (host to host) into one half
        // of a page-locked, GPU-addressable buffer that the reset kernel reads across PCIe (n bytes) - no copy call and no stream
        // synchronisation (which would also wait for a refill still running).  A half is reused only after the launch that
        // read it last has finished (an event per half; two halves, so this practically never waits).
        const size_t cap = ((size_t)h->n + 255) & ~(size_t)255;
        if (!h->mask_pin) {
            h->mask_pin = static_cast<uint8_t*>(ngw_host_alloc(2 * cap));
            if (!h->mask_pin) return NGW_E_HIP;
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->mask_pin_dev), h->mask_pin, 0));
            HIP_TRY(hipEventCreateWithFlags(&h->mask_ev[0], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&h->mask_ev[1], hipEventDisableTiming));
        }
        slot = h->mask_next; h->mask_next ^= 1;
        HIP_TRY(hipEventSynchronize(h->mask_ev[slot]));
        memcpy(h->mask_pin + (size_t)slot * cap, mask_host, (size_t)h->n);
        m = h->mask_pin_dev + (size_t)slot * cap;
    }
    const int rc = launch(h, NGW_MODE_RESET, 1, nullptr, m, 0, 0);
    if (slot >= 0 && !rc) HIP_TRY(hipEventRecord(h->mask_ev[slot], h->stream));
    return rc;
}

int ngw_reset_host(ngw_handle* h, const uint8_t* mask_host, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv, uint8_t* selected,
                   int32_t* step_count, uint32_t* error_flags) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)h->n, S2 = (size_t)h->proto.S2, K = (size_t)h->proto.K;
    if (h->hostres && !mask_host) {
        // Single-wavefront handle: ONE launch that ends by copying the wave's rows into the host mirror; completion polled on the
        // word the reset kernel writes when its stores are out (the refill that re-prepares the consumed episode follows on the
        // stream and is NOT waited for), results read from the mirror.
        h->step_seq = h->step_seq + 1u ? h->step_seq + 1u : 1u;
        h->launch_seq = h->step_seq;
        const int lrc = launch(h, NGW_MODE_RESET, 1, nullptr, nullptr, 0, 0);
        h->launch_seq = 0;
        if (lrc) return lrc;
        volatile uint32_t* sp = h->b.flags_host + NGW_SEQ_WORD;
        bool seen = false;
        for (uint32_t spin = 0; spin < (1u << 21); spin++) {
            if (*sp == h->step_seq) { seen = true; break; }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            __asm__ __volatile__("" ::: "memory");
#endif
        }
        if (!seen) HIP_TRY(hipStreamSynchronize(h->stream));
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const NgwMirror& m = h->mir;
        if (map) memcpy(map, m.map, n * S2);
        if (loc) memcpy(loc, m.loc, n * 8);
        if (facing) memcpy(facing, m.facing, n * 4);
        if (inv) memcpy(inv, m.inv, n * K * 4);
        if (selected) memcpy(selected, m.selected, n);
        if (step_count) memcpy(step_count, m.step_count, n * 4);
        if (error_flags) *error_flags = *h->b.flags_host;             // sticky: ngw_error_flags reads and clears it (with the device word)
        return NGW_OK;
    }
    if (int rc = ngw_reset(h, mask_host)) return rc;
    D2H(map, h->b.map, n * S2);
    D2H(loc, h->b.loc, n * 2 * sizeof(int32_t));
    D2H(facing, h->b.facing, n * sizeof(int32_t));
    D2H(inv, h->b.inv, n * K * sizeof(int32_t));
    D2H(selected, h->b.selected, n);
    D2H(step_count, h->b.step_count, n * sizeof(int32_t));
    if (error_flags) HIP_TRY(hipMemcpyAsync(error_flags, h->b.flags, sizeof(uint32_t), hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (error_flags && h->b.flags_host) *error_flags |= *h->b.flags_host;
    return NGW_OK;
}

int ngw_step(ngw_handle* h, const int32_t* actions_host) {
    if (!h || !actions_host) return fail(NGW_E_INVALID_ARG, "NULL argument");
    const int A = h->spec.n_actions;
    for (int64_t i = 0; i < h->n; i++)
        if (actions_host[i] < 0 || actions_host[i] >= A)
            return fail(NGW_E_INVALID_ACTION, "%d is not in list", (int)actions_host[i]);   // pogostick_v1_env.py:236
    HIP_TRY(hipSetDevice(h->device));
    // The caller's array may be pageable and is his again when this call returns: it goes (host to host) into one half of a
    // page-locked buffer and from there to the device by an asynchronous copy - no stream synchronisation (which would also wait
    // for a refill still running).  A half is rewritten only after the copy that read it last has finished (an event per half).
    const size_t bytes = (size_t)h->n * sizeof(int32_t), cap = (bytes + 255) & ~(size_t)255;
    if (!h->act_pin) {
        h->act_pin = static_cast<uint8_t*>(ngw_host_alloc(2 * cap));
        if (!h->act_pin) return NGW_E_HIP;
        HIP_TRY(hipEventCreateWithFlags(&h->act_ev[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->act_ev[1], hipEventDisableTiming));
    }
    const int slot = h->act_next; h->act_next ^= 1;
    HIP_TRY(hipEventSynchronize(h->act_ev[slot]));
    memcpy(h->act_pin + (size_t)slot * cap, actions_host, bytes);
    HIP_TRY(hipMemcpyAsync(h->actions_dev, h->act_pin + (size_t)slot * cap, bytes, hipMemcpyDefault, h->stream));
    HIP_TRY(hipEventRecord(h->act_ev[slot], h->stream));
    return launch(h, NGW_MODE_STEP, 1, h->actions_dev, nullptr, 0, 0);
}

int ngw_step_device(ngw_handle* h, const int32_t* actions_dev) {
    if (!h || !actions_dev) return fail(NGW_E_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    return launch(h, NGW_MODE_STEP, 1, actions_dev, nullptr, 0, 0);
}

int ngw_step_device_many(ngw_handle* h, const int32_t* actions_dev, int64_t step_stride, int32_t n_steps) {
    if (!h || !actions_dev) return fail(NGW_E_INVALID_ARG, "NULL argument");
    if (n_steps < 1) return fail(NGW_E_INVALID_ARG, "n_steps must be >= 1");
    HIP_TRY(hipSetDevice(h->device));
    for (int32_t i = 0; i < n_steps; i++)
        if (int rc = launch(h, NGW_MODE_STEP, 1, actions_dev + (int64_t)i * step_stride, nullptr, 0, 0)) return rc;
    return NGW_OK;
}

int ngw_rollout(ngw_handle* h, int32_t n_steps, uint64_t action_seed, int64_t t0) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (n_steps < 1) return fail(NGW_E_INVALID_ARG, "n_steps must be >= 1");
    HIP_TRY(hipSetDevice(h->device));
    return rollout_chunks(h, NGW_MODE_ROLLOUT, n_steps, nullptr, action_seed, t0, 0);
}

int ngw_rollout_actions(ngw_handle* h, const int32_t* actions_dev, int64_t step_stride, int32_t n_steps) {
    if (!h || !actions_dev) return fail(NGW_E_INVALID_ARG, "NULL argument");
    if (n_steps < 1) return fail(NGW_E_INVALID_ARG, "n_steps must be >= 1");
    if (step_stride < h->n) return fail(NGW_E_INVALID_ARG, "step_stride %lld is smaller than n_envs", (long long)step_stride);
    HIP_TRY(hipSetDevice(h->device));
    return rollout_chunks(h, NGW_MODE_ROLLOUT_ACT, n_steps, actions_dev, 0, 0, step_stride);
}

int ngw_set_terminal_capture(ngw_handle* h, int enable) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (enable && !h->term.map) {
        const size_t np = (size_t)h->n_pad, S2 = (size_t)h->proto.S2, K = (size_t)h->proto.K;
        NgwTerm t = {};
        int rc = dev_alloc(h, &t.map, np * S2);
        if (!rc) rc = dev_alloc(h, &t.loc, np * 2);
        if (!rc) rc = dev_alloc(h, &t.facing, np);
        if (!rc) rc = dev_alloc(h, &t.inv, np * K);
        if (rc) {
            void* const part[4] = {t.map, t.loc, t.facing, t.inv};
            for (void* q : part) if (q) dev_free(h, q);
            return rc;
        }
        h->term = t;
    }
    const NgwTerm on_device = enable ? h->term : NgwTerm{};          // null pointers switch the capture off; the buffers stay for the next switch-on
    HIP_TRY(hipMemcpyAsync(&h->dspec->term, &on_device, sizeof(NgwTerm), hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->term_on = enable != 0;
    return NGW_OK;
}

int ngw_get_terminal_obs(ngw_handle* h, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (!h->term.map) return fail(NGW_E_INVALID_ARG, "ngw_get_terminal_obs before ngw_set_terminal_capture(h, 1)");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)h->n, S2 = (size_t)h->proto.S2, K = (size_t)h->proto.K;
    D2H(map, h->term.map, n * S2);
    D2H(loc, h->term.loc, n * 2 * sizeof(int32_t));
    D2H(facing, h->term.facing, n * sizeof(int32_t));
    D2H(inv, h->term.inv, n * K * sizeof(int32_t));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NGW_OK;
}

int ngw_terminal_device_ptrs(ngw_handle* h, void** map, void** loc, void** facing, void** inv) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (!h->term.map) return fail(NGW_E_INVALID_ARG, "ngw_terminal_device_ptrs before ngw_set_terminal_capture(h, 1)");
    if (map) *map = h->term.map;
    if (loc) *loc = h->term.loc;
    if (facing) *facing = h->term.facing;
    if (inv) *inv = h->term.inv;
    return NGW_OK;
}

int ngw_rollout_outputs(ngw_handle* h, int32_t* reward_rows_dev, uint8_t* done_rows_dev, int64_t row_stride, int accumulate) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if ((reward_rows_dev || done_rows_dev) && row_stride < h->n) return fail(NGW_E_INVALID_ARG, "row_stride %lld is smaller than n_envs", (long long)row_stride);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->row_reward = reward_rows_dev; h->row_done = done_rows_dev; h->row_stride = row_stride;
    if (accumulate && !h->acc) { if (int rc = dev_alloc(h, &h->acc, (size_t)h->n_pad * 4)) return rc; }
    if (!accumulate && h->acc) { dev_free(h, h->acc); h->acc = nullptr; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NGW_OK;
}

int ngw_episode_stats(ngw_handle* h, int32_t* run_return, int32_t* run_length, int32_t* sum_return, int32_t* n_episodes, int clear) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (!h->acc) return fail(NGW_E_INVALID_ARG, "ngw_episode_stats before ngw_rollout_outputs(..., accumulate = 1)");
    HIP_TRY(hipSetDevice(h->device));
    int32_t* const dst[4] = {run_return, run_length, sum_return, n_episodes};
    for (int i = 0; i < 4; i++)
        if (dst[i]) HIP_TRY(hipMemcpyAsync(dst[i], h->acc + (size_t)i * h->n_pad, (size_t)h->n * sizeof(int32_t), hipMemcpyDefault, h->stream));
    if (clear) HIP_TRY(hipMemsetAsync(h->acc, 0, (size_t)h->n_pad * 4 * sizeof(int32_t), h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NGW_OK;
}

/* Diagnostics builds (-DNGW_STAMPS): device buffer [grid][16] uint64 the kernels write their clock stamps to; NULL = off. */
int ngw_debug_set_stamps(ngw_handle* h, void* stamps_dev) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    drop_graph(h);
    h->proto.stamps = static_cast<uint64_t*>(stamps_dev);
    return NGW_OK;
}

/* Diagnostics (not part of include/ngw.h): the refill cadence of the prepared episodes right now (0 = off); under the default
 * setting it adapts between 2 and 32 steps to how fast episodes end. */
int ngw_debug_refill_cadence(ngw_handle* h) { return h ? (h->prefetch_every > 0 ? h->cadence : 0) : -1; }

/* Diagnostics: resets that found no prepared episode (stale row) and ran the placement loop inside a step or rollout launch
 * since prepared episodes were switched on; -1 = off.  Waits for the stream. */
long long ngw_debug_slow_resets(ngw_handle* h) {
    if (!h || !h->nx.slow || h->prefetch_every <= 0) return -1;
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return -2;
    uint32_t v = 0;
    if (hipMemcpy(&v, h->nx.slow, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -2;
    return (long long)v;
}

/* Diagnostic launches (profiling only, not part of include/ngw.h): mode 8 = empty kernel, 9 = stage in/out only. */
int ngw_debug_launch(ngw_handle* h, int mode, int32_t n_launches) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    for (int i = 0; i < n_launches; i++)
        if (int rc = launch(h, mode, 1, h->actions_dev, nullptr, 0, 0)) return rc;
    return NGW_OK;
}

/* Diagnostics (bench.py's roofline.floor; not part of include/ngw.h): the launch period of an EMPTY kernel in the step kernel's launch
 * shape (same grid, 64 lanes per workgroup, the step kernel's LDS request, the same 600-byte argument block) issued back to back
 * on the handle's stream - eagerly from one host loop, or (graph != 0) as one captured hipGraph replayed once untimed and once
 * timed - measured with a HIP event pair.  What one launch per step() costs before a single instruction of the step runs. */
int ngw_debug_launch_floor(ngw_handle* h, int32_t n_launches, int graph, double* us_per_launch) {
    if (!h || !us_per_launch || n_launches < 1) return fail(NGW_E_INVALID_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    NgwLaunch a = h->nostage ? h->ns_proto : h->proto;
    a.b = h->b; a.mode = 13; a.actions = h->actions_dev;
    const size_t lds = h->nostage ? h->ns_lds : h->lds_bytes;
    const unsigned grid = (unsigned)(h->n_pad / NGW_EPB);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    int rc = NGW_OK;
    auto issue = [&]() -> int {
        for (int i = 0; i < n_launches; i++) HIP_TRY(ngw_launch(h->dspec, &a, h->map_mode, 0, grid, lds, h->stream));
        return NGW_OK;
    };
    if (graph) {
        if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) rc = fail(NGW_E_HIP, "capture failed");
        if (!rc) rc = issue();
        if (hipStreamEndCapture(h->stream, &g) != hipSuccess && !rc) rc = fail(NGW_E_HIP, "end capture failed");
        if (!rc && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) rc = fail(NGW_E_HIP, "instantiate failed");
        if (!rc && (hipGraphLaunch(ge, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)) rc = fail(NGW_E_HIP, "warm replay failed");
    } else rc = issue();                                               // (warm pass)
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(NGW_E_HIP, "sync failed");
    if (!rc) {
        (void)hipEventRecord(e0, h->stream);
        if (graph) { if (hipGraphLaunch(ge, h->stream) != hipSuccess) rc = fail(NGW_E_HIP, "replay failed"); }
        else rc = issue();
        (void)hipEventRecord(e1, h->stream);
        if (!rc && hipEventSynchronize(e1) != hipSuccess) rc = fail(NGW_E_HIP, "event sync failed");
        float ms = 0.f;
        if (!rc && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = fail(NGW_E_HIP, "elapsed failed");
        *us_per_launch = (double)ms * 1e3 / n_launches;
    }
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

int ngw_sync(ngw_handle* h) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NGW_OK;
}

// ngw_step_host: up to here the outputs go through mapped host memory (a kernel writing across PCIe sustains ~12 GB/s, the copy
// engine ~26 GB/s but costs ~15 us to get going: measured crossover 256-512 KB, tools/api_latency.py with NGW_ZC_BYTES - 16 384
// envs, 2.6 MB: 186-230 us through mapped memory, 125-140 us staged and copied)
#define NGW_ZERO_COPY_BYTES (h->zc_bytes)


int ngw_get_obs(ngw_handle* h, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)h->n, S2 = (size_t)h->proto.S2, K = (size_t)h->proto.K;
    D2H(map, h->b.map, n * S2);
    D2H(loc, h->b.loc, n * 2 * sizeof(int32_t));
    D2H(facing, h->b.facing, n * sizeof(int32_t));
    D2H(inv, h->b.inv, n * K * sizeof(int32_t));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NGW_OK;
}

int ngw_get_step_out(ngw_handle* h, int32_t* reward, uint8_t* done, uint8_t* result, uint8_t* cost_code, uint16_t* msg_code, uint16_t* msg_arg) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)h->n;
    D2H(reward, h->b.reward, n * sizeof(int32_t));
    D2H(done, h->b.done, n);
    std::vector<uint32_t> info;
    const bool want_info = result || cost_code || msg_code || msg_arg;
    if (want_info) {
        info.resize(n);
        HIP_TRY(hipMemcpyAsync(info.data(), h->b.info, n * sizeof(uint32_t), hipMemcpyDefault, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (want_info)
        for (size_t i = 0; i < n; i++) {
            const uint32_t w = info[i];
            if (result) result[i] = (uint8_t)NGW_INFO_RESULT(w);
            if (cost_code) cost_code[i] = (uint8_t)NGW_INFO_COST(w);
            if (msg_code) msg_code[i] = (uint16_t)NGW_INFO_MSG(w);
            if (msg_arg) msg_arg[i] = (uint16_t)NGW_INFO_ARG(w);
        }
    return NGW_OK;
}

int ngw_host_step_layout(ngw_handle* h, uint64_t* offsets11) {
    if (!h || !offsets11) return fail(NGW_E_INVALID_ARG, "NULL argument");
    host_step_layout(h, offsets11);
    return NGW_OK;
}

int ngw_host_mirror_invalidate(ngw_handle* h) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    h->mirror_valid = false;
    return NGW_OK;
}

int ngw_step_host(ngw_handle* h, const int32_t* actions_host, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv, int32_t* reward,
                  uint8_t* done, uint8_t* result, uint8_t* cost_code, uint16_t* msg_code, uint16_t* msg_arg, uint32_t* error_flags,
                  uint8_t* selected, int32_t* step_count) {
    if (!h || !actions_host) return fail(NGW_E_INVALID_ARG, "NULL argument");
    const int A = h->spec.n_actions;
    const size_t n = (size_t)h->n, S2 = (size_t)h->proto.S2, K = (size_t)h->proto.K;
    for (size_t i = 0; i < n; i++)
        if (actions_host[i] < 0 || actions_host[i] >= A)
            return fail(NGW_E_INVALID_ACTION, "%d is not in list", (int)actions_host[i]);   // pogostick_v1_env.py:236
    HIP_TRY(hipSetDevice(h->device));
    const bool want_info = result || cost_code || msg_code || msg_arg;
    const uint32_t* info_words = nullptr;
    // what the caller wants back: (host pointer, device source, bytes)
    struct Out { void* host; const void* dev; size_t bytes; };
    constexpr int NOUT = 10, INFO = NOUT - 1;
    const Out outs[NOUT] = {{map, h->b.map, n * S2}, {loc, h->b.loc, n * 8}, {facing, h->b.facing, n * 4}, {inv, h->b.inv, n * K * 4},
                            {reward, h->b.reward, n * 4}, {done, h->b.done, n}, {error_flags, h->b.flags, 4},
                            {selected, h->b.selected, n}, {step_count, h->b.step_count, n * 4},
                            {want_info ? (void*)h : nullptr, h->b.info, n * 4}};
    size_t total = 0;
    for (const Out& o : outs) if (o.host) total += (o.bytes + 255) & ~(size_t)255;
    if (h->hostres) {
        // Single-wavefront handle: ONE launch, no copy call, no stream synchronisation.  The action of a one-env handle travels
        // in the kernel's argument block (more envs: a page-locked array the kernel reads in place), the kernel steps the
        // state in HBM and copies the wave's rows into the host mirror, and the results are read there.
        if (!h->zc_host) {
            h->zc_host = static_cast<uint8_t*>(ngw_host_alloc(NGW_EPB * sizeof(int32_t)));
            if (!h->zc_host) return NGW_E_HIP;
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->zc_dev), h->zc_host, 0));
        }
#ifdef NGW_HOSTTRACE
        static double tA = 0, tB = 0, tC = 0, tD = 0; static int tn = 0;
        auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
        const double t0 = now();
#endif
        memcpy(h->zc_host, actions_host, n * sizeof(int32_t));
        // The kernel writes this launch's sequence number to host memory once its stores are out: polling that word costs a
        // PCIe write's latency, a stream synchronisation several microseconds (and would also wait for a refill launch that
        // follows the step on the stream).  Bounded: after ~20 ms without the word the stream is synchronised the usual way.
        h->step_seq = h->step_seq + 1u ? h->step_seq + 1u : 1u;
        h->launch_seq = h->step_seq;
        h->launch_use_action0 = n == 1; h->launch_action0 = actions_host[0];
        const int lrc = launch(h, NGW_MODE_STEP, 1, n == 1 ? h->actions_dev : reinterpret_cast<const int32_t*>(h->zc_dev), nullptr, 0, 0);
        h->launch_seq = 0; h->launch_use_action0 = false;
        if (lrc) return lrc;
#ifdef NGW_HOSTTRACE
        const double t1 = now();
#endif
        {
            volatile uint32_t* sp = h->b.flags_host + NGW_SEQ_WORD;
            bool seen = false;
            for (uint32_t spin = 0; spin < (1u << 21); spin++) {
                if (*sp == h->step_seq) { seen = true; break; }
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#else
                __asm__ __volatile__("" ::: "memory");
#endif
            }
            if (!seen) HIP_TRY(hipStreamSynchronize(h->stream));
            __atomic_thread_fence(__ATOMIC_ACQUIRE);                                // the state reads below stay behind the poll
        }
#ifdef NGW_HOSTTRACE
        const double t2 = now();
#endif
        const NgwMirror& m = h->mir;
        const void* const mirrors[INFO] = {m.map, m.loc, m.facing, m.inv, m.reward, m.done, nullptr, m.selected, m.step_count};
        for (int r = 0; r < INFO; r++)
            if (outs[r].host && r != 6) memcpy(outs[r].host, mirrors[r], outs[r].bytes);
        if (error_flags) *error_flags = *h->b.flags_host;             // sticky: ngw_error_flags reads and clears it (with the device word)
        if (want_info) info_words = m.info;
#ifdef NGW_HOSTTRACE
        const double t3 = now();
        tA += t1 - t0; tB += t2 - t1; tC += t3 - t2; (void)tD;
        if (++tn == 1000) { fprintf(stderr, "[hosttrace] launch %.2f us, wait %.2f us, copy-out %.2f us\n", tA / tn, tB / tn, tC / tn); tA = tB = tC = 0; tn = 0; }
#endif
    } else if (total > NGW_ZERO_COPY_BYTES && map && one_block(h, map, loc, facing, inv, reward, done, error_flags, selected, step_count)) {
        // Big batch whose output arrays are the sections of ONE page-locked block (ngw_host_step_layout).  The caller keeps that
        // block from call to call (VecNovelGridworld's host mirrors), so it already holds the previous step's observation:
        //   * map, inventory and selected rows change by a few bytes per env and step.  The device keeps a shadow of what the
        //     block holds; one launch compares, 16 bytes at a time, and writes only the pieces that differ - to the shadow and
        //     straight into the block across PCIe (~1 % of the 9 MB a full copy moves at 65 536 envs);
        //   * pose, reward, done, info, flags and step_count change for (nearly) every env: they are packed into a staging
        //     payload with one launch and ONE copy brings them across (26 B per env).
        // A step costs 1.7 MB + the deltas instead of 10 MB of PCIe traffic.  The first call on a block, and the first one after
        // anything else touched the state (ngw_reset, ngw_set_state, device steps, rollouts, graph replays), copies everything
        // and re-seeds the shadows; NGW_HOST_DELTA=0 makes every call do that.
        uint64_t off[11];
        host_step_layout(h, off);
        if (!h->step_stage) { if (int rc = dev_alloc(h, &h->step_stage, (size_t)off[10])) return rc; }
        const bool delta = h->host_delta && h->mirror_valid && h->mirror_block == map;
        HIP_TRY(hipMemcpyAsync(h->actions_dev, actions_host, n * sizeof(int32_t), hipMemcpyDefault, h->stream));
        if (int rc = launch(h, NGW_MODE_STEP, 1, h->actions_dev, nullptr, 0, 0)) return rc;
        const void* const srcs[10] = {h->b.map, h->b.loc, h->b.facing, h->b.inv, h->b.reward, h->b.done, h->b.info, h->b.flags, h->b.selected, h->b.step_count};
        const uint64_t nb[10] = {n * S2, n * 8, n * 4, n * K * 4, n * 4, n, n * 4, 4, n, n * 4};
        constexpr int SPARSE[3] = {0, 3, 8};                      // map, inventory, selected: the sections refreshed by deltas
        NgwPack p = {};
        for (int r = 0; r < 10; r++) {
            if (delta && (r == 0 || r == 3 || r == 8)) continue;
            p.src[p.n_regions] = static_cast<const uint8_t*>(srcs[r]); p.dst[p.n_regions] = h->step_stage + off[r]; p.nbytes[p.n_regions] = nb[r];
            p.n_regions++;
        }
        if (delta) {
            NgwDiff d = {};
            for (int k = 0; k < 3; k++) {
                const int r = SPARSE[k];
                d.cur[k] = static_cast<const uint8_t*>(srcs[r]); d.shadow[k] = h->shadow[k]; d.host[k] = h->mirror_dev + off[r]; d.nbytes[k] = nb[r];
            }
            d.n_regions = 3;
            HIP_TRY(ngw_diff_launch(&d, h->stream));
            HIP_TRY(ngw_pack_launch(&p, h->stream));
            const uint64_t d0 = off[HS_DENSE_FIRST];
            HIP_TRY(hipMemcpyAsync(reinterpret_cast<uint8_t*>(map) + d0, h->step_stage + d0, (size_t)(off[10] - d0), hipMemcpyDefault, h->stream));
        } else {
            HIP_TRY(ngw_pack_launch(&p, h->stream));
            HIP_TRY(hipMemcpyAsync(map, h->step_stage, (size_t)off[10], hipMemcpyDefault, h->stream));
            if (h->host_delta) {                                      // (re-)seed the shadows: the block mirrors the state from here on
                void* dev = nullptr;
                bool ok = hipHostGetDevicePointer(&dev, map, 0) == hipSuccess && dev;
                if (!ok) (void)hipGetLastError();                     // (a block that is not mapped into the GPU's address space: full copies)
                for (int k = 0; k < 3 && ok; k++) {
                    const int r = SPARSE[k];
                    if (!h->shadow[k]) ok = dev_alloc(h, &h->shadow[k], (size_t)((nb[r] + 255) & ~(uint64_t)255)) == NGW_OK;
                    if (ok) HIP_TRY(hipMemcpyAsync(h->shadow[k], srcs[r], (size_t)nb[r], hipMemcpyDeviceToDevice, h->stream));
                }
                h->mirror_block = ok ? map : nullptr; h->mirror_dev = static_cast<uint8_t*>(dev);
            }
        }
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->mirror_valid = h->host_delta && h->mirror_block == map;
        if (want_info) info_words = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(map) + off[6]);
    } else if (total <= NGW_ZERO_COPY_BYTES) {
        // Small batch: no copy calls at all.  The kernel reads the actions from, and a pack kernel writes every output into,
        // page-locked host memory that is mapped into the GPU's address space; one synchronisation, then plain memcpys.
        if (!h->zc_host) {
            const size_t cap = NGW_ZERO_COPY_BYTES + n * sizeof(int32_t) + 4096;
            h->zc_host = static_cast<uint8_t*>(ngw_host_alloc(cap));
            if (!h->zc_host) return NGW_E_HIP;
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->zc_dev), h->zc_host, 0));
        }
        memcpy(h->zc_host, actions_host, n * sizeof(int32_t));
        if (int rc = launch(h, NGW_MODE_STEP, 1, reinterpret_cast<const int32_t*>(h->zc_dev), nullptr, 0, 0)) return rc;
        NgwPack p = {};
        size_t off = (n * sizeof(int32_t) + 255) & ~(size_t)255, offs[NOUT] = {0};
        for (int r = 0; r < NOUT; r++)
            if (outs[r].host) {
                p.src[p.n_regions] = static_cast<const uint8_t*>(outs[r].dev);
                p.dst[p.n_regions] = h->zc_dev + off; p.nbytes[p.n_regions] = outs[r].bytes; p.n_regions++;
                offs[r] = off; off += (outs[r].bytes + 255) & ~(size_t)255;
            }
        if (p.n_regions) HIP_TRY(ngw_pack_launch(&p, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (int r = 0; r < INFO; r++) if (outs[r].host) memcpy(outs[r].host, h->zc_host + offs[r], outs[r].bytes);
        if (want_info) info_words = reinterpret_cast<const uint32_t*>(h->zc_host + offs[INFO]);
    } else {
        // actions in, launch, everything out, ONE synchronisation (ngw_step + ngw_get_obs + ngw_get_step_out take three)
        HIP_TRY(hipMemcpyAsync(h->actions_dev, actions_host, n * sizeof(int32_t), hipMemcpyDefault, h->stream));
        if (int rc = launch(h, NGW_MODE_STEP, 1, h->actions_dev, nullptr, 0, 0)) return rc;
        for (int r = 0; r < INFO; r++) D2H(outs[r].host, outs[r].dev, outs[r].bytes);
        if (want_info) {
            if (!h->info_host) {
                h->info_host = static_cast<uint32_t*>(ngw_host_alloc(n * sizeof(uint32_t)));
                if (!h->info_host) return NGW_E_HIP;
            }
            HIP_TRY(hipMemcpyAsync(h->info_host, h->b.info, n * sizeof(uint32_t), hipMemcpyDefault, h->stream));
            info_words = h->info_host;
        }
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (want_info)
        for (size_t i = 0; i < n; i++) {
            const uint32_t w = info_words[i];
            if (result) result[i] = (uint8_t)NGW_INFO_RESULT(w);
            if (cost_code) cost_code[i] = (uint8_t)NGW_INFO_COST(w);
            if (msg_code) msg_code[i] = (uint16_t)NGW_INFO_MSG(w);
            if (msg_arg) msg_arg[i] = (uint16_t)NGW_INFO_ARG(w);
        }
    return NGW_OK;
}

/* ---- the host step in its narrow wire format (big batches)
 * Block sections (index: 0 map int8 [n][S*S], 1 inventory int32 [n][K], 2 pose uint32 [n] = r | c << 8 | facing << 16 | selected << 24,
 * 3 reward int16 [n], 4 done uint8 [n], 5 info uint32 [n] (NGW_INFO_*), 6 error flags uint32), each padded to 256 bytes; offsets7[7] =
 * the block's size.  Sections 0-1 are refreshed by deltas (only the 16-byte pieces a step changed cross PCIe), 2-6 are dense and
 * come back with ONE copy: 11 B per env against the 26 B of the int32 SoA arrays of ngw_step_host. */
namespace {
void host_step_layout_packed(const ngw_handle* h, uint64_t off[8]) {
    const uint64_t n = (uint64_t)h->n, S2 = (uint64_t)h->proto.S2, K = (uint64_t)h->proto.K;
    const uint64_t bytes[7] = {n * S2, n * K * 4, n * 4, n * 2, n, n * 4, 4};
    uint64_t o = 0;
    for (int i = 0; i < 7; i++) { off[i] = o; o += (bytes[i] + 255) & ~(uint64_t)255; }
    off[7] = o;
}
}  // namespace

int ngw_host_step_layout_packed(ngw_handle* h, uint64_t* offsets8) {
    if (!h || !offsets8) return fail(NGW_E_INVALID_ARG, "NULL argument");
    host_step_layout_packed(h, offsets8);
    return NGW_OK;
}

int ngw_step_host_packed(ngw_handle* h, const int32_t* actions_host, void* block, int with_map) {
    if (!h || !actions_host || !block) return fail(NGW_E_INVALID_ARG, "NULL argument");
    const ngw_spec& sp = h->spec;
    {   // rewards travel as int16 here
        const int rw[5] = {sp.reward_step, sp.reward_done, sp.fire_reward, sp.place_reward, sp.ext_reward};
        for (int v : rw) if (v < -32768 || v > 32767) return fail(NGW_E_INVALID_ARG, "a reward of %d does not fit the narrow wire format (int16): use ngw_step_host", v);
        for (int i = 0; i < sp.n_items; i++) if (sp.break_reward[i] < -32768 || sp.break_reward[i] > 32767) return fail(NGW_E_INVALID_ARG, "break_reward does not fit int16: use ngw_step_host");
    }
    const size_t n = (size_t)h->n, S2 = (size_t)h->proto.S2, K = (size_t)h->proto.K;
    if (h->proto.S > 255) return fail(NGW_E_INVALID_ARG, "map_size beyond the pose bytes");
    HIP_TRY(hipSetDevice(h->device));
    // ---- actions: validated and narrowed to one byte per env in ONE pass, into a page-locked, GPU-addressable buffer (two halves, an
    //      event per half as in ngw_step) that the step kernel reads in place: no copy call for 64 KB.  (The buffer is 4 n bytes long:
    //      the kernel's int32 load of the same lanes must stay in bounds.)
    const int A = sp.n_actions;
    const size_t cap = (n * sizeof(int32_t) + 255) & ~(size_t)255;
    if (!h->act_pin) {
        h->act_pin = static_cast<uint8_t*>(ngw_host_alloc(2 * cap));
        if (!h->act_pin) return NGW_E_HIP;
        HIP_TRY(hipEventCreateWithFlags(&h->act_ev[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->act_ev[1], hipEventDisableTiming));
    }
    if (!h->act_pin_dev) {
        void* d = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&d, h->act_pin, 0));
        h->act_pin_dev = static_cast<uint8_t*>(d);
    }
    const int slot = h->act_next; h->act_next ^= 1;
#ifdef NGW_HOSTTRACE
    static double pA = 0, pB = 0, pC = 0, pD = 0; static int pn = 0;
    auto pnow = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
    const double p0 = pnow();
#endif
    HIP_TRY(hipEventSynchronize(h->act_ev[slot]));
    uint8_t* const a8 = h->act_pin + (size_t)slot * cap;
    uint32_t bad = 0;
    for (size_t i = 0; i < n; i++) {                                  // (branch-free: vectorises)
        const uint32_t a = (uint32_t)actions_host[i];
        bad |= a >= (uint32_t)A ? 1u : 0u;
        a8[i] = (uint8_t)a;
    }
    if (bad) {
        h->act_next ^= 1;
        for (size_t i = 0; i < n; i++)
            if (actions_host[i] < 0 || actions_host[i] >= A) return fail(NGW_E_INVALID_ACTION, "%d is not in list", (int)actions_host[i]);   // pogostick_v1_env.py:236
    }
    uint64_t off[8];
    host_step_layout_packed(h, off);
    if (!h->wire_stage) { if (int rc = dev_alloc(h, &h->wire_stage, (size_t)(off[7] - off[2]))) return rc; }
#ifdef NGW_HOSTTRACE
    const double p1 = pnow();                                          // actions validated and narrowed
#endif
    const bool delta = h->host_delta && h->mirror_valid && h->mirror_block == block;
    h->launch_use_action0 = false; h->launch_act_u8 = true;
    const int lrc = launch(h, NGW_MODE_STEP, 1, reinterpret_cast<const int32_t*>(h->act_pin_dev + (size_t)slot * cap), nullptr, 0, 0);
    h->launch_act_u8 = false;
    if (lrc) return lrc;
    HIP_TRY(hipEventRecord(h->act_ev[slot], h->stream));
    uint8_t* const blk = static_cast<uint8_t*>(block);
    const void* const srcs[2] = {h->b.map, h->b.inv};
    const uint64_t nb[2] = {n * S2, n * K * 4};
    NgwDiff d = {};
    const bool merged = delta && h->wire_direct && h->wire_merge;     // steady state: delta refresh + narrowing in one launch (below)
    if (delta) {
        int k = 0;
        for (int r = with_map ? 0 : 1; r < 2; r++, k++) {
            d.cur[k] = static_cast<const uint8_t*>(srcs[r]); d.shadow[k] = h->shadow[r]; d.host[k] = h->mirror_dev + off[r]; d.nbytes[k] = nb[r];
        }
        d.n_regions = k;
        if (!merged) HIP_TRY(ngw_diff_launch(&d, h->stream));
    } else {
        for (int r = 0; r < 2; r++) HIP_TRY(hipMemcpyAsync(blk + off[r], srcs[r], (size_t)nb[r], hipMemcpyDefault, h->stream));
        if (h->host_delta) {                                          // (re-)seed the shadows: the block mirrors the state from here on
            void* dev = nullptr;
            bool ok = hipHostGetDevicePointer(&dev, block, 0) == hipSuccess && dev;
            if (!ok) (void)hipGetLastError();                         // (a block that is not mapped into the GPU's address space: full copies)
            for (int r = 0; r < 2 && ok; r++) {
                if (!h->shadow[r]) ok = dev_alloc(h, &h->shadow[r], (size_t)((nb[r] + 255) & ~(uint64_t)255)) == NGW_OK;
                if (ok) HIP_TRY(hipMemcpyAsync(h->shadow[r], srcs[r], (size_t)nb[r], hipMemcpyDeviceToDevice, h->stream));
            }
            h->mirror_block = ok ? block : nullptr; h->mirror_dev = static_cast<uint8_t*>(dev);
        }
    }
    NgwWire w = {};
    w.loc = h->b.loc; w.facing = h->b.facing; w.selected = h->b.selected; w.reward = h->b.reward; w.done = h->b.done; w.info = h->b.info; w.flags = h->b.flags;
    // In delta mode the block is mapped into the GPU's address space (mirror_dev): the narrowing kernel stores pose / reward / done /
    // info straight into it across PCIe, like the delta kernel before it - no staging, no copy operation behind the kernels.
    const bool direct = h->wire_direct && delta;
    uint8_t* const st = direct ? h->mirror_dev : h->wire_stage - off[2];   // (staging holds sections 2 .. 6 at their block offsets)
    w.pose = reinterpret_cast<uint32_t*>(st + off[2]); w.reward16 = reinterpret_cast<int16_t*>(st + off[3]); w.done8 = st + off[4];
    w.info32 = reinterpret_cast<uint32_t*>(st + off[5]); w.flags_out = reinterpret_cast<uint32_t*>(st + off[6]);
    w.n = (int64_t)n;
    if (merged) HIP_TRY(ngw_diff_wire_launch(&d, &w, h->stream));
    else HIP_TRY(ngw_wire_launch(&w, h->stream));
    if (!direct) HIP_TRY(hipMemcpyAsync(blk + off[2], h->wire_stage, (size_t)(off[7] - off[2]), hipMemcpyDefault, h->stream));
#ifdef NGW_HOSTTRACE
    const double p2 = pnow();                                          // everything enqueued
#endif
    HIP_TRY(hipStreamSynchronize(h->stream));
#ifdef NGW_HOSTTRACE
    {
        const double p3 = pnow();                                      // the device is done and the block is written
        pA += p1 - p0; pB += p2 - p1; pC += p3 - p2; (void)pD;
        if (++pn == 200) {
            fprintf(stderr, "[hosttrace packed, %zu envs] narrow actions %.2f us, enqueue %.2f us, wait for the device %.2f us (merged %d, direct %d)\n", n, pA / pn, pB / pn, pC / pn,
                    (int)merged, (int)direct);
            pA = pB = pC = 0; pn = 0;
        }
    }
#endif
    // (a delta step that skipped the map leaves the map's shadow describing what the block holds: the next step that wants the map
    //  brings every change since across)
    h->mirror_valid = h->host_delta && h->mirror_block == block;
    return NGW_OK;
}

/* Payload of the multi-GPU observation gather: the seven SoA arrays back to back, each section padded to 16 bytes. */
namespace {
struct PackSection { const void* dev; uint64_t bytes; };
int pack_sections(const ngw_handle* h, PackSection sec[7], uint64_t offs[8]) {
    const uint64_t n = (uint64_t)h->n, S2 = (uint64_t)h->proto.S2, K = (uint64_t)h->proto.K;
    const PackSection s[7] = {{h->b.map, n * S2}, {h->b.loc, n * 8}, {h->b.facing, n * 4}, {h->b.inv, n * K * 4},
                              {h->b.reward, n * 4}, {h->b.done, n}, {h->b.info, n * 4}};
    uint64_t off = 0;
    for (int i = 0; i < 7; i++) { sec[i] = s[i]; offs[i] = off; off += (s[i].bytes + 15) & ~(uint64_t)15; }
    offs[7] = off;
    return NGW_OK;
}
}  // namespace

int ngw_pack_layout(ngw_handle* h, uint64_t* offsets8) {
    if (!h || !offsets8) return fail(NGW_E_INVALID_ARG, "NULL argument");
    PackSection sec[7];
    return pack_sections(h, sec, offsets8);
}

int ngw_pack_obs(ngw_handle* h, void* payload_dev) {
    if (!h || !payload_dev) return fail(NGW_E_INVALID_ARG, "NULL argument");
    if ((uintptr_t)payload_dev & 15u) return fail(NGW_E_INVALID_ARG, "payload must be 16-byte aligned");
    HIP_TRY(hipSetDevice(h->device));
    PackSection sec[7]; uint64_t offs[8];
    pack_sections(h, sec, offs);
    NgwPack p = {};
    for (int i = 0; i < 7; i++) {
        p.src[i] = static_cast<const uint8_t*>(sec[i].dev); p.dst[i] = static_cast<uint8_t*>(payload_dev) + offs[i]; p.nbytes[i] = sec[i].bytes;
    }
    p.n_regions = 7;
    HIP_TRY(ngw_pack_launch(&p, h->stream));
    return NGW_OK;
}

int ngw_unpack_obs(ngw_handle* h, const void* payloads_dev, int32_t world, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv,
                   int32_t* reward, uint8_t* done, uint32_t* info) {
    if (!h || !payloads_dev) return fail(NGW_E_INVALID_ARG, "NULL argument");
    if (world < 1) return fail(NGW_E_INVALID_ARG, "world %d must be >= 1", world);
    HIP_TRY(hipSetDevice(h->device));
    PackSection sec[7]; uint64_t offs[8];
    pack_sections(h, sec, offs);
    uint8_t* const dsts[7] = {reinterpret_cast<uint8_t*>(map), reinterpret_cast<uint8_t*>(loc), reinterpret_cast<uint8_t*>(facing),
                              reinterpret_cast<uint8_t*>(inv), reinterpret_cast<uint8_t*>(reward), done, reinterpret_cast<uint8_t*>(info)};
    // one launch moves up to NGW_PACK_MAX regions = 9 ranks' payloads; bigger worlds (several nodes) take more launches
    NgwPack p = {};
    for (int r = 0; r < world; r++) {
        for (int i = 0; i < 7; i++) {
            if (!dsts[i]) continue;
            p.src[p.n_regions] = static_cast<const uint8_t*>(payloads_dev) + (uint64_t)r * offs[7] + offs[i];
            p.dst[p.n_regions] = dsts[i] + (uint64_t)r * sec[i].bytes; p.nbytes[p.n_regions] = sec[i].bytes; p.n_regions++;
        }
        if (p.n_regions + 7 > NGW_PACK_MAX || r == world - 1) {
            HIP_TRY(ngw_pack_launch(&p, h->stream));
            p = NgwPack{};
        }
    }
    return NGW_OK;
}

int ngw_get_state(ngw_handle* h, int64_t first, int64_t count, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv,
                  int32_t* selected, int32_t* step_count, uint32_t* episode) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (first < 0 || count < 0 || first + count > h->n) return fail(NGW_E_INVALID_ARG, "env range [%lld, +%lld) out of bounds", (long long)first, (long long)count);
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)count, f = (size_t)first, S2 = (size_t)h->proto.S2, K = (size_t)h->proto.K;
    D2H(map, h->b.map + f * S2, n * S2);
    D2H(loc, h->b.loc + f * 2, n * 2 * sizeof(int32_t));
    D2H(facing, h->b.facing + f, n * sizeof(int32_t));
    D2H(inv, h->b.inv + f * K, n * K * sizeof(int32_t));
    D2H(step_count, h->b.step_count + f, n * sizeof(int32_t));
    D2H(episode, h->b.episode + f, n * sizeof(uint32_t));
    std::vector<uint8_t> sel;
    if (selected) {
        sel.resize(n);
        HIP_TRY(hipMemcpyAsync(sel.data(), h->b.selected + f, n, hipMemcpyDefault, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (selected)
        for (size_t i = 0; i < n; i++) selected[i] = sel[i];
    return NGW_OK;
}

#define H2D(dst, src, bytes)                                                                             \
    do {                                                                                                 \
        if (src) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDefault, h->stream));        \
    } while (0)

int ngw_set_state(ngw_handle* h, int64_t first, int64_t count, const int8_t* map, const int32_t* loc, const int32_t* facing,
                  const int32_t* inv, const int32_t* selected, const int32_t* step_count, const uint32_t* episode) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (first < 0 || count < 0 || first + count > h->n) return fail(NGW_E_INVALID_ARG, "env range [%lld, +%lld) out of bounds", (long long)first, (long long)count);
    const int S = h->proto.S, K = h->proto.K;
    const size_t n = (size_t)count, f = (size_t)first, S2 = (size_t)h->proto.S2;
    // the kernel indexes LUTs and the map with these values: reject anything that could go out of bounds
    if (map)
        for (size_t i = 0; i < n * S2; i++)
            if (map[i] < 0 || map[i] >= K) return fail(NGW_E_INVALID_ARG, "map cell value %d outside [0, %d)", (int)map[i], K);
    if (loc)
        for (size_t i = 0; i < n; i++)
            if (loc[2 * i] < 1 || loc[2 * i] > S - 2 || loc[2 * i + 1] < 1 || loc[2 * i + 1] > S - 2)
                return fail(NGW_E_INVALID_ARG, "agent_location (%d, %d) outside the walled interior", loc[2 * i], loc[2 * i + 1]);
    if (facing)
        for (size_t i = 0; i < n; i++)
            if (facing[i] < 0 || facing[i] > 3) return fail(NGW_E_INVALID_ARG, "agent_facing_id %d outside [0, 3]", facing[i]);
    if (inv)
        for (size_t i = 0; i < n * (size_t)K; i++)
            if (inv[i] < 0) return fail(NGW_E_INVALID_ARG, "inventory quantity %d is negative", inv[i]);
    std::vector<uint8_t> sel;
    if (selected) {
        sel.resize(n);
        for (size_t i = 0; i < n; i++) {
            if (selected[i] < 0 || selected[i] >= K) return fail(NGW_E_INVALID_ARG, "selected item %d outside [0, %d)", selected[i], K);
            sel[i] = (uint8_t)selected[i];
        }
    }
    HIP_TRY(hipSetDevice(h->device));
    h->mirror_valid = false;
    H2D(h->b.map + f * S2, map, n * S2);
    H2D(h->b.loc + f * 2, loc, n * 2 * sizeof(int32_t));
    H2D(h->b.facing + f, facing, n * sizeof(int32_t));
    H2D(h->b.inv + f * K, inv, n * K * sizeof(int32_t));
    H2D(h->b.step_count + f, step_count, n * sizeof(int32_t));
    H2D(h->b.episode + f, episode, n * sizeof(uint32_t));
    if (selected) HIP_TRY(hipMemcpyAsync(h->b.selected + f, sel.data(), n, hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NGW_OK;
}

void* ngw_host_alloc(uint64_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); fail(NGW_E_HIP, "hipHostMalloc(%llu) failed", (unsigned long long)bytes); return nullptr; }
    memset(p, 0, bytes);
    return p;
}

int ngw_host_free(void* p) {
    if (p) HIP_TRY(hipHostFree(p));
    return NGW_OK;
}

int ngw_obs_device_ptrs(ngw_handle* h, void** map, void** loc, void** facing, void** inv) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (map) *map = h->b.map;
    if (loc) *loc = h->b.loc;
    if (facing) *facing = h->b.facing;
    if (inv) *inv = h->b.inv;
    return NGW_OK;
}

int ngw_out_device_ptrs(ngw_handle* h, void** reward, void** done, void** info) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (reward) *reward = h->b.reward;
    if (done) *done = h->b.done;
    if (info) *info = h->b.info;
    return NGW_OK;
}

int ngw_error_flags(ngw_handle* h, uint32_t* flags) {
    if (!h || !flags) return fail(NGW_E_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(flags, h->b.flags, sizeof(uint32_t), hipMemcpyDefault, h->stream));
    HIP_TRY(hipMemsetAsync(h->b.flags, 0, sizeof(uint32_t), h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->b.flags_host) { *flags |= *h->b.flags_host; *h->b.flags_host = 0; }
    return NGW_OK;
}

namespace {
// LDS of the stand-alone lidar launch: item tables | per-lane ray table (only when the rays are not world-frame) | observation
// tile | guard | maps | guard | inventory rows.  After its hit a ray's remaining cells may fall outside the lane's own map: the
// guards keep those (ignored) reads inside the allocation.
int layout_lidar(ngw_handle* h) {
    NgwLaunch& q = h->lidar_proto;
    q = h->proto;
    q.b = h->b;
    lidar_format(h, q);
    const uint32_t guard = (uint32_t)((h->lidar_range * (q.S + 1) + 15) / 16 * 4);        // dwords
    uint32_t off = 0;
    q.off_litem = off; off += 2 * NGW_MAX_ITEMS / 4;
    off = (off + 3u) & ~3u;
    q.off_ltab = off; if (!h->lidar_world) off += 4 * NGW_LIDAR_MAX_BEAMS * NGW_LIDAR_MAX_RANGE * 2 / 4;
    q.off_ltile = off; off += (uint32_t)(NGW_EPB * q.l_rb / 4) + NGW_EPB / 4;             // + one dump byte per lane
    off = ((off + 3u) & ~3u) + guard;
    q.off_map = off; off += (uint32_t)(NGW_EPB * q.MS / 4) + guard;
    off = (off + 3u) & ~3u;
    q.off_inv = off; off += (uint32_t)(q.KP * NGW_EPB);
    if ((size_t)off * 4 > 160 * 1024) return fail(NGW_E_INVALID_ARG, "lidar observation of %d values needs %zu B of LDS (> 160 KiB)", h->lidar_len, (size_t)off * 4);
    h->lidar_lds = (size_t)off * 4;
    return NGW_OK;
}
}  // namespace

int ngw_lidar_configure(ngw_handle* h, const ngw_lidar_cfg* cfg) {
    if (!h || !cfg) return fail(NGW_E_INVALID_ARG, "NULL argument");
    const int K = h->proto.K;
    if (cfg->num_beams < 1 || cfg->num_beams > NGW_LIDAR_MAX_BEAMS || cfg->max_range < 1 || cfg->max_range > NGW_LIDAR_MAX_RANGE ||
        cfg->n_chan < 1 || cfg->n_chan > NGW_MAX_ITEMS || cfg->n_inv < 0 || cfg->n_inv > NGW_MAX_ITEMS)
        return fail(NGW_E_INVALID_ARG, "lidar configuration out of range");
    for (int i = 0; i < NGW_MAX_ITEMS; i++)
        if (cfg->chan_of_item[i] > cfg->n_chan || (i < cfg->n_inv && cfg->inv_item[i] >= K))
            return fail(NGW_E_INVALID_ARG, "lidar item table out of range");
    for (int f = 0; f < 4; f++)                           /* ray offsets beyond the guard band would read outside the wave's LDS */
        for (int b = 0; b < cfg->num_beams; b++)
            for (int k = 0; k < cfg->max_range; k++)
                if (cfg->dr[f][b][k] > cfg->max_range || cfg->dr[f][b][k] < -cfg->max_range || cfg->dc[f][b][k] > cfg->max_range ||
                    cfg->dc[f][b][k] < -cfg->max_range)
                    return fail(NGW_E_INVALID_ARG, "lidar ray offset (%d, %d) beyond max_range %d", cfg->dr[f][b][k], cfg->dc[f][b][k], cfg->max_range);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    drop_graph(h);
    const int L = cfg->num_beams * cfg->n_chan + cfg->n_inv;
    if (!h->lidar_cfg) { if (int rc = dev_alloc(h, &h->lidar_cfg, 1)) return rc; }
    if (L > h->lidar_cap) {                              /* grow only: a smaller observation reuses the buffer (int32 rows are the widest format) */
        if (h->lidar_out) dev_free(h, h->lidar_out);
        h->lidar_out = nullptr; h->lidar_cap = 0;
        if (int rc = dev_alloc(h, &h->lidar_out, (size_t)h->n_pad * L)) return rc;
        h->lidar_cap = L;
    }
    int world = 0;
    {
        static thread_local NgwLidarDev hd;
        memset(&hd, 0, sizeof(hd));
        const int S = h->proto.S, B = cfg->num_beams, R = cfg->max_range;
        for (int f = 0; f < 4; f++)
            for (int b = 0; b < B; b++)
                for (int k = 0; k < NGW_LIDAR_MAX_RANGE; k++) {
                    const int kk = k < R ? k : R - 1;                                 // pad: repeats the last in-range cell
                    hd.off[f][b][k] = (int16_t)(cfg->dr[f][b][kk] * S + cfg->dc[f][b][kk]);
                }
        // World-frame form (NgwLidarDev::woff): with B a multiple of 4, ray b of facing f should be world ray (u_f + b - B / 2) mod B,
        // u_f = B / 2, 0, 3 B / 4, B / 4 for NORTH, SOUTH, WEST, EAST.  The world table is read off facing SOUTH (u = 0) and every
        // entry of the other facings is compared with it: only an exact match switches the uniform-offset march on.
        if (B % 4 == 0) {
            const int uf[4] = {B / 2, 0, 3 * B / 4, B / 4};
            world = 1;
            for (int w = 0; w < B; w++)
                for (int k = 0; k < NGW_LIDAR_MAX_RANGE; k++) hd.woff[w][k] = hd.off[1][(w + B / 2) % B][k];
            for (int f = 0; f < 4 && world; f++)
                for (int b = 0; b < B && world; b++) {
                    const int w = ((uf[f] + b - B / 2) % B + B) % B;
                    for (int k = 0; k < R; k++)
                        if (cfg->dr[f][b][k] != cfg->dr[1][(w + B / 2) % B][k] || cfg->dc[f][b][k] != cfg->dc[1][(w + B / 2) % B][k]) { world = 0; break; }
                }
            if (const char* v = getenv("NGW_LIDAR_WORLD")) if (atoi(v) == 0) world = 0;     // A/B: the per-lane table march
        }
        if (world && B == 8 && S == NGW_LIDAR_CONST_S) {               // the reference's default rays on its default map: compile-time offsets?
            bool same = R == 11;                                       // int(sqrt(2 * (S - 2)^2)) for S = 10: what the kernels instantiate
            for (int w = 0; w < 8 && same; w++)
                for (int k = 1; k <= R; k++)
                    if (hd.woff[w][k - 1] != ngw_lidar8_dr(w, k) * S + ngw_lidar8_dc(w, k)) { same = false; break; }
            if (same) world = 2;
            if (const char* v = getenv("NGW_LIDAR_WORLD")) if (atoi(v) == 1) world = 1;   // A/B: the table-driven world march
        }
        hd.world = world;
        memcpy(hd.chan_of_item, cfg->chan_of_item, NGW_MAX_ITEMS);
        memcpy(hd.inv_item, cfg->inv_item, NGW_MAX_ITEMS);
        hd.num_beams = B; hd.max_range = R; hd.n_chan = cfg->n_chan; hd.n_inv = cfg->n_inv;
        HIP_TRY(hipMemcpyAsync(h->lidar_cfg, &hd, sizeof(hd), hipMemcpyDefault, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    h->lidar_len = L; h->lidar_world = world;
    h->lidar_range = cfg->max_range; h->lidar_beams = cfg->num_beams; h->lidar_chan = cfg->n_chan; h->lidar_ninv = cfg->n_inv;
    if (h->lidar_fused) { if (int rc = layout_lds(h)) { h->lidar_fused = 0; layout_lds(h); upload_reset_u(h); return rc; } }
    if (int rc = upload_reset_u(h)) return rc;
    h->lidar_lds = 0;
    if (h->general_ok) { if (int rc = layout_lidar(h)) return rc; }
    return NGW_OK;
}

int ngw_lidar_set_output(ngw_handle* h, int bits) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (bits != 8 && bits != 16 && bits != 32) return fail(NGW_E_INVALID_ARG, "lidar output format must be 32 (int32), 16 (int16) or 8 (packed: uint8 beams + int16 inventory)");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    drop_graph(h);                                   // captured launches bake the format in
    h->lidar_bits = bits;
    if (h->lidar_len) {
        if (h->lidar_fused) { if (int rc = layout_lds(h)) return rc; }
        if (int rc = upload_reset_u(h)) return rc;
        if (h->general_ok) { if (int rc = layout_lidar(h)) return rc; }
    }
    return NGW_OK;
}

int ngw_lidar_row_layout(ngw_handle* h, int32_t* row_bytes, int32_t* beam_bytes, int32_t* inv_offset, int32_t* inv_bytes) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (!h->lidar_len) return fail(NGW_E_INVALID_ARG, "ngw_lidar_row_layout before ngw_lidar_configure");
    NgwLaunch q{};
    lidar_format(h, q);
    if (row_bytes) *row_bytes = q.l_rb;
    if (beam_bytes) *beam_bytes = q.l_fmt == NGW_LFMT_I32 ? 4 : (q.l_fmt == NGW_LFMT_I16 ? 2 : 1);
    if (inv_offset) *inv_offset = q.l_invoff;
    if (inv_bytes) *inv_bytes = q.l_fmt == NGW_LFMT_I32 ? 4 : 2;
    return NGW_OK;
}

int ngw_lidar_fuse(ngw_handle* h, int enable) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (enable && !h->lidar_len) return fail(NGW_E_INVALID_ARG, "ngw_lidar_fuse before ngw_lidar_configure");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    drop_graph(h);                                   // captured launches bake the LDS layout in
    if (enable && !h->general_ok) return fail(NGW_E_INVALID_ARG, "map_size %d: the fused lidar epilogue keeps a wavefront's 64 maps in LDS (> 160 KiB)", h->proto.S);
    const int before = h->lidar_fused;
    h->lidar_fused = enable ? 1 : 0;
    if (int rc = layout_lds(h)) { h->lidar_fused = before; layout_lds(h); upload_reset_u(h); return rc; }
    if (int rc = upload_reset_u(h)) return rc;
    return NGW_OK;
}

int ngw_lidar(ngw_handle* h) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (!h->lidar_len) return fail(NGW_E_INVALID_ARG, "ngw_lidar before ngw_lidar_configure");
    if (!h->lidar_lds) return fail(NGW_E_INVALID_ARG, "map_size %d: the lidar observation keeps a wavefront's 64 maps in LDS (> 160 KiB)", h->proto.S);
    HIP_TRY(hipSetDevice(h->device));
    NgwLaunch a = h->lidar_proto;
    a.b = h->b;
    HIP_TRY(ngw_lidar_launch(&a, h->map_mode, (unsigned)(h->n_pad / NGW_EPB), h->lidar_lds, h->stream));
    return NGW_OK;
}

int ngw_get_lidar(ngw_handle* h, void* out_host) {
    if (!h || !out_host) return fail(NGW_E_INVALID_ARG, "NULL argument");
    if (!h->lidar_len) return fail(NGW_E_INVALID_ARG, "ngw_get_lidar before ngw_lidar_configure");
    HIP_TRY(hipSetDevice(h->device));
    NgwLaunch q{};
    lidar_format(h, q);
    HIP_TRY(hipMemcpyAsync(out_host, h->lidar_out, (size_t)h->n * (size_t)q.l_rb, hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NGW_OK;
}

int ngw_lidar_device_ptr(ngw_handle* h, void** out) {
    if (!h || !out) return fail(NGW_E_INVALID_ARG, "NULL argument");
    *out = h->lidar_out;
    return NGW_OK;
}

int ngw_agent_view(ngw_handle* h, int view_size) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (view_size < 1 || view_size > 127) return fail(NGW_E_INVALID_ARG, "view_size must be in 1..127");   // :99 'Increase the agent_view_size'
    const size_t W = 2 * (size_t)view_size + 1, bytes = (size_t)h->n * W * W;
    if (bytes + 4 > 0xffffffffull) return fail(NGW_E_INVALID_ARG, "agent view of %zu B exceeds the 4 GiB index range", bytes);
    HIP_TRY(hipSetDevice(h->device));
    if (view_size != h->view_size) {
        h->view_size = 0;
        if (bytes > h->view_cap) {
            if (h->view_out) dev_free(h, h->view_out);
            h->view_out = nullptr; h->view_cap = 0;
            if (int rc = dev_alloc(h, &h->view_out, (bytes + 3) / 4 * 4)) return rc;
            h->view_cap = bytes;
        }
        h->view_size = view_size;
    }
    HIP_TRY(ngw_agent_view_launch(h->b.map, h->b.loc, reinterpret_cast<uint32_t*>(h->view_out), (uint32_t)((bytes + 3) / 4),
                                  h->proto.S, view_size, h->stream));
    return NGW_OK;
}

int ngw_get_agent_view(ngw_handle* h, int8_t* out_host) {
    if (!h || !out_host) return fail(NGW_E_INVALID_ARG, "NULL argument");
    if (!h->view_size) return fail(NGW_E_INVALID_ARG, "ngw_get_agent_view before ngw_agent_view");
    HIP_TRY(hipSetDevice(h->device));
    const size_t W = 2 * (size_t)h->view_size + 1;
    HIP_TRY(hipMemcpyAsync(out_host, h->view_out, (size_t)h->n * W * W, hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return NGW_OK;
}

int ngw_agent_view_device_ptr(ngw_handle* h, void** out) {
    if (!h || !out) return fail(NGW_E_INVALID_ARG, "NULL argument");
    if (!h->view_size) return fail(NGW_E_INVALID_ARG, "ngw_agent_view_device_ptr before ngw_agent_view");
    *out = h->view_out;
    return NGW_OK;
}

int ngw_timing_begin(ngw_handle* h) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->ev0) { HIP_TRY(hipEventCreate(&h->ev0)); HIP_TRY(hipEventCreate(&h->ev1)); }
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    h->ev_marked = false;
    return NGW_OK;
}

int ngw_timing_mark(ngw_handle* h) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (!h->ev0) return fail(NGW_E_INVALID_ARG, "ngw_timing_mark without ngw_timing_begin");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->ev_marked = true;
    return NGW_OK;
}

int ngw_timing_end(ngw_handle* h, double* elapsed_ms) {
    if (!h || !elapsed_ms) return fail(NGW_E_INVALID_ARG, "NULL argument");
    if (!h->ev0) return fail(NGW_E_INVALID_ARG, "ngw_timing_end without ngw_timing_begin");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->ev_marked) HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->ev_marked = false;
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *elapsed_ms = ms;
    return NGW_OK;
}

static int capture_graph(ngw_handle* h, const int32_t* actions_dev, int64_t step_stride, int32_t n_steps) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    drop_graph(h);
    h->since_refill = 0;                              // the captured refill cadence starts from a known phase
    HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    h->capturing = true;                              // (the depth and cadence the handle has adapted to so far are the ones captured)
    int rc = NGW_OK;
    for (int i = 0; i < n_steps && !rc; i++) rc = launch(h, NGW_MODE_STEP, 1, actions_dev + (int64_t)i * step_stride, nullptr, 0, 0);
    // every replay must leave the refill cadence where it found it: a graph shorter than (or not a multiple of) the cadence
    // ends with one more refill, otherwise a replayed graph would never re-prepare the episodes its steps consume
    if (!rc && h->prefetch_every > 0 && h->since_refill > 0) {
        h->since_refill = h->prefetch_every;
        rc = launch_refill(h);
    }
    h->capturing = false;
    hipError_t e = hipStreamEndCapture(h->stream, &h->graph);
    if (rc) { drop_graph(h); return rc; }
    if (e != hipSuccess) { drop_graph(h); return fail(NGW_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e)); }
    e = hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { drop_graph(h); return fail(NGW_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
    (void)hipGraphUpload(h->graph_exec, h->stream);   // pre-stage the graph so the first replay does not pay for it
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->graph_steps = n_steps; h->graph_actions = actions_dev; h->graph_stride = step_stride;
    h->adapted = false;
    return NGW_OK;
}

int ngw_graph_build(ngw_handle* h, const int32_t* actions_dev, int64_t step_stride, int32_t n_steps) {
    if (!h || !actions_dev) return fail(NGW_E_INVALID_ARG, "NULL argument");
    if (n_steps < 1) return fail(NGW_E_INVALID_ARG, "n_steps must be >= 1");
    HIP_TRY(hipSetDevice(h->device));
    return capture_graph(h, actions_dev, step_stride, n_steps);
}

int ngw_graph_launch(ngw_handle* h, int32_t reps) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (!h->graph_exec) return fail(NGW_E_INVALID_ARG, "no graph: call ngw_graph_build first");
    HIP_TRY(hipSetDevice(h->device));
    for (int i = 0; i < reps; i++) {
        // A captured graph holds the prepared-episode depth and the refill cadence it was captured with.  The refills inside it
        // keep reporting, so the host keeps adapting between replays (default setting only); when that changed something the
        // graph is captured again - a few milliseconds, a handful of times in the life of a handle.
        if (h->prefetch_every > 0) adapt_cadence(h);
        if (h->adapted) {
            const int32_t* acts = h->graph_actions; const int64_t stride = h->graph_stride; const int32_t k = h->graph_steps;
            if (int rc = capture_graph(h, acts, stride, k)) return rc;
        }
        h->mirror_valid = false;
        HIP_TRY(hipGraphLaunch(h->graph_exec, h->stream));
    }
    return NGW_OK;
}

}  // extern "C"
