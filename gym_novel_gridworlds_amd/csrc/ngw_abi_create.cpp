// ngw_abi_create.cpp - a handle's life: spec checks, device buffers, LDS carve-ups, the spec blob in HBM (see ngw_host.h for the other units).
// Host side of the C-ABI declared in include/ngw.h (HIP runtime only; no torch types).  There is NO CPU execution path: without a GPU every entry
// point that computes returns NGW_E_NO_DEVICE / NGW_E_HIP.
#include "ngw_host.h"

using namespace ngwh;

namespace {
thread_local char g_err[512] = "";
}

namespace ngwh {

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
const char* last_error() { return g_err; }

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
        if (h->spec.n_passes == 0) {                                   // (register blocks here too: a C2 reset of every env 23.5 -> 25 us, profiles/r05_ab.md)
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
        q.l_boards = 0; q.lidar_len = 0;
        uint32_t off = q.off_act + NGW_MAX_PLACE / 4;
        if (h->boards_on) {                                            // the bit-row lidar epilogue: item tables | observation tile + one dump byte per lane
            lidar_format(h, q);
            q.l_boards = 1;
            off = (off + 3u) & ~3u;
            q.off_litem = off; off += 2 * NGW_MAX_ITEMS / 4;
            off = (off + 3u) & ~3u;
            q.off_ltile = off; off += (uint32_t)(NGW_EPB * q.l_rb / 4) + NGW_EPB;   // + one dump DWORD per lane (entries that report nothing are stored there)
            off = (off + 3u) & ~3u;
            const uint32_t nr = q.BS <= 12 ? 12u : (q.BS <= 20 ? 20u : 32u);       // the kernel's register rows (ngw_part_step_boards picks the same)
            q.off_ltab = off; off += NGW_EPB * (nr + 4u);                           // the lanes' bit rows, [64][NR + 4] words: rows come back by index from here
        }
        h->ns_lds = (size_t)off * 4;
        if (getenv("NGW_DEBUG_LDS"))
            fprintf(stderr, "[ngw] LDS per wavefront, in-place step kernel: %zu B (S = %d, bit-row lidar %d: inventory rows + candidate masks + placement sequence%s)\n", h->ns_lds, h->proto.S,
                    (int)h->boards_on, h->boards_on ? " + item tables + observation tile + bit-row scratch" : "");
        HIP_TRY(hipMemcpyAsync(&h->dspec->lp_ns, &q, sizeof(q), hipMemcpyDefault, h->stream));
    }
    if (h->proto.BS) {
        // ngw_boards_kernel: maps | word tile [64][BS + 1];  ngw_lidar_boards_kernel: inventory rows | item tables | observation tile
        NgwLaunch& r = h->brd_proto;
        r = h->proto; r.b = h->b;
        r.off_map = 0;
        uint32_t off = ((uint32_t)(NGW_EPB * r.MS / 4) + 3u) & ~3u;
        r.off_ltile = off; off += (uint32_t)(NGW_EPB * (r.BS + 1));
        r.magicK = (uint32_t)((0x100000000ull + (uint32_t)r.BS - 1) / (uint32_t)r.BS);
        h->brd_lds = (size_t)off * 4;
        NgwLaunch& l = h->lb_proto;
        l = h->proto; l.b = h->b;
        l.off_inv = 0; off = (uint32_t)(l.KP * NGW_EPB);
        h->lb_lds = 0;
        if (h->lidar_len) {
            lidar_format(h, l);
            l.l_boards = 1;
            off = (off + 3u) & ~3u;
            l.off_litem = off; off += 2 * NGW_MAX_ITEMS / 4;
            off = (off + 3u) & ~3u;
            l.off_ltile = off; off += (uint32_t)(NGW_EPB * l.l_rb / 4) + NGW_EPB;
            h->lb_lds = (size_t)off * 4;
        }
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
    {   // staging tile of the composed rows: the chunk's exact image up to 512-byte rows, else [64][128 + 16] bytes; in the boards mode the
        // occupancy bit rows of the same maps are staged there afterwards ([64][BS + 1] words)
        const uint32_t tile_dw = (uint32_t)((S2 <= 512 ? (S2 * NGW_EPB + 15) / 16 * 16 : 144 * NGW_EPB) / 4), brd_dw = (uint32_t)(NGW_EPB * (h->proto.BS + 1));
        a.off_tile = off; off += tile_dw > brd_dw ? tile_dw : brd_dw;
    }
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

}  // namespace ngwh

namespace {

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

}  // namespace

extern "C" {

int ngw_abi_version(void) { return NGW_ABI_VERSION; }
int ngw_spec_size(void) { return (int)sizeof(ngw_spec); }
const char* ngw_last_error(void) { return ngwh::last_error(); }

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
    if (const char* v = getenv("NGW_HOST_WRITE_THROUGH")) h->wt_enabled = atoi(v) != 0;
    if (const char* v = getenv("NGW_ZC_BYTES")) { h->zc_bytes = (size_t)atoll(v); if (!h->zc_bytes) h->zc_bytes = 1; }
    if (const char* v = getenv("NGW_API_SLICES")) h->api_slices = atoi(v);
    if (const char* v = getenv("NGW_SOLO")) h->solo_enabled = atoi(v) != 0;
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
    h->hostres = h->n_pad == NGW_EPB;                                 // a single-wavefront handle keeps a host mirror of its rows (NgwMirror)
    int rc = NGW_OK;
    {   // the six arrays a step's prologue reads are ONE allocation: the step kernel names them by a base + 32-bit offsets that
        // travel in the preloaded head of its argument block (ngw_lean.inc); map rows first (the base)
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t o_map = 0, o_inv = up(np * S2), o_loc = o_inv + up(np * K * 4), o_fac = o_loc + up(np * 8), o_sel = o_fac + up(np * 4),
                     o_stp = o_sel + up(np), o_brd = o_stp + up(np * 4);
        // ... and, for maps up to 32 x 32, the occupancy bit rows the O(1) lidar reads (NGW_BOARD_*: S words per env, rounded up to four)
        const size_t BS = S <= NGW_BOARD_MAX_S ? (size_t)NGW_BOARD_STRIDE(S) : 0, total = o_brd + up(np * BS * 4);
        uint8_t* slab = nullptr;
        if (!rc) rc = dev_alloc(h, &slab, total);
        if (!rc) {
            h->b.map = reinterpret_cast<int8_t*>(slab + o_map); h->b.inv = reinterpret_cast<int32_t*>(slab + o_inv);
            h->b.loc = reinterpret_cast<int32_t*>(slab + o_loc); h->b.facing = reinterpret_cast<int32_t*>(slab + o_fac);
            h->b.selected = slab + o_sel; h->b.step_count = reinterpret_cast<int32_t*>(slab + o_stp);
            h->b.brd = BS ? reinterpret_cast<uint32_t*>(slab + o_brd) : nullptr;
            h->proto.BS = (int32_t)BS;
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
    if (h->solo_running) (void)solo_stop(h);
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
    for (hipEvent_t e : h->slice_ev) if (e) (void)hipEventDestroy(e);
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
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

int ngw_sync(ngw_handle* h) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    if (h->solo_running) { if (int rc = solo_stop(h)) return rc; }
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
    if (h->solo_running) { if (int rc = solo_stop(h)) return rc; }
    HIP_TRY(hipMemcpyAsync(flags, h->b.flags, sizeof(uint32_t), hipMemcpyDefault, h->stream));
    HIP_TRY(hipMemsetAsync(h->b.flags, 0, sizeof(uint32_t), h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->b.flags_host) { *flags |= *h->b.flags_host; *h->b.flags_host = 0; }
    return NGW_OK;
}

}  // extern "C"
