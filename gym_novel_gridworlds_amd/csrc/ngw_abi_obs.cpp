// ngw_abi_obs.cpp - observation wrappers computed on the device: LidarInFront (ray tables, row formats, the marches over LDS maps and the O(1)
// form on occupancy bit rows), AgentMap windows (see ngw_host.h).
#include "ngw_host.h"

using namespace ngwh;

namespace {

// Boards mode on / off: decided from (fused, eligible table); call BEFORE upload_reset_u (the no-stage launch prototype carries the mode),
// and boards_mode_changed AFTER it.  Switching it on makes every bit row stale: the prepared maps' rows are rebuilt at once, the main set's
// before the next step launch.
bool boards_mode_update(ngw_handle* h) {
    const bool was = h->boards_on;
    h->boards_on = h->lidar_fused && h->lidar_boards && h->b.brd && h->nostage;
    return was;
}

int boards_mode_changed(ngw_handle* h, bool was) {
    if (was == h->boards_on) return NGW_OK;
    if (int rc = publish_nx(h, h->prefetch_every > 0)) return rc;
    if (h->boards_on) {
        h->brd_dirty = true;
        if (h->nx.map && h->nx.brd) { if (int rc = rebuild_boards(h, h->nx.map, h->nx.brd, (int64_t)h->n_pad * h->depth)) return rc; }
    }
    return NGW_OK;
}

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

extern "C" {

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
        // The O(1) form on the occupancy bit rows (ngw_boards.inc): the reference's default 8 rays - four axes, four true diagonals advancing by
        // round(0.71 k) - on a map of at most 32 x 32.  Checked, not assumed: every entry of the world table against ngw_lidar8_dr / _dc, a diagonal
        // never skips a cell, and the kernel's closed form of "the first range that reaches diagonal distance d" against the table for every d.
        h->lidar_boards = 0;
        if (world && B == 8 && h->proto.BS && h->nostage) {
            bool same = true;
            for (int w = 0; w < 8 && same; w++)
                for (int k = 1; k <= R; k++)
                    if (hd.woff[w][k - 1] != ngw_lidar8_dr(w, k) * S + ngw_lidar8_dc(w, k)) { same = false; break; }
            for (int k = 1; k <= NGW_LIDAR_MAX_RANGE && same; k++)
                if (ngw_lidar8_diag(k) - ngw_lidar8_diag(k - 1) < 0 || ngw_lidar8_diag(k) - ngw_lidar8_diag(k - 1) > 1) same = false;
            for (int d = 1; d <= S - 2 && same; d++) {
                int first = 0;
                for (int k = 1; k <= 2 * NGW_LIDAR_MAX_RANGE; k++) if (ngw_lidar8_diag(k) == d) { first = k; break; }
                if (first != (100 * d + 20) / 71) same = false;
            }
            if (same) h->lidar_boards = 1;
            if (const char* v = getenv("NGW_LIDAR_BOARDS")) if (atoi(v) == 0) h->lidar_boards = 0;   // A/B: the marches over maps staged through LDS
            if (getenv("NGW_LIDAR_WORLD")) h->lidar_boards = 0;                                     // (somebody chose a march by name)
        }
        hd.world = world;
        memcpy(hd.chan_of_item, cfg->chan_of_item, NGW_MAX_ITEMS);
        memcpy(hd.inv_item, cfg->inv_item, NGW_MAX_ITEMS);
        hd.num_beams = B; hd.max_range = R; hd.n_chan = cfg->n_chan; hd.n_inv = cfg->n_inv;
        HIP_TRY(hipMemcpyAsync(h->lidar_cfg, &hd, sizeof(hd), hipMemcpyDefault, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    h->lidar_len = L; h->lidar_world = world;
    h->lidar_host_rows = nullptr;
    h->wt_block = nullptr;
    h->lidar_range = cfg->max_range; h->lidar_beams = cfg->num_beams; h->lidar_chan = cfg->n_chan; h->lidar_ninv = cfg->n_inv;
    if (h->lidar_fused) { if (int rc = layout_lds(h)) { h->lidar_fused = 0; layout_lds(h); boards_mode_update(h); upload_reset_u(h); return rc; } }
    const bool was = boards_mode_update(h);
    if (int rc = upload_reset_u(h)) return rc;
    if (int rc = boards_mode_changed(h, was)) return rc;
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
    h->lidar_host_rows = nullptr;                    // (the row size changed: the caller registers a buffer of the new size)
    h->wt_block = nullptr;
    if (h->lidar_len) {
        if (h->lidar_fused) { if (int rc = layout_lds(h)) return rc; }
        const bool was = boards_mode_update(h);
        if (int rc = upload_reset_u(h)) return rc;
        if (int rc = boards_mode_changed(h, was)) return rc;
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
    h->wt_block = nullptr;
    if (int rc = layout_lds(h)) { h->lidar_fused = before; layout_lds(h); upload_reset_u(h); return rc; }
    const bool was = boards_mode_update(h);
    if (int rc = upload_reset_u(h)) return rc;
    if (int rc = boards_mode_changed(h, was)) return rc;
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

int ngw_lidar_host_rows(ngw_handle* h, void* rows_host) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (rows_host && !h->lidar_len) return fail(NGW_E_INVALID_ARG, "ngw_lidar_host_rows before ngw_lidar_configure");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->lidar_host_rows = static_cast<uint8_t*>(rows_host);
    h->wt_block = nullptr;                           // (the step kernel's write-through targets are set up again)
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

}  // extern "C"
