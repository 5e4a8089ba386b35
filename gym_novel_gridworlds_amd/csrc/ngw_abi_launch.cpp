// ngw_abi_launch.cpp - which kernel a call runs: steps, explicit resets, fused rollouts, prepared next episodes (refills, depth and cadence
// adaptation), hipGraph capture / replay, terminal-observation capture (see ngw_host.h).
#include "ngw_host.h"

using namespace ngwh;

int capture_graph(ngw_handle* h, const int32_t* actions_dev, int64_t step_stride, int32_t n_steps);

namespace {
int launch_reset_fast(ngw_handle* h, int mode, const uint8_t* mask_dev, bool* taken);
int launch_lidar_boards(ngw_handle* h);
int refill_launches(ngw_handle* h, bool* fast);
int rollout_chunks(ngw_handle* h, int mode, int32_t n_steps, const int32_t* actions_dev, uint64_t action_seed, int64_t t0, int64_t step_stride);
}

namespace ngwh {

int publish_nx(ngw_handle* h, bool on) {
    NgwNx on_device = on ? h->nx : NgwNx{};                                        // null pointers switch the consume path off
    if (!h->boards_on) on_device.brd = nullptr;                                    // (the cold path copies the prepared maps' bit rows only while somebody reads them)
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
    if (!rc && h->proto.BS) rc = dev_alloc(h, &nw.brd, np * (size_t)h->proto.BS);
    if (!rc) rc = dev_alloc(h, &nw.slow, 16);
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(NGW_E_HIP, "zero-fill of the prepared-episode buffers failed");
    auto drop = [&](const NgwNx& x) {
        void* const part[7] = {x.map, x.loc, x.facing, x.inv, x.episode, x.slow, x.brd};
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

// Boards mode: the occupancy bit rows of `rows` maps (a multiple of 64) rebuilt from the maps (ngw_boards_kernel), on the handle's stream.
int rebuild_boards(ngw_handle* h, const int8_t* map, uint32_t* brd, int64_t rows) {
    if (!map || !brd || rows <= 0) return NGW_OK;
    HIP_TRY(ngw_boards_launch(&h->brd_proto, h->map_mode, map, brd, rows, h->brd_lds, h->stream));
    return NGW_OK;
}

int launch_refill(ngw_handle* h) {
    bool fast = false;
    // (boards mode: both new-episode kernels write the bit rows of the maps they make - NgwBufs::brd of the slot)
    return refill_launches(h, &fast);
}

int launch(ngw_handle* h, int mode, int n_steps, const int32_t* actions_dev, const uint8_t* mask_dev, uint64_t action_seed, int64_t t0) {
    if (h->solo_running) { if (int rc = solo_stop(h)) return rc; }
    h->solo_mirror_valid = false;                     // (a per-launch step / reset refreshes the WHOLE mirror itself; the loop's host side starts from a copy)
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
    // Boards mode (fused lidar on the occupancy bit rows): steps run the in-place kernel with the bit-row epilogue; an explicit reset runs
    // the plain new-episode kernels and is followed by the rebuild of the bit rows and the observation launch; a fused rollout still ends in
    // the staged march (its maps are in LDS anyway) and leaves the bit rows stale until the next step launch wants them.
    const bool boards = h->boards_on;
    if (boards && h->brd_dirty && mode == NGW_MODE_STEP) {
        if (int rc = rebuild_boards(h, h->b.map, h->b.brd, h->n_pad)) return rc;
        h->brd_dirty = false;
    }
    bool fast_took = false;
    if (mode == NGW_MODE_RESET) { if (int rc = launch_reset_fast(h, NGW_MODE_RESET, mask_dev, &taken)) return rc; fast_took = taken; }
    if (!taken && mode == NGW_MODE_STEP && h->nostage && (!h->lidar_fused || boards)) {   // maps read in place: no-stage step kernel
        NgwLaunch q = h->ns_proto;
        q.b = h->b; q.mode = mode; q.n_steps = 1; q.actions = actions_dev; q.autoreset = h->autoreset; q.horizon = h->horizon; q.stamps = h->proto.stamps;
        q.seq = h->launch_wire ? h->wt_seq : h->launch_seq; q.action0 = h->launch_action0; q.use_action0 = h->launch_use_action0 ? 1 : (h->launch_act_u8 ? 2 : 0);
        HIP_TRY(ngw_launch(h->dspec, &q, h->map_mode, 8 | (h->ext ? 2 : 0) | (boards ? 1 : 0) | (h->launch_wire ? 16 : 0), grid, h->ns_lds, h->stream));
        taken = true;
    }
    if (!taken) {
        if (!h->general_ok)
            return fail(NGW_E_INVALID_ARG, "map_size %d: this call keeps a wavefront's 64 maps in LDS (fused rollouts, the fused lidar epilogue) "
                                           "and they need more than 160 KiB; per-launch steps and resets are available", h->proto.S);
        const bool march = h->lidar_fused && !(boards && mode == NGW_MODE_RESET);
        if (!boards) a.b.brd = nullptr;                                // (the general kernel writes bit rows only while somebody reads them)
        HIP_TRY(ngw_launch(h->dspec, &a, h->map_mode, (march ? 1 : 0) | (h->ext ? 2 : 0), grid, h->lds_bytes, h->stream));
    }
    if (boards && mode == NGW_MODE_RESET) {
        // (the dedicated new-episode kernel wrote the bit rows of the maps it made or copied; a masked reset leaves the others as they were,
        //  which is only right if they were right before: a stale set is rebuilt whole)
        (void)fast_took;
        if (h->brd_dirty) { if (int rc = rebuild_boards(h, h->b.map, h->b.brd, h->n_pad)) return rc; }
        h->brd_dirty = false;
        if (int rc = launch_lidar_boards(h)) return rc;
    }
    if (boards && (mode == NGW_MODE_ROLLOUT || mode == NGW_MODE_ROLLOUT_ACT)) h->brd_dirty = true;
    if (h->prefetch_every > 0 && (mode == NGW_MODE_STEP || mode == NGW_MODE_RESET || mode == NGW_MODE_ROLLOUT || mode == NGW_MODE_ROLLOUT_ACT)) {
        // Prepared next episodes: every `prefetch_every` batched steps (and right after an explicit reset) one more launch
        // refills the shadow rows that resets have consumed since.  Same stream, so it is ordered between the steps.
        h->since_refill += mode == NGW_MODE_RESET ? h->prefetch_every : n_steps;
        if (h->since_refill >= h->cadence) return launch_refill(h);
    }
    return NGW_OK;
}

// One SLICE [first, first + count) of a batched step through the in-place step kernel (first a multiple of 64): every array pointer of
// the launch block is shifted to the slice's first env - the kernel's hot path works on those -, bid0 tells the cold path (which reads the
// blob's unshifted arrays) where the slice starts.  Byte actions (ngw_step_host_packed's page-locked staging).  The caller issues the
// slices of ONE step back to back and then calls step_slices_done.
int launch_step_slice(ngw_handle* h, const uint8_t* actions_u8_dev, int64_t first, int64_t count) {
    if (first == 0) {
        h->mirror_valid = false;
        if (h->boards_on && h->brd_dirty) {
            if (int rc = rebuild_boards(h, h->b.map, h->b.brd, h->n_pad)) return rc;
            h->brd_dirty = false;
        }
    }
    const int64_t S2 = h->proto.S2, K = h->proto.K, BS = h->proto.BS;
    NgwLaunch q = h->ns_proto;
    q.b = h->b;
    q.b.map += first * S2; q.b.inv += first * K; q.b.loc += first * 2; q.b.facing += first; q.b.selected += first; q.b.step_count += first;
    q.b.reward += first; q.b.done += first; q.b.info += first;
    if (q.b.brd) q.b.brd += first * BS;
    if (q.lout) q.lout = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(q.lout) + first * q.l_rb);
    q.n = count; q.bid0 = (uint32_t)(first / NGW_EPB);
    q.mode = NGW_MODE_STEP; q.n_steps = 1; q.actions = reinterpret_cast<const int32_t*>(actions_u8_dev); q.autoreset = h->autoreset; q.horizon = h->horizon;
    q.stamps = nullptr; q.seq = 0; q.action0 = 0; q.use_action0 = 2;
    const unsigned grid = (unsigned)((count + NGW_EPB - 1) / NGW_EPB);
    HIP_TRY(ngw_launch(h->dspec, &q, h->map_mode, 8 | (h->ext ? 2 : 0) | (h->boards_on ? 1 : 0), grid, h->ns_lds, h->stream));
    return NGW_OK;
}

// ---------------------------------------------------------------- the one-env handle's resident step loop (ngw_solo.inc)
bool solo_ok(const ngw_handle* h) {
    return h->solo_enabled && h->n == 1 && h->hostres && !h->autoreset && !h->lidar_fused && !h->term_on && !h->capturing && h->proto.S <= 46;
}

namespace {
int solo_start(ngw_handle* h, int32_t commit0, uint32_t k0) {
    const int S2 = h->proto.S2, K = h->proto.K;
    if (!h->solo_mbox) {
        void* q = nullptr;
        HIP_TRY(hipHostMalloc(&q, 64, hipHostMallocMapped));
        memset(q, 0, 64);
        h->host_allocs.push_back(q);
        h->solo_mbox = static_cast<uint32_t*>(q);
        const size_t rec_dw = (size_t)((8 + K + 3) & ~3);
        const size_t bytes = (NGW_SOLO_REC0 + NGW_MAX_ACTIONS * rec_dw) * 4;
        HIP_TRY(hipHostMalloc(&q, bytes, hipHostMallocMapped));
        memset(q, 0, bytes);
        h->host_allocs.push_back(q);
        h->solo_out = static_cast<uint32_t*>(q);
        NgwSolo& p = h->solo_proto;
        p = NgwSolo{};
        p.mbox = h->solo_mbox; p.out = h->solo_out;
        p.S = h->proto.S; p.S2 = S2; p.K = K; p.A = h->spec.n_actions; p.MSp = (S2 + 15) & ~15;
        p.KP = K | 1; p.rec_dw = (int32_t)rec_dw;
        p.timeout_ticks = 30000;                                      // 300 us without a command: the loop ends, the next step() starts a new one
        if (const char* v = getenv("NGW_SOLO_TIMEOUT_US")) p.timeout_ticks = (uint32_t)atoi(v) * 100u;
        uint32_t off = 0;
        p.off_map = off; off += (uint32_t)(NGW_EPB * p.MSp / 4);
        p.off_master = off; off += (uint32_t)(p.MSp / 4);
        p.off_inv = off; off += (uint32_t)(NGW_EPB * p.KP);
        p.off_invb = off; off += (uint32_t)(NGW_EPB * p.KP);
        h->solo_lds = (size_t)off * 4;
    }
    if (!h->solo_mirror_valid) {
        // the host applies every step's outcome to its mirror of the state: it has to start from what the device holds (anything else that
        // ran on this handle since the last loop - resets, state injection, per-launch steps - may have left the mirror behind)
        const NgwMirror& m = h->mir;
        HIP_TRY(hipMemcpyAsync(m.map, h->b.map, (size_t)S2, hipMemcpyDefault, h->stream));
        HIP_TRY(hipMemcpyAsync(m.loc, h->b.loc, 8, hipMemcpyDefault, h->stream));
        HIP_TRY(hipMemcpyAsync(m.facing, h->b.facing, 4, hipMemcpyDefault, h->stream));
        HIP_TRY(hipMemcpyAsync(m.inv, h->b.inv, (size_t)K * 4, hipMemcpyDefault, h->stream));
        HIP_TRY(hipMemcpyAsync(m.selected, h->b.selected, 1, hipMemcpyDefault, h->stream));
        HIP_TRY(hipMemcpyAsync(m.step_count, h->b.step_count, 4, hipMemcpyDefault, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->solo_mirror_valid = true;
    }
    NgwSolo p = h->solo_proto;
    p.b = h->b; p.k0 = k0; p.commit0 = commit0;
    ((volatile uint32_t*)h->solo_out)[0] = k0 - 1u;                  // (nothing speculated yet)
    ((volatile uint32_t*)h->solo_out)[1] = 0u;
    ((volatile uint32_t*)h->solo_mbox)[0] = k0;                      // (no command posted beyond the state the launch starts from ... or the one it commits first)
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    HIP_TRY(ngw_solo_launch(h->dspec, &p, h->ext, h->solo_lds, h->stream));
    h->solo_running = true;
    return NGW_OK;
}
inline void cpu_pause() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}
}  // namespace

int solo_stop(ngw_handle* h) {
    if (!h->solo_running) return NGW_OK;
    volatile uint32_t* mb = h->solo_mbox; volatile uint32_t* out = h->solo_out;
    mb[2] = 1u;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->solo_running = false;
    if (out[0] != h->solo_seq) {
        // the loop had ended (idle limit) before it saw the last command: a fresh launch commits it and, the quit word still up, leaves at once
        if (out[0] + 1u != h->solo_seq) { mb[2] = 0u; return fail(NGW_E_HIP, "one-env step loop lost its place (device %u, host %u)", out[0], h->solo_seq); }
        if (int rc = solo_start(h, h->solo_last_action, h->solo_seq - 1u)) { mb[2] = 0u; return rc; }
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->solo_running = false;
    }
    mb[2] = 0u;
    return NGW_OK;
}

int solo_step(ngw_handle* h, int32_t action) {
    volatile uint32_t* mb = h->solo_mbox; volatile uint32_t* out = h->solo_out;
    if (!h->solo_running) { if (int rc = solo_start(h, -1, h->solo_seq)) return rc; mb = h->solo_mbox; out = h->solo_out; }
    // the records of the committed state: usually there already (the device speculated while the caller was busy)
    for (uint64_t spin = 0;; spin++) {
        if (out[0] == h->solo_seq) break;
        if (out[1]) {                                                // the loop ended (idle limit) ...
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            if (out[0] == h->solo_seq) break;                        // ... after it had speculated this state: the records stand
            HIP_TRY(hipStreamSynchronize(h->stream));                // ... before it saw the last command: a fresh launch commits it first
            h->solo_running = false;
            if (out[0] + 1u != h->solo_seq) return fail(NGW_E_HIP, "one-env step loop lost its place (device %u, host %u)", out[0], h->solo_seq);
            if (int rc = solo_start(h, h->solo_last_action, h->solo_seq - 1u)) return rc;
            spin = 0;
            continue;
        }
        if (spin > (1ull << 26)) return fail(NGW_E_HIP, "one-env step loop does not answer");
        cpu_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    // ---- the outcome of `action`, applied to the host's mirror of the state (what the kernel's commit does to the state in HBM)
    const NgwSolo& p = h->solo_proto;
    const uint32_t* rec = h->solo_out + NGW_SOLO_REC0 + (size_t)action * (size_t)p.rec_dw;
    const NgwMirror& m = h->mir;
    const int S = p.S, K = p.K;
    const uint32_t w2 = rec[2], w3 = rec[3];
    const int nr = (int)((w2 >> 8) & 255u), nc = (int)((w2 >> 16) & 255u);
    if ((w3 >> 8) & 1u) m.map[rec[5]] = (int8_t)((w3 >> 16) & 255u);
    if (rec[6]) {
        const int ac = nr * S + nc;
        for (int k = 0; k < 9; k++) if ((rec[6] >> k) & 1u) m.map[ac + (k / 3 - 1) * S + (k % 3 - 1)] = 0;
    }
    for (int k = 0; k < K; k++) m.inv[k] = (int32_t)rec[8 + k];
    m.loc[0] = nr; m.loc[1] = nc; m.facing[0] = (int32_t)(w2 >> 24); m.selected[0] = (uint8_t)(w3 & 255u); m.step_count[0] = (int32_t)rec[4];
    m.reward[0] = (int32_t)rec[0]; m.done[0] = (uint8_t)(w2 & 1u); m.info[0] = rec[1];
    if (rec[7]) *h->b.flags_host |= rec[7];
    // ---- post the command: the loop commits it and speculates from the new state
    h->solo_last_action = action;
    h->solo_seq++;
    if (out[1]) {                                                    // (the loop has ended meanwhile: start the next one with the commit)
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->solo_running = false;
        return solo_start(h, action, h->solo_seq - 1u);
    }
    mb[1] = (uint32_t)action;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    mb[0] = h->solo_seq;
    return NGW_OK;
}

int step_slices_done(ngw_handle* h) {
    if (h->prefetch_every > 0) {
        h->since_refill += 1;
        if (h->since_refill >= h->cadence) return launch_refill(h);
    }
    return NGW_OK;
}

void drop_graph(ngw_handle* h) {
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    h->graph_exec = nullptr;
    h->graph = nullptr;
    h->graph_steps = 0;
    h->graph_open = false;
}

}  // namespace ngwh

namespace {

// mode = NGW_MODE_RESET (mask_dev or nullptr) / NGW_MODE_REFILL; returns 1 if the dedicated kernel took the launch
int launch_reset_fast(ngw_handle* h, int mode, const uint8_t* mask_dev, bool* taken) {
    *taken = false;
    if (h->rf_nw < 0 || (h->lidar_fused && !h->boards_on)) return NGW_OK;   // (the march rides on the general kernel; the bit-row lidar is its own launch)
    NgwResetFast a = h->rf;
    a.main = h->b; a.nx = h->prefetch_every > 0 ? h->nx : NgwNx{}; a.mode = mode; a.reset_mask = mask_dev; a.stamps = h->proto.stamps;
    a.seq = mode == NGW_MODE_RESET ? h->launch_seq : 0u;
    a.boards = h->boards_on ? 1 : 0; a.BS = h->proto.BS;                            // (boards mode: the kernel writes the bit rows of its maps itself)
    HIP_TRY(ngw_reset_fast_launch(h->dspec, &a, h->rf_nw, h->rf_additem, (unsigned)(h->n_pad / NGW_EPB), h->rf_lds, h->stream));
    *taken = true;
    return NGW_OK;
}

// ... and the LidarInFront observation of the current state from them, as its own launch (what follows an explicit reset in the boards mode)
int launch_lidar_boards(ngw_handle* h) {
    NgwLaunch a = h->lb_proto;
    a.b = h->b;
    HIP_TRY(ngw_lidar_boards_launch(&a, (unsigned)(h->n_pad / NGW_EPB), h->lb_lds, h->stream));
    return NGW_OK;
}

int refill_launches(ngw_handle* h, bool* fast) {
    *fast = false;
    h->since_refill = 0;
    adapt_cadence(h);
    if (h->adapt_error) {                                           // (not fatal: the handle keeps working at the depth it has)
        h->adapt_error = false;
        fail(NGW_E_HIP, "prepared episodes: not enough device memory for %d rows per env, staying at %d (the call itself succeeded)", h->depth * 2, h->depth);
    }
    h->refill_count++;
    bool taken = false;
    if (int rc = launch_reset_fast(h, NGW_MODE_REFILL, nullptr, &taken)) return rc;
    if (taken) { *fast = true; return NGW_OK; }
    // the general kernel prepares one slot per launch (its shadow set is the launch's buffer set)
    const size_t np = (size_t)h->n_pad, S2 = (size_t)h->proto.S2, K = (size_t)h->proto.K;
    for (int slot = 0; slot < h->depth; slot++) {
        NgwLaunch rf = h->proto;
        rf.b = NgwBufs{};
        rf.b.map = h->nx.map + slot * np * S2; rf.b.loc = h->nx.loc + slot * np * 2; rf.b.facing = h->nx.facing + slot * np;
        rf.b.inv = h->nx.inv + slot * np * K; rf.b.episode = h->nx.episode + slot * np;
        rf.b.brd = h->boards_on && h->nx.brd ? h->nx.brd + slot * np * (size_t)h->proto.BS : nullptr;
        rf.b.flags = h->b.flags; rf.b.perm = h->b.perm;
        rf.mode = NGW_MODE_REFILL; rf.n_steps = 1;
        rf.actions = reinterpret_cast<const int32_t*>(h->b.episode);
        rf.reset_mask = nullptr; rf.action_seed = 0; rf.t0 = (int64_t)h->refill_count;   // (REFILL: t0 = the refill's number)
        rf.autoreset = slot; rf.horizon = h->depth - 1;                                    // (REFILL: the slot and depth - 1)
        HIP_TRY(ngw_launch(h->dspec, &rf, h->map_mode, 0, (unsigned)(h->n_pad / NGW_EPB), h->lds_bytes, h->stream));
    }
    return NGW_OK;
}

// A fused rollout as launches of at most `prefetch_every` steps with the refill launches between them: with prepared next
// episodes on, an env's reset inside the launch copies its prepared row - but only the first one, the shadow rows are
// re-prepared between launches.  Same action stream (keyed by the absolute step), same results as one launch.
int rollout_chunks(ngw_handle* h, int mode, int32_t n_steps, const int32_t* actions_dev, uint64_t action_seed, int64_t t0, int64_t step_stride) {
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

}  // namespace

int capture_graph(ngw_handle* h, const int32_t* actions_dev, int64_t step_stride, int32_t n_steps) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    drop_graph(h);
    if (h->boards_on && h->brd_dirty) {               // (a rebuild captured into the graph would run with every replay)
        if (int rc = rebuild_boards(h, h->b.map, h->b.brd, h->n_pad)) return rc;
        h->brd_dirty = false;
    }
    // A graph much shorter than the refill cadence is captured WITHOUT a refill: closing every replay of a 20-step graph with one (below) would
    // run the ~21 us launch four times as often as the cadence asks for.  Such a graph is "open": ngw_graph_launch counts its steps and issues
    // the refill between replays, eagerly, whenever the next replay would overrun the cadence.
    const bool open = h->prefetch_every > 0 && n_steps * 2 <= h->cadence;
    const int since0 = h->since_refill;
    h->since_refill = 0;                              // the captured refill cadence starts from a known phase
    HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    h->capturing = true;                              // (the depth and cadence the handle has adapted to so far are the ones captured)
    int rc = NGW_OK;
    for (int i = 0; i < n_steps && !rc; i++) rc = launch(h, NGW_MODE_STEP, 1, actions_dev + (int64_t)i * step_stride, nullptr, 0, 0);
    // every replay must leave the refill cadence where it found it: a graph shorter than (or not a multiple of) the cadence
    // ends with one more refill, otherwise a replayed graph would never re-prepare the episodes its steps consume
    if (!rc && h->prefetch_every > 0 && h->since_refill > 0 && !open) {
        h->since_refill = h->prefetch_every;
        rc = launch_refill(h);
    }
    if (open) h->since_refill = since0;               // (nothing ran: the steps since the last refill are what they were)
    h->capturing = false;
    hipError_t e = hipStreamEndCapture(h->stream, &h->graph);
    if (rc) { drop_graph(h); return rc; }
    if (e != hipSuccess) { drop_graph(h); return fail(NGW_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e)); }
    e = hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { drop_graph(h); return fail(NGW_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
    (void)hipGraphUpload(h->graph_exec, h->stream);   // pre-stage the graph so the first replay does not pay for it
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->graph_steps = n_steps; h->graph_actions = actions_dev; h->graph_stride = step_stride;
    h->graph_open = open;
    h->adapted = false;
    return NGW_OK;
}

extern "C" {

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

int ngw_get_reset_prefetch_depth(ngw_handle* h, int32_t* depth) {
    if (!h || !depth) return fail(NGW_E_INVALID_ARG, "NULL argument");
    *depth = h->depth;
    return NGW_OK;
}

int ngw_step_kernel_info(ngw_handle* h, int32_t* map_in_place) {
    if (!h || !map_in_place) return fail(NGW_E_INVALID_ARG, "NULL argument");
    *map_in_place = (h->nostage && (!h->lidar_fused || h->boards_on)) ? 1 : 0;     // (the rule launch() applies to NGW_MODE_STEP)
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
        h->solo_mirror_valid = true;                                    // (the reset kernel has just written every row of the mirror: a one-env step loop starts from it without a copy)
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

int ngw_graph_build(ngw_handle* h, const int32_t* actions_dev, int64_t step_stride, int32_t n_steps) {
    if (!h || !actions_dev) return fail(NGW_E_INVALID_ARG, "NULL argument");
    if (n_steps < 1) return fail(NGW_E_INVALID_ARG, "n_steps must be >= 1");
    HIP_TRY(hipSetDevice(h->device));
    if (h->solo_running) { if (int rc = solo_stop(h)) return rc; }    // (a one-env handle's resident loop: its stream is busy until it ends)
    return capture_graph(h, actions_dev, step_stride, n_steps);
}

int ngw_graph_launch(ngw_handle* h, int32_t reps) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    if (!h->graph_exec) return fail(NGW_E_INVALID_ARG, "no graph: call ngw_graph_build first");
    HIP_TRY(hipSetDevice(h->device));
    if (h->solo_running) { if (int rc = solo_stop(h)) return rc; }
    h->solo_mirror_valid = false;
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
        if (h->graph_open && h->prefetch_every > 0 && h->since_refill + h->graph_steps > h->cadence) {
            if (int rc = launch_refill(h)) return rc;                               // (an open graph: the cadence is kept between its replays)
        }
        HIP_TRY(hipGraphLaunch(h->graph_exec, h->stream));
        if (h->graph_open) h->since_refill += h->graph_steps;
    }
    return NGW_OK;
}

}  // extern "C"
