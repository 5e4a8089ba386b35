// ngw_host.h - internals shared by the translation units of the C-ABI's host side (ngw_abi_*.cpp).  Not installed, not part of include/ngw.h.
//   ngw_abi_create.cpp   spec checks, allocation and layout of a handle (state slab, LDS carve-ups, the HBM spec blob), small accessors
//   ngw_abi_launch.cpp   which kernel a call runs: steps, resets, rollouts, prepared next episodes and their cadence, hipGraph capture
//   ngw_abi_host.cpp     the host API's wire formats (ngw_step_host, ngw_step_host_packed), state in / out, the multi-GPU payload
//   ngw_abi_obs.cpp      observation wrappers on the device: LidarInFront (marches and the bit-row form), AgentMap
//   ngw_abi_debug.cpp    timing pair and diagnostics entry points (not in include/ngw.h)
#ifndef NGW_HOST_H
#define NGW_HOST_H
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ngw.h"
#include "ngw_device.h"

namespace ngwh {
int fail(int code, const char* fmt, ...);           // sets the thread-local message ngw_last_error() returns; returns `code`
const char* last_error();
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return ngwh::fail(NGW_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define D2H(dst, src, bytes)                                                                             \
    do {                                                                                                 \
        if (dst) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDefault, h->stream));        \
    } while (0)
#define H2D(dst, src, bytes)                                                                             \
    do {                                                                                                 \
        if (src) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDefault, h->stream));        \
    } while (0)

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
    int wt_enabled = 1;                   // NGW_HOST_WRITE_THROUGH=0: the delta kernel behind every step instead of the step kernel's own stores into the block (A/B)
    bool shadow_stale = false;            // write-through steps ran since the shadows were seeded: the delta kernel needs them seeded again
    void* wt_block = nullptr;             // the block NgwDevSpec::wt points into
    uint32_t* wt_count = nullptr;         // device counter of finished blocks
    uint32_t wt_seq = 0;
    bool wt_rows = false;                 // NgwDevSpec::wt also points at the caller's lidar row buffer
    int host_delta = 1;                   // NGW_HOST_DELTA=0: every ngw_step_host copies the whole observation (A/B)
    size_t zc_bytes = (size_t)256 << 10;  // NGW_ZC_BYTES: largest ngw_step_host result written straight into mapped host memory (read at ngw_create)
    std::vector<void*> allocs;
    std::vector<void*> host_allocs;       // single-wavefront handles: the host mirror of the state (GPU-addressable page-locked memory)
    NgwMirror mir = {};                   // ... its arrays (host addresses = device addresses under unified addressing)
    uint8_t* mask_pin = nullptr; uint8_t* mask_pin_dev = nullptr;   // ngw_reset's mask: two page-locked halves the kernel reads in place
    hipEvent_t mask_ev[2] = {nullptr, nullptr};
    int mask_next = 0;
    uint8_t* act_pin_dev = nullptr;            // ... the same buffer as the GPU addresses it (ngw_step_host_packed: the kernel reads the actions in place)
    bool launch_wire = false;                  // the launch being issued is the host write-through form (feat 16), its sequence number wt_seq
    bool launch_act_u8 = false;                // the launch being issued reads one byte per env from `actions`
    uint8_t* wire_stage = nullptr;             // ngw_step_host_packed: device staging of the dense sections
    uint8_t* act_pin = nullptr;                // ngw_step's actions: two page-locked halves feeding the asynchronous copy
    hipEvent_t act_ev[2] = {nullptr, nullptr};
    int act_next = 0;
    int hostres = 0;                         // single-wavefront handle with a host mirror (NgwMirror)
    uint32_t step_seq = 0, launch_seq = 0;   // hostres: sequence number the next step launch reports (launch_seq: only while ngw_step_host issues it)
    int32_t launch_action0 = 0; bool launch_use_action0 = false;   // one-env handles: the action of the launch ngw_step_host is issuing
    // ngw_step_host_packed, pipelined: the batch steps in slices on the handle's stream while a second stream brings the finished slices'
    // results across PCIe (api_slices: 0 / 1 = off - the default: measured slower, profiles/r05_ab.md -, NGW_API_SLICES=<n> selects n slices)
    hipStream_t stream2 = nullptr;
    hipEvent_t slice_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int api_slices = 0;
    // The one-env handle's resident step loop (ngw_solo.inc; NGW_SOLO=0 switches it off): a kernel that stays on the stream between the steps
    // of a host loop, speculates every action's outcome into `solo_out` and commits what the host posts into `solo_mbox`.
    int solo_enabled = 1;
    bool solo_running = false, solo_mirror_valid = false;
    uint32_t* solo_mbox = nullptr;            // page-locked, GPU-addressable: [0] command sequence, [1] action, [2] quit
    uint32_t* solo_out = nullptr;             // page-locked, GPU-addressable: [0] speculated sequence, [1] exited, records from dword NGW_SOLO_REC0
    uint32_t solo_seq = 0;                    // sequence number of the committed state as the host counts it
    int32_t solo_last_action = -1;            // the last action posted (re-posted to a fresh launch if the loop ended before it took it)
    NgwSolo solo_proto{};
    size_t solo_lds = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // timing pair
    hipEvent_t order_ev = nullptr;             // ngw_stream_order
    bool ev_marked = false;                    // ngw_timing_mark recorded the closing event already
    // LidarInFront observation
    NgwLidarDev* lidar_cfg = nullptr;     // device tables
    int32_t* lidar_out = nullptr;
    int lidar_len = 0, lidar_cap = 0;         // lidar_cap: row length lidar_out was allocated for
    int lidar_bits = 32;                  // row format of the lidar observation (ngw_lidar_set_output): 32 (the default: the reference's integers), 16 or 8 = packed
    int lidar_world = 0;                  // the ray table is one world-frame table rotated by the facing (NgwLidarDev::woff)
    uint8_t* lidar_host_rows = nullptr;   // ngw_lidar_host_rows: the packed host step also brings the observation rows across (the caller's page-locked buffer)
    NgwLaunch lidar_proto{};              // the stand-alone lidar launch: its own LDS layout
    // The O(1) lidar on occupancy bit rows (ngw_boards.inc).  lidar_boards: the configured ray table is the reference's default 8 beams and
    // the map is at most 32 x 32 (ngw_lidar_configure checks it entry by entry); boards_on: that, and the observation is fused - step launches
    // then run the in-place kernel with the bit-row epilogue, and whoever rewrites maps wholesale is followed by ngw_boards_kernel.
    int lidar_boards = 0;
    bool boards_on = false, brd_dirty = false;   // brd_dirty: the main set's bit rows do not describe its maps (ngw_set_state, a fused rollout): rebuilt before the next step launch
    NgwLaunch brd_proto{}, lb_proto{};    // launch layouts of ngw_boards_kernel / ngw_lidar_boards_kernel
    size_t brd_lds = 0, lb_lds = 0;
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
    int fast_reset = 1;                   // dedicated new-episode kernel where it applies (NGW_FAST_RESET=0 / 2: the test suite's hook to run every reset through the general kernel / the dedicated one)
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
    bool graph_open = false;              // a SHORT graph (steps * 2 <= refill cadence) holds no refill: ngw_graph_launch keeps the cadence between its replays
    const int32_t* graph_actions = nullptr;   // what ngw_graph_build captured: an adaptation re-captures it
    int64_t graph_stride = 0;
};

namespace ngwh {

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
void dev_free(ngw_handle* h, void* p);

// ngw_abi_create.cpp
void lidar_format(const ngw_handle* h, NgwLaunch& p);
int layout_lds(ngw_handle* h);
int upload_reset_u(ngw_handle* h);
void layout_reset_fast(ngw_handle* h);
// ngw_abi_launch.cpp
int launch(ngw_handle* h, int mode, int n_steps, const int32_t* actions_dev, const uint8_t* mask_dev, uint64_t action_seed, int64_t t0);
int launch_refill(ngw_handle* h);
int solo_stop(ngw_handle* h);                       // ends the one-env handle's resident step loop (no-op when it is not running); every other entry point calls it first
int solo_step(ngw_handle* h, int32_t action);      // one step() through the loop: the host mirror (h->mir) holds the new state afterwards
bool solo_ok(const ngw_handle* h);
int launch_step_slice(ngw_handle* h, const uint8_t* actions_u8_dev, int64_t first, int64_t count);   // one slice of a batched step (byte actions), on the handle's stream
int step_slices_done(ngw_handle* h);                // the bookkeeping of ONE batched step (refill cadence) once its slices are out
int publish_nx(ngw_handle* h, bool on);
int alloc_nx(ngw_handle* h, int depth, bool on);
void adapt_cadence(ngw_handle* h);
int rebuild_boards(ngw_handle* h, const int8_t* map, uint32_t* brd, int64_t rows);
void drop_graph(ngw_handle* h);
// ngw_abi_host.cpp
void host_step_layout(const ngw_handle* h, uint64_t off[11]);

}  // namespace ngwh
#endif
