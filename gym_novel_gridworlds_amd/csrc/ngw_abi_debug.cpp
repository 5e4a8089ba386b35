// ngw_abi_debug.cpp - the timing pair of bench.py's roofline leg and the diagnostics entry points (profiling tools only; none of the latter is
// declared in include/ngw.h) (see ngw_host.h).
#include "ngw_host.h"

using namespace ngwh;

extern "C" {

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
    // the SHORTEST of a few regions (a region of 20 launches is a ~50 us sample: one of them says little), more of them the shorter the region
    const int regions = n_launches >= 512 ? 2 : (n_launches >= 64 ? 4 : 8);
    double best = 1e30;
    for (int rep = 0; rep < regions && !rc; rep++) {
        (void)hipEventRecord(e0, h->stream);
        if (graph) { if (hipGraphLaunch(ge, h->stream) != hipSuccess) rc = fail(NGW_E_HIP, "replay failed"); }
        else rc = issue();
        (void)hipEventRecord(e1, h->stream);
        if (!rc && hipEventSynchronize(e1) != hipSuccess) rc = fail(NGW_E_HIP, "event sync failed");
        float ms = 0.f;
        if (!rc && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = fail(NGW_E_HIP, "elapsed failed");
        if (!rc && (double)ms < best) best = (double)ms;
    }
    if (!rc) *us_per_launch = best * 1e3 / n_launches;
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

}  // extern "C"
