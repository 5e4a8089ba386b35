// ngw_abi_host.cpp - the host API: ngw_step_host and its narrow wire format ngw_step_host_packed (delta refresh of the caller's page-locked
// block), observations / outputs / state to and from host arrays, the multi-GPU observation payload (see ngw_host.h).
#include "ngw_host.h"

using namespace ngwh;

// ngw_step_host: up to here the outputs go through mapped host memory (a kernel writing across PCIe sustains ~12 GB/s, the copy
// engine ~26 GB/s but costs ~15 us to get going: measured crossover 256-512 KB, tools/api_latency.py with NGW_ZC_BYTES - 16 384
// envs, 2.6 MB: 186-230 us through mapped memory, 125-140 us staged and copied)
#define NGW_ZERO_COPY_BYTES (h->zc_bytes)

namespace ngwh {

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

}  // namespace ngwh

namespace {

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

void host_step_layout_packed(const ngw_handle* h, uint64_t off[8]) {
    const uint64_t n = (uint64_t)h->n, S2 = (uint64_t)h->proto.S2, K = (uint64_t)h->proto.K;
    const uint64_t bytes[7] = {n * S2, n * K * 4, n * 4, n * 4, n, n * 4, 4};
    uint64_t o = 0;
    for (int i = 0; i < 7; i++) { off[i] = o; o += (bytes[i] + 255) & ~(uint64_t)255; }
    off[7] = o;
}

// Actions: int32 ids from the caller's array, validated (the reference raises for a bad id before it touches any state: pogostick_v1_env.py:236)
// and narrowed to one byte per env in ONE pass into the page-locked buffer the step kernel reads in place.  Returns nonzero when an id lies
// outside [0, A).  AVX2 where the CPU has it (32 ids per round: 2.5 us per 65 536 against the 7 us of the compiler's own vectorisation).
#if (defined(__x86_64__) || defined(__i386__)) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
__attribute__((target("avx2"))) uint32_t narrow_actions_avx2(const int32_t* a, uint8_t* o, size_t n, int A) {
    const __m256i top = _mm256_set1_epi32(A - 1), idx = _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7);
    __m256i bad = _mm256_setzero_si256();                              // sign bits of a and of (A - 1 - a): set for a < 0 or a >= A
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i x0 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(a + i)), x1 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(a + i + 8));
        const __m256i x2 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(a + i + 16)), x3 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(a + i + 24));
        bad = _mm256_or_si256(bad, _mm256_or_si256(_mm256_or_si256(x0, x1), _mm256_or_si256(x2, x3)));
        bad = _mm256_or_si256(bad, _mm256_or_si256(_mm256_or_si256(_mm256_sub_epi32(top, x0), _mm256_sub_epi32(top, x1)),
                                                   _mm256_or_si256(_mm256_sub_epi32(top, x2), _mm256_sub_epi32(top, x3))));
        const __m256i p = _mm256_packus_epi16(_mm256_packs_epi32(x0, x1), _mm256_packs_epi32(x2, x3));   // (bytes of valid ids survive both saturations)
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(o + i), _mm256_permutevar8x32_epi32(p, idx));
    }
    uint32_t b = (uint32_t)_mm256_movemask_ps(_mm256_castsi256_ps(bad));
    for (; i < n; i++) { const uint32_t v = (uint32_t)a[i]; b |= v >= (uint32_t)A ? 1u : 0u; o[i] = (uint8_t)v; }
    return b;
}
#define NGW_HAVE_AVX2_NARROW 1
#endif
uint32_t narrow_actions(const int32_t* a, uint8_t* o, size_t n, int A) {
#ifdef NGW_HAVE_AVX2_NARROW
    static const bool avx2 = __builtin_cpu_supports("avx2") != 0;
    if (avx2 && A >= 1) return narrow_actions_avx2(a, o, n, A);      // (A - 1 - a cannot wrap for a >= 0: A <= 64)
#endif
    uint32_t bad = 0;
    for (size_t i = 0; i < n; i++) {                                  // (branch-free: vectorises)
        const uint32_t v = (uint32_t)a[i];
        bad |= v >= (uint32_t)A ? 1u : 0u;
        o[i] = (uint8_t)v;
    }
    return bad;
}

/* Payload of the multi-GPU observation gather: the seven SoA arrays back to back, each section padded to 16 bytes. */
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

extern "C" {

int ngw_get_obs(ngw_handle* h, int8_t* map, int32_t* loc, int32_t* facing, int32_t* inv) {
    if (!h) return fail(NGW_E_INVALID_ARG, "handle is NULL");
    HIP_TRY(hipSetDevice(h->device));
    if (h->solo_running) { if (int rc = solo_stop(h)) return rc; }
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
    if (h->solo_running) { if (int rc = solo_stop(h)) return rc; }
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
    if (h->hostres && solo_ok(h)) {
        // ONE env, reference semantics (the gym.Env adapter): the resident step loop (ngw_solo.inc) has speculated the outcome of every
        // action from the committed state - this step is a look-up in its records plus a command posted for it to commit.  No launch, no
        // wait (unless the caller steps faster than the device speculates).
        if (int rc = solo_step(h, actions_host[0])) return rc;
        const NgwMirror& m = h->mir;
        const void* const mirrors[INFO] = {m.map, m.loc, m.facing, m.inv, m.reward, m.done, nullptr, m.selected, m.step_count};
        for (int r = 0; r < INFO; r++)
            if (outs[r].host && r != 6) memcpy(outs[r].host, mirrors[r], outs[r].bytes);
        if (error_flags) *error_flags = *h->b.flags_host;
        if (want_info) info_words = m.info;
    } else if (h->hostres) {
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
        const bool delta = h->host_delta && h->mirror_valid && h->mirror_block == map && !h->shadow_stale;
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
                h->shadow_stale = false;
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
 * 3 reward int32 [n] (ABI 3; int16 before: widening it on the host meant reading 128 KB the device had just written - ~12 us of cache
 * misses per call at 65 536 envs - where two more bytes per env on the wire cost ~2), 4 done uint8 [n], 5 info uint32 [n] (NGW_INFO_*),
 * 6 error flags uint32), each padded to 256 bytes; offsets8[7] = the block's size.  Sections 0-1 are refreshed by deltas (only the 16-byte
 * pieces a step changed cross PCIe), 2-6 are dense: 13 B per env against the 26 B of the int32 SoA arrays of ngw_step_host. */
int ngw_host_step_layout_packed(ngw_handle* h, uint64_t* offsets8) {
    if (!h || !offsets8) return fail(NGW_E_INVALID_ARG, "NULL argument");
    host_step_layout_packed(h, offsets8);
    return NGW_OK;
}

int ngw_step_host_packed(ngw_handle* h, const int32_t* actions_host, void* block, int with_map) {
    if (!h || !actions_host || !block) return fail(NGW_E_INVALID_ARG, "NULL argument");
    const ngw_spec& sp = h->spec;
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
    uint64_t off[8];
    host_step_layout_packed(h, off);
    if (!h->wire_stage) { if (int rc = dev_alloc(h, &h->wire_stage, (size_t)(off[7] - off[2]))) return rc; }
    // the fused lidar observation's rows ride along when the caller registered a buffer for them (ngw_lidar_host_rows)
    const bool lrows = h->lidar_host_rows && h->lidar_fused && h->lidar_len;
    size_t lrb = 0;
    if (lrows) { NgwLaunch lq{}; lidar_format(h, lq); lrb = (size_t)lq.l_rb; }
    const bool mirrored = h->host_delta && h->mirror_valid && h->mirror_block == block;
    // Steady state on the in-place step kernel: ONE launch.  The block is a mirror of the state and mapped into the GPU's address space; the
    // step kernel's write-through form (NgwWT, ngw_lean.inc) stores what the step changes straight into it - cells, inventory slots, whole rows of
    // the envs that start an episode, pose / reward / done / info of every env - and its last block publishes the error flags and this call's
    // sequence number behind section 6's flags word, which is polled here: no delta kernel, no stream synchronisation (a refill launch that
    // follows the step on the stream runs while the caller already works on the results).  The delta kernel's shadows go stale meanwhile.
    if (mirrored && h->wt_enabled && h->api_slices <= 1 && h->nostage && (!h->lidar_fused || h->boards_on) && !h->hostres && !h->capturing && !h->proto.stamps) {
        const uint32_t bad = narrow_actions(actions_host, a8, n, A);
        if (bad) {
            h->act_next ^= 1;
            for (size_t i = 0; i < n; i++)
                if (actions_host[i] < 0 || actions_host[i] >= A) return fail(NGW_E_INVALID_ACTION, "%d is not in list", (int)actions_host[i]);   // pogostick_v1_env.py:236
        }
#ifdef NGW_HOSTTRACE
        static double wA = 0, wB = 0, wC = 0; static int wn = 0;
        const double w1 = pnow();                                      // actions validated and narrowed
#endif
        uint8_t* const blk = static_cast<uint8_t*>(block);
        if (h->wt_block != block) {
            if (!h->wt_count) {
                if (int rc = dev_alloc(h, &h->wt_count, 64)) return rc;
                HIP_TRY(hipMemsetAsync(h->wt_count, 0, 64, h->stream));
            }
            uint8_t* const st = h->mirror_dev;
            NgwWT w;
            w.map = reinterpret_cast<int8_t*>(st + off[0]); w.inv = reinterpret_cast<int32_t*>(st + off[1]); w.pose = reinterpret_cast<uint32_t*>(st + off[2]);
            w.reward = reinterpret_cast<int32_t*>(st + off[3]); w.done = st + off[4]; w.info = reinterpret_cast<uint32_t*>(st + off[5]);
            w.flags = reinterpret_cast<uint32_t*>(st + off[6]); w.count = h->wt_count;
            w.rows = nullptr;                                         // the observation rows go the same way when the buffer is mapped and whole blocks fit it
            h->wt_rows = false;
            if (lrows && h->n == h->n_pad) {
                void* rd = nullptr;
                if (hipHostGetDevicePointer(&rd, h->lidar_host_rows, 0) == hipSuccess && rd) { w.rows = static_cast<uint8_t*>(rd); h->wt_rows = true; }
                else (void)hipGetLastError();
            }
            HIP_TRY(hipMemcpyAsync(&h->dspec->wt, &w, sizeof(w), hipMemcpyDefault, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));                 // (`w` is on this frame)
            h->wt_block = block;
        }
        h->wt_seq = h->wt_seq + 1u ? h->wt_seq + 1u : 1u;
        volatile uint32_t* const sw = reinterpret_cast<volatile uint32_t*>(blk + off[6]) + 1;
        *sw = 0u;                                                       // (whatever the block held there)
        h->launch_use_action0 = false; h->launch_act_u8 = true; h->launch_wire = true;
        const int lrc = launch(h, NGW_MODE_STEP, 1, reinterpret_cast<const int32_t*>(h->act_pin_dev + (size_t)slot * cap), nullptr, 0, 0);
        h->launch_act_u8 = false; h->launch_wire = false;
        if (lrc) return lrc;
        // (no event for the action buffer's slot: this call returns only after the kernel has published its sequence number)
#ifdef NGW_HOSTTRACE
        const double w2 = pnow();                                      // the launch is enqueued
#endif
        if (lrows && !h->wt_rows) {
            HIP_TRY(hipMemcpyAsync(h->lidar_host_rows, h->lidar_out, n * lrb, hipMemcpyDefault, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        } else {
            bool seen = false;
            for (uint32_t spin = 0; spin < (1u << 22); spin++) {          // (bounded: ~100 ms, then the stream is synchronised the usual way)
                if (*sw == h->wt_seq) { seen = true; break; }
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#else
                __asm__ __volatile__("" ::: "memory");
#endif
            }
            if (!seen) HIP_TRY(hipStreamSynchronize(h->stream));
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);                       // the block's reads stay behind the poll
#ifdef NGW_HOSTTRACE
        {
            const double w3 = pnow();                                  // the sequence word has arrived: the block is written
            wA += w1 - p0; wB += w2 - w1; wC += w3 - w2;
            if (++wn == 200) {
                fprintf(stderr, "[hosttrace packed, write-through, %zu envs] narrow actions %.2f us, enqueue %.2f us, wait for the device %.2f us (lidar rows %s)\n", n, wA / wn, wB / wn, wC / wn,
                        lrows ? (h->wt_rows ? "by write-through" : "copied behind the launch") : "none");
                wA = wB = wC = 0; wn = 0;
            }
        }
#endif
        h->mirror_valid = true; h->shadow_stale = true;
        return NGW_OK;
    }
    const bool delta = mirrored && !h->shadow_stale;
    // Pipelined form (steady state of a big batch on the in-place step kernel): the batch steps in 2 - 4 slices on the handle's stream, and as
    // soon as a slice's step kernel is done a second stream refreshes that slice's part of the caller's block across PCIe - while the next
    // slice's actions are still being narrowed on the host and its step kernel runs.  The reference raises for a bad action id BEFORE it
    // touches any state (pogostick_v1_env.py:236), so every id is checked first (read-only pass); narrowing then goes slice by slice.
    int nsl = 1;
    if (delta && h->nostage && (!h->lidar_fused || h->boards_on) && !h->hostres && !h->capturing) {
        // MEASURED AND NOT THE DEFAULT (profiles/r05_ab.md): at 65 536 envs one slice 64.8 us per call, two 84.4, four 100.0 - every slice costs
        // two launches, an event record and a cross-stream wait (~10 us), more than the overlap returns.  NGW_API_SLICES=<n> still selects it.
        nsl = h->api_slices > 0 ? h->api_slices : 1;
        if (nsl > 4) nsl = 4;
        while (nsl > 1 && n / (size_t)nsl < 1024) nsl >>= 1;
    }
    if (nsl > 1) {
        uint32_t bad = 0;
        for (size_t i = 0; i < n; i++) bad |= (uint32_t)actions_host[i] >= (uint32_t)A ? 1u : 0u;     // (branch-free: vectorises)
        if (bad) {
            h->act_next ^= 1;
            for (size_t i = 0; i < n; i++)
                if (actions_host[i] < 0 || actions_host[i] >= A) return fail(NGW_E_INVALID_ACTION, "%d is not in list", (int)actions_host[i]);   // pogostick_v1_env.py:236
        }
        if (!h->stream2) {
            HIP_TRY(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
            for (hipEvent_t& e : h->slice_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        const size_t per = (n / (size_t)nsl + 63) / 64 * 64;
        int s_i = 0;
        for (size_t first = 0; first < n; first += per, s_i++) {
            const size_t count = n - first < per ? n - first : per;
            for (size_t i = first; i < first + count; i++) a8[i] = (uint8_t)actions_host[i];
            if (int rc = launch_step_slice(h, h->act_pin_dev + (size_t)slot * cap + first, (int64_t)first, (int64_t)count)) return rc;
            HIP_TRY(hipEventRecord(h->slice_ev[s_i], h->stream));
            HIP_TRY(hipStreamWaitEvent(h->stream2, h->slice_ev[s_i], 0));
            NgwDiff d = {};
            const uint8_t* const srcs[2] = {reinterpret_cast<const uint8_t*>(h->b.map) + first * S2, reinterpret_cast<const uint8_t*>(h->b.inv) + first * K * 4};
            const uint64_t so[2] = {first * S2, first * K * 4}, nb[2] = {count * S2, count * K * 4};
            int k = 0;
            for (int r = with_map ? 0 : 1; r < 2; r++, k++) {
                d.cur[k] = srcs[r]; d.shadow[k] = h->shadow[r] + so[r]; d.host[k] = h->mirror_dev + off[r] + so[r]; d.nbytes[k] = nb[r];
            }
            d.n_regions = k;
            NgwWire w = {};
            w.loc = h->b.loc + 2 * first; w.facing = h->b.facing + first; w.selected = h->b.selected + first; w.reward = h->b.reward + first;
            w.done = h->b.done + first; w.info = h->b.info + first; w.flags = h->b.flags;
            uint8_t* const st = h->mirror_dev;
            w.pose = reinterpret_cast<uint32_t*>(st + off[2]) + first; w.reward32 = reinterpret_cast<int32_t*>(st + off[3]) + first; w.done8 = st + off[4] + first;
            w.info32 = reinterpret_cast<uint32_t*>(st + off[5]) + first; w.flags_out = reinterpret_cast<uint32_t*>(st + off[6]);
            w.n = (int64_t)count;
            HIP_TRY(ngw_diff_wire_launch(&d, &w, h->stream2));
            if (lrows) HIP_TRY(hipMemcpyAsync(h->lidar_host_rows + first * lrb, reinterpret_cast<const uint8_t*>(h->lidar_out) + first * lrb, count * lrb, hipMemcpyDefault, h->stream2));
        }
        if (int rc = step_slices_done(h)) return rc;
        HIP_TRY(hipEventRecord(h->act_ev[slot], h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream2));                     // every slice's results are in the block (each waited for its step kernel)
        h->mirror_valid = h->host_delta && h->mirror_block == block;
        return NGW_OK;
    }
    const uint32_t bad = narrow_actions(actions_host, a8, n, A);
    if (bad) {
        h->act_next ^= 1;
        for (size_t i = 0; i < n; i++)
            if (actions_host[i] < 0 || actions_host[i] >= A) return fail(NGW_E_INVALID_ACTION, "%d is not in list", (int)actions_host[i]);   // pogostick_v1_env.py:236
    }
#ifdef NGW_HOSTTRACE
    const double p1 = pnow();                                          // actions validated and narrowed
#endif
    h->launch_use_action0 = false; h->launch_act_u8 = true;
    const int lrc = launch(h, NGW_MODE_STEP, 1, reinterpret_cast<const int32_t*>(h->act_pin_dev + (size_t)slot * cap), nullptr, 0, 0);
    h->launch_act_u8 = false;
    if (lrc) return lrc;
    HIP_TRY(hipEventRecord(h->act_ev[slot], h->stream));
    uint8_t* const blk = static_cast<uint8_t*>(block);
    const void* const srcs[2] = {h->b.map, h->b.inv};
    const uint64_t nb[2] = {n * S2, n * K * 4};
    NgwDiff d = {};
    const bool merged = delta;                                             // steady state: delta refresh + narrowing in one launch (below)
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
            h->shadow_stale = false;
        }
    }
    NgwWire w = {};
    w.loc = h->b.loc; w.facing = h->b.facing; w.selected = h->b.selected; w.reward = h->b.reward; w.done = h->b.done; w.info = h->b.info; w.flags = h->b.flags;
    // In delta mode the block is mapped into the GPU's address space (mirror_dev): the narrowing kernel stores pose / reward / done /
    // info straight into it across PCIe, like the delta kernel before it - no staging, no copy operation behind the kernels.
    const bool direct = delta;
    uint8_t* const st = direct ? h->mirror_dev : h->wire_stage - off[2];   // (staging holds sections 2 .. 6 at their block offsets)
    w.pose = reinterpret_cast<uint32_t*>(st + off[2]); w.reward32 = reinterpret_cast<int32_t*>(st + off[3]); w.done8 = st + off[4];
    w.info32 = reinterpret_cast<uint32_t*>(st + off[5]); w.flags_out = reinterpret_cast<uint32_t*>(st + off[6]);
    w.n = (int64_t)n;
    if (merged) HIP_TRY(ngw_diff_wire_launch(&d, &w, h->stream));
    else HIP_TRY(ngw_wire_launch(&w, h->stream));
    if (!direct) HIP_TRY(hipMemcpyAsync(blk + off[2], h->wire_stage, (size_t)(off[7] - off[2]), hipMemcpyDefault, h->stream));
    if (lrows) HIP_TRY(hipMemcpyAsync(h->lidar_host_rows, h->lidar_out, n * lrb, hipMemcpyDefault, h->stream));
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
    if (h->solo_running) { if (int rc = solo_stop(h)) return rc; }
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
    if (h->solo_running) { if (int rc = solo_stop(h)) return rc; }
    h->solo_mirror_valid = false;
    h->mirror_valid = false;
    if (map) h->brd_dirty = true;                    // (boards mode: the bit rows are rebuilt before the next step launch)
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

}  // extern "C"
