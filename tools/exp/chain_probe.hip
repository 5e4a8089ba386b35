// Feasibility probe (not product code): do consecutive dependent "step-like" launches overlap when they alternate between two
// streams and each wave waits for ITS OWN predecessor through a device flag instead of the stream's kernel-to-kernel barrier?
//   build: hipcc --offload-arch=gfx950 -O3 -o chain_probe chain_probe.hip ; run: ./chain_probe [n_wg] [launches] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(64) step_like(uint32_t* flag, uint32_t* state, const uint32_t* table, uint32_t t, int chained, uint32_t* xcc_log, uint32_t* err) {
    extern __shared__ uint32_t lds[];
    const uint32_t w = blockIdx.x, tid = threadIdx.x;
    if (chained) {
        uint32_t spins = 0;
        while (__hip_atomic_load(&flag[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != t - 1u) {
            if (++spins > (1u << 18)) { if (tid == 0) atomicOr(err, 1u); break; }        // bounded: never hang
            __builtin_amdgcn_s_sleep(2);
        }
    }
    // a dependent chain shaped like the step: state load -> table lookup -> second lookup -> store
    uint32_t s = __hip_atomic_load(&state[w * 64 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s != (t - 1u) * 3u) atomicOr(err, 2u);                                          // the predecessor's value must be visible
    uint32_t a = table[(s + tid) & 4095];
    lds[tid] = a; __syncthreads();
    uint32_t b = table[(lds[(tid + 1) & 63] + a) & 4095];
    uint32_t acc = b;
    for (int i = 0; i < 150; i++) acc = acc * 1664525u + 1013904223u;                    // ~dependent VALU
    __hip_atomic_store(&state[w * 64 + tid], t * 3u + (acc & 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (xcc_log) { const uint32_t prev = xcc_log[w]; if (t > 1 && prev != (xcc & 15u)) atomicOr(err, 4u); xcc_log[w] = xcc & 15u; }
    }
    __builtin_amdgcn_s_waitcnt(0);                                                      // the state stores are out (agent-scope stores: visible device-wide)
    if (tid == 0) __hip_atomic_store(&flag[w], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 1024, L = argc > 2 ? atoi(argv[2]) : 2000, ldsb = argc > 3 ? atoi(argv[3]) : 25600;
    uint32_t *flag, *state, *table, *xcc, *err;
    CK(hipMalloc(&flag, nwg * 4)); CK(hipMalloc(&state, nwg * 256)); CK(hipMalloc(&table, 4096 * 4)); CK(hipMalloc(&xcc, nwg * 4)); CK(hipMalloc(&err, 4));
    CK(hipMemset(table, 0, 4096 * 4));
    hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    for (int mode = 0; mode < 3; mode++) {   // 0: one stream, barrier between launches; 1: two streams + flags; 2: one stream + flags (control)
        CK(hipMemset(flag, 0, nwg * 4)); CK(hipMemset(state, 0, nwg * 256)); CK(hipMemset(err, 0, 4)); CK(hipMemset(xcc, 0, nwg * 4));
        CK(hipDeviceSynchronize());
        double best = 1e30;
        uint32_t t = 0;
        for (int rep = 0; rep < 3; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < L; i++) {
                t++;
                hipStream_t st = mode == 1 ? s[t & 1] : s[0];
                hipLaunchKernelGGL(step_like, dim3(nwg), dim3(64), ldsb, st, flag, state, table, t, mode != 0 ? 1 : 0, xcc, err);
            }
            CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        uint32_t e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        printf("mode %d (%s): %.2f us per launch, err flags %u (1 = spin bound hit, 2 = stale predecessor state, 4 = XCC of a workgroup changed)\n", mode,
               mode == 0 ? "one stream, kernel barrier" : mode == 1 ? "two streams, per-wave flags" : "one stream + flags", best / L, e);
    }
    return 0;
}
