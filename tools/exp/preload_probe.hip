// Feasibility probe (not product code): does kernel-argument PRELOAD (-mllvm -amdgpu-kernarg-preload-count=N: the first N dwords of
// the argument block arrive in SGPRs with the wave instead of through an s_load) work on this GPU / firmware, and how many cycles
// after wave entry are the arguments usable with and without it?
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=14 -o preload_probe preload_probe.hip && ./preload_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct Big { uint64_t* out; const uint32_t* src; uint32_t pad[140]; uint32_t k; };

// arguments behind a 600-byte struct: never preloaded (beyond the first dwords of the struct itself)
__global__ void __launch_bounds__(64) by_struct(uint32_t dummy0, uint32_t dummy1, uint32_t dummy2, uint32_t dummy3, uint32_t d4, uint32_t d5, uint32_t d6, uint32_t d7,
                                                uint32_t d8, uint32_t d9, uint32_t d10, uint32_t d11, uint32_t d12, uint32_t d13, Big a) {
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t v = a.src[threadIdx.x + a.k];                 // address needs a.src and a.k: both behind the preloaded window
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    asm volatile("" :: "v"(v));
    if (threadIdx.x == 0) a.out[blockIdx.x] = t1 - t0;
}
// the same two arguments first in the list: inside the preload window
__global__ void __launch_bounds__(64) by_scalars(const uint32_t* src, uint32_t k, uint64_t* out, Big a) {
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t v = src[threadIdx.x + k];
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    asm volatile("" :: "v"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    const int nwg = 1024;
    uint64_t* out; uint32_t* src;
    CK(hipMalloc(&out, nwg * 8)); CK(hipMalloc(&src, 4096));
    CK(hipMemset(src, 0, 4096));
    Big a = {}; a.out = out; a.src = src; a.k = 3;
    std::vector<uint64_t> h(nwg);
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 20; rep++) {
            if (which == 0) hipLaunchKernelGGL(by_struct, dim3(nwg), dim3(64), 0, 0, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, a);
            else hipLaunchKernelGGL(by_scalars, dim3(nwg), dim3(64), 0, 0, (const uint32_t*)src, 3u, out, a);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), out, nwg * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("%s: cycles from wave entry until a load whose address needs the arguments has been ISSUED: min %llu median %llu p90 %llu\n",
               which == 0 ? "arguments behind the preload window (s_load)" : "arguments in the preload window", (unsigned long long)h[0],
               (unsigned long long)h[nwg / 2], (unsigned long long)h[nwg * 9 / 10]);
    }
    return 0;
}
