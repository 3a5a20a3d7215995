// What ONE hand-off timeout costs (tools/ubench_src; build: hipcc --offload-arch=gfx950 -O3 -I las_pytorch_amd/csrc): a single wave runs the
// library's bounded spin (persist_common.h::spin_expired, PS_SPIN_LIMIT) on a word nobody ever writes, exactly as a consumer whose
// producer workgroup is not resident would, and the elapsed time is the dead time before the step is re-run on the generic kernels.
#include "persist_common.h"
#include <stdio.h>
using namespace las;
__global__ void k(unsigned* word, unsigned* err, unsigned long long* ticks) {
    unsigned spins = 0;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != PS_SENT) break;
        if (spin_expired(spins, err, 0xDEAD00FFu)) break;
    }
    if (threadIdx.x == 0) { ticks[0] = wall_clock64() - t0; ticks[1] = spins; }
}
int main() {
    unsigned *word, *err; unsigned long long* ticks;
    (void)hipMalloc(&word, 4); (void)hipMalloc(&err, 4); (void)hipMalloc(&ticks, 16);
    (void)hipMemset(word, 0xFF, 4); (void)hipMemset(err, 0, 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, word, err, ticks);
    unsigned long long h[2]; unsigned e = 0;
    (void)hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
    printf("bounded spin: %llu spins, %.1f ms until the error word is raised (0x%08x)\n", h[1], h[0] / 100.0 / 1000.0, e);
    return 0;
}
