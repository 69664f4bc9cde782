// order_probe.cpp -- self-test of the emulator's HIPEMU_ORDER modes (tests/test_hipemu_orders.py): a kernel that is MISSING a barrier in a way the
// natural fibre order 0, 1, 2, ... hides (every thread reads what the thread before it wrote), and the same kernel with the barrier.
#include <hip/hip_runtime.h>

template <bool BARRIER>
__global__ void neighbour_kernel(int* out)
{
    __shared__ int s[256];
    const int t = threadIdx.x;
    s[t] = t + 1;
    if (BARRIER) __syncthreads();
    out[t] = s[(t + 255) & 255];          // the value thread t - 1 stored
}

extern "C" int probe(int barrier, int* out)
{
    for (int i = 0; i < 256; i++) out[i] = -1;
    if (barrier) hipLaunchKernelGGL((neighbour_kernel<true>), dim3(1), dim3(256), 0, 0, out);
    else hipLaunchKernelGGL((neighbour_kernel<false>), dim3(1), dim3(256), 0, 0, out);
    return 0;
}
