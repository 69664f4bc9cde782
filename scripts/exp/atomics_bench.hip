// Development micro-benchmark (not part of the product): throughput of global atomics onto ~1200 hot
// counters, as the tile-count / tile-cursor binning scheme would issue them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int STRIDE, bool RET>
__global__ void k_atomic(uint32_t* cnt, uint64_t* out, const uint32_t* base, int n, int tiles)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // Gaussian-like locality: instance i belongs to gaussian i/2, tiles adjacent
    uint32_t g = i >> 1;
    uint32_t t = (hash(g) % (tiles - 1)) + (i & 1);
    if (RET) { uint32_t slot = atomicAdd(&cnt[t * STRIDE], 1u); out[base[t] + slot] = ((uint64_t)i << 32) | t; }
    else atomicAdd(&cnt[t * STRIDE], 1u);
}
int main()
{
    const int n = 1210413, tiles = 1200;
    uint32_t *cnt, *base; uint64_t* out;
    CK(hipMalloc(&cnt, tiles * 64 * 4)); CK(hipMalloc(&base, tiles * 4)); CK(hipMalloc(&out, (size_t)n * 8 + 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<uint32_t> h(tiles * 64), hb(tiles);
    auto run = [&](auto kern, int stride, bool ret, const char* name) {
        float best = 1e9;
        for (int it = 0; it < 5; it++) {
            hipMemset(cnt, 0, tiles * 64 * 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3((n + 255) / 256), dim3(256), 0, 0, cnt, out, base, n, tiles);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-28s %8.1f us\n", name, best * 1e3);
        return 0;
    };
    run(k_atomic<1, false>, 1, false, "count stride1 noret");
    // build bases from counts
    CK(hipMemcpy(h.data(), cnt, tiles * 4, hipMemcpyDeviceToHost));
    uint32_t acc = 0; for (int t = 0; t < tiles; t++) { hb[t] = acc; acc += h[t]; }
    CK(hipMemcpy(base, hb.data(), tiles * 4, hipMemcpyHostToDevice));
    run(k_atomic<16, false>, 16, false, "count stride16(64B) noret");
    run(k_atomic<1, true>, 1, true, "cursor stride1 ret+scatter");
    run(k_atomic<16, true>, 16, true, "cursor stride16 ret+scatter");
    return 0;
}
