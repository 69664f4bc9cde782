// Development experiment: which (XCC, SE, CU) does workgroup b of a 1200 x 256-thread launch land on?
// hipcc --offload-arch=gfx950 -O2 -o /tmp/dispatch_map scripts/exp/dispatch_map.hip && /tmp/dispatch_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256) void probe(uint32_t* out, int spin)
{
    __shared__ float pad[3300];                    // ~13 KB like the blend kernels
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = threadIdx.x;
    for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;      // keep the block resident for a while
    pad[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = (xcc & 0xf) | (pad[17] > 1e30f ? 16 : 0); }
}
int main()
{
    const int nb = 1200;
    uint32_t* d; hipMalloc(&d, nb * 8);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d, 20000);
        hipDeviceSynchronize();
    }
    std::vector<uint32_t> h(nb * 2); hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> per_cu;
    for (int b = 0; b < nb; b++) {
        const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        const uint32_t key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_cu[key]++;
        if (b < 48 || (b % 100) == 0) printf("block %4d -> xcc %u se %u sh %u cu %2u   (b%%8=%d)\n", b, xcc, se, sh, cu, b % 8);
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("distinct CUs used: %zu\n", per_cu.size());
    for (auto& kv : hist) printf("  %d CUs hold %d blocks\n", kv.second, kv.first);
    return 0;
}
