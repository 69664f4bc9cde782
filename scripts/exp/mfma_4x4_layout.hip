// Development check: operand layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (16 independent 4x4 outer products per instruction).
// Hypothesis: lane l = 4 b + i holds A_b[i]; lane l = 4 b + j holds B_b[j]; D register i of lane 4 b + j holds D_b[i][j].
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 scripts/exp/mfma_4x4_layout.hip -o /tmp/mfma_layout && /tmp/mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d)
{
    const int l = threadIdx.x;
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l] * 0.5f, b[l], c, 0, 0, 0);     // accumulation: D = 1.5 a b
    for (int i = 0; i < 4; i++) d[l * 4 + i] = c[i];
}
int main()
{
    float ha[64], hb[64], hd[256], *a, *b, *d;
    for (int l = 0; l < 64; l++) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3 * l; }
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 4; i++) {
            const int blk = l / 4, j = l % 4;
            const float want = 1.5f * ha[blk * 4 + i] * hb[blk * 4 + j];
            if (hd[l * 4 + i] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, i, hd[l * 4 + i], want); bad++; }
        }
    printf("mfma_f32_4x4x1_16b layout: %d mismatches of 256\n", bad);
    return bad != 0;
}
