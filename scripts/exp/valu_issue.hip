// valu_issue.hip -- what one wave64 vector instruction costs on gfx950 (issue slots per SIMD), by opcode class.
// Standalone micro-benchmark behind DESIGN.md's "VALU issue" roofline of the blend kernels:
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
// Every kernel runs ITER iterations of a block of 64 independent instructions of one class (8 chains x 8) per wave;
// grid = 256 CUs x (waves per SIMD) so that every SIMD holds exactly `w` waves.  Reported: G wave-instructions/s of
// the whole chip and the cycles one instruction occupies a SIMD (at the 2.4 GHz nominal clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITER = 2000;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY8(STMT) REP8(STMT) REP8(STMT) REP8(STMT) REP8(STMT) REP8(STMT) REP8(STMT) REP8(STMT) REP8(STMT)

typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(float* out, float seed)
{
    float a[8];
    v2f p[8];
    double dd[8];
    __shared__ __attribute__((aligned(16))) float lds[256 * 20];
    const int tid = threadIdx.x;
    for (int i = 0; i < 8; i++) { a[i] = seed + i + tid * 1e-3f; p[i] = v2f{a[i], a[i] * 0.5f}; dd[i] = (double)a[i]; }
    const double dc = 1.0 + (double)seed * 1e-12;
    for (int i = tid; i < 256 * 20; i += 256) lds[i] = seed;
    __syncthreads();
    const float c1 = 0.999f + seed * 1e-9f, c2 = 1e-6f;
    const v2f pc1 = {c1, c1}, pc2 = {c2, c2};
    float* my = lds + tid * 20;
    float* mine17 = lds + (tid & 63) * 17 + (tid >> 6) * 1100;
    unsigned long long acc_mask = 0; const unsigned long long smask = __ballot((tid & 1) != 0 && seed > 0.f);
    for (int it = 0; it < ITER; it++) {
        if (KIND == 0) {            // v_fma_f32
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
            BODY8(S)
#undef S
        } else if (KIND == 1) {     // v_pk_fma_f32
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc1), "v"(pc2));
            BODY8(S)
#undef S
        } else if (KIND == 2) {     // v_pk_mul_f32
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc1));
            BODY8(S)
#undef S
        } else if (KIND == 3) {     // v_pk_add_f32
#define S(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc2));
            BODY8(S)
#undef S
        } else if (KIND == 4) {     // v_add_f32_dpp row_ror:8
#define S(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            BODY8(S)
#undef S
        } else if (KIND == 5) {     // v_exp_f32
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            BODY8(S)
#undef S
        } else if (KIND == 6) {     // v_rcp_f32
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            BODY8(S)
#undef S
        } else if (KIND == 7) {     // v_cndmask_b32 (vcc)
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c1));
            BODY8(S)
#undef S
        } else if (KIND == 8) {     // v_mul_f32
#define S(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            BODY8(S)
#undef S
        } else if (KIND == 9) {     // v_mov_b32 dpp quad_perm
#define S(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            BODY8(S)
#undef S
        } else if (KIND == 10) {    // mix: 3 fma : 1 exp
#define S(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(a[i]), "+v"(p[i].x) : "v"(c1), "v"(c2)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[i].y) : "v"(c1), "v"(c2));
            REP8(S) REP8(S)
#undef S
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            REP8(S) REP8(S)
#undef S
        } else if (KIND == 11) {    // ds_read_b128 (conflict-free, stride 20 dwords)
#define S(i) { float4 v = *reinterpret_cast<float4*>(my + ((i & 3) << 2)); asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
            BODY8(S)
#undef S
        } else if (KIND == 12) {    // ds_write_b32 (stride 1)
#define S(i) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"((int)(tid * 4)), "v"(a[i]), "n"(i * 1024) : "memory");
            BODY8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 13) {    // ds_add_f32 distinct addresses (LDS float atomic, no return)
#define S(i) asm volatile("ds_add_f32 %0, %1 offset:%2" :: "v"((int)(tid * 4)), "v"(c2), "n"(i * 1024) : "memory");
            BODY8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 14) {    // ds_read_b32 stride 17 (lane = row)
#define S(i) { float v = mine17[i]; asm volatile("" :: "v"(v)); }
            BODY8(S)
#undef S
        } else if (KIND == 15) {    // v_fmac_f32 with 2 independent chains only (dependent-issue check)
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 1]) : "v"(c1), "v"(c2));
            BODY8(S)
#undef S
        } else if (KIND == 16) {    // ds_write_b128
#define S(i) { typedef float v4f __attribute__((ext_vector_type(4))); v4f w4 = {a[i], a[(i + 1) & 7], a[(i + 2) & 7], a[(i + 3) & 7]}; \
              asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"((int)(tid * 16)), "v"(w4), "n"((i & 3) * 4096) : "memory"); }
            BODY8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 17) {    // ds_bpermute_b32
#define S(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"((int)((tid * 4 + 68) & 255)));
            BODY8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 18) {    // v_cndmask_b32_e64 with an SGPR-pair mask that no VALU wrote
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "s"(smask));
            BODY8(S)
#undef S
        } else if (KIND == 19) {    // v_cmp (-> vcc) + v_cndmask (vcc) pairs: 32 + 32 instructions
#define S(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c1) : "vcc");
            REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
        } else if (KIND == 20) {    // v_cmp_gt_f32_e64 -> SGPR pair only
#define S(i) { unsigned long long m_; asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m_) : "v"(a[i]), "v"(c1)); acc_mask ^= m_; }
            BODY8(S)
#undef S
        } else if (KIND == 21) {    // v_max_f32
#define S(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            BODY8(S)
#undef S
        } else if (KIND == 22) {    // v_fmac_f32 (VOP2)
#define S(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
            BODY8(S)
#undef S
        } else if (KIND == 23) {    // v_mov_b32
#define S(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(c1));
            BODY8(S)
#undef S
        } else if (KIND == 24) {    // v_add_f32
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
            BODY8(S)
#undef S
        } else if (KIND == 25) {    // s_nop 1 between v_mul (DPP hazard idiom): 32 + 32
#define S(i) asm volatile("s_nop 1\n\tv_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
        } else if (KIND == 26) {    // v_cndmask_b32 vop2 with vcc written once per iteration by a v_cmp
            asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[0]), "v"(c1) : "vcc");
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c1) : );
            BODY8(S)
#undef S
        } else if (KIND == 27) {    // v_mul_f32 by an SGPR operand
#define S(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
            BODY8(S)
#undef S
        } else if (KIND == 28) {    // v_readlane_b32 + s use
#define S(i) { unsigned r_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(r_) : "v"(a[i])); acc_mask += r_; }
            BODY8(S)
#undef S
        } else if (KIND == 30) {    // ds_add_u32, distinct addresses (integer LDS atomic, no return)
#define S(i) asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"((int)(tid * 4)), "v"(1), "n"(i * 1024) : "memory");
            BODY8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 31) {    // ds_add_rtn_u32, distinct addresses
#define S(i) { int r_; asm volatile("ds_add_rtn_u32 %0, %1, %2 offset:%3" : "=v"(r_) : "v"((int)(tid * 4)), "v"(1), "n"(i * 1024) : "memory"); acc_mask += r_; }
            BODY8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 32) {    // ds_add_u32, random-ish addresses within 1200 bins (histogram pattern)
#define S(i) asm volatile("ds_add_u32 %0, %1" :: "v"((int)((((tid * 2654435761u) >> (7 + i)) % 1200u) * 4)), "v"(1) : "memory");
            BODY8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 33) {    // v_min_f64
#define S(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(dc));
            BODY8(S)
#undef S
        } else if (KIND == 34) {    // v_fma_f64
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd[i]) : "v"(dc));
            BODY8(S)
#undef S
        } else if (KIND == 35) {    // v_cmp_lt_u64 -> sgpr pair
#define S(i) { unsigned long long m_; asm volatile("v_cmp_lt_u64_e64 %0, %1, %2" : "=s"(m_) : "v"(dd[i]), "v"(dc)); acc_mask ^= m_; }
            BODY8(S)
#undef S
        } else if (KIND == 36) {    // v_permlane32_swap_b32
#define S(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
            BODY8(S)
#undef S
        } else if (KIND == 37) {    // v_xor_b32
#define S(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            BODY8(S)
#undef S
        } else if (KIND == 29) {    // v_fma_f32 with operands spread over register banks: a[i] = a[i] * a[(i+1)&7] + a[(i+2)&7]
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            BODY8(S)
#undef S
        }
    }
    if (acc_mask == 0x123456789ull) out[1] = 1.f;
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)dd[i];
    if (s == 12345.678f) out[0] = s + lds[tid];
}

const char* kNames[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_add_f32_dpp(row_ror)", "v_exp_f32", "v_rcp_f32",
                        "v_cndmask_b32", "v_mul_f32", "v_mov_b32_dpp(quad)", "mix 3fma:1exp", "ds_read_b128", "ds_write_b32", "ds_add_f32",
                        "ds_read_b32 s17", "v_fma_f32 2 chains", "ds_write_b128", "ds_bpermute_b32",
                        "v_cndmask_e64 sgpr mask", "v_cmp+v_cndmask vcc (pairs)", "v_cmp_e64 -> sgpr", "v_max_f32", "v_fmac_f32", "v_mov_b32", "v_add_f32",
                        "s_nop1+v_mul (pairs)", "v_cndmask vcc (cmp hoisted)", "v_mul_f32 sgpr src", "v_readlane_b32", "v_fma_f32 mixed banks",
                        "ds_add_u32", "ds_add_rtn_u32", "ds_add_u32 histogram",
                        "v_min_f64", "v_fma_f64", "v_cmp_lt_u64 -> sgpr", "v_permlane32_swap_b32", "v_xor_b32"};

template <int KIND>
void run(float* d_out, int waves_per_simd)
{
    const int grid = 256 * waves_per_simd;      // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(256), 0, 0, d_out, 1.0f);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(256), 0, 0, d_out, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double per_block = KIND == 10 ? 64.0 : 64.0;
    const double insts = (double)grid * 4 * ITER * per_block;          // wave-instructions
    const double rate = insts / (best * 1e-3);
    const double cyc = 1024.0 * 2.4e9 / rate;                         // SIMD-cycles per wave-instruction at 2.4 GHz
    printf("%-26s waves/SIMD %d : %8.1f G wave-inst/s  %6.2f cyc/inst/SIMD  (%.3f ms)\n", kNames[KIND], waves_per_simd, rate / 1e9, cyc, best);
    fflush(stdout);
}

template <int KIND>
void sweep(float* d_out) { run<KIND>(d_out, 1); run<KIND>(d_out, 5); run<KIND>(d_out, 8); }

int main(int argc, char** argv)
{
    float* d_out;
    CHECK(hipMalloc(&d_out, 1024));
    const bool all = argc < 2;
    if (all) {
    sweep<0>(d_out); sweep<1>(d_out); sweep<2>(d_out); sweep<3>(d_out); sweep<4>(d_out); sweep<5>(d_out); sweep<6>(d_out); sweep<7>(d_out);
    sweep<8>(d_out); sweep<9>(d_out); sweep<10>(d_out); sweep<11>(d_out); sweep<12>(d_out); sweep<13>(d_out); sweep<14>(d_out);
    sweep<15>(d_out); sweep<16>(d_out); sweep<17>(d_out);
    }
    if (argc < 3) {
    sweep<18>(d_out); sweep<19>(d_out); sweep<20>(d_out); sweep<21>(d_out); sweep<22>(d_out); sweep<23>(d_out); sweep<24>(d_out);
    sweep<25>(d_out); sweep<26>(d_out); sweep<27>(d_out); sweep<28>(d_out); sweep<29>(d_out);
    }
    sweep<30>(d_out); sweep<31>(d_out); sweep<32>(d_out);
    sweep<33>(d_out); sweep<34>(d_out); sweep<35>(d_out); sweep<36>(d_out); sweep<37>(d_out);
    return 0;
}
