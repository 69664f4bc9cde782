// hip_runtime.h -- TEST-ONLY functional emulator of the HIP device programming model on the host.
//
// Purpose: this container has no GPU and gpurun minutes are scarce, so the *unmodified* kernel
// sources under activesplat_amd/csrc are also compiled with g++ against this header
// (g++ -I tests/hipemu -x c++ ...), which lets the CPU test-suite execute the real kernel logic
// (LDS staging, wave-64 ballots/shuffles, barriers, atomics) block by block.  It is a debugging aid
// for tests/ only: nothing in activesplat_amd/ includes, links or falls back to it, and the product
// loader (activesplat_amd/_lib.py) only ever opens the hipcc-built libgsplat_hip.so.
//
// Model: one workgroup = one OS thread running blockDim fibers (a register-only context switch on x86-64, ucontext elsewhere).  A fiber runs until it
// reaches __syncthreads(), a wave collective, or the end of the kernel.  Wave collectives
// (__shfl*, __ballot, ...) rendezvous the live lanes of a 64-lane wavefront.  Workgroups run in
// parallel over OpenMP threads.  Not modelled: memory-ordering subtleties, LDS bank conflicts,
// timing.
#pragma once
#include <ucontext.h>
#include <time.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __constant__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIPEMU 1

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToDevice 3

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }

namespace hipemu {
enum St { READY, AT_BARRIER, AT_WAVE, DONE };
struct Fiber {
#if defined(__x86_64__) && !defined(HIPEMU_UCONTEXT)
    void* sp;                        // saved stack pointer (hipemu.cpp's register-only switch: no signal-mask system call per yield)
#else
    ucontext_t ctx;
#endif
    St st;
    dim3 tid;
    unsigned long long wave_in;      // payload deposited for a wave collective
    int wave_op;                     // tag of the collective being waited on
    char* stack;
};
struct Block {
    std::vector<Fiber> fibers;
#if defined(__x86_64__) && !defined(HIPEMU_UCONTEXT)
    void* sched_sp;
#else
    ucontext_t sched;
#endif
    int cur = -1;
    dim3 bid, bdim, gdim;
    unsigned long long wave_out[16][64];    // snapshot per wave (<=1024 threads)
    unsigned long long wave_mask[16];       // active lanes of the last released collective
    long long barrier_count = 0;            // for __syncthreads_count/or
    long long barrier_acc = 0;
    const std::function<void()>* body = nullptr;
};
extern thread_local Block* g_blk;
static inline Fiber& cur() { return g_blk->fibers[g_blk->cur]; }
void yield_to_sched();
void run_grid(dim3 grid, dim3 block, const std::function<void()>& body);
static inline unsigned lane_id() { auto& f = cur(); return (f.tid.x + f.tid.y * g_blk->bdim.x + f.tid.z * g_blk->bdim.x * g_blk->bdim.y) & 63; }
static inline unsigned wave_id() { auto& f = cur(); return (f.tid.x + f.tid.y * g_blk->bdim.x + f.tid.z * g_blk->bdim.x * g_blk->bdim.y) >> 6; }
// deposit v, wait for the wave, return pointer to the 64-entry snapshot; *mask = participating lanes
static inline const unsigned long long* wave_exchange(unsigned long long v, int op, unsigned long long* mask)
{
    Fiber& f = cur();
    f.wave_in = v; f.wave_op = op; f.st = AT_WAVE;
    unsigned w = wave_id();
    yield_to_sched();
    if (mask) *mask = g_blk->wave_mask[w];
    return g_blk->wave_out[w];
}
template <class T> static inline unsigned long long pack(T v) { unsigned long long u = 0; static_assert(sizeof(T) <= 8, ""); std::memcpy(&u, &v, sizeof(T)); return u; }
template <class T> static inline T unpack(unsigned long long u) { T v; std::memcpy(&v, &u, sizeof(T)); return v; }
}  // namespace hipemu

#define threadIdx (hipemu::cur().tid)
#define blockIdx (hipemu::g_blk->bid)
#define blockDim (hipemu::g_blk->bdim)
#define gridDim (hipemu::g_blk->gdim)
#define warpSize 64

// ---- barriers --------------------------------------------------------------------------------
static inline void __syncthreads()
{
    hipemu::cur().st = hipemu::AT_BARRIER;
    hipemu::yield_to_sched();
}
static inline int __syncthreads_count(int pred)
{
    // two-phase: accumulate, barrier, read, barrier (so the accumulator can be reset safely)
    hipemu::g_blk->barrier_acc += pred ? 1 : 0;
    __syncthreads();
    int r = (int)hipemu::g_blk->barrier_count;
    __syncthreads();
    return r;
}
static inline int __syncthreads_or(int pred) { return __syncthreads_count(pred) != 0; }
static inline int __syncthreads_and(int pred) { return __syncthreads_count(!pred) == 0; }

// ---- wave collectives --------------------------------------------------------------------------
template <class T> static inline T __shfl(T v, int src, int width = 64)
{
    unsigned lane = hipemu::lane_id();
    const unsigned long long* o = hipemu::wave_exchange(hipemu::pack(v), 1, nullptr);
    int base = (int)lane & ~(width - 1);
    return hipemu::unpack<T>(o[base + (src & (width - 1))]);
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64)
{
    unsigned lane = hipemu::lane_id();
    const unsigned long long* o = hipemu::wave_exchange(hipemu::pack(v), 2, nullptr);
    int base = (int)lane & ~(width - 1);
    int s = ((int)lane ^ m);
    if (s >= base + width || s < base) s = (int)lane;
    return hipemu::unpack<T>(o[s]);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64)
{
    unsigned lane = hipemu::lane_id();
    const unsigned long long* o = hipemu::wave_exchange(hipemu::pack(v), 3, nullptr);
    int base = (int)lane & ~(width - 1);
    int s = (int)lane - (int)d;
    if (s < base) s = (int)lane;
    return hipemu::unpack<T>(o[s]);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64)
{
    unsigned lane = hipemu::lane_id();
    const unsigned long long* o = hipemu::wave_exchange(hipemu::pack(v), 4, nullptr);
    int base = (int)lane & ~(width - 1);
    int s = (int)lane + (int)d;
    if (s >= base + width) s = (int)lane;
    return hipemu::unpack<T>(o[s]);
}
static inline unsigned long long __ballot(int pred)
{
    unsigned long long mask;
    const unsigned long long* o = hipemu::wave_exchange(pred ? 1ull : 0ull, 5, &mask);
    unsigned long long r = 0;
    for (int i = 0; i < 64; i++) if (((mask >> i) & 1) && o[i]) r |= 1ull << i;
    return r;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred)
{
    unsigned long long mask;
    const unsigned long long* o = hipemu::wave_exchange(pred ? 1ull : 0ull, 6, &mask);
    for (int i = 0; i < 64; i++) if (((mask >> i) & 1) && !o[i]) return 0;
    return 1;
}
template <class T> static inline T hipemu_readfirstlane(T v)
{
    unsigned long long mask;
    const unsigned long long* o = hipemu::wave_exchange(hipemu::pack(v), 7, &mask);
    int first = __builtin_ctzll(mask);
    return hipemu::unpack<T>(o[first]);
}
template <class T> static inline T hipemu_readlane(T v, int l)
{
    const unsigned long long* o = hipemu::wave_exchange(hipemu::pack(v), 8, nullptr);
    return hipemu::unpack<T>(o[l & 63]);
}
static inline void hipemu_wave_barrier() { hipemu::wave_exchange(0ull, 12, nullptr); }
#define __builtin_amdgcn_wave_barrier() hipemu_wave_barrier()
#define __builtin_amdgcn_readfirstlane(v) hipemu_readfirstlane(v)
#define __builtin_amdgcn_readlane(v, l) hipemu_readlane(v, l)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
static inline unsigned hipemu_mbcnt_lo(unsigned mask, unsigned base)
{
    const unsigned l = hipemu::lane_id();
    return base + (unsigned)__builtin_popcount(mask & (l >= 32 ? 0xffffffffu : ((1u << l) - 1u)));
}
static inline unsigned hipemu_mbcnt_hi(unsigned mask, unsigned base)
{
    const unsigned l = hipemu::lane_id();
    return base + (unsigned)__builtin_popcount(l <= 32 ? 0u : (mask & ((1u << (l - 32)) - 1u)));
}
#define __builtin_amdgcn_mbcnt_lo(a, b) hipemu_mbcnt_lo(a, b)
#define __builtin_amdgcn_mbcnt_hi(a, b) hipemu_mbcnt_hi(a, b)
#define __lane_id() (hipemu::lane_id())
// the lane's own bit of a wave-uniform mask (v_cndmask with the mask as its condition on the device)
#define __builtin_amdgcn_inverse_ballot_w64(m) (((((unsigned long long)(m)) >> hipemu::lane_id()) & 1ull) != 0ull)

// DPP emulation (the subset of controls the kernels use); semantics of v_mov_b32_dpp with
// old=`old`, bound_ctrl as given, row_mask/bank_mask honoured.
static inline int hipemu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    unsigned lane = hipemu::lane_id();
    const unsigned long long* o = hipemu::wave_exchange(hipemu::pack(src), 9, nullptr);
    int row = lane >> 4, in_row = lane & 15, s = -1;
    bool valid = true;
    if (ctrl >= 0x000 && ctrl <= 0x0FF) { int sel = (ctrl >> (2 * (lane & 3))) & 3; s = (lane & ~3u) + sel; }
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { int n = ctrl & 15; if (in_row + n > 15) valid = false; else s = lane + n; }           // row_shl
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { int n = ctrl & 15; if (in_row - n < 0) valid = false; else s = lane - n; }            // row_shr
    else if (ctrl >= 0x121 && ctrl <= 0x12F) { int n = ctrl & 15; s = (lane & ~15u) + ((in_row - n) & 15); }                         // row_ror
    else if (ctrl == 0x130) { if (lane + 1 > 63) valid = false; else s = lane + 1; }                                                 // wave_shl1
    else if (ctrl == 0x134) { s = (lane + 1) & 63; }                                                                                  // wave_rol1
    else if (ctrl == 0x138) { if (lane == 0) valid = false; else s = lane - 1; }                                                     // wave_shr1
    else if (ctrl == 0x13C) { s = (lane - 1) & 63; }                                                                                  // wave_ror1
    else if (ctrl == 0x140) { s = (lane & ~15u) + (15 - in_row); }                                                                    // row_mirror
    else if (ctrl == 0x141) { s = (lane & ~7u) + (7 - (lane & 7)); }                                                                  // row_half_mirror
    else if (ctrl == 0x142) { if (row == 0) valid = false; else s = (row - 1) * 16 + 15; }                                            // row_bcast15
    else if (ctrl == 0x143) { if (row < 2) valid = false; else s = 31; }                                                              // row_bcast31
    else { std::fprintf(stderr, "hipemu: unsupported dpp ctrl 0x%x\n", ctrl); std::abort(); }
    bool en = ((row_mask >> row) & 1) && ((bank_mask >> (in_row >> 2)) & 1);
    if (ctrl == 0x142 || ctrl == 0x143) { /* bcast: lanes of rows not receiving keep old */ }
    if (!en) return old;
    if (!valid) return bound_ctrl ? 0 : old;
    return hipemu::unpack<int>(o[s]);
}
typedef unsigned hipemu_v2u __attribute__((vector_size(8)));
// v_permlane32_swap: lanes 32-63 of a swap with lanes 0-31 of b.  Returns {new a, new b}.
static inline hipemu_v2u hipemu_permlane32_swap(unsigned a, unsigned b)
{
    unsigned lane = hipemu::lane_id();
    const unsigned long long* o = hipemu::wave_exchange(((unsigned long long)b << 32) | a, 10, nullptr);
    hipemu_v2u r;
    if (lane < 32) { r[0] = a; r[1] = (unsigned)(o[lane + 32] & 0xffffffffu); }          // b_lo <- a_hi
    else { r[0] = (unsigned)(o[lane - 32] >> 32); r[1] = b; }                             // a_hi <- b_lo
    return r;
}
// v_permlane16_swap: odd rows of a swap with even rows of b.
static inline hipemu_v2u hipemu_permlane16_swap(unsigned a, unsigned b)
{
    unsigned lane = hipemu::lane_id();
    const unsigned long long* o = hipemu::wave_exchange(((unsigned long long)b << 32) | a, 11, nullptr);
    hipemu_v2u r;
    if (((lane >> 4) & 1) == 0) { r[0] = a; r[1] = (unsigned)(o[lane + 16] & 0xffffffffu); }  // b even row <- a odd row
    else { r[0] = (unsigned)(o[lane - 16] >> 32); r[1] = b; }                                  // a odd row <- b even row
    return r;
}
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipemu_permlane32_swap(a, b)
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) hipemu_permlane16_swap(a, b)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu_update_dpp(old, src, ctrl, rm, bm, bc)
#define __builtin_amdgcn_mov_dpp(src, ctrl, rm, bm, bc) hipemu_update_dpp(0, src, ctrl, rm, bm, bc)
// ---- scalar helpers ----------------------------------------------------------------------------
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float hipemu_expf(float x) { return expf(x); }
static inline float hipemu_exp2f(float x) { return exp2f(x); }
static inline float hipemu_logf(float x) { return logf(x); }
#define __expf hipemu_expf
#define __exp2f hipemu_exp2f
#define __logf hipemu_logf
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
using std::max;
using std::min;

// ---- atomics -----------------------------------------------------------------------------------
static inline float atomicAdd(float* p, float v)
{
    unsigned* up = (unsigned*)p; unsigned old = __atomic_load_n(up, __ATOMIC_RELAXED), nw;
    do { float f; std::memcpy(&f, &old, 4); f += v; std::memcpy(&nw, &f, 4); } while (!__atomic_compare_exchange_n(up, &old, nw, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    float r; std::memcpy(&r, &old, 4); return r;
}
static inline float unsafeAtomicAdd(float* p, float v) { return atomicAdd(p, v); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
static inline int atomicMax(int* p, int v) { int o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
#define __HIP_MEMORY_SCOPE_AGENT 0
template <class T> static inline T hipemu_atomic_load(const T* p, int order) { T v; __atomic_load(p, &v, order); return v; }
template <class T, class U> static inline void hipemu_atomic_store(T* p, U val, int order) { T v = (T)val; __atomic_store(p, &v, order); }
#define __hip_atomic_load(p, order, scope) hipemu_atomic_load(p, order)
#define __hip_atomic_store(p, v, order, scope) hipemu_atomic_store(p, v, order)
#define __HIP_MEMORY_SCOPE_SYSTEM 1
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, order)
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or(p, v, order)
// a waiting wavefront on the device sleeps ~64 x clocks per s_sleep; here the poller gives its host thread away for a moment, so that a bounded
// poll loop of the kernels (2^21..2^22 polls) stays a bound of seconds, as on the device, instead of milliseconds
static inline void hipemu_sleep() { struct timespec ts = {0, 2000}; nanosleep(&ts, nullptr); }
#define __builtin_amdgcn_s_sleep(x) hipemu_sleep()
#define GS_WAIT_VMEM() ((void)0)
static inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- runtime API subset --------------------------------------------------------------------------
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memmove(d, s, n); return 0; }
static inline hipError_t hipHostGetDevicePointer(void** dp, void* hp, unsigned) { *dp = hp; return 0; }   // host memory IS device memory here
#define hipHostMallocMapped 2u
#define hipHostMallocPortable 1u
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n); return *p ? 0 : 2; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return *p ? 0 : 2; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
typedef void* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }

// v_min_f64 / v_max_f64 of the tile sort (inline assembly in the device build)
#define GS_MIN_F64(a, b) std::fmin(a, b)
#define GS_MAX_F64(a, b) std::fmax(a, b)

template <class K, class... A>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, A... args)
{
    std::function<void()> body = [=]() { kernel(args...); };
    hipemu::run_grid(grid, block, body);
}
