# Development helper: build activesplat_amd/libgsplat_hip_traceb.so -- the library with a TRACE variant of the backward blend (per wavefront: start /
# end on the 100 MHz clock, shader cycles in the scans, the rounds' set-up, phase A, phase B + gather and the flush, and their trip counts, in a
# __device__ array exported as gs_debug_bwd_trace) for scripts/exp/bwd_trace.py.  Not part of the product build.
# usage (repo root, after `make -C activesplat_amd/csrc`): bash scripts/exp/make_trace_bwd.sh
set -e
R=$PWD; T=$(mktemp -d)
python3 - "$R/activesplat_amd/csrc/blend.hip" "$R/activesplat_amd/csrc/blend_traceb_tmp.hip" <<'PY'
import sys
s = open(sys.argv[1]).read()
def sub(old, new, count=1):
    global s
    assert s.count(old) >= 1, old
    s = s.replace(old, new, count)
sub('namespace gs {', 'namespace gs {\n__device__ unsigned long long g_bwd_trace[14 * 32768];\n')
sub('''    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TileCtx c;
    const int nseg = FEW''', '''    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long cy[7] = {0, 0, 0, 0, 0, 0, 0}, nn[3] = {0, 0, 0};      // cycles: scan, round set-up, phase A, phase B + gather, flush, (spare); counts: scans, rounds, batches
    struct TraceEnd { unsigned long long t0, c0; unsigned long long* cy; unsigned long long* nn; unsigned w; int lane; __device__ ~TraceEnd() { if (lane == 0 && w < 32768) {
        unsigned long long* o = g_bwd_trace + 14 * (size_t)w;
        o[0] = t0; o[1] = wall_clock64(); o[2] = __builtin_readcyclecounter() - c0; for (int i = 0; i < 5; i++) o[3 + i] = cy[i]; for (int i = 0; i < 3; i++) o[8 + i] = nn[i]; o[11] = 1; o[12] = cy[5]; o[13] = cy[6]; } } }
        trace_end{(unsigned long long)wall_clock64(), (unsigned long long)__builtin_readcyclecounter(), cy, nn, blockIdx.x, (int)(threadIdx.x & 63)};
    TileCtx c;
    const int nseg = FEW''')
sub('''    if (pieces > 1 && piece > 0) {
        // (as late as possible''', '''    cy[5] = __builtin_readcyclecounter() - trace_end.c0;          // prologue up to the hand-over wait
    if (pieces > 1 && piece > 0) {
        // (as late as possible''')
sub('''        if (!handed) T = __builtin_nanf("");
    }''', '''        if (!handed) T = __builtin_nanf("");
    }
    nn[2] = 0; cy[6] = __builtin_readcyclecounter() - trace_end.c0 - cy[5];     // the wait itself (+ reading the state)''')
sub('''        const float4 q0 = r0, q1 = r1, q2 = r2;
        const uint32_t id_cur = id_next;
        id_next = id_next2;
        id_next2 = ch >= cmin_u + 2 ?''', '''        const unsigned long long c_s0 = __builtin_readcyclecounter(); nn[0]++;
        const float4 q0 = r0, q1 = r1, q2 = r2;
        const uint32_t id_cur = id_next;
        id_next = id_next2;
        id_next2 = ch >= cmin_u + 2 ?''')
sub('''        if (cnt == 0 || !(split || ch < cmin_u || (m_any != 0ull && cnt + n_hit + kRoundSlack > kWave))) continue;
''', '''        cy[0] += __builtin_readcyclecounter() - c_s0;
        if (cnt == 0 || !(split || ch < cmin_u || (m_any != 0ull && cnt + n_hit + kRoundSlack > kWave))) continue;
        unsigned long long c_r0 = __builtin_readcyclecounter(); nn[1]++;
''')
sub('''            // ---- phase A: up to kBT list positions, two per iteration ----
            const int tend = min(kBT, ntrips - t0);''', '''            // ---- phase A: up to kBT list positions, two per iteration ----
            { const unsigned long long c_ = __builtin_readcyclecounter(); if (t0 == 0) cy[1] += c_ - c_r0; c_r0 = c_; nn[2]++; }
            const int tend = min(kBT, ntrips - t0);''')
sub('''            __builtin_amdgcn_wave_barrier();
            // ---- phase B: lane = (half hb, position tb, row rb) ----''', '''            __builtin_amdgcn_wave_barrier();
            { const unsigned long long c_ = __builtin_readcyclecounter(); cy[2] += c_ - c_r0; c_r0 = c_; }
            // ---- phase B: lane = (half hb, position tb, row rb) ----''')
sub('''            __builtin_amdgcn_wave_barrier();
        }
        // flush: the staged records (dense slots''', '''            __builtin_amdgcn_wave_barrier();
            { const unsigned long long c_ = __builtin_readcyclecounter(); cy[3] += c_ - c_r0; c_r0 = c_; }
        }
        // flush: the staged records (dense slots''')
sub('''        __builtin_amdgcn_wave_barrier();
        cnt = 0;
    }''', '''        __builtin_amdgcn_wave_barrier();
        cy[4] += __builtin_readcyclecounter() - c_r0;
        cnt = 0;
    }''')
sub('''    if (pieces > 1 && piece < pieces - 1) {            // hand the state on''', '''    const unsigned long long c_e0 = __builtin_readcyclecounter();
    struct EpiEnd { unsigned long long c0; unsigned long long* cy; __device__ ~EpiEnd() { cy[4] = cy[4]; cy[0] = cy[0]; cy[6] |= (__builtin_readcyclecounter() - c0) << 32; } } epi_end{c_e0, cy};
    if (pieces > 1 && piece < pieces - 1) {            // hand the state on''')
sub('''            float* fls = s_m[wave][0];                          // 64 x 12 floats <= 2 planes''', '''            float* fls = s_m[wave][0];                          // 64 x 12 floats <= 2 planes
            { bool z = true; for (int k_ = 0; k_ < 10; k_++) z = z && racc[k_] == 0.0f;
              nn[1] += ((unsigned long long)__popcll(__ballot(lane < cnt && z)) << 40) | ((unsigned long long)cnt << 20); }''')
s += '''
extern "C" int gs_debug_bwd_trace(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gs::g_bwd_trace), (size_t)n * 14 * 8, 0, hipMemcpyDeviceToHost);
}
'''
open(sys.argv[2], 'w').write(s)
PY
cd $R/activesplat_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics -fno-slp-vectorize -c blend_traceb_tmp.hip -o $T/blend_traceb.o
rm -f blend_traceb_tmp.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgsplat_hip_traceb.so api.o preprocess.o preprocess_bwd.o binning.o tilebin.o $T/blend_traceb.o adam.o compact.o loss.o activate.o grow.o stats.o densify.o rows.o sort_rocprim.o
echo built activesplat_amd/libgsplat_hip_traceb.so
