# Development helper: build activesplat_amd/libgsplat_hip_tracef.so -- the library with a TRACE variant of the streams FORWARD blend (per wavefront:
# start / end timestamps of the 100 MHz clock, hardware id (XCC, SE, CU, SIMD), chunks fetched and chunks blended, in a __device__ array exported as
# gs_debug_fwd_trace) for scripts/exp/fwd_trace.py.  Not part of the product build.  usage (repo root, after `make -C activesplat_amd/csrc`):
# bash scripts/exp/make_trace_fwd.sh
set -e
R=$PWD; T=$(mktemp -d)
python3 - "$R/activesplat_amd/csrc/blend.hip" "$R/activesplat_amd/csrc/blend_tracef_tmp.hip" <<'PY'
import sys
s = open(sys.argv[1]).read()
def sub(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)
sub('namespace gs {', 'namespace gs {\n__device__ unsigned long long g_fwd_trace[10 * 32768];\n')
sub('''    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // side job (SEG = 0 only)''', '''    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long n_scans = 0, n_rounds = 0, c_trips = 0, n_trips = 0, c_wait = 0; const unsigned long long c_start = __builtin_readcyclecounter();
    struct TraceEnd { unsigned long long t0; unsigned long long& ns; unsigned long long& nr; unsigned long long& ct; unsigned long long& nt; unsigned long long& cw; unsigned long long c0; unsigned w; int lane; __device__ ~TraceEnd() { if (lane == 0 && w < 32768) {
        uint32_t hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_fwd_trace[10 * w] = t0; g_fwd_trace[10 * w + 1] = wall_clock64(); g_fwd_trace[10 * w + 2] = hw; g_fwd_trace[10 * w + 3] = xcc & 0xf; g_fwd_trace[10 * w + 4] = ns; g_fwd_trace[10 * w + 5] = nr; g_fwd_trace[10 * w + 6] = ct; g_fwd_trace[10 * w + 7] = nt; g_fwd_trace[10 * w + 8] = __builtin_readcyclecounter() - c0; g_fwd_trace[10 * w + 9] = cw; } } }
        trace_end{(unsigned long long)wall_clock64(), n_scans, n_rounds, c_trips, n_trips, c_wait, c_start, SEG == 0 ? blockIdx.x * NW + (unsigned)(threadIdx.x >> 6) : 0xffffffffu, (int)(threadIdx.x & 63)};
    // side job (SEG = 0 only)''')
sub('''            const float4 q0 = r0, q1 = r1, q2 = r2;
            const uint32_t id_cur = id_next;
            id_next = id_next2;
            id_next2 = base + 128u''', '''            n_scans++;
            const float4 q0 = r0, q1 = r1, q2 = r2;
            const uint32_t id_cur = id_next;
            id_next = id_next2;
            id_next2 = base + 128u''')
sub('''            stage_record(s0, s1, s2, lane, q0, q1, q2, id_cur);
            // per-stream lists''', '''            n_rounds++;
            stage_record(s0, s1, s2, lane, q0, q1, q2, id_cur);
            // per-stream lists''')
sub('''            uint32_t jj2_next = *reinterpret_cast<const uint16_t*>(my_list);                // two list entries per read''', '''            const unsigned long long c_t0 = __builtin_readcyclecounter(); n_trips += (unsigned long long)((ntrips + 1) / 2);
            uint32_t jj2_next = *reinterpret_cast<const uint16_t*>(my_list);                // two list entries per read''')
sub('''            last = lj >= 0 ? base + (uint32_t)lj + 1u : last;''', '''            c_trips += __builtin_readcyclecounter() - c_t0;
            last = lj >= 0 ? base + (uint32_t)lj + 1u : last;''')
sub('''            const unsigned long long going = SEG == 1 ? ~0ull : ~__ballot(done);
            unsigned long long m_any, mb[4];''', '''            { const unsigned long long c_w0 = __builtin_readcyclecounter(); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); c_wait += __builtin_readcyclecounter() - c_w0; }
            const unsigned long long going = SEG == 1 ? ~0ull : ~__ballot(done);
            unsigned long long m_any, mb[4];''')
s += '''
extern "C" int gs_debug_fwd_trace(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gs::g_fwd_trace), (size_t)n * 10 * 8, 0, hipMemcpyDeviceToHost);
}
'''
open(sys.argv[2], 'w').write(s)
PY
cd $R/activesplat_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics -fno-slp-vectorize -c blend_tracef_tmp.hip -o $T/blend_tracef.o
rm -f blend_tracef_tmp.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgsplat_hip_tracef.so api.o preprocess.o preprocess_bwd.o binning.o tilebin.o $T/blend_tracef.o adam.o compact.o loss.o activate.o grow.o stats.o densify.o rows.o sort_rocprim.o
echo built activesplat_amd/libgsplat_hip_tracef.so
