# Development helper: build activesplat_amd/libgsplat_hip_trace.so -- the library with a TRACE variant of the backward blend (per workgroup:
# start / end / before-the-wait / after-the-wait timestamps of the 100 MHz clock in a __device__ array, exported as gs_debug_wave_trace) for
# scripts/exp/wave_trace.py.  Not part of the product build.  usage (repo root, after `make -C activesplat_amd/csrc`): bash scripts/exp/make_trace_lib.sh
set -e
R=$PWD; T=$(mktemp -d)
python3 - "$R/activesplat_amd/csrc/blend.hip" "$R/activesplat_amd/csrc/blend_trace_tmp.hip" <<'PY'
import sys
s = open(sys.argv[1]).read()
def sub(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)
sub('namespace gs {', 'namespace gs {\n__device__ unsigned long long g_wave_trace[5 * 16384];\n')
sub('''    const bool split = FEW && cam.split != 0;
    // Chained walks''', '''    const bool split = FEW && cam.split != 0;
    const unsigned long long t_start = wall_clock64();
    unsigned long long t_loop = 0, t_waited = 0;
    struct TraceEnd { unsigned long long t0; unsigned long long& tl; unsigned long long& tw; unsigned b; int lane; __device__ ~TraceEnd() { if (lane == 0 && b < 16384) { g_wave_trace[5 * b] = t0; g_wave_trace[5 * b + 1] = wall_clock64(); g_wave_trace[5 * b + 2] = tl; g_wave_trace[5 * b + 3] = tw; g_wave_trace[5 * b + 4] = 1; } } } trace_end{t_start, t_loop, t_waited, blockIdx.x, (int)(threadIdx.x & 63)};
    // Chained walks''')
sub('''    if (pieces > 1 && piece > 0) {
        // (as late as possible''', '''    t_loop = wall_clock64();
    if (pieces > 1 && piece > 0) {
        // (as late as possible''')
sub('''    for (int ch = cmax; ch >= cmin; ch--) {
        const float4 q0 = r0, q1 = r1, q2 = r2;
        const uint32_t id_cur = id_next;
        id_next = id_next2;
        id_next2 = ch >= cmin + 2 ?''', '''    t_waited = wall_clock64();
    for (int ch = cmax; ch >= cmin; ch--) {
        const float4 q0 = r0, q1 = r1, q2 = r2;
        const uint32_t id_cur = id_next;
        id_next = id_next2;
        id_next2 = ch >= cmin + 2 ?''')
s += '''
extern "C" int gs_debug_wave_trace(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gs::g_wave_trace), (size_t)n * 5 * 8, 0, hipMemcpyDeviceToHost);
}
'''
open(sys.argv[2], 'w').write(s)
PY
cd $R/activesplat_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics -fno-slp-vectorize -c blend_trace_tmp.hip -o $T/blend_trace.o
rm -f blend_trace_tmp.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgsplat_hip_trace.so api.o preprocess.o preprocess_bwd.o binning.o tilebin.o $T/blend_trace.o adam.o compact.o loss.o activate.o grow.o stats.o densify.o rows.o sort_rocprim.o
echo built activesplat_amd/libgsplat_hip_trace.so
