"""Development helper: registers / scratch / occupancy / LDS of every kernel of one source file, from the compiler's resource-usage remarks.
    python scripts/exp/kernel_regs.py activesplat_amd/csrc/blend.hip [extra hipcc flags, e.g. -munsafe-fp-atomics -fno-slp-vectorize]"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{R}/include", "-Rpass-analysis=kernel-resource-usage"]
                     + sys.argv[2:] + ["-c", os.path.join(R, sys.argv[1]), "-o", "/dev/null"], capture_output=True, text=True).stderr
rows, cur = [], None
for ln in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void gs::", "")
    print(f"{n:72s} VGPR {r.get('VGPRs'):>4} AGPR {r.get('AGPRs'):>3} scratch {r.get('ScratchSize'):>4} waves/SIMD {r.get('Occupancy'):>2} LDS {r.get('LDS'):>6}")
