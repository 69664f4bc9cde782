"""Per-kernel means of the PMC passes of scripts/gpu_round_end.sh -> profiles/<tag>_pmc_summary.json.
usage: python scripts/pmc_summarise.py gpurun_out/final r01
Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE count KiB; on gfx950 FETCH_SIZE counts a
128-byte line fetched by 16 B/lane reads as 64 B, so the read bytes are doubled.  Each counter group comes from its own
rocprofv3 pass (never together with a trace domain)."""
import glob
import json
import os
import re
import sys

import pandas as pd

src, tag = sys.argv[1], sys.argv[2]
out = {}
for f in sorted(glob.glob(os.path.join(src, f"{tag}_pmc_*_counter_collection.csv"))):
    d = pd.read_csv(f)
    d = d[d["Kernel_Name"].str.contains("gs::|nccl|rccl", regex=True)]      # this library's kernels and RCCL's (configs[3]'s exchange)
    d["k"] = d["Kernel_Name"].map(lambda n: re.sub(r"<.*", "", re.sub(r"^void ", "", n)).split("(")[0].replace("gs::", ""))
    # the same with the template arguments kept ("preprocess_backward_kernel<3, true, true>"): one loop can run several instantiations of a kernel
    d["kfull"] = d["Kernel_Name"].map(lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n)).replace("gs::", "").strip())
    # a dispatch appears once per counter (values already summed over the device's XCDs / SEs); skip each kernel's warm-up launches
    for col in ("k", "kfull"):
        for (k, c), g in d.groupby([col, "Counter_Name"]):
            if col == "kfull" and "<" not in k:
                continue
            vals = g.sort_values("Dispatch_Id")["Counter_Value"].tolist()
            vals = vals[len(vals) // 4:] or vals
            out.setdefault(k, {})[c] = sum(vals) / len(vals)
            out[k]["launches_averaged"] = len(vals)
for k, v in out.items():
    if "FETCH_SIZE" in v:
        v["hbm_read_bytes_raw"] = v["FETCH_SIZE"] * 1024
        v["hbm_read_bytes_x2_gfx950"] = v["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in v:
        v["hbm_write_bytes"] = v["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["traffic_bytes"] = v["hbm_read_bytes_x2_gfx950"] + v["hbm_write_bytes"]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_pmc_summary.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for k in sorted(out):
    print(k, {c: round(x) for c, x in out[k].items() if c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES",
                                                             "GRBM_GUI_ACTIVE", "traffic_bytes", "hbm_write_bytes", "hbm_read_bytes_x2_gfx950")})
print("wrote", dst)
