#!/usr/bin/env python3
"""profiles/r01_pmc_hbm_traffic.json from the two rocprofv3 --pmc passes (tools/rocprof_pmc.sh).

FETCH_SIZE and WRITE_SIZE are reported in KiB-like units of 1000 B by rocprofv3 on this image ("KB"); on gfx950 FETCH_SIZE
counts 64 B per 128-B request and is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported (it matches
the AdamW kernel's algorithmic 0.89 GB of writes to 1 %).  Kernel names are written the way bench.py names them."""
import json
import re
import sys

raw = json.load(open(sys.argv[1]))
B, T, P, math = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
out = {"workload": [B, T, P, math], "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on "
       "`bench.py --no-graph --steps 2 --warmup 2`; FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), units of 1000 B; "
       "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE averaged over its 8 per-XCD instances x 1024 "
       "SIMDs) from a third pass; mfma_busy_sq = the same over SQ_BUSY_CYCLES (averaged over its 32 instances), which excludes the "
       "dispatch / drain time GRBM counts; valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA", "kernels": {}}
for mangled, c in raw.items():
    w8 = re.search(r"gemm16_kernel_w8ILb(\d)ELb(\d)ELi(\d+)E", mangled)
    m = re.search(r"gemm16_(group_)?kernelILb(\d)ELb(\d)ELi(\d+)ELi(\d+)ELi(\d+)E(?:Li(\d+)E)?", mangled)
    gx = re.search(r"g16x_(group_)?kernelILb(\d)ELb(\d)E((?:Li\d+E)+)", mangled)
    if gx:                                                        # large-tile family (kk_gemm16x.hip): every template argument, as bench.py names it
        b = lambda v: "true" if v == "1" else "false"
        ints = ",".join(re.findall(r"Li(\d+)E", gx.group(4)))
        name = f"g16x_{gx.group(1) or ''}kernel<{b(gx.group(2))},{b(gx.group(3))},{ints}>"
    elif w8:
        b = lambda v: "true" if v == "1" else "false"
        name = f"gemm16_kernel_w8<{b(w8.group(1))},{b(w8.group(2))},{w8.group(3)}>"
    elif m:
        b = lambda v: "true" if v == "1" else "false"
        epi = f",{m.group(7)}" if m.group(7) not in (None, "0") else ""       # epilogue variant (0 = plain) as bench.py names it
        name = f"gemm16_{m.group(1) or ''}kernel<{b(m.group(2))},{b(m.group(3))},{m.group(4)},{m.group(5)},{m.group(6)}{epi}>"
    else:
        m = re.search(r"_GLOBAL__N_1\d+([a-z_0-9]+?)(I|E)", mangled)
        name = m.group(1) if m else mangled
        name = {"attn_fwd_kernel": "kk_attn_fwd", "attn_bwd_dq_kernel": "kk_attn_bwd_dq", "attn_bwd_dkv_kernel": "kk_attn_bwd_dkv"}.get(name, name)
    f, w = c.get("FETCH_SIZE", [0, 0.0]), c.get("WRITE_SIZE", [0, 0.0])
    n = max(f[0], w[0], 1)
    e = out["kernels"].setdefault(name, {"dispatches": 0, "fetch": 0.0, "write": 0.0, "mfma": 0.0, "gui": 0.0, "gui_rows": 0, "sqb": 0.0,
                                         "sqb_rows": 0, "valu": 0.0, "nmfma": 0.0})
    e["dispatches"] += n
    e["fetch"] += 2.0 * f[1] * 1000.0
    e["write"] += w[1] * 1000.0
    e["mfma"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0])[1]
    e["gui"] += c.get("GRBM_GUI_ACTIVE", [0, 0.0])[1]
    e["gui_rows"] += c.get("GRBM_GUI_ACTIVE", [0, 0.0])[0]
    e["sqb"] += c.get("SQ_BUSY_CYCLES", [0, 0.0])[1]
    e["sqb_rows"] += c.get("SQ_BUSY_CYCLES", [0, 0.0])[0]
    e["valu"] += c.get("SQ_INSTS_VALU", [0, 0.0])[1]
    e["nmfma"] += c.get("SQ_INSTS_MFMA", [0, 0.0])[1]
for e in out["kernels"].values():
    e["fetch_bytes_per_launch"] = e.pop("fetch") / e["dispatches"]
    e["write_bytes_per_launch"] = e.pop("write") / e["dispatches"]
    mf, gui, rows, sqb, srows, valu, nm = (e.pop(k) for k in ("mfma", "gui", "gui_rows", "sqb", "sqb_rows", "valu", "nmfma"))
    if rows and gui > 0:            # rows = dispatches x counter instances: gui / rows = mean active cycles of one launch on one XCD
        e["mfma_busy"] = round(mf / ((gui / rows) * (rows / 8.0) * 1024.0), 4) if mf else 0.0
        e["mfma_busy_sq"] = round(mf / ((sqb / srows) * (srows / 32.0) * 1024.0), 4) if (mf and srows and sqb > 0) else (0.0 if not mf else None)
        e["valu_per_mfma"] = round(valu / nm, 1) if nm else None
json.dump(out, open(sys.argv[6], "w"), indent=1)
print(json.dumps({k: v for k, v in list(out["kernels"].items())[:4]}, indent=1))
