#!/usr/bin/env python3
"""Sum the rocprofv3 --pmc passes written by profiles/collect_pmc.sh per kernel and derive the per-modexp figures quoted in
DESIGN.md and read by bench.py (`roofline.traffic`):
    python profiles/aggregate_pmc.py <dir with pass*/...counter_collection.csv> <kernel name substring, e.g. "k_enc<4>"> [modexps]
modexps = exponentiations done by ALL dispatches of that kernel in one pass (default: "all_modexps_of_the_kernel" of <dir>/shape.json)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root, kernel = sys.argv[1], sys.argv[2]
    if len(sys.argv) > 3:
        modexps = float(sys.argv[3])
    else:
        modexps = float(json.load(open(os.path.join(root, "shape.json")))["all_modexps_of_the_kernel"])
    agg = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k].add((f, row["Dispatch_Id"]))
    out = {}
    for k, v in agg.items():
        if not k.startswith(("void zkp::", "zkp::")):
            continue
        rec = dict(v)
        passes = len({f for f, _ in disp[k]})
        rec["dispatches"] = len(disp[k]) // max(passes, 1)
        out[k] = rec
    name = next((k for k in out if kernel in k), None)
    if name and modexps:
        r = out[name]
        lanes = 64 // int(kernel.split("<")[1].split(">")[0].split(",")[0]) if "<" in kernel else 1      # modexps per wavefront
        d = {"modexps_in_these_dispatches": modexps, "modexps_per_wavefront": lanes}
        if "SQ_INSTS_VALU" in r:
            d["valu_wave_instr_per_wave_modexp"] = r["SQ_INSTS_VALU"] / (modexps / lanes)
            if "GRBM_GUI_ACTIVE" in r:   # summed over the 8 XCDs: x 1024 SIMDs / 8 = busy cycles of all SIMDs
                d["simd_cycles_per_valu_instr"] = r["GRBM_GUI_ACTIVE"] * 128 / r["SQ_INSTS_VALU"]
            d["valu_active_fraction_of_wave_cycles"] = r["SQ_ACTIVE_INST_VALU"] / r["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in r:
            d["fetch_bytes_per_modexp (FETCH_SIZE in KB x 1024, uncorrected)"] = r["FETCH_SIZE"] * 1024 / modexps
        if "WRITE_SIZE" in r:
            d["write_bytes_per_modexp (WRITE_SIZE in KB x 1024)"] = r["WRITE_SIZE"] * 1024 / modexps
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            d["hbm_bytes_per_modexp (gfx950 correction: 2 x FETCH_SIZE + WRITE_SIZE)"] = (2 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024 / modexps
        if "SQ_LDS_BANK_CONFLICT" in r and r.get("SQ_ACTIVE_INST_LDS"):
            d["lds_bank_conflict_cycles_over_lds_active"] = r["SQ_LDS_BANK_CONFLICT"] / r["SQ_ACTIVE_INST_LDS"]
        r["_derived"] = d
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
