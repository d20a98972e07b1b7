#!/usr/bin/env python3
"""Sum the rocprofv3 --pmc passes written by profiles/collect_pmc.sh per kernel and derive the per-Enc figures quoted in
DESIGN.md section 8:  python profiles/aggregate_pmc.py <dir with pass*/...counter_collection.csv> <modexps in the k_enc dispatches>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    modexps = float(sys.argv[2]) if len(sys.argv) > 2 else None
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
    enc = next((k for k in out if "k_enc<8>" in k), None)
    if enc and modexps:
        r = out[enc]
        d = {"modexps_in_these_dispatches": modexps}
        if "SQ_INSTS_VALU" in r:
            d["valu_wave_instr_per_wave_modexp (8 modexps per wave)"] = r["SQ_INSTS_VALU"] / (modexps / 8)
            if "GRBM_GUI_ACTIVE" in r:   # summed over the 8 XCDs: x 1024 SIMDs / 8 = busy cycles of all SIMDs
                d["simd_cycles_per_valu_instr"] = r["GRBM_GUI_ACTIVE"] * 128 / r["SQ_INSTS_VALU"]
            d["valu_active_fraction_of_wave_cycles"] = r["SQ_ACTIVE_INST_VALU"] / r["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in r:
            d["fetch_bytes_per_modexp (FETCH_SIZE in KB x 1024, uncorrected)"] = r["FETCH_SIZE"] * 1024 / modexps
        if "WRITE_SIZE" in r:
            d["write_bytes_per_modexp (WRITE_SIZE in KB x 1024)"] = r["WRITE_SIZE"] * 1024 / modexps
        if "SQ_LDS_BANK_CONFLICT" in r and r.get("SQ_ACTIVE_INST_LDS"):
            d["lds_bank_conflict_cycles_over_lds_active"] = r["SQ_LDS_BANK_CONFLICT"] / r["SQ_ACTIVE_INST_LDS"]
        r["_derived"] = d
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
