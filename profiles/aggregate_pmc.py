#!/usr/bin/env python3
"""Sum the rocprofv3 --pmc passes written by profiles/collect_pmc.sh per kernel and derive the per-modexp figures quoted in
DESIGN.md and read by bench.py (`roofline.traffic`, `roofline.clock_ghz`):
    python profiles/aggregate_pmc.py <dir with pass*/...counter_collection.csv> <kernel name substring, e.g. "k_enc<4, true"> [modexps] [--calib calib.json]
    python profiles/aggregate_pmc.py --calibrate <dir of the `tabread` shape>        -> calib.json on stdout
modexps = exponentiations done by ALL dispatches of that kernel in one pass (default: "all_modexps_of_the_kernel" of <dir>/shape.json).

Calibration (round 3): the `tabread` shape launches k_table_traffic<4> twice — launch 1 READS, launch 2 WRITES a known number of
bytes in the ladders' own table access pattern (144 B per lane, 576 B contiguous per group, slots 18 KB apart, 600 MB in all: past L2
and MALL).  known bytes / (counter x 1024) is the factor that turns FETCH_SIZE / WRITE_SIZE (KB) of THIS pattern into HBM-side
bytes; with a calib file the derived `hbm_bytes_per_modexp` uses those factors instead of the guide's blanket 2 x FETCH_SIZE.

Effective shader clock: GRBM_GUI_ACTIVE is summed over the 8 XCDs, so cycles per XCD / wall time of the dispatches (End - Start
timestamps of the same pass) = the clock the kernel really ran at while the counters were collected."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

XCDS = 8


def read_passes(root):
    agg = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    wall = defaultdict(lambda: defaultdict(float))         # kernel -> counter -> summed (End - Start) ns of the dispatches that carried it
    per_dispatch = defaultdict(lambda: defaultdict(dict))  # kernel -> counter -> {dispatch id: value}
    for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k, cn = row["Kernel_Name"], row["Counter_Name"]
                agg[k][cn] += float(row["Counter_Value"])
                disp[k].add((f, row["Dispatch_Id"]))
                d = per_dispatch[k][cn]
                d[int(row["Dispatch_Id"])] = d.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
                key = (k, cn, row["Dispatch_Id"])
                if key not in seen and row.get("End_Timestamp"):
                    seen.add(key)
                    wall[k][cn] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return agg, disp, wall, per_dispatch


def calibrate(root):
    agg, disp, wall, per = read_passes(root)
    name = next((k for k in agg if "k_table_traffic" in k), None)
    if not name:
        sys.exit("no k_table_traffic dispatches under " + root)
    shape = json.load(open(os.path.join(root, "shape.json")))
    rd, wr = per[name]["FETCH_SIZE"], per[name]["WRITE_SIZE"]
    first, second = min(rd), max(rd)
    out = {"kernel": name, "known_bytes_read_by_launch_1": shape["bytes_read_by_launch_1"], "known_bytes_written_by_launch_2": shape["bytes_written_by_launch_2"],
           "FETCH_SIZE_KB_launch_1": rd[first], "WRITE_SIZE_KB_launch_1": wr[min(wr)], "FETCH_SIZE_KB_launch_2": rd[second], "WRITE_SIZE_KB_launch_2": wr[max(wr)],
           "fetch_factor": shape["bytes_read_by_launch_1"] / (rd[first] * 1024.0), "write_factor": shape["bytes_written_by_launch_2"] / (wr[max(wr)] * 1024.0),
           "note": "HBM-side bytes of the window-table pattern = factor x counter x 1024; the guide's blanket figure for wide reads is fetch_factor = 2"}
    json.dump(out, sys.stdout, indent=1)


def main():
    if sys.argv[1] == "--calibrate":
        return calibrate(sys.argv[2])
    args = [a for a in sys.argv[1:]]
    calib = None
    if "--calib" in args:
        i = args.index("--calib")
        calib = json.load(open(args[i + 1]))
        del args[i:i + 2]
    root, kernel = args[0], args[1]
    if len(args) > 2:
        modexps = float(args[2])
    else:
        modexps = float(json.load(open(os.path.join(root, "shape.json")))["all_modexps_of_the_kernel"])
    agg, disp, wall, _ = read_passes(root)
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
        if "GRBM_GUI_ACTIVE" in r and wall[name].get("GRBM_GUI_ACTIVE"):
            d["kernel_wall_ms_in_the_counter_pass"] = wall[name]["GRBM_GUI_ACTIVE"] / 1e6
            d["effective_clock_ghz"] = (r["GRBM_GUI_ACTIVE"] / XCDS) / wall[name]["GRBM_GUI_ACTIVE"]
        if "FETCH_SIZE" in r:
            d["fetch_bytes_per_modexp (FETCH_SIZE in KB x 1024, uncorrected)"] = r["FETCH_SIZE"] * 1024 / modexps
        if "WRITE_SIZE" in r:
            d["write_bytes_per_modexp (WRITE_SIZE in KB x 1024)"] = r["WRITE_SIZE"] * 1024 / modexps
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            d["hbm_bytes_per_modexp (gfx950 correction: 2 x FETCH_SIZE + WRITE_SIZE)"] = (2 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024 / modexps
            if calib:
                d["fetch_factor"], d["write_factor"] = calib["fetch_factor"], calib["write_factor"]
                d["hbm_bytes_per_modexp"] = (calib["fetch_factor"] * r["FETCH_SIZE"] + calib["write_factor"] * r["WRITE_SIZE"]) * 1024 / modexps
                d["hbm_bytes_note"] = "factors calibrated on k_table_traffic (known bytes in the table access pattern), see calib file"
        if "SQ_LDS_BANK_CONFLICT" in r and r.get("SQ_ACTIVE_INST_LDS"):
            d["lds_bank_conflict_cycles_over_lds_active"] = r["SQ_LDS_BANK_CONFLICT"] / r["SQ_ACTIVE_INST_LDS"]
        r["_derived"] = d
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
