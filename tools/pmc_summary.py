#!/usr/bin/env python3
"""tools/pmc_summary.py <prof dir> [frames_per_launch] — condense the rocprofv3 PMC passes of tools/gpu_profile.sh (separate passes: SQ instruction counts, VALU busy,
lane utilisation, FETCH_SIZE, WRITE_SIZE with their calibration runs) into the per-frame figures bench.py quotes in its roofline block.  Prints one JSON object."""
import csv, collections, json, os, sys
d = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
kernel = sys.argv[3] if len(sys.argv) > 3 else "oa_encode"
def mean(f, match=kernel):
    agg = collections.defaultdict(list)
    p = os.path.join(d, f)
    if not os.path.exists(p): return {}
    for r in csv.DictReader(open(p)):
        if match in r.get("Kernel_Name", ""): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
def calib(f, counter, kern):
    for r in csv.DictReader(open(os.path.join(d, f))):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == counter: return float(r["Counter_Value"])
    return None
ins, busy, lanes, fe, wr = mean("pmc_sq_insts.csv"), mean("pmc_valu_busy.csv"), mean("pmc_lanes.csv"), mean("pmc_fetch.csv"), mean("pmc_write.csv")
GiB = 1 << 30
cf = calib("calib_FETCH_SIZE.csv", "FETCH_SIZE", "calib_read4"); cw = calib("calib_WRITE_SIZE.csv", "WRITE_SIZE", "calib_write4")
fetch_bytes = fe["FETCH_SIZE"] * GiB / cf; write_bytes = wr["WRITE_SIZE"] * GiB / cw      # counter units from the calibration: 1 GiB read / written per calibration kernel
waves = ins["SQ_WAVES"]
out = {"source": "%s (rocprofv3 --pmc, separate passes, kernel %s*, %d frames per launch)" % (d, kernel, frames),
       "calibration": "tools/pmc_calibrate: 1 GiB read -> FETCH_SIZE %.1f, 1 GiB written -> WRITE_SIZE %.1f" % (cf, cw),
       "frames_per_launch": frames,
       "fetch_bytes_per_frame": round(fetch_bytes / frames), "write_bytes_per_frame": round(write_bytes / frames), "hbm_bytes_per_frame": round((fetch_bytes + write_bytes) / frames),
       "issue": {"valu_insts_per_frame": round(ins["SQ_INSTS_VALU"] / frames), "salu_insts_per_frame": round(ins["SQ_INSTS_SALU"] / frames), "lds_insts_per_frame": round(ins["SQ_INSTS_LDS"] / frames),
                 "valu_active_fraction_of_wave_cycles": round(busy["SQ_ACTIVE_INST_VALU"] / ins["SQ_WAVE_CYCLES"], 3), "resident_waves": waves},
       "lane_utilisation": {"active_lanes_per_valu_cycle": round(lanes["SQ_THREAD_CYCLES_VALU"] / lanes["SQ_ACTIVE_INST_VALU"], 1)}}
print(json.dumps(out, indent=1))
