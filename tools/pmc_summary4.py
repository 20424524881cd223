#!/usr/bin/env python3
"""tools/pmc_summary4.py <dir with config dirs from tools/gpu_pmc.sh> <out.json> — condense the rocprofv3 counter passes (rounds 4 and 5) (one directory per bench leg: pmc2,
pmc3, pmc4, pmcd2 ...; separate passes per counter group; FETCH_SIZE / WRITE_SIZE with their calibration runs next to them) into per-frame figures, per kernel and summed over the
kernels of one call: what bench.py quotes in roofline.traffic / roofline.issue.  16,384 frames per launch."""
import csv, collections, json, os, sys
root, out = sys.argv[1], sys.argv[2]
FR = 16384; GiB = 1 << 30
# resident waves per SIMD of each kernel (LDS / VGPR footprint: tools/kernel_resources.sh + the launch's dynamic LDS); front: 12 waves per CU mono, 10 stereo
WPS = {"oa_encode_kernel": 4.0, "oa_celt_front_kernel": 4.0, "oa_celt_pvq_kernel": 3.0, "oa_celt_back_kernel": 4.0, "oa_sh_back2_kernel": 3.0, "oa_celt_sort_kernel": 8.0, "oa_sh_front_kernel": 4.0, "oa_sh_quant_kernel": 2.0, "oa_sh_back_kernel": 3.0, "oa_decode_kernel": 2.0, "oa_decode_fast_kernel": 4.0, "oa_sh_encode_kernel": 1.75,
       "oa_decode_look_kernel": 8.0, "oa_decode_hyb_kernel": 4.0, "oa_celt_dpvq_kernel": 3.0, "oa_celt_dback_kernel": 4.0, "oa_sdec_lane_kernel": 2.0, "oa_ms_split_kernel": 8.0, "oa_ms_pack_kernel": 8.0,
       "oa_sh_pred_kernel": 8.0, "oa_sh_predc_kernel": 8.0, "oa_sh_preda_kernel": 1.0, "oa_sh_predb_kernel": 1.0, "oa_celt_transient_kernel": 2.0, "oa_sh_transient_kernel": 2.0}      # (round 5: the front kernel holds 16 waves per CU mono, 12 stereo -- config 4 below)
WPS_BY_LEG = {"config_4": {"oa_sh_front_kernel": 3.0}}
DEC_KERNELS = {"oa_sdec_lane_kernel", "oa_celt_dpvq_kernel", "oa_celt_dback_kernel", "oa_celt_deemph_kernel"}
def is_dec(k): return "decode" in k or k in DEC_KERNELS
def table(d, f):
    """{kernel: {counter: mean value per dispatch}}, dispatch counts"""
    p = os.path.join(d, f); agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(p): return {}
    for r in csv.DictReader(open(p)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
def calib(f, counter, kern):
    p = os.path.join(root, f)
    if not os.path.exists(p): return None
    for r in csv.DictReader(open(p)):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == counter: return float(r["Counter_Value"])
    return None
cf, cw = calib("calib_FETCH_SIZE.csv", "FETCH_SIZE", "calib_read4"), calib("calib_WRITE_SIZE.csv", "WRITE_SIZE", "calib_write4")
doc = {"source": "rocprofv3 --pmc passes (separate runs per counter group, tools/gpu_pmc.sh: bench.py --steps 3 --warmup 1 --streams 16384, every oa_* kernel of a call), condensed by tools/pmc_summary4.py; raw CSVs next to this file's directory name in profiles/",
       "calibration": None if not cf else "tools/pmc_calibrate: 1 GiB read -> FETCH_SIZE %.1f, 1 GiB written -> WRITE_SIZE %.1f" % (cf, cw), "frames_per_launch": FR}
for sub in sorted(os.listdir(root)):
    d = os.path.join(root, sub)
    if not (os.path.isdir(d) and sub.startswith("pmc")): continue
    key = ("decode_%s" % sub[4:]) if sub.startswith("pmcd") else ("config_%s" % sub[3:])
    ins, busy, lanes, fe, wr = table(d, "pmc_sq_insts.csv"), table(d, "pmc_valu_busy.csv"), table(d, "pmc_lanes.csv"), table(d, "pmc_fetch.csv"), table(d, "pmc_write.csv")
    kernels = sorted(k for k in ins if is_dec(k) == key.startswith("decode"))     # (a decoder leg encodes its packets first: those launches are not its own)
    ins = {k: ins[k] for k in kernels}; busy = {k: v for k, v in busy.items() if k in kernels}; lanes = {k: v for k, v in lanes.items() if k in kernels}
    per = {}
    tot = collections.Counter()
    for k in kernels:
        e = {"valu_insts_per_frame": round(ins[k].get("SQ_INSTS_VALU", 0) / FR), "salu_insts_per_frame": round(ins[k].get("SQ_INSTS_SALU", 0) / FR), "lds_insts_per_frame": round(ins[k].get("SQ_INSTS_LDS", 0) / FR),
             "vmem_insts_per_frame": round((ins[k].get("SQ_INSTS_VMEM_RD", 0) + ins[k].get("SQ_INSTS_VMEM_WR", 0)) / FR), "waves_per_launch": round(ins[k].get("SQ_WAVES", 0))}
        if k in busy and busy[k].get("SQ_WAVE_CYCLES"): e["valu_active_fraction_of_wave_cycles"] = round(busy[k]["SQ_ACTIVE_INST_VALU"] / busy[k]["SQ_WAVE_CYCLES"], 3)
        if k in lanes and lanes[k].get("SQ_ACTIVE_INST_VALU"): e["active_lanes_per_valu_cycle"] = round(lanes[k]["SQ_THREAD_CYCLES_VALU"] / lanes[k]["SQ_ACTIVE_INST_VALU"], 1)
        if cf and k in fe: e["fetch_bytes_per_frame"] = round(fe[k]["FETCH_SIZE"] * GiB / cf / FR)
        if cw and k in wr: e["write_bytes_per_frame"] = round(wr[k]["WRITE_SIZE"] * GiB / cw / FR)
        per[k] = e
        for c in ("valu_insts_per_frame", "salu_insts_per_frame", "lds_insts_per_frame", "fetch_bytes_per_frame", "write_bytes_per_frame"): tot[c] += e.get(c, 0)
    tl = sum(lanes[k]["SQ_THREAD_CYCLES_VALU"] for k in lanes if "SQ_THREAD_CYCLES_VALU" in lanes[k]); ta = sum(lanes[k]["SQ_ACTIVE_INST_VALU"] for k in lanes if "SQ_ACTIVE_INST_VALU" in lanes[k])
    va = sum(busy[k].get("SQ_ACTIVE_INST_VALU", 0) for k in busy); wc = sum(busy[k].get("SQ_WAVE_CYCLES", 0) for k in busy); bc = sum(busy[k].get("SQ_BUSY_CYCLES", 0) for k in busy)
    doc[key] = {"source": "profiles/%s/%s/*.csv (%s)" % (os.path.basename(os.path.abspath(root)), sub, ", ".join(kernels)),
                "fetch_bytes_per_frame": tot["fetch_bytes_per_frame"] or None, "write_bytes_per_frame": tot["write_bytes_per_frame"] or None,
                "hbm_bytes_per_frame": (tot["fetch_bytes_per_frame"] + tot["write_bytes_per_frame"]) or None,
                "issue": {"valu_insts_per_frame": tot["valu_insts_per_frame"], "salu_insts_per_frame": tot["salu_insts_per_frame"], "lds_insts_per_frame": tot["lds_insts_per_frame"],
                          "valu_active_fraction_of_wave_cycles": None if not wc else round(va / wc, 3),     # x resident waves per SIMD = VALU busy per SIMD
                          "valu_busy_per_simd": None if not wc else round(va / sum(busy[k].get("SQ_WAVE_CYCLES", 0) / WPS_BY_LEG.get(key, {}).get(k, WPS.get(k, 2.0)) for k in busy), 3),     # VALU-active cycles / SIMD-resident cycles (wave cycles / resident waves per SIMD), over the call's kernels
                          "waves_per_simd": {k: WPS_BY_LEG.get(key, {}).get(k, WPS.get(k)) for k in kernels}},
                "lane_utilisation": {"active_lanes_per_valu_cycle": None if not ta else round(tl / ta, 1)},
                "kernels": per}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: {"hbm": v.get("hbm_bytes_per_frame"), "valu": v["issue"]["valu_insts_per_frame"], "lanes": v["lane_utilisation"]["active_lanes_per_valu_cycle"], "valu_frac_of_wave_cycles": v["issue"]["valu_active_fraction_of_wave_cycles"]} for k, v in doc.items() if isinstance(v, dict) and "issue" in v}, indent=1))
