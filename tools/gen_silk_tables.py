#!/usr/bin/env python3
"""tools/gen_silk_tables.py — emit the SILK resampler constant tables (reference silk/resampler_rom.c:41-96) in this repo's layout.

They are DATA (filter designs fixed by the codec), read out of the *compiled* reference (oracle/_ref/libopus_ref_fx.so exports
the arrays) and written to
    oracle/oc_silk_tables.h           (CPU oracle, prefix ocs_)
    opus_amd/csrc/silk_tables.h       (HIP product, prefix sk_)
Run only where oracle/_ref was built; outputs are committed."""
import ctypes, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "oracle/_ref/libopus_ref_fx.so"))
def arr(name, n): return list((ctypes.c_int16 * n).in_dll(L, name))
T = [("resampler_3_4_coefs", "silk_Resampler_3_4_COEFS", 2 + 27), ("resampler_2_3_coefs", "silk_Resampler_2_3_COEFS", 2 + 18),
     ("resampler_1_2_coefs", "silk_Resampler_1_2_COEFS", 2 + 12), ("resampler_1_3_coefs", "silk_Resampler_1_3_COEFS", 2 + 18),
     ("resampler_1_4_coefs", "silk_Resampler_1_4_COEFS", 2 + 18), ("resampler_1_6_coefs", "silk_Resampler_1_6_COEFS", 2 + 18),
     ("resampler_frac_fir_12", "silk_resampler_frac_FIR_12", 48), ("resampler_2_3_coefs_lq", "silk_Resampler_2_3_COEFS_LQ", 6)]
# pitch estimator contour codebooks (reference silk/pitch_est_tables.c:35-99), int8
T8 = [("cb_lags_stage2_10ms", "silk_CB_lags_stage2_10_ms", 2 * 3), ("cb_lags_stage3_10ms", "silk_CB_lags_stage3_10_ms", 2 * 12),
      ("lag_range_stage3_10ms", "silk_Lag_range_stage3_10_ms", 2 * 2), ("cb_lags_stage2", "silk_CB_lags_stage2", 4 * 11),
      ("cb_lags_stage3", "silk_CB_lags_stage3", 4 * 34), ("lag_range_stage3", "silk_Lag_range_stage3", 3 * 4 * 2), ("nb_cbk_searchs_stage3", "silk_nb_cbk_searchs_stage3", 3)]
def arr8(name, n): return list((ctypes.c_int8 * n).in_dll(L, name))
# header-only constants (silk/resampler_rom.h:41-47), not exported symbols
UP2 = {"resampler_up2_hq_0": [1746, 14986, 39083 - 65536], "resampler_up2_hq_1": [6854, 25769, 55542 - 65536]}
# silk/resampler_rom.h:41-42 (2x decimator allpass coefficients) are emitted as macros below
# ---- decoder tables (reference silk/tables_*.c, table_LSF_cos.c): uint8 iCDFs, codebooks ----
U8 = [("gain_icdf", "silk_gain_iCDF", 24), ("delta_gain_icdf", "silk_delta_gain_iCDF", 41), ("pitch_lag_icdf", "silk_pitch_lag_iCDF", 32),
      ("pitch_delta_icdf", "silk_pitch_delta_iCDF", 21), ("pitch_contour_icdf", "silk_pitch_contour_iCDF", 34), ("pitch_contour_nb_icdf", "silk_pitch_contour_NB_iCDF", 11),
      ("pitch_contour_10ms_icdf", "silk_pitch_contour_10_ms_iCDF", 12), ("pitch_contour_10ms_nb_icdf", "silk_pitch_contour_10_ms_NB_iCDF", 3),
      ("pulses_per_block_icdf", "silk_pulses_per_block_iCDF", 180), ("rate_levels_icdf", "silk_rate_levels_iCDF", 18),
      ("shell_code_table0", "silk_shell_code_table0", 152), ("shell_code_table1", "silk_shell_code_table1", 152), ("shell_code_table2", "silk_shell_code_table2", 152),
      ("shell_code_table3", "silk_shell_code_table3", 152), ("shell_code_table_offsets", "silk_shell_code_table_offsets", 17), ("lsb_icdf", "silk_lsb_iCDF", 2),
      ("sign_icdf", "silk_sign_iCDF", 42), ("uniform3_icdf", "silk_uniform3_iCDF", 3), ("uniform4_icdf", "silk_uniform4_iCDF", 4), ("uniform5_icdf", "silk_uniform5_iCDF", 5),
      ("uniform6_icdf", "silk_uniform6_iCDF", 6), ("uniform8_icdf", "silk_uniform8_iCDF", 8), ("nlsf_ext_icdf", "silk_NLSF_EXT_iCDF", 7),
      ("ltp_per_index_icdf", "silk_LTP_per_index_iCDF", 3), ("ltpscale_icdf", "silk_LTPscale_iCDF", 3), ("type_offset_vad_icdf", "silk_type_offset_VAD_iCDF", 4),
      ("type_offset_no_vad_icdf", "silk_type_offset_no_VAD_iCDF", 2), ("stereo_pred_joint_icdf", "silk_stereo_pred_joint_iCDF", 25),
      ("stereo_only_code_mid_icdf", "silk_stereo_only_code_mid_iCDF", 2), ("nlsf_interpolation_factor_icdf", "silk_NLSF_interpolation_factor_iCDF", 5)]
I16 = [("ltpscales_table_q14", "silk_LTPScales_table_Q14", 3), ("stereo_pred_quant_q13", "silk_stereo_pred_quant_Q13", 16), ("lsf_cos_tab_q12", "silk_LSFCosTab_FIX_Q12", 129)]
def arru8(name, n): return list((ctypes.c_uint8 * n).in_dll(L, name))
def ptr_table(name, n, elem, sizes):
    """an exported array of n pointers -> concatenated data + offsets"""
    ptrs = (ctypes.c_void_p * n).in_dll(L, name); data = []; offs = []
    for k in range(n):
        offs.append(len(data)); data += list((elem * sizes[k]).from_address(ptrs[k]))
    return data, offs
class NlsfCb(ctypes.Structure):
    _fields_ = [("nVectors", ctypes.c_int16), ("order", ctypes.c_int16), ("quantStepSize_Q16", ctypes.c_int16), ("invQuantStepSize_Q6", ctypes.c_int16),
                ("CB1_NLSF_Q8", ctypes.c_void_p), ("CB1_Wght_Q9", ctypes.c_void_p), ("CB1_iCDF", ctypes.c_void_p), ("pred_Q8", ctypes.c_void_p),
                ("ec_sel", ctypes.c_void_p), ("ec_iCDF", ctypes.c_void_p), ("ec_Rates_Q5", ctypes.c_void_p), ("deltaMin_Q15", ctypes.c_void_p)]
def nlsf_cb(sym):
    cb = NlsfCb.in_dll(L, sym); nv, od = cb.nVectors, cb.order
    g = lambda p, t, n: list((t * n).from_address(p))
    return dict(nVectors=nv, order=od, quantStepSize_Q16=cb.quantStepSize_Q16, cb1_nlsf_q8=g(cb.CB1_NLSF_Q8, ctypes.c_uint8, nv * od), cb1_wght_q9=g(cb.CB1_Wght_Q9, ctypes.c_int16, nv * od),
                cb1_icdf=g(cb.CB1_iCDF, ctypes.c_uint8, 2 * nv), pred_q8=g(cb.pred_Q8, ctypes.c_uint8, 2 * (od - 1)), ec_sel=g(cb.ec_sel, ctypes.c_uint8, nv * od // 2),
                ec_icdf=g(cb.ec_iCDF, ctypes.c_uint8, 72), deltamin_q15=g(cb.deltaMin_Q15, ctypes.c_int16, od + 1))

def emit(path, prefix, qual):
    o = ["/* GENERATED by tools/gen_silk_tables.py — SILK resampler filter constants (reference silk/resampler_rom.c:41-96, resampler_rom.h:41-47) and pitch-estimator contour codebooks (silk/pitch_est_tables.c:35-99). */",
         "#ifndef %sSILK_TABLES_H\n#define %sSILK_TABLES_H\n#include <stdint.h>" % (prefix.upper(), prefix.upper())]
    for name, sym, n in T:
        v = arr(sym, n)
        o.append("%s int16_t %s%s[%d] = { %s };" % (qual, prefix, name, n, ", ".join(map(str, v))))
    for name, sym, n in T8:
        o.append("%s int8_t %s%s[%d] = { %s };" % (qual, prefix, name, n, ", ".join(map(str, arr8(sym, n)))))
    for name, sym, n in U8:
        o.append("%s uint8_t %s%s[%d] = { %s };" % (qual, prefix, name, n, ", ".join(map(str, arru8(sym, n)))))
    for name, sym, n in I16:
        o.append("%s int16_t %s%s[%d] = { %s };" % (qual, prefix, name, n, ", ".join(map(str, arr(sym, n)))))
    d, offs = ptr_table("silk_LTP_gain_iCDF_ptrs", 3, ctypes.c_uint8, [8, 16, 32])
    o.append("%s uint8_t %sltp_gain_icdf[%d] = { %s };   /* three codebooks, offsets %s */" % (qual, prefix, len(d), ", ".join(map(str, d)), offs))
    d, offs = ptr_table("silk_LTP_vq_ptrs_Q7", 3, ctypes.c_int8, [40, 80, 160])
    o.append("%s int8_t %sltp_vq_q7[%d] = { %s };   /* three codebooks x 5 taps, offsets %s */" % (qual, prefix, len(d), ", ".join(map(str, d)), offs))
    d, offs = ptr_table("silk_LBRR_flags_iCDF_ptr", 2, ctypes.c_uint8, [3, 7])
    o.append("%s uint8_t %slbrr_flags_icdf[%d] = { %s };   /* 2 and 3 frames per packet, offsets %s */" % (qual, prefix, len(d), ", ".join(map(str, d)), offs))
    for tag, sym in (("nb_mb", "silk_NLSF_CB_NB_MB"), ("wb", "silk_NLSF_CB_WB")):
        cb = nlsf_cb(sym)
        o.append("/* NLSF codebook %s: nVectors %d, order %d, quantStepSize_Q16 %d */" % (tag, cb["nVectors"], cb["order"], cb["quantStepSize_Q16"]))
        o.append("#define %sNLSF_%s_QSTEP_Q16 %d" % (prefix.upper(), tag.upper(), cb["quantStepSize_Q16"]))
        for k in ("cb1_nlsf_q8", "cb1_icdf", "pred_q8", "ec_sel", "ec_icdf"):
            o.append("%s uint8_t %snlsf_%s_%s[%d] = { %s };" % (qual, prefix, tag, k, len(cb[k]), ", ".join(map(str, cb[k]))))
        for k in ("cb1_wght_q9", "deltamin_q15"):
            o.append("%s int16_t %snlsf_%s_%s[%d] = { %s };" % (qual, prefix, tag, k, len(cb[k]), ", ".join(map(str, cb[k]))))
    for name, v in UP2.items():
        o.append("%s int16_t %s%s[3] = { %s };" % (qual, prefix, name, ", ".join(map(str, v))))
    o.append("#define %sRESAMPLER_DOWN2_0 9872\n#define %sRESAMPLER_DOWN2_1 (39809 - 65536)" % (prefix.upper(), prefix.upper()))
    o.append("#endif")
    open(path, "w").write("\n".join(o) + "\n")
emit(os.path.join(ROOT, "oracle/oc_silk_tables.h"), "ocs_", "static const")
emit(os.path.join(ROOT, "opus_amd/csrc/silk_tables.h"), "sk_", "WV_TABLE")
print("ok")
