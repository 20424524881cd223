"""tests/reflib.py — loaders for the checker libraries (TEST INFRASTRUCTURE).

ref_fx()    : unmodified reference, fixed-point build (oracle/_ref/libopus_ref_fx.so)
ref_expose(): wrappers reaching reference statics (oracle/_ref/libref_expose_fx.so)
oracle()    : this repo's plain-C restatement (oracle/libcelt_oracle.so)
"""
import ctypes, os, functools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def _load(rel, mode=ctypes.RTLD_LOCAL):
    p = os.path.join(ROOT, rel)
    if not os.path.exists(p):
        return None
    return ctypes.CDLL(p, mode=mode)

@functools.lru_cache(None)
def ref_fx():
    # RTLD_LOCAL: the reference exports the same opus_* names as the product library; they must never interpose.
    # libref_expose_fx.so is linked against this file (rpath $ORIGIN), so the loader shares the one copy.
    return _load("oracle/_ref/libopus_ref_fx.so")

@functools.lru_cache(None)
def ref_fl():
    return _load("oracle/_ref/libopus_ref_fl.so")

@functools.lru_cache(None)
def ref_fxa():
    """fixed-point arithmetic with the float API: analysis.c + mlp.c active (the oracle of the analysis row; today only the gap is measured, tools/analysis_gap.py)"""
    return _load("oracle/_ref/libopus_ref_fxa.so")

@functools.lru_cache(None)
def ref_expose_fxa():
    """run_analysis of the compiled reference, frame by frame (oracle/ref_expose_fxa/x_analysis.c)"""
    ref_fxa()
    return _load("oracle/_ref/libref_expose_fxa.so")

@functools.lru_cache(None)
def ref_expose():
    ref_fx()
    return _load("oracle/_ref/libref_expose_fx.so")

@functools.lru_cache(None)
def oracle():
    return _load("oracle/libcelt_oracle.so")
