"""opus_amd — MI355X-native batched Opus encoder and decoder (CELT, SILK, hybrid; multistream, projection) behind the libopus C ABI.

Host side: a thin ctypes mirror of the reference's encoder / decoder interface (`opus_encoder_create / opus_encode /
opus_encoder_ctl`, `opus_decoder_create / opus_decode`, reference/include/opus.h:174-520) plus the additive batch API of include/opus_amd.h.
All compute happens in the hand-written HIP kernels of opus_amd/csrc (one wavefront per stream-frame);
there is NO CPU fallback: if the shared library or a GPU is missing, construction raises.

PyTorch is only plumbing here (device buffers, streams, torch.distributed for the multi-GPU gather).
"""
import ctypes, os, subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
PRODUCT_LIB_PATH = os.path.join(_HERE, "libopus_amd.so")                               # what build() writes, always
LIB_PATH = os.environ.get("OPUS_AMD_LIB") or PRODUCT_LIB_PATH                          # what lib() loads (OPUS_AMD_LIB / assignment: A/B experiments with variant builds, the emulated library of the CPU tests)

OPUS_OK, OPUS_BAD_ARG, OPUS_BUFFER_TOO_SMALL, OPUS_INTERNAL_ERROR = 0, -1, -2, -3
OPUS_INVALID_PACKET, OPUS_UNIMPLEMENTED, OPUS_INVALID_STATE, OPUS_ALLOC_FAIL = -4, -5, -6, -7
OPUS_AUTO, OPUS_BITRATE_MAX = -1000, -1
OPUS_APPLICATION_VOIP, OPUS_APPLICATION_AUDIO, OPUS_APPLICATION_RESTRICTED_LOWDELAY = 2048, 2049, 2051
OPUS_APPLICATION_RESTRICTED_SILK, OPUS_APPLICATION_RESTRICTED_CELT = 2052, 2053
OPUS_SET_BITRATE_REQUEST, OPUS_GET_BITRATE_REQUEST = 4002, 4003
OPUS_SET_MAX_BANDWIDTH_REQUEST, OPUS_SET_VBR_REQUEST, OPUS_SET_BANDWIDTH_REQUEST, OPUS_GET_BANDWIDTH_REQUEST = 4004, 4006, 4008, 4009
OPUS_SET_COMPLEXITY_REQUEST, OPUS_GET_COMPLEXITY_REQUEST = 4010, 4011
OPUS_SET_VBR_CONSTRAINT_REQUEST, OPUS_SET_FORCE_CHANNELS_REQUEST, OPUS_RESET_STATE = 4020, 4022, 4028
OPUS_GET_FINAL_RANGE_REQUEST, OPUS_SET_LSB_DEPTH_REQUEST, OPUS_SET_PHASE_INVERSION_DISABLED_REQUEST = 4031, 4036, 4046
OPUS_SET_FORCE_MODE_REQUEST, OPUS_SET_SIGNAL_REQUEST, OPUS_SET_PACKET_LOSS_PERC_REQUEST, OPUS_SET_INBAND_FEC_REQUEST, OPUS_SET_DTX_REQUEST = 11002, 4024, 4014, 4012, 4016
OPUS_MODE_SILK_ONLY, OPUS_MODE_HYBRID, OPUS_MODE_CELT_ONLY = 1000, 1001, 1002
OPUS_AMD_SET_KERNEL_TIMING_REQUEST = 11904       # HIP events around every kernel of a batch's calls (EncoderBatch.kernel_times)
OPUS_AMD_SET_FLOAT_ANALYSIS_REQUEST, OPUS_AMD_GET_FLOAT_ANALYSIS_REQUEST = 11900, 11901     # private: 0 = encode like a reference built with DISABLE_FLOAT_API
OPUS_AMD_SET_KERNEL_PIPELINE_REQUEST, OPUS_AMD_GET_KERNEL_PIPELINE_REQUEST = 11902, 11903       # private: -1 the library chooses, 0 one kernel, 1 / 2 the front / quantiser / back kernel pipeline (include/opus_amd.h)
OPUS_BANDWIDTH_NARROWBAND, OPUS_BANDWIDTH_MEDIUMBAND, OPUS_BANDWIDTH_WIDEBAND, OPUS_BANDWIDTH_SUPERWIDEBAND, OPUS_BANDWIDTH_FULLBAND = 1101, 1102, 1103, 1104, 1105

SOURCES = [os.path.join(_HERE, "csrc", "opus_amd.hip")]

def source_hash():
    """sha256 over the sources the library is compiled from (opus_amd/csrc/*, include/opus_amd.h, in name order), first 16 hex digits"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    if not os.path.isdir(csrc): return None                                                                # a box that received only the .so: nothing to hash, nothing to rebuild from
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".hip"))) + [os.path.join(_ROOT, "include", "opus_amd.h")]
    for p in files:
        h.update(os.path.basename(p).encode()); h.update(open(p, "rb").read())
    return h.hexdigest()[:16]

def built_source_hash(path=None):
    """the source hash the built library carries (the string `OA_SRC_HASH=<hex>` compiled into it, opus_amd.hip: opusgpu_build_info); None when there is no library"""
    path = path or LIB_PATH
    if not os.path.exists(path): return None
    data = open(path, "rb").read()
    i = data.find(b"OA_SRC_HASH=")
    return None if i < 0 else data[i + 12:i + 28].decode("ascii", "replace")

def build(force=False, verbose=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU).  Skipped only when the library on disk was built from exactly these sources
    (the hash it carries == the hash of the files; when the source tree is absent -- a box that received only the .so -- there is nothing to rebuild from)."""
    want = source_hash()
    if want is None:
        if os.path.exists(PRODUCT_LIB_PATH): return PRODUCT_LIB_PATH
        raise RuntimeError("opus_amd: neither the sources (opus_amd/csrc) nor a built %s are here" % PRODUCT_LIB_PATH)
    if not force and built_source_hash(PRODUCT_LIB_PATH) == want and not os.environ.get("OPUS_AMD_EXTRA_CFLAGS"):
        return PRODUCT_LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wl,-Bsymbolic", "-fvisibility=hidden", "-DOA_SOURCE_HASH=\"%s\"" % want, "-I" + os.path.join(_HERE, "csrc"),
           "-I" + os.path.join(_ROOT, "include")] + os.environ.get("OPUS_AMD_EXTRA_CFLAGS", "").split() + SOURCES + ["-o", PRODUCT_LIB_PATH]   # (extra flags: profiling experiments only)
    if verbose: print(" ".join(cmd))
    subprocess.check_call(cmd)
    return PRODUCT_LIB_PATH

_lib = None
def lib():
    """The C-ABI library; raises if it is missing (the product path never falls back to the CPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("opus_amd: %s is not built — run opus_amd.build() (hipcc --offload-arch=gfx950)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        vp, i32, u32p, i32p = ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int32)
        L.opus_encoder_create.restype = vp; L.opus_encoder_create.argtypes = [i32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.opus_encoder_init.argtypes = [vp, i32, ctypes.c_int, ctypes.c_int]
        L.opus_encode.restype = i32; L.opus_encode.argtypes = [vp, vp, ctypes.c_int, vp, i32]
        L.opus_encoder_destroy.argtypes = [vp]; L.opus_encoder_destroy.restype = None
        L.opus_strerror.restype = ctypes.c_char_p; L.opus_get_version_string.restype = ctypes.c_char_p
        if hasattr(L, "opusgpu_build_info"): L.opusgpu_build_info.restype = ctypes.c_char_p           # (variant / older libraries through OPUS_AMD_LIB may lack the newer entry points: guarded)
        L.opusgpu_enc_batch_create.restype = vp; L.opusgpu_enc_batch_create.argtypes = [i32, i32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.opusgpu_enc_batch_destroy.argtypes = [vp]; L.opusgpu_enc_batch_destroy.restype = None
        L.opusgpu_enc_batch_ctl.argtypes = [vp, i32, ctypes.c_int, i32]
        L.opusgpu_enc_batch_get.argtypes = [vp, i32, ctypes.c_int, i32p]
        L.opusgpu_encode_batch.argtypes = [vp, vp, ctypes.c_int, vp, i32, i32, vp, vp]
        L.opusgpu_encode_batch_dev.argtypes = [vp, vp, ctypes.c_int, vp, i32, i32, vp, vp, vp]
        L.opusgpu_time_encode_dev.argtypes = [vp, vp, ctypes.c_int, vp, i32, i32, vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        L.opusgpu_enc_batch_export_state.argtypes = [vp, i32, vp]; L.opusgpu_enc_batch_import_state.argtypes = [vp, i32, vp]
        L.opusgpu_enc_batch_sync.argtypes = [vp]; L.opusgpu_enc_batch_reset.argtypes = [vp]
        if hasattr(L, "opusgpu_enc_batch_copy_states"): L.opusgpu_enc_batch_copy_states.argtypes = [vp, i32, vp, i32, i32]
        if hasattr(L, "opusgpu_encode_batch_lookahead"):
            L.opusgpu_encode_batch_lookahead.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, i32, i32, vp, vp]
            L.opusgpu_encode_batch_lookahead_dev.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, i32, i32, vp, vp, vp]
        L.opusgpu_pack_packets_dev.argtypes = [vp, i32, vp, vp, vp, i32, vp]
        if hasattr(L, "opusgpu_pack_packets_cap_dev"): L.opusgpu_pack_packets_cap_dev.argtypes = [vp, i32, vp, vp, vp, i32, ctypes.c_longlong, vp]
        L.opusgpu_enc_moved_state_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        # decoder
        L.opus_decoder_create.restype = vp; L.opus_decoder_create.argtypes = [i32, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.opus_decoder_destroy.argtypes = [vp]
        L.opus_decode.argtypes = [vp, ctypes.c_char_p, i32, vp, ctypes.c_int, ctypes.c_int]
        L.opusgpu_dec_batch_create.restype = vp; L.opusgpu_dec_batch_create.argtypes = [i32, i32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.opusgpu_dec_batch_destroy.argtypes = [vp]
        if hasattr(L, "opusgpu_dec_batch_set_fast_kernel"): L.opusgpu_dec_batch_set_fast_kernel.argtypes = [vp, ctypes.c_int]
        if hasattr(L, "opusgpu_dec_batch_set_lane_kernel"): L.opusgpu_dec_batch_set_lane_kernel.argtypes = [vp, ctypes.c_int]
        if hasattr(L, "opusgpu_dec_batch_lane_stats"): L.opusgpu_dec_batch_lane_stats.argtypes = [vp, vp, vp]
        if hasattr(L, "opusgpu_dec_batch_set_pvq_stage"): L.opusgpu_dec_batch_set_pvq_stage.argtypes = [vp, ctypes.c_int]; L.opusgpu_dec_batch_pvq_stats.argtypes = [vp, vp]
        L.opusgpu_decode_batch.argtypes = [vp, vp, i32, vp, vp, ctypes.c_int, vp, vp]
        L.opusgpu_decode_batch_dev.argtypes = [vp, vp, i32, vp, vp, ctypes.c_int, vp, vp, vp]
        L.opusgpu_time_decode_dev.argtypes = [vp, vp, i32, vp, vp, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        L.opusgpu_dec_batch_export_state.argtypes = [vp, i32, vp]; L.opusgpu_dec_batch_import_state.argtypes = [vp, i32, vp]
        L.opusgpu_dec_batch_sync.argtypes = [vp]; L.opusgpu_dec_batch_reset.argtypes = [vp]
        # SILK building blocks
        L.opusgpu_nsq_batch_create.restype = vp; L.opusgpu_nsq_batch_create.argtypes = [i32, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.opusgpu_nsq_batch_destroy.argtypes = [vp]; L.opusgpu_nsq_batch_destroy.restype = None
        L.opusgpu_nsq_batch_reset.argtypes = [vp]; L.opusgpu_nsq_batch_sync.argtypes = [vp]
        L.opusgpu_nsq_batch_import_state.argtypes = [vp, i32, vp]; L.opusgpu_nsq_batch_export_state.argtypes = [vp, i32, vp]
        L.opusgpu_nsq_batch_run.argtypes = [vp, vp, vp, vp, vp]
        L.opusgpu_nsq_batch_run_dev.argtypes = [vp, vp, vp, vp, vp, vp]
        L.opusgpu_nsq_time_dev.argtypes = [vp, vp, vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        L.opusgpu_silk_lpc_analysis_filter_batch.argtypes = [ctypes.c_int, i32, vp, vp, vp, i32, i32]
        L.opusgpu_silk_lpc_analysis_filter_batch_dev.argtypes = [ctypes.c_int, i32, vp, vp, vp, i32, i32, vp]
        L.opusgpu_resampler_batch_create.restype = vp; L.opusgpu_resampler_batch_create.argtypes = [i32, i32, i32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.opusgpu_resampler_batch_destroy.argtypes = [vp]; L.opusgpu_resampler_batch_destroy.restype = None
        L.opusgpu_resampler_batch_reset.argtypes = [vp]; L.opusgpu_resampler_batch_sync.argtypes = [vp]
        L.opusgpu_resampler_batch_out_len.argtypes = [vp, i32]
        L.opusgpu_resampler_batch_run.argtypes = [vp, vp, vp, i32]; L.opusgpu_resampler_batch_run_dev.argtypes = [vp, vp, vp, i32, vp]
        L.opusgpu_resampler_batch_export_state.argtypes = [vp, i32, vp]; L.opusgpu_resampler_batch_import_state.argtypes = [vp, i32, vp]
        L.opusgpu_silk_pitch_analysis_batch.argtypes = [ctypes.c_int, i32, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.opusgpu_silk_pitch_analysis_batch_dev.argtypes = [ctypes.c_int, i32, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
        _lib = L
    return _lib

class OpusError(Exception):
    def __init__(self, code): super().__init__("%s (%d)" % (lib().opus_strerror(code).decode(), code)); self.code = code

class OpusEncoder:
    """Mirror of the reference encoder object: OpusEncoder(Fs, channels, application); .encode(pcm_int16, frame_size);
    .ctl(request, value) / .get(request).  Each encode runs on the GPU as a batch of one."""
    def __init__(self, Fs, channels, application):
        err = ctypes.c_int()
        self._L = lib()
        self._st = self._L.opus_encoder_create(Fs, channels, application, ctypes.byref(err))
        if not self._st: raise OpusError(err.value)
        self.channels = channels
        self._out = (ctypes.c_ubyte * 1500)()
        self._L.opus_encoder_ctl.restype = ctypes.c_int
    def ctl(self, request, value=0):
        self._L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int32]
        r = self._L.opus_encoder_ctl(self._st, request, value)
        if r != OPUS_OK: raise OpusError(r)
    def get(self, request):
        v = ctypes.c_int32()
        self._L.opus_encoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        r = self._L.opus_encoder_ctl(self._st, request, ctypes.byref(v))
        if r != OPUS_OK: raise OpusError(r)
        return v.value
    def encode(self, pcm, frame_size, max_data_bytes=1276):
        import numpy as np
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        n = self._L.opus_encode(self._st, pcm.ctypes.data, frame_size, self._out, max_data_bytes)
        if n < 0: raise OpusError(n)
        return bytes(self._out[:n])
    def final_range(self): return self.get(OPUS_GET_FINAL_RANGE_REQUEST) & 0xffffffff
    def __del__(self):
        if getattr(self, "_st", None): self._L.opus_encoder_destroy(self._st); self._st = None

class EncoderBatch:
    """S independent streams stepped together on one GPU (include/opus_amd.h batch API)."""
    def __init__(self, nstreams, channels=2, application=OPUS_APPLICATION_RESTRICTED_LOWDELAY, Fs=48000, device=0):
        err = ctypes.c_int()
        self._L = lib()
        self._b = self._L.opusgpu_enc_batch_create(nstreams, Fs, channels, application, device, ctypes.byref(err))
        if not self._b: raise OpusError(err.value)
        self.S, self.channels, self.device, self.Fs = nstreams, channels, device, Fs
        # record kind: the SILK-capable applications (VOIP / AUDIO / RESTRICTED_SILK) use the larger OaShStream record
        self.kind = 1 if application in (OPUS_APPLICATION_VOIP, OPUS_APPLICATION_AUDIO, OPUS_APPLICATION_RESTRICTED_SILK) else 0
        self._L.opusgpu_enc_sh_state_size.restype = ctypes.c_int
        self.state_size = self._L.opusgpu_enc_sh_state_size() if self.kind else self._L.opusgpu_enc_state_size()
    def ctl(self, request, value=0, stream=-1):
        r = self._L.opusgpu_enc_batch_ctl(self._b, stream, request, value)
        if r != OPUS_OK: raise OpusError(r)
    def get(self, request, stream):
        v = ctypes.c_int32()
        r = self._L.opusgpu_enc_batch_get(self._b, stream, request, ctypes.byref(v))
        if r != OPUS_OK: raise OpusError(r)
        return v.value
    def encode(self, pcm, frame_size, max_data_bytes=1276):
        """pcm: int16 array [S, frame_size*channels] (host).  Returns (list of packet bytes, final ranges)."""
        import numpy as np
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        assert pcm.size == self.S * frame_size * self.channels
        nf = max(1, -(-frame_size * 50 // self.Fs)) if frame_size > self.Fs // 50 else 1       # calls above 20 ms may come back as multi-frame packets
        stride = max(1280, ((min(max_data_bytes, 1276 * 6) + 15) // 16) * 16) if nf == 1 else ((max_data_bytes + 48 + 15) // 16) * 16   # oa_enc_out_stride_needed
        out = np.zeros((self.S, stride), np.uint8); lens = np.zeros(self.S, np.int32); rng = np.zeros(self.S, np.uint32)
        r = self._L.opusgpu_encode_batch(self._b, pcm.ctypes.data, frame_size, out.ctypes.data, stride, max_data_bytes, lens.ctypes.data, rng.ctypes.data)
        if r != OPUS_OK: raise OpusError(r)
        return [bytes(out[s, :max(int(lens[s]), 0)]) for s in range(self.S)], lens, rng
    def encode_dev(self, d_pcm_ptr, frame_size, d_out_ptr, out_stride, d_lens_ptr, d_rng_ptr, max_data_bytes=1276, hip_stream=None):
        r = self._L.opusgpu_encode_batch_dev(self._b, d_pcm_ptr, frame_size, d_out_ptr, out_stride, max_data_bytes, d_lens_ptr, d_rng_ptr, hip_stream)
        if r != OPUS_OK: raise OpusError(r)
    def time_encode_dev(self, d_pcm_ptr, frame_size, d_out_ptr, out_stride, d_lens_ptr, d_rng_ptr, steps, max_data_bytes=1276):
        ms = ctypes.c_float()
        r = self._L.opusgpu_time_encode_dev(self._b, d_pcm_ptr, frame_size, d_out_ptr, out_stride, max_data_bytes, d_lens_ptr, d_rng_ptr, steps, ctypes.byref(ms))
        if r != OPUS_OK: raise OpusError(r)
        return ms.value
    def kernel_times(self):
        """{kernel name: ms} of the batch's last call, in launch order (after ctl(OPUS_AMD_SET_KERNEL_TIMING_REQUEST, 1)); {} when the library predates the entry"""
        L = lib()
        if not hasattr(L, "opusgpu_enc_batch_kernel_times"): return {}
        L.opusgpu_enc_batch_kernel_times.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_int]
        names = ctypes.create_string_buffer(1024); ms = (ctypes.c_float * 20)()
        n = L.opusgpu_enc_batch_kernel_times(self._b, names, 1024, ms, 20)
        if n <= 0: return {}
        out = {}
        for k, v in zip(names.value.decode().split(","), list(ms)[:n]): out[k] = out.get(k, 0.0) + float(v)
        return out
    def export_state(self, stream):
        buf = ctypes.create_string_buffer(self.state_size)
        r = self._L.opusgpu_enc_batch_export_state(self._b, stream, buf)
        if r != OPUS_OK: raise OpusError(r)
        return buf.raw
    def import_state(self, stream, blob):
        if len(blob) != self.state_size: raise ValueError("state blob of %d bytes, this batch's records are %d bytes" % (len(blob), self.state_size))
        r = self._L.opusgpu_enc_batch_import_state(self._b, stream, blob)
        if r != OPUS_OK: raise OpusError(r)
    def copy_states_from(self, src, n, dst_first=0, src_first=0):
        """n stream records (configuration + state) of batch `src` into this one, device to device (opusgpu_enc_batch_copy_states)"""
        r = self._L.opusgpu_enc_batch_copy_states(self._b, dst_first, src._b, src_first, n)
        if r != OPUS_OK: raise OpusError(r)
    def reset(self):
        r = self._L.opusgpu_enc_batch_reset(self._b)
        if r != OPUS_OK: raise OpusError(r)
    def sync(self): self._L.opusgpu_enc_batch_sync(self._b)
    def close(self):
        if getattr(self, "_b", None): self._L.opusgpu_enc_batch_destroy(self._b); self._b = None
    def __del__(self): self.close()


class OpusDecoder:
    """Mirror of the reference decoder object: OpusDecoder(Fs, channels); .decode(packet, frame_size) -> int16 [n, channels].
    CELT-only packets; each decode runs on the GPU as a batch of one."""
    def __init__(self, Fs, channels):
        err = ctypes.c_int()
        self._L = lib()
        self._st = self._L.opus_decoder_create(Fs, channels, ctypes.byref(err))
        if not self._st: raise OpusError(err.value)
        self.channels = channels
    def decode(self, packet, frame_size=5760, decode_fec=0):
        import numpy as np
        pcm = np.zeros((frame_size, self.channels), np.int16)
        n = self._L.opus_decode(self._st, packet, len(packet) if packet else 0, pcm.ctypes.data, frame_size, decode_fec)
        if n < 0: raise OpusError(n)
        return pcm[:n].copy()
    def final_range(self):
        v = ctypes.c_uint32()
        self._L.opus_decoder_ctl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        r = self._L.opus_decoder_ctl(self._st, OPUS_GET_FINAL_RANGE_REQUEST, ctypes.byref(v))
        if r != OPUS_OK: raise OpusError(r)
        return v.value
    def __del__(self):
        if getattr(self, "_st", None): self._L.opus_decoder_destroy(self._st); self._st = None

DEC_PVQ_STAGE_DEFAULT = None      # tests: opusgpu_dec_batch_set_pvq_stage of every DecoderBatch created from here on (None: the library's default, wide calls only)
class DecoderBatch:
    """S independent streams decoded together on one GPU (include/opus_amd.h batch decoder API)."""
    def __init__(self, nstreams, channels=2, Fs=48000, device=0):
        err = ctypes.c_int()
        self._L = lib()
        self._b = self._L.opusgpu_dec_batch_create(nstreams, Fs, channels, device, ctypes.byref(err))
        if not self._b: raise OpusError(err.value)
        self.S, self.channels, self.device = nstreams, channels, device
        if DEC_PVQ_STAGE_DEFAULT is not None: self.set_pvq_stage(DEC_PVQ_STAGE_DEFAULT)
    def decode(self, packets, frame_size=960):
        """packets: list of S bytes objects (b"" or None = lost packet: conceal frame_size samples).  Returns (pcm int16 [S, frame_size, channels],
        nsamples [S], final ranges [S])."""
        import numpy as np
        assert len(packets) == self.S
        packets = [p or b"" for p in packets]
        stride = (max(len(p) for p in packets) + 8 + 3) & ~3
        buf = np.zeros((self.S, stride), np.uint8)
        lens = np.array([len(p) for p in packets], np.int32)
        for s, p in enumerate(packets): buf[s, :len(p)] = np.frombuffer(p, np.uint8)
        pcm = np.zeros((self.S, frame_size, self.channels), np.int16); ns = np.zeros(self.S, np.int32); rng = np.zeros(self.S, np.uint32)
        r = self._L.opusgpu_decode_batch(self._b, buf.ctypes.data, stride, lens.ctypes.data, pcm.ctypes.data, frame_size, ns.ctypes.data, rng.ctypes.data)
        if r != OPUS_OK: raise OpusError(r)
        return pcm, ns, rng
    def set_fast_kernel(self, enable):
        """False: the following calls skip the CELT-only fast kernel (for batches that carry no CELT-only packets); the output is the same either way"""
        r = self._L.opusgpu_dec_batch_set_fast_kernel(self._b, 1 if enable else 0)
        if r != OPUS_OK: raise OpusError(r)
    def lane_stats(self):
        """(packets of the last call given to the lane = stream SILK kernel, of those handed on to the general kernel)"""
        a, b = ctypes.c_uint32(), ctypes.c_uint32()
        r = self._L.opusgpu_dec_batch_lane_stats(self._b, ctypes.byref(a), ctypes.byref(b))
        if r != OPUS_OK: raise OpusError(r)
        return a.value, b.value
    def set_lane_kernel(self, enable):
        """False: the following calls skip the lane = stream SILK kernel (oa_sdec_lane_kernel); the output is the same either way"""
        r = self._L.opusgpu_dec_batch_set_lane_kernel(self._b, 1 if enable else 0)
        if r != OPUS_OK: raise OpusError(r)
    def set_pvq_stage(self, mode):
        """-1: wide calls (the default), 0: never, 1: always -- the bands of steady-state CELT frames on oa_celt_dpvq_kernel (four streams per wave); the output is the same either way"""
        r = self._L.opusgpu_dec_batch_set_pvq_stage(self._b, int(mode))
        if r != OPUS_OK: raise OpusError(r)
    def pvq_stats(self):
        """frames of the last call whose bands oa_celt_dpvq_kernel decoded"""
        a = ctypes.c_uint32()
        r = self._L.opusgpu_dec_batch_pvq_stats(self._b, ctypes.byref(a))
        if r != OPUS_OK: raise OpusError(r)
        return a.value
    def decode_dev(self, d_pkt_ptr, stride, d_lens_ptr, d_pcm_ptr, frame_size, d_ns_ptr, d_rng_ptr, hip_stream=None):
        r = self._L.opusgpu_decode_batch_dev(self._b, d_pkt_ptr, stride, d_lens_ptr, d_pcm_ptr, frame_size, d_ns_ptr, d_rng_ptr, hip_stream)
        if r != OPUS_OK: raise OpusError(r)
    def export_state(self, stream):
        buf = ctypes.create_string_buffer(self._L.opusgpu_dec_state_size())
        r = self._L.opusgpu_dec_batch_export_state(self._b, stream, buf)
        if r != OPUS_OK: raise OpusError(r)
        return buf.raw
    def import_state(self, stream, blob):
        r = self._L.opusgpu_dec_batch_import_state(self._b, stream, blob)
        if r != OPUS_OK: raise OpusError(r)
    def reset(self):
        r = self._L.opusgpu_dec_batch_reset(self._b)
        if r != OPUS_OK: raise OpusError(r)
    def sync(self): self._L.opusgpu_dec_batch_sync(self._b)
    def close(self):
        if getattr(self, "_b", None): self._L.opusgpu_dec_batch_destroy(self._b); self._b = None
    def __del__(self): self.close()


class NsqBatch:
    """N independent SILK channels' noise-shaping quantisers on one GPU (include/opus_amd.h opusgpu_nsq_*; reference silk/NSQ.c:76,
    silk/NSQ_del_dec.c:114).  cfg = (fs_kHz, nb_subfr, predictLPCOrder, shapingLPCOrder, nStatesDelayedDecision, warping_Q16)."""
    def __init__(self, nstreams, cfg, device=0):
        import numpy as np
        err = ctypes.c_int()
        self._L = lib()
        self.cfg = np.ascontiguousarray(cfg, dtype=np.int32)
        self._b = self._L.opusgpu_nsq_batch_create(nstreams, self.cfg.ctypes.data, device, ctypes.byref(err))
        if not self._b: raise OpusError(err.value)
        self.n, self.device = nstreams, device
        self.frame_length = int(self.cfg[1]) * 5 * int(self.cfg[0])
    def run(self, frames, x16):
        """frames: structured array [n] laid out as OpusGpuNsqFrame (388 B); x16: int16 [n, frame_length].  Returns (pulses int8 [n, frame_length], seed int8 [n])."""
        import numpy as np
        assert frames.shape == (self.n,) and frames.dtype.itemsize == 388 and x16.shape == (self.n, self.frame_length) and x16.dtype == np.int16
        frames = np.ascontiguousarray(frames); x16 = np.ascontiguousarray(x16)
        pulses = np.zeros((self.n, self.frame_length), np.int8); seed = np.zeros(self.n, np.int8)
        r = self._L.opusgpu_nsq_batch_run(self._b, frames.ctypes.data, x16.ctypes.data, pulses.ctypes.data, seed.ctypes.data)
        if r != OPUS_OK: raise OpusError(r)
        return pulses, seed
    def run_dev(self, d_frames_ptr, d_x16_ptr, d_pulses_ptr, d_seed_ptr=None, hip_stream=None):
        r = self._L.opusgpu_nsq_batch_run_dev(self._b, d_frames_ptr, d_x16_ptr, d_pulses_ptr, d_seed_ptr, hip_stream)
        if r != OPUS_OK: raise OpusError(r)
    def time_dev(self, d_frames_ptr, d_x16_ptr, d_pulses_ptr, steps):
        ms = ctypes.c_float()
        r = self._L.opusgpu_nsq_time_dev(self._b, d_frames_ptr, d_x16_ptr, d_pulses_ptr, steps, ctypes.byref(ms))
        if r != OPUS_OK: raise OpusError(r)
        return ms.value
    def export_state(self, stream):
        buf = ctypes.create_string_buffer(self._L.opusgpu_nsq_state_size())
        r = self._L.opusgpu_nsq_batch_export_state(self._b, stream, buf)
        if r != OPUS_OK: raise OpusError(r)
        return buf.raw
    def import_state(self, stream, blob):
        r = self._L.opusgpu_nsq_batch_import_state(self._b, stream, bytes(blob))
        if r != OPUS_OK: raise OpusError(r)
    def reset(self):
        r = self._L.opusgpu_nsq_batch_reset(self._b)
        if r != OPUS_OK: raise OpusError(r)
    def sync(self): self._L.opusgpu_nsq_batch_sync(self._b)
    def close(self):
        if getattr(self, "_b", None): self._L.opusgpu_nsq_batch_destroy(self._b); self._b = None
    def __del__(self): self.close()


def silk_lpc_analysis_filter(x, B, device=0):
    """silk_LPC_analysis_filter (silk/LPC_analysis_filter.c:49) for n signals: x int16 [n, len], B int16 [n, d] (Q12) -> int16 [n, len]."""
    import numpy as np
    x = np.ascontiguousarray(x, dtype=np.int16); B = np.ascontiguousarray(B, dtype=np.int16)
    assert x.ndim == 2 and B.ndim == 2 and x.shape[0] == B.shape[0]
    out = np.zeros_like(x)
    r = lib().opusgpu_silk_lpc_analysis_filter_batch(device, x.shape[0], out.ctypes.data, x.ctypes.data, B.ctypes.data, x.shape[1], B.shape[1])
    if r != OPUS_OK: raise OpusError(r)
    return out


class ResamplerBatch:
    """n independent channels of one SILK rate pair (include/opus_amd.h opusgpu_resampler_*; reference silk/resampler.c:79,:183)."""
    def __init__(self, nchannels, Fs_in, Fs_out, for_enc=1, device=0):
        err = ctypes.c_int()
        self._L = lib()
        self._b = self._L.opusgpu_resampler_batch_create(nchannels, Fs_in, Fs_out, for_enc, device, ctypes.byref(err))
        if not self._b: raise OpusError(err.value)
        self.n, self.Fs_in, self.Fs_out = nchannels, Fs_in, Fs_out
    def run(self, x):
        """x int16 [n, inLen] (whole milliseconds) -> int16 [n, inLen * Fs_out / Fs_in]"""
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.int16); assert x.ndim == 2 and x.shape[0] == self.n
        ol = self._L.opusgpu_resampler_batch_out_len(self._b, x.shape[1])
        if ol < 0: raise OpusError(ol)
        out = np.zeros((self.n, ol), np.int16)
        r = self._L.opusgpu_resampler_batch_run(self._b, out.ctypes.data, x.ctypes.data, x.shape[1])
        if r != OPUS_OK: raise OpusError(r)
        return out
    def run_dev(self, d_out_ptr, d_in_ptr, in_len, hip_stream=None):
        r = self._L.opusgpu_resampler_batch_run_dev(self._b, d_out_ptr, d_in_ptr, in_len, hip_stream)
        if r != OPUS_OK: raise OpusError(r)
    def export_state(self, channel):
        buf = ctypes.create_string_buffer(self._L.opusgpu_resampler_state_size())
        r = self._L.opusgpu_resampler_batch_export_state(self._b, channel, buf)
        if r != OPUS_OK: raise OpusError(r)
        return buf.raw
    def import_state(self, channel, blob):
        r = self._L.opusgpu_resampler_batch_import_state(self._b, channel, bytes(blob))
        if r != OPUS_OK: raise OpusError(r)
    def reset(self):
        r = self._L.opusgpu_resampler_batch_reset(self._b)
        if r != OPUS_OK: raise OpusError(r)
    def sync(self): self._L.opusgpu_resampler_batch_sync(self._b)
    def close(self):
        if getattr(self, "_b", None): self._L.opusgpu_resampler_batch_destroy(self._b); self._b = None
    def __del__(self): self.close()


def silk_pitch_analysis(frames, params, Fs_kHz, complexity, nb_subfr, device=0):
    """silk_pitch_analysis_core (silk/fixed/pitch_analysis_core_FIX.c:82) for n buffers: frames int16 [n, (20+5*nb_subfr)*Fs_kHz]; params: structured
    array [n] laid out as OpusGpuPitchIn (4 x int32).  Returns a structured array [n] laid out as OpusGpuPitchOut (24 bytes)."""
    import numpy as np
    frames = np.ascontiguousarray(frames, dtype=np.int16); params = np.ascontiguousarray(params)
    n = frames.shape[0]
    assert frames.shape == (n, (20 + 5 * nb_subfr) * Fs_kHz) and params.shape == (n,) and params.dtype.itemsize == 16
    out = np.zeros(n, np.dtype([("pitch", "<i4", 4), ("LTPCorr_Q15", "<i4"), ("lagIndex", "<i2"), ("contourIndex", "i1"), ("unvoiced", "i1")], align=True))
    r = lib().opusgpu_silk_pitch_analysis_batch(device, n, frames.ctypes.data, params.ctypes.data, out.ctypes.data, Fs_kHz, complexity, nb_subfr)
    if r != OPUS_OK: raise OpusError(r)
    return out
