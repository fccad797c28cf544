"""ctypes binding of libeegldm.so (the C ABI declared in include/eegldm.h).

The HIP library is the product: if it is missing or fails to load this module
raises -- there is no CPU or PyTorch fallback for any compute entry point.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EEGLDM_LIB") or os.path.join(_HERE, "libeegldm.so")   # EEGLDM_LIB: developer override (instrumented builds)

F32, BF16, F16 = 0, 1, 2
PRED = {"epsilon": 0, "v_prediction": 1, "sample": 2}


class LibraryMissing(RuntimeError):
    pass


def _load():
    # torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so).  If libeegldm.so is loaded before torch, the loader binds
    # the system runtime instead and the process ends up with two HIP runtimes -- hipSetDevice then reports "no ROCm-capable device"
    # (seen when __graft_entry__.build() and smoke() ran in one process).  torch is plumbing here (device memory, streams).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(f"{LIB_PATH} not found: build it with `make` (python -c 'import __graft_entry__ as g; g.build()')")
    return C.CDLL(LIB_PATH)


lib = _load()
lib.eegldm_last_error.restype = C.c_char_p
if hasattr(lib, "eegldm_ctx_stream"):
    lib.eegldm_ctx_stream.restype = C.c_void_p
if hasattr(lib, 'eegldm_unet_num_params'):
    lib.eegldm_unet_num_params.restype = C.c_long

# (name, argtypes) -- every symbol include/eegldm.h declares; tests/test_abi.py checks the list against the header
_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float
SIGNATURES = {
    "eegldm_abi_version": [],
    "eegldm_last_error": [],
    "eegldm_ctx_create": [_i, _vp, _i, C.POINTER(_vp)],
    "eegldm_ctx_stream": [_vp],
    "eegldm_ctx_destroy": [_vp],
    "eegldm_ctx_sync": [_vp],
    "eegldm_timer_start": [_vp],
    "eegldm_timer_stop_ms": [_vp, C.POINTER(_f)],
    "eegldm_prof_enable": [_vp, _i],
    "eegldm_prof_summary": [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i)],
    "eegldm_prof_bracket_overhead_ms": [_vp, C.POINTER(C.c_double)],
    "eegldm_prof_dump": [_vp, C.c_char_p],
    "eegldm_debug_reload_env": [],
    "eegldm_ncl_to_nlc": [_vp, _vp, _vp, _l, _i, _i, _i, _i],
    "eegldm_nlc_to_ncl": [_vp, _vp, _l, _vp, _i, _i, _i, _i],
    "eegldm_pack_conv_weight": [_vp, _vp, _vp, _i, _i, _i],
    "eegldm_conv1d_pack_kblocked": [_vp, _vp, _vp, _i, _i, _i],
    "eegldm_conv1d_pack_kblocked_k": [_vp, _vp, _vp, _i, _i, _i, _i],
    "eegldm_batchnorm_lrelu_fwd": [_vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _i, _f, _i, _i],
    "eegldm_batchnorm_lrelu_bwd": [_vp, _vp, _l, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _vp, _l, _i, _f, _i],
    "eegldm_kl_reparam_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i],
    "eegldm_kl_reparam_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _f, _i],
    "eegldm_avgpool2_fwd": [_vp, _vp, _l, _vp, _l, _i, _i, _i, _i],
    "eegldm_avgpool2_bwd": [_vp, _vp, _l, _vp, _l, _i, _i, _i, _i],
    "eegldm_nearest2_fwd": [_vp, _vp, _l, _vp, _l, _i, _i, _i, _i],
    "eegldm_nearest2_bwd": [_vp, _vp, _l, _vp, _l, _i, _i, _i, _i],
    "eegldm_conv1d_skip_fwd": [_vp, _vp, _l, _vp, _vp, _vp, _l, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _vp, _l, _i],
    "eegldm_comm_unique_id": [_vp],
    "eegldm_comm_create": [_vp, _vp, _i, _i, _vp],
    "eegldm_comm_destroy": [_vp],
    "eegldm_comm_rank": [_vp],
    "eegldm_comm_world": [_vp],
    "eegldm_comm_allreduce_mean_f32": [_vp, _vp, _l, _l],
    "eegldm_comm_broadcast_f32": [_vp, _vp, _l, _i],
    "eegldm_comm_wait": [_vp],
    "eegldm_conv1d_forget_kblocked": [_vp, _vp],
    "eegldm_conv1d_pack_stride2": [_vp, _vp, _vp, _vp, _i, _i, _i],
    "eegldm_conv1d_pack_dgrad": [_vp, _vp, _vp, _i, _i, _i],
    "eegldm_conv1d_pack_dgrad_k": [_vp, _vp, _vp, _i, _i, _i, _i],
    "eegldm_unpack_conv_weight": [_vp, _vp, _vp, _i, _i, _i],
    "eegldm_cast": [_vp, _vp, _vp, _l, _i],
    "eegldm_conv1d_fwd": [_vp, _vp, _l, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _l, _vp, _l, _i],
    "eegldm_conv1d_bwd_data": [_vp, _vp, _l, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _l, _i],
    "eegldm_conv1d_bwd_weight": [_vp, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i],
    "eegldm_linear_fwd": [_vp, _vp, _l, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i],
    "eegldm_linear_bwd": [_vp, _vp, _l, _vp, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _i, _i],
    "eegldm_groupnorm_fwd": [_vp, _vp, _l, _vp, _vp, _vp, _l, _vp, _i, _i, _i, _i, _f, _i, _i, _vp, _l, _i],
    "eegldm_groupnorm_bwd": [_vp, _vp, _l, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _l, _i],
    "eegldm_attention_fwd": [_vp, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _i],
    "eegldm_attention_bwd": [_vp, _vp, _l, _vp, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _i],
    "eegldm_add_noise": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _l],
    "eegldm_get_velocity": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _l],
    "eegldm_ddim_step": [_vp, _vp, _vp, _f, _f, _i, _i, _vp, _vp, _l],
    "eegldm_ddpm_step": [_vp, _vp, _vp, _vp, _f, _f, _f, _i, _i, _vp, _vp, _l],
    "eegldm_ddim_step_eta": [_vp, _vp, _vp, _vp, _f, _f, _f, _i, _i, _vp, _vp, _l],
    "eegldm_ddpm_step_var": [_vp, _vp, _vp, _vp, _f, _f, _f, _i, _i, _i, _vp, _vp, _l],
    "eegldm_mse_loss": [_vp, _vp, _vp, _vp, _vp, _l, _f],
    "eegldm_adam_step": [_vp, _vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _i, _f],
    "eegldm_grad_check_finite": [_vp, _vp, _l, _vp],
    "eegldm_randn": [_vp, _vp, _l, C.c_uint64, C.c_uint64],
    "eegldm_randint": [_vp, _vp, _l, C.c_int64, C.c_uint64, C.c_uint64],
    "eegldm_fill": [_vp, _vp, _l, _f],
    "eegldm_unet_create": [_vp, _vp, C.POINTER(_vp)],
    "eegldm_unet_destroy": [_vp],
    "eegldm_unet_set_dropout": [_vp, _f, C.c_uint64],
    "eegldm_dropout": [_vp, _vp, _l, _l, _i, _f, C.c_uint64, C.c_uint64, _i],
    "eegldm_unet_num_entries": [_vp],
    "eegldm_unet_num_params": [_vp],
    "eegldm_unet_entry": [_vp, _i, C.c_char_p, _i, C.POINTER(_l), C.POINTER(_l), C.POINTER(_i), C.POINTER(_i)],
    "eegldm_unet_bind": [_vp, _vp, _vp],
    "eegldm_unet_sync_weights": [_vp],
    "eegldm_unet_set_grad_hook": [_vp, _vp, _vp],
    "eegldm_unet_forward": [_vp, _vp, _vp, _vp, _i, _i, _i],
    "eegldm_unet_backward": [_vp, _vp, _vp],
    "eegldm_ldm_train_step": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "eegldm_l1_loss": [_vp, _vp, _vp, _vp, _vp, _l, _f],
    "eegldm_lsgan_loss": [_vp, _vp, _i, _vp, _vp, _l, _f],
    "eegldm_spectral_loss": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f],
    "eegldm_axpy": [_vp, _vp, _vp, _f, _l],
    "eegldm_aekl_create": [_vp, _vp, C.POINTER(_vp)],
    "eegldm_aekl_destroy": [_vp],
    "eegldm_aekl_num_entries": [_vp],
    "eegldm_aekl_num_params": [_vp],
    "eegldm_aekl_entry": [_vp, _i, C.c_char_p, _i, C.POINTER(_l), C.POINTER(_l), C.POINTER(_i), C.POINTER(_i)],
    "eegldm_aekl_bind": [_vp, _vp, _vp],
    "eegldm_aekl_sync_weights": [_vp],
    "eegldm_aekl_encode": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i],
    "eegldm_aekl_decode": [_vp, _vp, _vp, _i, _i],
    "eegldm_aekl_forward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i],
    "eegldm_aekl_backward": [_vp, _vp, _f, _vp],
    "eegldm_aekl_backward_ex": [_vp, _vp, _vp, _vp, _f, _vp],
    "eegldm_ms_ssim_1d": [_vp, _vp, _vp, _vp, _i, _i, _i, C.POINTER(_f), _i, C.POINTER(_f), _i, _f, _f, _f],
    "eegldm_psd_multitaper": [_vp, _vp, _vp, C.POINTER(_f), _i, _f, _i, _vp, _i, _i],
    "eegldm_sample": [_vp, _vp, _vp, C.POINTER(C.c_int64), C.POINTER(_f), C.POINTER(_f), C.POINTER(_f), _i, _i, _i, _i, _f, C.c_uint64, _vp, _vp,
                      _i, _i, _i, C.POINTER(_i)],
    "eegldm_disc_create": [_vp, _vp, C.POINTER(_vp)],
    "eegldm_disc_destroy": [_vp],
    "eegldm_disc_num_entries": [_vp],
    "eegldm_disc_num_params": [_vp],
    "eegldm_disc_entry": [_vp, _i, C.c_char_p, _i, C.POINTER(_l), C.POINTER(_l), C.POINTER(_i), C.POINTER(_i)],
    "eegldm_disc_num_buffer_entries": [_vp],
    "eegldm_disc_num_buffers": [_vp],
    "eegldm_disc_buffer_entry": [_vp, _i, C.c_char_p, _i, C.POINTER(_l), C.POINTER(_l), C.POINTER(_i), C.POINTER(_i)],
    "eegldm_disc_bind": [_vp, _vp, _vp, _vp],
    "eegldm_disc_sync_weights": [_vp],
    "eegldm_disc_forward": [_vp, _vp, _vp, _i, _i, _i],
    "eegldm_disc_feature": [_vp, _i, _vp, _vp, _vp],
    "eegldm_disc_backward": [_vp, _vp, _vp, _i],
    "eegldm_aekl_train_step": [_vp, _vp, _vp, _vp, _f, _f, _f, _i, _vp, _vp, _i, _i],
    "eegldm_usleep_create": [_vp, _vp, _vp],
    "eegldm_usleep_destroy": [_vp],
    "eegldm_usleep_num_params": [_vp],
    "eegldm_usleep_num_buffers": [_vp],
    "eegldm_usleep_num_entries": [_vp],
    "eegldm_usleep_channel": [_vp, _i],
    "eegldm_usleep_entry": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "eegldm_usleep_bind": [_vp, _vp, _vp],
    "eegldm_usleep_forward": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "eegldm_feature_moments": [_vp, _vp, _l, _i, _vp, _vp],
    "eegldm_resblock_create": [_vp, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)],
    "eegldm_attnblock_create": [_vp, _i, _i, _i, C.POINTER(_vp)],
    "eegldm_block_destroy": [_vp],
    "eegldm_block_num_entries": [_vp],
    "eegldm_block_num_params": [_vp],
    "eegldm_block_entry": [_vp, _i, C.c_char_p, _i, C.POINTER(_l), C.POINTER(_l), C.POINTER(_i), C.POINTER(_i)],
    "eegldm_block_bind": [_vp, _vp, _vp],
    "eegldm_block_forward": [_vp, _vp, _vp, _vp, _i, _i],
    "eegldm_block_backward": [_vp, _vp, _vp, _vp],
    "eegldm_timestep_embedding": [_vp, _vp, _vp, _i, _i],
    "eegldm_conv1d_fwd_gn": [_vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _l, _i, _i, _i, _i, _vp, _l, _vp, _l, _i],
}
for _n in ("eegldm_block_num_params", "eegldm_aekl_num_params", "eegldm_disc_num_params", "eegldm_disc_num_buffers", "eegldm_usleep_num_params", "eegldm_usleep_num_buffers"):
    if hasattr(lib, _n):
        getattr(lib, _n).restype = C.c_long


class AeklCfg(C.Structure):
    _fields_ = [("in_channels", _i), ("out_channels", _i), ("n_levels", _i), ("num_channels", _i * 8),
                ("latent_channels", _i), ("num_res_blocks", _i), ("norm_num_groups", _i), ("dtype", _i)]


class DiscCfg(C.Structure):
    _fields_ = [("in_channels", _i), ("out_channels", _i), ("num_channels", _i), ("num_layers_d", _i),
                ("kernel_size", _i), ("padding", _i), ("bias", _i), ("dtype", _i)]


class USleepCfg(C.Structure):
    _fields_ = [("in_chans", _i), ("depth", _i), ("n_time_filters", _i), ("n_classes", _i), ("kernel_size", _i), ("input_size", _i),
                ("with_skip_connection", _i), ("complexity_factor", C.c_float)]


class UNetCfg(C.Structure):
    _fields_ = [("in_channels", _i), ("out_channels", _i), ("model_channels", _i), ("num_res_blocks", _i),
                ("n_mult", _i), ("channel_mult", _i * 8), ("n_attn", _i), ("attention_resolutions", _i * 8),
                ("num_heads", _i), ("dtype", _i),
                # ABI 8 (zero = the config_ldm.yaml behaviour)
                ("num_head_channels", _i), ("num_heads_upsample", _i), ("use_scale_shift_norm", _i), ("resample_layers", _i),
                ("resample_pool_only", _i)]


def _bind():
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            continue          # reported by tests/test_abi.py; compute calls fail loudly below
        fn.argtypes = args


_bind()


def check(rc):
    if rc != 0:
        raise RuntimeError(f"libeegldm error {rc}: {lib.eegldm_last_error().decode(errors='replace')}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


class Context:
    """One per process / GPU; enqueues on torch's current stream so torch ops and library
    kernels are ordered with respect to each other."""

    def __init__(self, device=0, use_torch_stream=True):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("eegldm needs an MI355X (gfx950) GPU: no HIP device is visible; there is no CPU fallback")
        self.device = device
        stream = torch.cuda.current_stream(device).cuda_stream if use_torch_stream else 0
        h = C.c_void_p()
        check(lib.eegldm_ctx_create(device, C.c_void_p(stream), 0 if use_torch_stream else 1, C.byref(h)))
        self.h = h
        # the REAL stream (with use_torch_stream=False the library made its own; 0 would wrongly look like torch's default stream)
        self.stream_handle = int(lib.eegldm_ctx_stream(h) or 0)
        self.owns_stream = not use_torch_stream

    def sync(self):
        check(lib.eegldm_ctx_sync(self.h))

    def timer_start(self):
        check(lib.eegldm_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float()
        check(lib.eegldm_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    PROF_CLASSES = ["conv3_fwd_implicit_gemm", "conv3_dgrad_implicit_gemm", "conv_wgrad_splitk_gemm", "gemm_nt", "gemm_nn", "gemm_tn"]

    def prof_enable(self, on=True):
        check(lib.eegldm_prof_enable(self.h, 1 if on else 0))

    def prof_summary(self):
        out = {}
        for i, name in enumerate(self.PROF_CLASSES):
            f, ms, n = C.c_double(), C.c_double(), C.c_int()
            check(lib.eegldm_prof_summary(self.h, i, C.byref(f), C.byref(ms), C.byref(n)))
            out[name] = dict(flops=f.value, ms=ms.value, launches=n.value)
        return out

    def prof_bracket_overhead_us(self):
        ms = C.c_double()
        check(lib.eegldm_prof_bracket_overhead_ms(self.h, C.byref(ms)))
        return 1e3 * ms.value

    def __del__(self):
        try:
            lib.eegldm_ctx_destroy(self.h)
        except Exception:
            pass


_default = {}


def set_deterministic(on=True):
    """Bit-reproducible train steps (EEGLDM_DETERMINISTIC): every order-dependent reduction of the library -- fp32 atomics of the bias /
    GroupNorm / thin-conv gradients and of the loss sums, fused column sums inside the weight-gradient GEMM, split-K without a workspace --
    takes a written-partials + fixed-order-fold route.  Same arithmetic up to summation order; a few per cent slower.  The reference's
    counterpart is `torch.use_deterministic_algorithms(True)` around /root/reference/src/train_ldm.py / train_autoencoderkl.py."""
    if on:
        os.environ["EEGLDM_DETERMINISTIC"] = "1"
    else:
        os.environ.pop("EEGLDM_DETERMINISTIC", None)
    lib.eegldm_debug_reload_env()


def default_context(device=0):
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]
