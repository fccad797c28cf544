"""UNet golden-case table shared by make_golden.py and the tests (data only)."""
_BASE = dict(dropout=0.0, conv_resample=True, num_heads=1, use_scale_shift_norm=False, resblock_updown=True,
             attention_resolutions=[8, 4], channel_mult=[1, 2, 4], model_channels=32)
UNET_CASES = {
    "tiny_l64": (dict(_BASE, image_size=64, in_channels=1, out_channels=1, num_res_blocks=1), 2, 64),
    "tiny_l96_lat3": (dict(_BASE, image_size=96, in_channels=3, out_channels=3, num_res_blocks=2), 2, 96),
    "small_l256": (dict(_BASE, image_size=256, in_channels=1, out_channels=1, num_res_blocks=2), 3, 256),
}
# config/config_ldm.yaml:30-43 (latent_channels = 1), the BASELINE UNet: (kwargs, B, L)
UNET_FULL = (dict(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
                  channel_mult=[1, 2, 4], dropout=0.0, conv_resample=True, num_heads=1, use_scale_shift_norm=False,
                  resblock_updown=True), 2, 768)
# Round 6: constructor branches no reference yaml takes (unet.py:132-166 heads, :177-224 + :462-470 conv / pool resampling outside the
# ResBlocks, :318-322 scale-shift norm) -- goldens by tests/golden/make_golden_r6.py, files unet_<name>.npz like the cases above
_OPT = dict(_BASE, image_size=64, in_channels=1, out_channels=1, num_res_blocks=1)
UNET_OPTION_CASES = {
    "opt_heads4": (dict(_OPT, num_heads=4), 2, 64),
    "opt_headch16": (dict(_OPT, num_head_channels=16), 2, 64),
    "opt_heads2_up4": (dict(_OPT, num_heads=2, num_heads_upsample=4), 2, 64),
    "opt_ssn": (dict(_OPT, use_scale_shift_norm=True), 2, 64),
    "opt_conv_resample": (dict(_OPT, resblock_updown=False, conv_resample=True), 2, 64),
    "opt_pool_resample": (dict(_OPT, resblock_updown=False, conv_resample=False), 2, 64),
    "opt_all": (dict(_OPT, in_channels=3, out_channels=3, num_res_blocks=2, num_head_channels=8, use_scale_shift_norm=True, resblock_updown=False,
                     conv_resample=True), 2, 96),
}
