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
