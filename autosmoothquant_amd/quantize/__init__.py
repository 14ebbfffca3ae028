"""Weight-side conversion and calibration helpers: what runs once, before the hot path."""
from .smooth import smooth_ln_fcs, smooth_lm  # noqa: F401
from .calibration import (decoder_layer_scales, get_act_scales, get_io_absmax, get_static_decoder_layer_scales,  # noqa: F401
                          parse_quant_config, dataset_batches, get_act_scales_from_dataset, get_static_decoder_layer_scales_from_dataset,
                          replace_module, get_layers_to_ignore, quantize_activations_fp8)
