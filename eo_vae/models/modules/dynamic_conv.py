from eo_vae_amd.dynamic_conv import (DynamicConv, DynamicConv_decoder, FCResLayer,  # noqa: F401
                                     TransformerWeightGenerator, TransformerWeightGenerator_decoder,
                                     get_1d_sincos_pos_embed_from_grid_torch)
