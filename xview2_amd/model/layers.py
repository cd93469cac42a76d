from ..decoder import (ASPP, PPM, ASPPModule, AttentionLayer, ConvBlock, ConvLayer, ConvTranspose,  # noqa: F401
                       FusionBlock, OutputBlock, UpsampleBlock)
