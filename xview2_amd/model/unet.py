from ..networks import *  # noqa: F401,F403
from ..networks import get_decoder, get_dmg_unet, get_encoder, get_nclass  # noqa: F401
