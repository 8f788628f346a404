from .. import AutoencoderKL
