"""TEST-ONLY stand-in for the parts of diffusers==0.18.2 that /root/reference imports.

This is NOT diffusers and NOT part of the product. It exists so that the
unmodified reference modules under /root/reference/models can be imported in
the build container (which has no diffusers) to (a) validate the oracle
restatement in oracle/*.py and (b) generate the golden fixtures under
tests/golden/ (see oracle/make_golden.py). It only carries infrastructure
symbols (config mixins, logging, output containers) plus the two pieces of real
arithmetic the UNet needs from diffusers: `Timesteps` / `TimestepEmbedding`
(restated from the published diffusers 0.18.2 `models/embeddings.py`).
Nothing under rich-text-to-image_amd/ may import this.
"""
class _Placeholder:
    def __init__(self, *a, **k):
        raise RuntimeError("diffusers stand-in: this symbol is import-only")

class AutoencoderKL(_Placeholder): pass
class PNDMScheduler(_Placeholder): pass
class EulerDiscreteScheduler(_Placeholder): pass
class DPMSolverMultistepScheduler(_Placeholder): pass
