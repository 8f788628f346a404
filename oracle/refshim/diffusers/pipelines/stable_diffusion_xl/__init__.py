from dataclasses import dataclass
from typing import Any
from ...utils import BaseOutput

@dataclass
class StableDiffusionXLPipelineOutput(BaseOutput):
    images: Any
