"""lora_amd — MI355X (gfx950) native LoRA hot path behind the ``lora_diffusion`` API.

``from lora_amd import *`` gives the names ``from lora_diffusion import *`` gives for the
injection / patch / merge / serialisation surface (reference ``lora_diffusion/__init__.py:1-5``,
``lora.py``).
"""
from .lora import *  # noqa: F401,F403
from .lora import (  # noqa: F401  (underscore names the reference's CLIs import)
    _find_modules, _find_modules_v2, _find_children, _text_lora_path, _ti_lora_path,
)
from . import _C  # noqa: F401
from .lora_manager import DummySafeTensorObject, LoRAManager, lora_join  # noqa: F401  (reference __init__.py:3)

__version__ = "0.2.0"
