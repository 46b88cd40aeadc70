"""Drop-in `models` package: put `<repo>/u-llava_amd/shim` in front of the reference checkout on sys.path / PYTHONPATH and
`from models import UllavaForCausalLM, KeywordsStoppingCriteria, DEFAULT_IMG_TOKEN, ...` (inference_ullava.py:19-20,
train_ullava.py, evaluation/eval_ullava.py, webui/gradio_chat.py) resolves to the MI355X implementation with no edit of the
callers.  Exports exactly the names of the reference's models/__init__.py:17-72.
"""
import importlib as _il
import os as _os
import sys as _sys

_REPO = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _REPO not in _sys.path:
    _sys.path.insert(0, _REPO)
_cfg = _il.import_module("u-llava_amd.configuration")
_core = _il.import_module("u-llava_amd.modeling_core")
_full = _il.import_module("u-llava_amd.modeling_ullava")
_tools = _il.import_module("u-llava_amd.tools")
_hf = _il.import_module("u-llava_amd.hf_integration")

UllavaConfig, UllavaCoreConfig = _cfg.UllavaConfig, _cfg.UllavaCoreConfig
UllavaForCausalLM, UllavaCoreForCausalLM = _full.UllavaForCausalLM, _core.UllavaCoreForCausalLM
KeywordsStoppingCriteria = _tools.KeywordsStoppingCriteria
smart_resize_token_embedding = _tools.smart_resize_token_embedding
smart_special_token_and_embedding_resize = _tools.smart_special_token_and_embedding_resize
multi_modal_resize_token_embedding = _tools.multi_modal_resize_token_embedding

DEFAULT_IMG_TOKEN = '<image>'
DEFAULT_IMG_PATCH_TOKEN = "<image_patch>"
DEFAULT_IMG_START_TOKEN = "<img_beg>"
DEFAULT_IMG_END_TOKEN = "</img_end>"
DEFAULT_VID_PATCH_TOKEN = "<video_patch>"
DEFAULT_VID_START_TOKEN = "<vid_beg>"
DEFAULT_VID_END_TOKEN = "</vid_end>"
DEFAULT_SEG_TOKEN = '[SEG]'
DEFAULT_LOC_TOKEN = '[LOC]'
DEFAULT_TAG_START = '[tag]'
DEFAULT_TAG_END = '[/tag]'
DEFAULT_BOS_TOKEN = '<s>'
DEFAULT_EOS_TOKEN = '</s>'
DEFAULT_UNK_TOKEN = '<unk>'
DEFAULT_PAD_TOKEN = '[PAD]'
IGNORE_INDEX = -100

# models/ullava_core.py:398-399, models/ullava.py:437-438: AutoConfig / AutoModelForCausalLM registration (when transformers is there)
_hf.register_with_transformers()
# models/ullava_core.py:78, models/ullava.py:69: @registry.register_model (when the reference's utils.registry is importable)
_hf.register_with_reference_registry()

__all__ = ["UllavaConfig", "UllavaForCausalLM", "UllavaCoreConfig", "UllavaCoreForCausalLM", "KeywordsStoppingCriteria",
           "smart_resize_token_embedding", "multi_modal_resize_token_embedding", "smart_special_token_and_embedding_resize",
           "DEFAULT_IMG_TOKEN", "DEFAULT_SEG_TOKEN", "DEFAULT_LOC_TOKEN", "DEFAULT_IMG_PATCH_TOKEN", "DEFAULT_IMG_START_TOKEN",
           "DEFAULT_IMG_END_TOKEN", "DEFAULT_VID_PATCH_TOKEN", "DEFAULT_VID_START_TOKEN", "DEFAULT_VID_END_TOKEN", "DEFAULT_BOS_TOKEN",
           "DEFAULT_EOS_TOKEN", "DEFAULT_UNK_TOKEN", "DEFAULT_PAD_TOKEN", "IGNORE_INDEX", "DEFAULT_TAG_START", "DEFAULT_TAG_END"]
