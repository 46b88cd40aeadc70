"""Optional glue to transformers' Auto classes (reference: `AutoConfig.register("ullava_core", UllavaCoreConfig)` /
`AutoModelForCausalLM.register(UllavaCoreConfig, UllavaCoreForCausalLM)` at models/ullava_core.py:398-399 and the same pair for
"ullava" at models/ullava.py:437-438).  The hot path never imports transformers; this module is only touched by the `models` shim.

AutoConfig wants PretrainedConfig subclasses, so thin ones are generated here that carry the same fields as the plain config
objects of configuration.py; `AutoModelForCausalLM.from_pretrained(dir)` then dispatches to the model classes' own
`from_pretrained`, which accept such a config through `config=`.
"""
import importlib

_DONE = False


def register_with_transformers() -> bool:
    """Idempotent; returns False when transformers is not importable."""
    global _DONE
    if _DONE:
        return True
    try:
        from transformers import AutoConfig, AutoModelForCausalLM, PretrainedConfig
    except Exception:
        return False
    C = importlib.import_module("u-llava_amd.configuration")
    MC = importlib.import_module("u-llava_amd.modeling_core")
    MU = importlib.import_module("u-llava_amd.modeling_ullava")

    def make(plain_cls, model_type):
        class _HF(PretrainedConfig):
            def __init__(self, **kwargs):
                known = plain_cls(**{k: v for k, v in kwargs.items() if k not in ("architectures", "torch_dtype", "dtype", "transformers_version")})
                super().__init__(**{k: v for k, v in kwargs.items() if k in ("architectures", "torch_dtype")})
                for k, v in known.to_dict().items():
                    if k == "model_type":
                        continue
                    try:
                        setattr(self, k, v)
                    except AttributeError:
                        pass                                    # read-only PretrainedConfig properties (use_return_dict, ...)

            def to_plain(self):
                return plain_cls(**{k: v for k, v in self.to_dict().items() if k in plain_cls().to_dict() and k != "model_type"})
        _HF.model_type = model_type
        _HF.__name__ = plain_cls.__name__
        return _HF

    pairs = ((make(C.UllavaCoreConfig, "ullava_core"), MC.UllavaCoreForCausalLM), (make(C.UllavaConfig, "ullava"), MU.UllavaForCausalLM))
    for hf_cfg, model_cls in pairs:
        try:
            AutoConfig.register(hf_cfg.model_type, hf_cfg)
            model_cls.hf_config_class = hf_cfg
            saved = model_cls.config_class
            model_cls.config_class = hf_cfg                    # register() checks model_class.config_class against the config class
            try:
                AutoModelForCausalLM.register(hf_cfg, model_cls)
            finally:
                model_cls.config_class = saved
        except ValueError:
            pass                                                # already registered in this process (e.g. by the reference itself)
    _DONE = True
    return True
