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


def register_with_reference_registry() -> bool:
    """models/ullava_core.py:78, models/ullava.py:69 (`@registry.register_model('ullava_core' / 'ullava')`): when the reference's own
    `utils.registry` is importable -- the callers run from the reference checkout with the shim in front of it on sys.path -- the two
    model classes are entered under the reference's names, so `registry.get_model_class(cfg.model.arch)` (utils/config_builder.py:46,
    tasks/base_task.py:15) resolves to the MI355X implementation.  Idempotent; returns False when there is no such registry."""
    try:
        reg = importlib.import_module("utils.registry").registry
        mapping = reg.mapping["model_name_mapping"]
    except Exception:
        return False
    MC = importlib.import_module("u-llava_amd.modeling_core")
    MU = importlib.import_module("u-llava_amd.modeling_ullava")
    for name, cls in (("ullava_core", MC.UllavaCoreForCausalLM), ("ullava", MU.UllavaForCausalLM)):
        if mapping.get(name) is not cls:
            mapping.pop(name, None)                             # (the reference's own class, had its models package been imported first)
            reg.register_model(name)(cls)
    return True
