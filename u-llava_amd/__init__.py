"""u-llava_amd: MI355X-native u-LLaVA multimodal forward path (HIP kernels behind a C-ABI).

The directory name carries a hyphen (mandated layout), so import it with
``importlib.import_module("u-llava_amd")``; inside the package only relative imports are used.
Nothing here imports `oracle/`; the HIP library is loaded lazily on first kernel call and its
absence is a hard error (no CPU fallback exists).
"""
__version__ = "0.1.0"
