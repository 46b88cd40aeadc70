"""Configuration objects mirroring the reference's (same field names / defaults / to_dict()).

reference: models/ullava_core.py:40-75 (UllavaCoreConfig, a LlamaConfig subclass + CLIPVisionConfig),
models/ullava.py:17-66 (UllavaConfig), models/segment_anything/build_sam.py:15-22,56-102 (SAM ViT-H dims).
They are plain Python objects (no transformers import on the hot path); `to_dict()` emits the same keys the
reference writes into config.json, so a checkpoint's config can be fed back in as kwargs.
"""
import copy
from typing import Optional


class _Cfg:
    def to_dict(self):
        out = {}
        for k, v in self.__dict__.items():
            out[k] = v.to_dict() if isinstance(v, _Cfg) else copy.deepcopy(v)
        return out

    def __repr__(self):
        return f"{self.__class__.__name__}({self.to_dict()})"


class CLIPVisionConfig(_Cfg):
    """transformers CLIPVisionConfig fields used by the forward (defaults = openai/clip-vit-large-patch14)."""
    model_type = "clip_vision_model"

    def __init__(self, hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, num_channels=3,
                 image_size=224, patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, **kwargs):
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_channels = num_channels
        self.image_size = image_size
        self.patch_size = patch_size
        self.hidden_act = hidden_act
        self.layer_norm_eps = layer_norm_eps
        if hidden_act != "quick_gelu":
            raise NotImplementedError("only CLIP's quick_gelu vision tower is on the u-LLaVA path")


class UllavaCoreConfig(_Cfg):
    """LLaMA fields + the reference's multimodal fields (defaults = LLaMA-7B)."""
    model_type = "ullava_core"
    is_composition = True

    def __init__(self, vision_config=None, vision_hidden_layer=-1, projector_type="mlp", projector_from_scratch=True,
                 mm_token_ids=None, vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=None, hidden_act="silu", max_position_embeddings=2048,
                 rms_norm_eps=1e-6, use_cache=True, pad_token_id=None, bos_token_id=1, eos_token_id=2, rope_theta=10000.0,
                 output_hidden_states=False, output_attentions=False, use_return_dict=True, **kwargs):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads or num_attention_heads
        if self.num_key_value_heads != num_attention_heads:
            raise NotImplementedError("LLaMA-7B has no GQA; num_key_value_heads must equal num_attention_heads")
        self.hidden_act = hidden_act
        self.max_position_embeddings = max_position_embeddings
        self.rms_norm_eps = rms_norm_eps
        self.use_cache = use_cache
        self.pad_token_id = pad_token_id
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.rope_theta = rope_theta
        self.output_hidden_states = output_hidden_states
        self.output_attentions = output_attentions
        self.use_return_dict = use_return_dict
        self.vision_hidden_layer = vision_hidden_layer
        self.mm_token_ids = mm_token_ids
        self.projector_type = projector_type
        self.projector_from_scratch = projector_from_scratch
        if isinstance(vision_config, CLIPVisionConfig):
            self.vision_config = vision_config
        else:
            self.vision_config = CLIPVisionConfig(**vision_config) if vision_config else CLIPVisionConfig()

    def to_dict(self):
        out = super().to_dict()
        out["model_type"] = self.model_type
        return out


class SamConfig(_Cfg):
    """build_sam_vit_h constants (build_sam.py:15-22,56-102); tests shrink the image encoder only."""

    def __init__(self, embed_dim=1280, depth=32, num_heads=16, global_attn_indexes=(7, 15, 23, 31), window_size=14, patch_size=16,
                 img_size=1024, out_chans=256, mlp_ratio=4.0, decoder_depth=2, decoder_heads=8, decoder_mlp_dim=2048,
                 num_multimask_outputs=3, iou_head_depth=3, iou_head_hidden_dim=256, mask_in_chans=16):
        self.embed_dim = embed_dim
        self.depth = depth
        self.num_heads = num_heads
        self.global_attn_indexes = list(global_attn_indexes)
        self.window_size = window_size
        self.patch_size = patch_size
        self.img_size = img_size
        self.out_chans = out_chans
        self.mlp_ratio = mlp_ratio
        self.decoder_depth = decoder_depth
        self.decoder_heads = decoder_heads
        self.decoder_mlp_dim = decoder_mlp_dim
        self.num_multimask_outputs = num_multimask_outputs
        self.iou_head_depth = iou_head_depth
        self.iou_head_hidden_dim = iou_head_hidden_dim
        self.mask_in_chans = mask_in_chans


class UllavaConfig(_Cfg):
    model_type = "ullava"
    is_composition = True

    def __init__(self, llm_config=None, ce_weight=1.0, bce_weight=2.0, dice_weight=0.5, l1_weight=1.0, iou_weight=1.0, out_dim=256,
                 seg_token_idx=32007, loc_token_idx=32008, train_mask_decoder=True, sam_config: Optional[dict] = None, **kwargs):
        if isinstance(llm_config, UllavaCoreConfig):
            self.llm_config = llm_config
        else:
            self.llm_config = UllavaCoreConfig(**llm_config) if llm_config else UllavaCoreConfig()
        self.ce_weight = ce_weight
        self.bce_weight = bce_weight
        self.out_dim = out_dim
        self.dice_weight = dice_weight
        self.l1_weight = l1_weight
        self.iou_weight = iou_weight
        self.seg_token_idx = seg_token_idx
        self.loc_token_idx = loc_token_idx
        self.train_mask_decoder = train_mask_decoder
        # not in the reference (it hard-wires build_sam_vit_h): lets tests run a small image encoder
        self.sam_config = sam_config if isinstance(sam_config, SamConfig) else SamConfig(**(sam_config or {}))

    def to_dict(self):
        out = super().to_dict()
        out["model_type"] = self.model_type
        return out
