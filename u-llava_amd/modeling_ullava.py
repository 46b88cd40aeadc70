"""UllavaForCausalLM on MI355X: u-LLaVA core + [SEG]/[LOC] projectors + SAM prompt-encoder / mask-decoder / postprocess.

Host-side mirror of reference `models/ullava.py:69-434` (same constructor, attribute names `.llm .seg_projector
.visual_model .det_projector .det_decoder`, state-dict keys, `forward(..., inference=True)` dict keys and `evaluate()`
tuple).  Differences that are deliberate and output-preserving:
  * the [SEG]/[LOC] rows are gathered BEFORE the projector MLPs (the reference projects all B*S rows and then masks);
  * the SAM image encoder runs batched instead of one image per python iteration with empty_cache();
  * the dense positional encoding is evaluated once (the reference recomputes it per image).
"""
from typing import List, Optional

import torch
import torch.nn as nn

from . import autograd_ops as A
from . import ops
from .configuration import UllavaConfig
from .modeling_core import BF16, Linear, UllavaCoreForCausalLM, _Holder
from .sam import SamEngine, build_sam_holder



def _mlp_seq(dims, device, dtype, dropout_tail=False):
    """nn.Sequential(Linear, ReLU, Linear, ...) with the reference's child indices (ReLU / Dropout hold no parameters)."""
    mods = []
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        mods.append(Linear(a, b, device=device, dtype=dtype))
        if i < len(dims) - 2:
            mods.append(nn.ReLU(inplace=True))
    if dropout_tail:
        mods.append(nn.Dropout(0.0))
    return nn.Sequential(*mods)


class UllavaForCausalLM(nn.Module):
    config_class = UllavaConfig

    def __init__(self, config: UllavaConfig, device=None, dtype=BF16):
        super().__init__()
        self.config = config
        llm_config = config.llm_config
        self.llm = UllavaCoreForCausalLM(llm_config, device=device, dtype=dtype)
        D, O = llm_config.hidden_size, config.out_dim
        self.seg_projector = _mlp_seq([D, D, O], device, dtype, dropout_tail=True)            # ullava.py:113-118
        self.visual_model = build_sam_holder(config.sam_config, device=device, dtype=dtype)   # ullava.py:124 (build_sam_vit_h)
        self.det_projector = _mlp_seq([D, D, O], device, dtype, dropout_tail=True)            # ullava.py:86-91
        self.det_decoder = _mlp_seq([O, O, O // 2, 4], device, dtype)                         # ullava.py:96-102
        self._sam = SamEngine(self.visual_model, config.sam_config)
        # trainability defaults of the reference's constructor (ullava.py:84-130): the projectors and the box decoder are trainable,
        # SAM is frozen except -- with config.train_mask_decoder -- its mask decoder.  (The language model's parameters default to
        # frozen here; train_ullava.py:239-261 switches on what it trains by name, exactly as it does for the reference.)
        for mod in (self.seg_projector, self.det_projector, self.det_decoder):
            for p_ in mod.parameters():
                p_.requires_grad = True
        if getattr(config, "train_mask_decoder", True):
            for p_ in self.visual_model.mask_decoder.parameters():
                p_.requires_grad = True
        self.overlap_sam_encoder = True     # SAM image encoder on a second HIP stream beside CLIP + LLaMA (False: same stream)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, device=None, **kwargs):
        """reference: inference_ullava.py:37, evaluation/eval_ullava.py:135, webui/gradio_chat.py:26."""
        from .checkpoint import ullava_from_pretrained
        return ullava_from_pretrained(cls, pretrained_model_name_or_path, torch_dtype, device, **kwargs)

    def save_pretrained(self, save_directory, **kwargs):
        from .checkpoint import save_pretrained
        return save_pretrained(self, save_directory, **kwargs)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        sd = {k.replace("llm.vision_encoder.vision_model.", "llm.vision_encoder."): v for k, v in state_dict.items()}
        sd = {k: v for k, v in sd.items() if not k.endswith("position_ids") and "rotary_emb.inv_freq" not in k}
        self.llm._packed = None
        self._sam.invalidate()
        return super().load_state_dict(sd, strict=strict, assign=assign)

    def load_visual_checkpoint(self, checkpoint):
        """reference ullava.py:134-137."""
        with open(checkpoint, "rb") as f:
            sd = torch.load(f, map_location="cpu", weights_only=True)      # a plain tensor state-dict (sam_vit_h_4b8939.pth)
        self._sam.invalidate()
        return self.visual_model.load_state_dict(sd, strict=False)

    @property
    def dtype(self):
        return self.llm.dtype

    @property
    def device(self):
        return self.llm.device

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)           # (the core model invalidates its own packs in its _apply)
        self._sam.invalidate()
        self._side = None
        return out

    def _side_stream(self):
        if not self.overlap_sam_encoder:
            return torch.cuda.current_stream()
        st = getattr(self, "_side", None)
        if st is None:
            st = self._side = torch.cuda.Stream()
        return st

    # -- SAM image encoder -------------------------------------------------------------------------------------------
    def _visual_embs_tm(self, pixel_values: torch.Tensor) -> torch.Tensor:
        return self._sam.encode(pixel_values)                       # [B, g*g, 256] token-major

    def get_visual_embs(self, pixel_values: torch.FloatTensor) -> torch.Tensor:
        """reference ullava.py:139-150 -> [B, 256, g, g] (NCHW, as the reference returns it)."""
        tm = self._visual_embs_tm(pixel_values)
        B, P, C = tm.shape
        g = int(P ** 0.5)
        return tm.view(B, g, g, C).permute(0, 3, 1, 2).contiguous()

    # -- shared tail of forward / evaluate -----------------------------------------------------------------------------
    def _heads_training_graph(self) -> bool:
        """gradients enabled and one of the heads on top of the language model is trainable (train_ullava.py:248-261: seg / det
        projectors, det_decoder, mask_decoder)."""
        if not torch.is_grad_enabled() or self.dtype == torch.float32:        # (the fp32 build is inference-only: csrc/f32.hip has no backward)
            return False
        mods = (self.seg_projector, self.det_projector, self.det_decoder, self.visual_model.mask_decoder)
        return any(p.requires_grad for m in mods for p in m.parameters())

    def _run_mlp(self, seq, x, train: bool = False):
        lin = [m for m in seq if isinstance(m, Linear)]
        for i, l in enumerate(lin):
            if train:
                x = A.linear(x, l.weight, l.bias, relu=i < len(lin) - 1)
            else:
                x = ops.linear(x, l.weight, l.bias, act="relu" if i < len(lin) - 1 else None)
        return x

    def _select(self, last_hidden: torch.Tensor, token_mask: torch.Tensor, projector, train: bool = False) -> List[torch.Tensor]:
        """rows of last_hidden [B, L, D] where token_mask [B, L] is set -> per-sample [n_i, out_dim] after the projector."""
        B, L, D = last_hidden.shape
        counts = token_mask.sum(-1).tolist()                       # host sync (the reference indexes device offsets too)
        idx = token_mask.reshape(-1).nonzero().squeeze(-1)
        if train:
            rows = last_hidden.reshape(B * L, D)[idx]               # gather (data movement) recorded by autograd
        else:
            rows = ops.gather_rows(last_hidden.reshape(B * L, D), idx)
        emb = self._run_mlp(projector, rows, train) if rows.shape[0] else rows.new_empty(0, self.config.out_dim)
        out, o = [], 0
        for c in counts:
            out.append(emb[o:o + c])
            o += c
        return out

    def _decode(self, image_embeddings_tm, pred_embeddings, resize_list, size_list, train: bool = False):
        """prompt encoder + mask decoder + postprocess.  The reference decodes one image at a time (ullava.py:228-252); every op of
        the decoder is independent per prompt, so all prompts of the batch go through one chain of launches here (the per-image
        chains were host-launch-bound: ~70 small kernels each) and only the two resizes of postprocess_masks stay per image."""
        counts = [int(e.shape[0]) for e in pred_embeddings]
        dev = image_embeddings_tm.device
        low_all = None
        if sum(counts):
            idx = torch.tensor([i for i, c in enumerate(counts) for _ in range(c)], dtype=torch.int64, device=dev)
            text = torch.cat([e for e in pred_embeddings if e.shape[0]], dim=0).contiguous()
            if train:
                masks = self._sam.decode_train(image_embeddings_tm, text, idx)
            else:
                masks, _iou = self._sam.decode(image_embeddings_tm, text, idx)
            low_all = masks[:, 0].contiguous()                      # multimask_output=False -> mask 0
        pred_masks, o = [], 0
        for i, c in enumerate(counts):
            H, W = int(size_list[i][0]), int(size_list[i][1])
            if c == 0:
                pred_masks.append(torch.empty(0, H, W, device=dev, dtype=torch.float32))
                continue
            post = self._sam.postprocess_train if train else self._sam.postprocess
            pred_masks.append(post(low_all[o:o + c].contiguous(), resize_list[i], (H, W)))
            o += c
        return pred_masks

    def forward(self, images_sam: torch.FloatTensor, images: torch.FloatTensor, input_ids: torch.LongTensor, labels: torch.LongTensor,
                attention_mask: torch.LongTensor, mask_list: List[torch.FloatTensor], size_list: List[torch.Tensor],
                resize_list: List[tuple], bbox_list: List[torch.FloatTensor], inference: bool = False):
        """reference ullava.py:152-333.  inference=False returns the training-loss dict; with gradients enabled and trainable
        parameters (train_ullava.py:207-261) the losses carry an autograd graph whose backward runs HIP kernels (autograd_ops.py)."""
        B = input_ids.shape[0]
        # the SAM image encoder does not depend on the LLM: it runs on a second HIP stream and fills the gaps (tile-quantisation
        # tails, small kernels, launch latency) of the CLIP + LLaMA stream; joined before the mask decoder.  Each stream has its own
        # stream-K workspace (ops._streamk_ws); while both are busy the K-split of partial tile rounds is limited to K >= 8192 because the
        # CUs it would fill are not idle (policy only -- correctness does not depend on it).
        main = torch.cuda.current_stream()
        side = self._side_stream()
        side.wait_stream(main)
        with ops.streamk_policy(8192 if side is not main else 2048):
            with torch.cuda.stream(side), torch.no_grad():          # SAM image encoder: frozen (reference get_visual_embs: no_grad)
                image_embeddings = self._visual_embs_tm(images_sam)
            pad = torch.zeros((B, 1), dtype=torch.bool, device=input_ids.device)
            seg_token_mask = torch.cat([input_ids[:, 1:] == self.config.seg_token_idx, pad], dim=1)    # row t selected iff ids[t+1]==[SEG]
            loc_token_mask = torch.cat([input_ids[:, 1:] == self.config.loc_token_idx, pad], dim=1)
            output = self.llm.forward(images=images, attention_mask=attention_mask, input_ids=input_ids, labels=labels,
                                      output_hidden_states=True)
        last = output.hidden_states[-1]
        main.wait_stream(side)
        image_embeddings.record_stream(main)
        train = self._heads_training_graph() and not inference
        pred_embeddings = self._select(last, seg_token_mask, self.seg_projector, train)
        pred_loc_embeddings = self._select(last, loc_token_mask, self.det_projector, train)
        pred_masks = self._decode(image_embeddings, pred_embeddings, resize_list, size_list, train)
        pred_boxes = [self._run_mlp(self.det_decoder, e, train) if e.shape[0] else e.new_empty(0, 4) for e in pred_loc_embeddings]
        if inference:
            return {"pred_masks": pred_masks, "pred_boxes": pred_boxes, "gt_masks": mask_list, "gt_boxes": bbox_list, "logits": output.logits}
        return self._losses(output.loss, pred_masks, pred_boxes, mask_list, bbox_list, train)

    def _losses(self, ce, pred_masks, pred_boxes, gt_masks, gt_boxes, train: bool = False):
        """reference ullava.py:268-333 + models/loss.py.  The per-pixel / per-box sums are HIP kernels; the handful of scalar
        combinations below run as 0-dim device ops.  Like the reference, the total is accumulated IN PLACE into the tensor that
        `ce_loss` names, so the returned "ce_loss" equals "loss" (reproduced, not corrected)."""
        cfg = self.config
        if ce is None:
            raise ValueError("forward(inference=False) needs `labels` (the reference multiplies output.loss by ce_weight)")
        ce_loss = ce * cfg.ce_weight
        loss = ce_loss
        dev = ce_loss.device
        mask_bce = torch.zeros((), device=dev)
        mask_dice = torch.zeros((), device=dev)
        box_l1 = torch.zeros((), device=dev)
        box_giou = torch.zeros((), device=dev)
        num_masks = num_boxes = 0
        for i in range(len(pred_masks)):
            gm, pm = gt_masks[i], pred_masks[i]
            if gm.shape[0] != pm.shape[0]:
                raise AssertionError(f"gt_mask.shape: {tuple(gm.shape)}, pred_mask.shape: {tuple(pm.shape)}")
            n = gm.shape[0]
            if n:
                sums = (A.mask_loss_sums if train else ops.mask_loss_sums)(pm.contiguous(), gm.to(device=dev, dtype=torch.float32).contiguous())   # [n, 4]
                hw = pm[0].numel()
                mask_bce = mask_bce + (sums[:, 0] / hw).sum() / (n + 1e-8) * n
                dice = 1 - (2 * sums[:, 1] + 1e-6) / (sums[:, 2] + sums[:, 3] + 1e-6)
                mask_dice = mask_dice + dice.sum() / (n + 1e-8) * n
            num_masks += n
            gb, pb = gt_boxes[i], pred_boxes[i]
            if gb.shape[0] != pb.shape[0]:
                raise AssertionError(f"gt_box.shape: {tuple(gb.shape)}, pred_box.shape: {tuple(pb.shape)}")
            nb = gb.shape[0]
            if nb:
                bl = (A.box_losses if train else ops.box_losses)(pb, gb.to(device=dev, dtype=torch.float32))
                box_l1 = box_l1 + bl[0] / (nb + 1e-8)
                box_giou = box_giou + bl[1] / (nb + 1e-8)
            num_boxes += nb
        mask_bce_loss = cfg.bce_weight * mask_bce / (num_masks + 1e-8)
        mask_dice_loss = cfg.dice_weight * mask_dice / (num_masks + 1e-8)
        mask_loss = mask_bce_loss + mask_dice_loss
        box_l1_loss = cfg.l1_weight * box_l1 / (num_boxes + 1e-8)
        box_giou_loss = cfg.iou_weight * box_giou / (num_boxes + 1e-8)
        bbox_loss = box_l1_loss + box_giou_loss
        loss += mask_loss
        loss += bbox_loss
        return {"loss": loss, "ce_loss": ce_loss, "mask_bce_loss": mask_bce_loss, "mask_dice_loss": mask_dice_loss, "mask_loss": mask_loss,
                "bbox_loss": bbox_loss}

    @torch.no_grad()
    def evaluate(self, images_sam, images, input_ids, raw_size_list, resize_list, max_new_tokens=32, temperature=0.2, top_p=None,
                 num_beams=1, no_repeat_ngram_size=None, stopping_criteria=None):
        """reference ullava.py:335-434 -> (output_ids, pred_masks, pred_boxes)."""
        main = torch.cuda.current_stream()                          # SAM image encoder on the second stream, under the generation loop
        side = self._side_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            image_embeddings = self._visual_embs_tm(images_sam)     # its GEMMs use the side stream's own stream-K workspace
        outputs = self.llm.generate(input_ids=input_ids, images=images, max_new_tokens=max_new_tokens, num_beams=num_beams, top_p=top_p,
                                    do_sample=True if temperature > 0 else False, temperature=temperature, output_hidden_states=True,
                                    return_dict_in_generate=True, no_repeat_ngram_size=no_repeat_ngram_size,
                                    stopping_criteria=stopping_criteria, keep_last_step_only=True)
        output_ids = outputs.sequences
        last = outputs.hidden_states[-1][-1]                          # last step, last layer: [B, L-1, D]
        seg_token_mask = output_ids[:, 1:] == self.config.seg_token_idx
        loc_token_mask = output_ids[:, 1:] == self.config.loc_token_idx
        L1 = last.shape[1]
        pred_embeddings = self._select(last, seg_token_mask[:, :L1], self.seg_projector)
        pred_loc_embeddings = self._select(last, loc_token_mask[:, :L1], self.det_projector)
        main.wait_stream(side)
        image_embeddings.record_stream(main)
        pred_masks = self._decode(image_embeddings, pred_embeddings, resize_list, raw_size_list)
        pred_boxes = [self._run_mlp(self.det_decoder, e) if e.shape[0] else e.new_empty(0, 4) for e in pred_loc_embeddings]
        return output_ids, pred_masks, pred_boxes
