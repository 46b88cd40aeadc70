#!/usr/bin/env python
"""Golden vectors for the pre/post-processing kernels.  Run in the build container (needs Pillow,
transformers and /root/reference):  python tests/golden/gen_golden_preprocess.py

Sources of truth, in the order the reference calls them:
  * CLIP branch: transformers.CLIPImageProcessor (the PIL backend of the installed transformers; dataset/processors/
    clip_processor.py:31,93 instantiates exactly this class) on PIL images, with and without the reference's pad_pil.
  * SAM branch: ResizeLongestSide.apply_image is `np.array(torchvision.resize(to_pil_image(img), size))`; torchvision is not in
    this image and for PIL inputs that call is `img.resize(size[::-1], PIL.Image.BILINEAR)`, which is what is executed here.
    SegToolBox.preprocess (dataset/tools/mask_toolbox.py:15-25) is restated with the same torch ops (the module itself imports
    pycocotools, absent here).
  * evaluation: /root/reference/evaluation/tools.py cannot be imported here (it imports torchvision at module level), so
    intersectionAndUnionGPU (tools.py:29-41) is restated below with the identical torch calls (view / masked assignment /
    torch.histc), and the accumulation loop of trainers/ullava_trainer.py:40-52 around it.
Stored: inputs (uint8 images, logits, targets) and expected outputs only."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def pad_pil(pil_img, background_color=(255, 255, 255)):          # clip_processor.py:35-52, verbatim behaviour
    width, height = pil_img.size
    if width == height:
        return pil_img
    if width > height:
        result = Image.new(pil_img.mode, (width, width), background_color)
        result.paste(pil_img, (0, (width - height) // 2))
        return result
    result = Image.new(pil_img.mode, (height, height), background_color)
    result.paste(pil_img, ((height - width) // 2, 0))
    return result


def main():
    from transformers import CLIPImageProcessor
    rs = np.random.RandomState(1234)
    out = {"versions": {"pillow": Image.__version__ if hasattr(Image, "__version__") else __import__("PIL").__version__,
                        "transformers": __import__("transformers").__version__, "torch": str(torch.__version__)}}
    sys.path.insert(0, HERE)
    from synth import synth_image
    seeds = iter(range(100, 1000))

    import hashlib
    sha = lambda t: hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()
    clip_cases = []
    # small crops carry the full expected tensor; the production sizes (224 / 336) carry a sha256 of the fp32 bytes
    for size in (48, 70, 224, 336):
        proc = CLIPImageProcessor(size={"shortest_edge": size}, crop_size={"height": size, "width": size})
        for (h, w), aspect in (((97, 131), None), ((150, 90), "pad"), ((size, size), None), ((61, 200), "pad")):
            seed = next(seeds)
            img = synth_image(h, w, seed)
            pil = Image.fromarray(img)
            if aspect == "pad":
                pil = pad_pil(pil)
            ref = proc.preprocess(pil, return_tensors="pt")["pixel_values"][0]
            case = dict(size=size, aspect_ratio=aspect, image_hw=(h, w), seed=seed, sha256=sha(ref))
            if size < 100:
                case["pixel_values"] = ref.clone()
            clip_cases.append(case)
    out["clip"] = clip_cases

    sam_cases = []
    mean = torch.Tensor([123.675, 116.28, 103.53]).view(-1, 1, 1)
    std = torch.Tensor([58.395, 57.12, 57.375]).view(-1, 1, 1)
    for L, (h, w) in ((96, (120, 160)), (96, (333, 97)), (96, (64, 64)), (96, (40, 300)), (1024, (480, 640)), (1024, (375, 500))):
        seed = next(seeds)
        img = synth_image(h, w, seed)
        scale = L * 1.0 / max(h, w)
        nh, nw = int(h * scale + 0.5), int(w * scale + 0.5)
        resized = np.array(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        x = torch.from_numpy(resized).permute(2, 0, 1).contiguous()
        x = (x - mean) / std
        x = F.pad(x, (0, L - nw, 0, L - nh))
        case = dict(long_side=L, image_hw=(h, w), seed=seed, resized_hw=(nh, nw), sha256=sha(x), sha256_resized=sha(torch.from_numpy(resized)),
                    sha256_bf16=sha(x.to(torch.bfloat16).view(torch.int16)))
        if L < 200:
            case.update(resized=torch.from_numpy(resized), pixel_values=x.clone())
        sam_cases.append(case)
    out["sam"] = sam_cases

    def intersectionAndUnionGPU(output, target, K, ignore_index=255):          # evaluation/tools.py:29-41, same torch ops
        output = output.view(-1)
        target = target.view(-1)
        output[target == ignore_index] = ignore_index
        intersection = output[output == target]
        area_intersection = torch.histc(intersection, bins=K, min=0, max=K - 1)
        area_output = torch.histc(output, bins=K, min=0, max=K - 1)
        area_target = torch.histc(target, bins=K, min=0, max=K - 1)
        return area_intersection, area_output + area_target - area_intersection, area_target

    iou_cases = []
    for n, (h, w), with_ignore, empty in ((3, (48, 64), False, False), (2, (31, 17), True, False), (2, (16, 16), False, True)):
        logits = torch.from_numpy(rs.randn(n, h, w).astype(np.float32))
        target = torch.from_numpy((rs.rand(n, h, w) > 0.6).astype(np.uint8))
        if with_ignore:
            target[torch.from_numpy(rs.rand(n, h, w) > 0.9)] = 255
        if empty:
            target[0] = 0
            logits[0] = -1.0                                      # no-object target, nothing predicted: union == 0 for class 1
        masks_list = target.int()
        output_list = (logits > 0).int()
        intersection, union, acc_iou = 0.0, 0.0, 0.0
        for mask_i, output_i in zip(masks_list, output_list):
            i_, u_, _ = intersectionAndUnionGPU(output_i.contiguous().clone().float(), mask_i.contiguous().float(), 2, ignore_index=255)
            intersection += i_
            union += u_
            acc_iou += i_ / (u_ + 1e-5)
            acc_iou[u_ == 0] += 1.0
        iou_cases.append(dict(logits=logits, target=target, intersection=intersection.clone(), union=union.clone(),
                              acc_iou=(acc_iou / masks_list.shape[0]).clone()))
    out["iou"] = iou_cases
    fn = os.path.join(HERE, "p1_preprocess.pt")
    torch.save(out, fn)
    print("wrote", fn, os.path.getsize(fn) // 1024, "KiB")


if __name__ == "__main__":
    main()
