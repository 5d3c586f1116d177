"""PIL -> normalised tensor transforms with the semantics of the reference's torchvision pipeline
(P/src/model.py:347-357): Resize(bicubic) -> CenterCrop -> ToTensor -> Normalize(ImageNet), and the
"no crop" variant that resizes straight to a square.  torchvision is not installed on the target;
these are host-side helpers (device-side preprocessing is a 'next' row).
"""
from __future__ import annotations

import numpy as np
import torch

_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(3, 1, 1)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1)


# OpenAI-CLIP statistics, the transforms of the reference's timm CLIP backbones (P/src/model.py:377-391)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _to_tensor_normalized(img, mean=_MEAN, std=_STD) -> torch.Tensor:
    arr = np.asarray(img.convert("RGB"), dtype=np.uint8).transpose(2, 0, 1).astype(np.float32) / 255.0
    return torch.from_numpy((arr - mean) / std)


def _stat(v):
    return np.array(v, dtype=np.float32).reshape(3, 1, 1)


def center_crop_origin(w: int, h: int, crop: int):
    """torchvision F.center_crop: (left, top) of the crop window in the coordinates of the (w, h) image; negative where
    the image is smaller than the crop (zero padding of (crop - dim) // 2 before and (crop - dim + 1) // 2 after)."""
    pl = (crop - w) // 2 if crop > w else 0
    pt = (crop - h) // 2 if crop > h else 0
    pw = w + pl + ((crop - w + 1) // 2 if crop > w else 0)
    ph = h + pt + ((crop - h + 1) // 2 if crop > h else 0)
    if pw == crop and ph == crop:
        return -pl, -pt
    return int(round((pw - crop) / 2.0)) - pl, int(round((ph - crop) / 2.0)) - pt


class ResizeCropTransform:
    """T.Compose([T.Resize(resize_dim, BICUBIC), T.CenterCrop(crop_dim), T.ToTensor(), T.Normalize(...)])"""

    def __init__(self, resize_dim: int, crop_dim: int, mean=None, std=None):
        self.resize_dim, self.crop_dim = resize_dim, crop_dim
        self.mean, self.std = (_MEAN if mean is None else _stat(mean)), (_STD if std is None else _stat(std))

    def __call__(self, img) -> torch.Tensor:
        from PIL import Image
        w, h = img.size
        s = self.resize_dim
        if w <= h:                      # torchvision: the SHORTER side becomes `size`, the other int(size*long/short)
            nw, nh = s, int(s * h / w)
        else:
            nw, nh = int(s * w / h), s
        img = img.resize((nw, nh), Image.BICUBIC)
        left, top = center_crop_origin(nw, nh, self.crop_dim)
        c = self.crop_dim
        img = img.crop((left, top, left + c, top + c))      # PIL fills with zeros outside the image = the pad
        return _to_tensor_normalized(img, self.mean, self.std)


class SquareResizeTransform:
    """T.Compose([T.Resize((resize_dim, resize_dim), BICUBIC), T.ToTensor(), T.Normalize(...)])"""

    def __init__(self, resize_dim: int, mean=None, std=None):
        self.resize_dim = resize_dim
        self.mean, self.std = (_MEAN if mean is None else _stat(mean)), (_STD if std is None else _stat(std))

    def __call__(self, img) -> torch.Tensor:
        from PIL import Image
        return _to_tensor_normalized(img.resize((self.resize_dim, self.resize_dim), Image.BICUBIC), self.mean, self.std)


def make_transforms(resize_dim: int, crop_dim: int, mean=None, std=None):
    return ResizeCropTransform(resize_dim, crop_dim, mean, std), SquareResizeTransform(resize_dim, mean, std)


def process_bboxes(imgs, bboxes, transform) -> torch.Tensor:
    """P/src/bbox_utils.py:406-421: crop every xywh box from its PIL image and transform the crop."""
    out = []
    for img, img_boxes in zip(imgs, bboxes.tolist()):
        for x_min, y_min, w, h in img_boxes:
            out.append(transform(img.crop((x_min, y_min, x_min + w, y_min + h))))
    return torch.stack(out)


def adjust_bbox_for_transform(image, bbox, resize_dim, crop_dim):
    """xywh box in original-image pixels -> crop-pixel coordinates of ResizeCropTransform
    (P/src/bbox_utils.py:170-218; the eval drivers call this before batching boxes)."""
    x1, y1, w, h = bbox
    ow, oh = image.size
    if ow < oh:
        sw = resize_dim / ow
        sh = (resize_dim * oh) / ow / oh
    else:
        sh = resize_dim / oh
        sw = (resize_dim * ow) / oh / ow
    nw, nh = int(ow * sw), int(oh * sh)
    x1, y1, w, h = x1 * sw, y1 * sh, w * sw, h * sh
    x1 -= max(0, (nw - crop_dim) // 2)
    y1 -= max(0, (nh - crop_dim) // 2)
    x1 = max(0, min(x1, crop_dim - 1))
    y1 = max(0, min(y1, crop_dim - 1))
    w = max(0, min(w, crop_dim - x1))
    h = max(0, min(h, crop_dim - y1))
    return [x1, y1, w, h]


def adjust_bbox_for_transform_no_scale(image, bbox, target_width, target_height):
    """P/src/bbox_utils.py:222-250."""
    x1, y1, w, h = bbox
    ow, oh = image.size
    sw, sh = target_width / ow, target_height / oh
    return [x1 * sw, y1 * sh, w * sw, h * sh]
