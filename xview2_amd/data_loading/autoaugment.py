"""`--autoaugment`: the ImageNet AutoAugment policy as used by the reference (data_loading/autoaugment.py): one of 25
two-operation sub-policies per sample, applied to the pre image, the post image (damage task) and - for the geometric
operations only - the label mask.

Table-driven re-implementation on PIL.  Behaviour kept: the sub-policy table (the published ImageNet policy plus the
reference's five repeated rows), the magnitude grids (:52-67), rotation composited over transparent black (:69-71),
which operations touch the mask (:131-133), the call signature and return arity (:40-42,:136-151).  Deliberate
difference: the reference draws the random SIGN of a shear/translate/colour change separately for the image, the
second image and the mask (the `random.choice` sits inside each lambda), so a mask can be sheared the opposite way to
its image; here one sign is drawn per operation and shared.  Randomness comes from the loader's per-worker numpy
Generator."""
import numpy as np
from PIL import Image, ImageEnhance, ImageOps

GEOMETRIC = ("shearX", "shearY", "translateX", "translateY", "rotate")

# (p1, op1, magnitude index 1, p2, op2, magnitude index 2)
POLICY = (
    (0.4, "posterize", 8, 0.6, "rotate", 9), (0.6, "solarize", 5, 0.6, "autocontrast", 5),
    (0.8, "equalize", 8, 0.6, "equalize", 3), (0.6, "posterize", 7, 0.6, "posterize", 6),
    (0.4, "equalize", 7, 0.2, "solarize", 4), (0.4, "equalize", 4, 0.8, "rotate", 8),
    (0.6, "solarize", 3, 0.6, "equalize", 7), (0.8, "posterize", 5, 1.0, "equalize", 2),
    (0.2, "rotate", 3, 0.6, "solarize", 8), (0.6, "equalize", 8, 0.4, "posterize", 6),
    (0.8, "rotate", 8, 0.4, "color", 0), (0.4, "rotate", 9, 0.6, "equalize", 2),
    (0.0, "equalize", 7, 0.8, "equalize", 8), (0.6, "invert", 4, 1.0, "equalize", 8),
    (0.6, "color", 4, 1.0, "contrast", 8), (0.8, "rotate", 8, 1.0, "color", 2),
    (0.8, "color", 8, 0.8, "solarize", 7), (0.4, "sharpness", 7, 0.6, "invert", 8),
    (0.6, "shearX", 5, 1.0, "equalize", 9), (0.4, "color", 0, 0.6, "equalize", 3),
    (0.4, "equalize", 7, 0.2, "solarize", 4), (0.6, "solarize", 5, 0.6, "autocontrast", 5),
    (0.6, "invert", 4, 1.0, "equalize", 8), (0.6, "color", 4, 1.0, "contrast", 8),
    (0.8, "equalize", 8, 0.6, "equalize", 3),
)


def magnitude(op, idx):
    grids = {
        "shearX": np.linspace(0, 0.3, 10), "shearY": np.linspace(0, 0.3, 10),
        "translateX": np.linspace(0, 150 / 331, 10), "translateY": np.linspace(0, 150 / 331, 10),
        "rotate": np.linspace(0, 30, 10), "color": np.linspace(0.0, 0.9, 10),
        "posterize": np.round(np.linspace(8, 4, 10), 0).astype(int), "solarize": np.linspace(256, 0, 10),
        "contrast": np.linspace(0.0, 0.9, 10), "sharpness": np.linspace(0.0, 0.9, 10),
        "brightness": np.linspace(0.0, 0.9, 10),
    }
    return grids[op][idx] if op in grids else 0


def apply_op(img, op, mag, sign, fillcolor=0):
    """one operation on one PIL image (RGB tile or single-channel mask)"""
    w, h = img.size
    if op == "shearX":
        return img.transform(img.size, Image.AFFINE, (1, mag * sign, 0, 0, 1, 0), Image.BICUBIC, fillcolor=fillcolor)
    if op == "shearY":
        return img.transform(img.size, Image.AFFINE, (1, 0, 0, mag * sign, 1, 0), Image.BICUBIC, fillcolor=fillcolor)
    if op == "translateX":
        return img.transform(img.size, Image.AFFINE, (1, 0, mag * w * sign, 0, 1, 0), fillcolor=fillcolor)
    if op == "translateY":
        return img.transform(img.size, Image.AFFINE, (1, 0, 0, 0, 1, mag * h * sign), fillcolor=fillcolor)
    if op == "rotate":
        rot = img.convert("RGBA").rotate(mag)
        return Image.composite(rot, Image.new("RGBA", rot.size, 0), rot).convert(img.mode)
    if op == "color":
        return ImageEnhance.Color(img).enhance(1 + mag * sign)
    if op == "contrast":
        return ImageEnhance.Contrast(img).enhance(1 + mag * sign)
    if op == "sharpness":
        return ImageEnhance.Sharpness(img).enhance(1 + mag * sign)
    if op == "brightness":
        return ImageEnhance.Brightness(img).enhance(1 + mag * sign)
    if op == "posterize":
        return ImageOps.posterize(img, int(mag))
    if op == "solarize":
        return ImageOps.solarize(img, mag)
    if op == "autocontrast":
        return ImageOps.autocontrast(img)
    if op == "equalize":
        return ImageOps.equalize(img)
    if op == "invert":
        return ImageOps.invert(img)
    raise ValueError("unknown AutoAugment operation %r" % op)


class ImageNetPolicy:
    def __init__(self, fillcolor=0, rng=None):
        self.fillcolor = fillcolor
        self.rng = rng

    def _rng(self):
        if self.rng is not None:
            return self.rng
        from .pytorch_loader import _rng
        return _rng()

    def __call__(self, img, lbl, img2=None):
        r = self._rng()
        p1, op1, m1, p2, op2, m2 = POLICY[int(r.integers(0, len(POLICY)))]
        for p, op, mi in ((p1, op1, m1), (p2, op2, m2)):
            if r.random() >= p:
                continue
            mag, sign = magnitude(op, mi), (1 if r.random() < 0.5 else -1)
            img = apply_op(img, op, mag, sign, self.fillcolor)
            if img2 is not None:
                img2 = apply_op(img2, op, mag, sign, self.fillcolor)
            if op in GEOMETRIC:
                lbl = apply_op(lbl, op, mag, sign, self.fillcolor)
        if img2 is not None:
            return img, lbl, img2
        return img, lbl

    def __repr__(self):
        return "AutoAugment ImageNet Policy"
