"""torchvision.transforms.functional: the two names the reference uses (resolution-schedule downscaling)."""
import enum
import torch.nn.functional as F
class InterpolationMode(enum.Enum):
    NEAREST = "nearest"
    BILINEAR = "bilinear"
def resize(img, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=None):
    mode = interpolation.value
    kw = {} if mode == "nearest" else dict(align_corners=False, antialias=bool(antialias))
    squeeze = img.dim() == 3
    out = F.interpolate(img[None] if squeeze else img, size=list(size), mode=mode, **kw)
    return out[0] if squeeze else out
