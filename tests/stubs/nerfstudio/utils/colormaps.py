def apply_colormap(image, *a, **k): return image.expand(*image.shape[:-1], 3)
def apply_depth_colormap(depth, accumulation=None, *a, **k): return depth.expand(*depth.shape[:-1], 3)
