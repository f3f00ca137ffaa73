"""jaxtyping stand-in: `Float[Tensor, "..."]` style annotations evaluate to the tensor type."""
class _Ann:
    def __class_getitem__(cls, item):
        return item[0] if isinstance(item, tuple) else item
class Float(_Ann): pass
class Int(_Ann): pass
class Bool(_Ann): pass
