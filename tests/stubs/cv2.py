"""cv2 stand-in: imported by the reference's data utilities for image I/O the tests never run; the interpolation
constants appear as default arguments at import time."""
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA = 0, 1, 2, 3
