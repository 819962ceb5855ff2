"""MI355X-native `pdm` package: the bilevel fine-tune / unlearn step of the pruned SD-2.1 U-Net on hand-written HIP
kernels (libpdmk.so, C ABI in include/pdmk.h).  Same import names as the reference package so its scripts and configs
drive this one unchanged; the version string carries the reference version it mirrors plus the backend tag."""
REFERENCE_API_VERSION = (2, 2, 0)
BACKEND = "mi355x.r1"
__version__ = ".".join(str(v) for v in REFERENCE_API_VERSION) + "+" + BACKEND
