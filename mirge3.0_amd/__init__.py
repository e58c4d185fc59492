"""mirge3.0_amd -- MI355X-native hot path of miRge3.0.

Scope: read collapse -> 9/10-pass small-RNA annotation cascade
-> per-class / per-miRNA count join.  The compute runs in hand-written HIP kernels
behind the C ABI declared in ``include/mirge_native.h``; this package is the Python
host side that mirrors the reference call sites (``mirge/__main__.py:140,157,166``):

* ``collapse.baking``      <- ``mirge/libs/digest.py:105``        (collapse + matrix)
* ``cascade.bwt_align``    <- ``mirge/libs/manifoldAlign.py:68``  (annotation cascade)
* ``countjoin.summarize``  <- ``mirge/libs/summary.py:677``       (count join + CSVs)

There is no CPU fallback: every compute entry point goes through ``_ffi`` and raises if
``libmirge_native.so`` or a GPU is missing.
"""

__version__ = "0.1.0"

PASS_COLUMNS = [
    "exact miRNA", "hairpin miRNA", "mature tRNA", "primary tRNA", "snoRNA",
    "rRNA", "ncrna others", "mRNA", "isomiR miRNA", "spike-in",
]  # reference: mirge/libs/digest.py:253
