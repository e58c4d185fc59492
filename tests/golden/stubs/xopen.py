"""Import-time stand-in (see Bio/__init__.py)."""


def xopen(*a, **k):
    raise RuntimeError("xopen stand-in: FASTQ reading is not part of golden generation")
