"""Import-time stand-in so that the reference's summary.py can be imported for golden-vector
generation (tests/golden/make_golden.py).  Nothing of Biopython is used on the hot path."""
