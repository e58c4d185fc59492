def warn_duplicate_adapters(*a, **k):
    pass
