class _M:
    def __init__(self, *a, **k):
        pass


class ModificationInfo(_M):
    pass


for _n in ("LengthTagModifier", "SuffixRemover", "PrefixSuffixAdder", "ZeroCapper", "QualityTrimmer",
           "UnconditionalCutter", "NEndTrimmer", "AdapterCutter", "PairedAdapterCutterError",
           "PairedAdapterCutter", "NextseqQualityTrimmer", "Shortener"):
    globals()[_n] = type(_n, (_M,), {})
