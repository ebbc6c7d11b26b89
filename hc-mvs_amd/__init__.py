"""hc-mvs_amd: MI355X-native PatchMatch MVS densifier (hot path of Liaoyongjian1/HC-MVS).

The product is the C-ABI library built from csrc/ (include/hcmvs_hip.h); this package holds the thin
ctypes binding used by tests and bench.py, and the synthetic scene generator.
Import with importlib.import_module("hc-mvs_amd") (the directory name carries a hyphen).
"""
