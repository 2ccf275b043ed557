"""asvspoof2021_air_amd -- MI355X-native hot path of yzyouzhang/ASVspoof2021_AIR.

LFCC front-end -> ResNet / ECAPA-TDNN forward+backward -> OC-Softmax (ang_iso),
behind the reference's Python module surface:

    from asvspoof2021_air_amd.feature_extraction import LFCC
    from asvspoof2021_air_amd.resnet import ResNet
    from asvspoof2021_air_amd.loss import AngularIsoLoss, OCSoftmax

All device arithmetic runs in hand-written gfx950 HIP kernels reached through
the C-ABI in include/air_hip.h (libair_hip.so, loaded with ctypes).
"""
__version__ = "0.1.0"
