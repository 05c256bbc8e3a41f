"""coskad_amd -- MI355X-native (gfx950) implementation of COSKAD's STS-GCN encoder hot path.

Host side: PyTorch-ROCm modules with the reference's names; device side: the C-ABI HIP
library coskad_amd/libcoskad_hip.so (sources in coskad_amd/csrc, ABI in include/coskad_hip.h).
"""
__version__ = "0.1.0"
