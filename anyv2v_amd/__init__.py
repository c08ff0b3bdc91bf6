"""anyv2v_amd: MI355X-native hot path of AnyV2V (I2VGen-XL DDIM inversion + PnP edit).

The compute path is ``libanyv2v_hip.so`` (hand-written HIP kernels for gfx950) behind the C ABI of
``include/anyv2v_hip.h``; this package is the Python host that mirrors the reference's interfaces.
"""
__version__ = "0.1.0"
