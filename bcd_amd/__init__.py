"""bcd_amd -- MI355X-native Bayesian collaborative denoiser (hot path of superboubek/bcd).

Python is plumbing only: `bcd_amd.hip` binds the C ABI of include/bcd_hip.h (libbcd_hip.so) with ctypes and
uses torch for device memory / streams / torch.distributed.  The product is the HIP library and the C++
host library (libbcdcore.so + bcd_cli) built by `python -m bcd_amd.build`."""
