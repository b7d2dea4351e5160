"""spartan2_amd — MI355X-native Spartan prover hot path (HIP kernels behind a C ABI).

Python here is only the loader/harness glue (ctypes over include/spartan_hip.h); the host side
above the C ABI is C++ (spartan2_amd/csrc/host_*.cpp), as the reference is compiled code.
"""
