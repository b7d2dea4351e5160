"""MI355X-native Spartan2 prover hot path.

  csrc/      HIP kernels (kernels_*.hpp) and the C ABI of include/spartan_hip.h (capi_*.hip) -> lib/libspartan_hip.so
  host/      C++ protocol drivers above the ABI (spartan_snark.cpp, neutronnova_nifs.cpp, sharded_snark.cpp + comm.hpp) -> lib/libspartan_host.so
  frontend/  integer R1CS generators for the bench circuits (inputs only)
  hip.py, host.py, dist.py   ctypes views of the two libraries and the torch.distributed plumbing of the harness
"""
