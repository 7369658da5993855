"""MI355X-native streaming Conformer-ASR forward path (drop-in for the hot path of
m1el/nemotron-asr.cpp).

The directory name (`nemotron-asr.cpp_amd`) is not a Python identifier; load it with
`__graft_entry__.load_package()` which registers it as `nemotron_asr_amd`.

Submodules:
  synth  -- seeded synthetic weights / PCM (workload definition)
  capi   -- ctypes binding of the C-ABI in include/nemotron_asr_amd.h (HIP engine)
"""
__all__ = ["synth", "capi"]
