"""twopaco_amd -- MI355X-native junction enumeration (TwoPaCo's two-pass hot path).

Layout:
  csrc/   hand-written HIP kernels for gfx950 + the C-ABI (include/twopaco_hip.h) -> lib/libtwopaco_hip.so
  host/   C++ host layer mirroring the reference interface (CreateEnumerator, JunctionPosition
          API, FASTA parser, `twopaco` CLI)                                    -> lib/libtwopaco_host.so, bin/twopaco
  capi.py ctypes bindings over both libraries (tests, bench.py; plumbing only)
  synth.py  seeded synthetic genome workloads of BASELINE.json's configs
  dist.py   multi-GPU (one process per GPU) Bloom sharding by bit address over torch.distributed

There is no CPU fallback: importing works anywhere, computing needs the HIP library and a GPU.
"""
from .build import build_all, lib_dir  # noqa: F401
