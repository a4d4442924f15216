"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's (AaronWatters/contourist) tetrahedral voxel march and its
mesh post-passes.  Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package `contourist_amd` never imports it.

  march_oracle.c     plain-C Level-0 march (classification, interpolation, triangle emit)
  level0.py          ctypes wrapper + canonicalisation helpers for Level-0 comparisons
  postpass.py        Python/numpy restatement of the Level-1 post-passes (weld, tiny collapse,
                     clean, orient) with a canonical processing order
  make_goldens.py    runs the REAL reference (only where /root/reference exists) and writes
                     tests/golden/*.npz
  build.py           gcc recipe for libmarch_oracle.so
"""
