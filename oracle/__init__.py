"""CPU oracle for the T-bar detection hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package - as the checker, never as the thing measured or shipped (the
measurement / debug scripts under `tools/` use it the same way: to diff detections
and to time the CPU side; they are not part of the product either).
The product (`flypylib_amd/`) never imports it and fails loudly when its HIP
library is missing.

Pinning:
  * `voxel2obj_oracle`, `infer_oracle`, `set_filter`: checked against outputs of
    the reference's own code run in the build container
    (`tests/golden/make_golden.py` -> `tests/golden/*.npz`).
  * `cnn_oracle` (Keras/TensorFlow layer arithmetic): the reference's Keras and
    TensorFlow dependencies are unpinned (`conda-recipe/meta.yaml:18-19`) and
    absent, and the reference has no tests -> PARITY UNPINNED at that boundary;
    the restatement is pinned by hand-derived known-answer tests only
    (tests/test_oracle_cnn.py).
"""
