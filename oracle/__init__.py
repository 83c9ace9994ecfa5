"""oracle — TEST INFRASTRUCTURE ONLY.

CPU checker for the HIP hot path: `oracle.py` binds the plain-C restatement
(tvmi_oracle.c), `build_ref.py` compiles the real reference CPU kernels into `_ref/`.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; nothing under vision_amd/ does (tests/test_layout.py enforces it).
"""
