"""CPU oracle of the CGLB hot path: TEST INFRASTRUCTURE ONLY.

`cglb_oracle.py` (numpy) and `cglb_oracle.c` (blocked C/OpenMP, bound through `cglb_oracle_c.py`) restate the reference's
algorithm for checking the HIP path; `gen_golden.py` produced `tests/golden/` from the reference's own solver.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this package - never the product path
(`cglb_amd/`), which has no CPU fallback.
"""
