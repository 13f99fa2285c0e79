"""TEST-ONLY stand-in for the `moleculekit` package, which the reference's driver imports (`torchmd/run.py:4`,
`torchmd/npzmol.py:2`) and this image does not have.  It exists for one test — tests/test_gpu_reference_driver.py, the
reference's own `run.py` executed unchanged on the MI355X classes — and reads files with `torchmd_amd.io`.  Nothing in
the product imports it."""
