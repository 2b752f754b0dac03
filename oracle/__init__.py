"""TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the U-Net-CA training hot path of
Createroner/InSAR-Unet-CA. Nothing in the product package
(`insar_unet_ca_amd/`) may import from here: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and
there only as the checker / reported baseline, never as the thing shipped.
"""
