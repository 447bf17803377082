"""oracle/ -- CPU restatement of the reference Far3D inference path.  TEST INFRASTRUCTURE ONLY.

Nothing under far3d_amd/ may import this package.  The only legitimate users are tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg, and there only as the checker / the
reported CPU baseline -- never as the thing measured as the product.

Pinning status: the reference ships no tests and no golden vectors
(SURVEY.md §4), so the oracle is pinned against outputs of the reference's own files run in the
build container (oracle/refload.py + tools/gen_golden.py -> tests/golden/*.npz).  Third-party
arithmetic the reference calls but does not contain (mmcv MSDA/MHA/FFN/ConvModule, mmdet FPN /
MlvlPointGenerator / inverse_sigmoid) is restated from its published semantics; for those pieces
parity is pinned to that restatement + torch CPU behaviour, not to the original binaries.
"""
