"""ORACLE — test infrastructure only (never imported by the product package).

CPU restatements of the reference's algorithms for the north-star hot path; each module cites the
reference file:line it follows.  Allowed importers: tests/, __graft_entry__.smoke(), and bench.py's
cpu_baseline leg.  The product package `contexture_nerf_amd` must never import from here and fails
loudly when its HIP library is missing.

Pinning status (details in DESIGN.md):
  * pinned against vectors produced by importing the reference itself (tests/golden/make_golden.py):
    positional encoding, NeRF2D, get_rays/ndc_rays/sample_pdf, face-view map / view-weight masks,
    DreamTime table, crop boxes, grid split/merge, camera pose lists, depth normalisation,
    mesh normalisation, config defaults.
  * PARITY UNPINNED (third-party dependency absent, reference holds no fixtures): kaolin raster /
    prepare_vertices / texture_mapping, torch-scatter scatter_max (semantically exact max),
    diffusers UNet / PNDM, nerf-pytorch raw2outputs.
"""
