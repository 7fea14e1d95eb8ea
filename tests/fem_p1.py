"""BASELINE config 4 helpers now live in the package (host-side set-up): parmgmc_amd/unstructured.py."""
from parmgmc_amd.unstructured import assemble_p1, greedy_aggregation, read_gmsh41_triangles, refine_uniform, build_hierarchy  # noqa: F401
