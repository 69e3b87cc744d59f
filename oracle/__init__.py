"""CPU oracle for the kaolin-wisp volumetric hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU (numpy for integer/byte work, torch-CPU fp32/fp64 for the
differentiable float stages), the algorithm of the reference path

    PackedRFTracer.trace -> OctreeAS.raymarch -> HashGrid.interpolate -> NeuralRadianceField.rgba
        -> exponential_integration / sum_reduce

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it, and only as the checker.  Nothing under ``kaolin-wisp_amd/`` imports it; the product path raises
if the HIP library is missing.

PARITY UNPINNED.  The reference's own tests hold no op-level golden vectors for this path
(reference ``tests/core/*`` never touch hashgrid_interpolate, raymarch or compositing; see
SURVEY.md section 4 and 8c), and neither ``kaolin`` (pinned ``kaolin==0.13.0``, INSTALL.md:14,64,69;
not vendored under /root/reference, no wheel, no network) nor ``wisp._C`` (CUDA-only, setup.py:89-90)
can be built or imported here.  The restatement is anchored on:
  * the in-tree CUDA sources for the hash grid (``wisp/csrc/ops/hashgrid_interpolate_cuda.cu``,
    ``wisp/csrc/ops/hash_utils.cuh``) and the lattice sampler (``wisp/csrc/ops/uniform_sample_cuda.cu``),
  * the wisp call sites of every Kaolin leaf (``wisp/accelstructs/octree_as.py``,
    ``wisp/tracers/packed_rf_tracer.py``, ``wisp/ops/spc/*.py``) and the published Kaolin 0.13 semantics
    summarised in SURVEY.md Appendix A,
  * hand-computed known-answer vectors and brute-force cross-checks committed under ``tests/golden``.
Where the exact upstream float ordering cannot be known (slab test of ``unbatched_raytrace``, the
quantisation inside ``unbatched_query``), the choice made here IS the definition and each such choice
is stated in the function's docstring.
"""
