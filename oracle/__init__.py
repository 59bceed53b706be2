"""CPU oracle for the UrsoNet hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference's hot path
(``/root/reference/net.py`` graph + losses + optimizer step, and the NumPy pose
codec in ``se3lib.py`` / ``utils.py``).  It exists so that the HIP kernels can
be checked against something that follows the reference line by line.

Rules (enforced by tests/test_layout_rules.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here;
  * nothing under ``ursonet_amd/`` (the product) may import it -- the product
    fails loudly when the HIP extension is missing instead of falling back.

Pinning status
  * pose codec / decode / metrics (``pose_math.py``): PINNED against golden
    vectors generated in the build container by importing the reference's own
    NumPy modules (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
  * NN graph (``graph_ref.py``): the arithmetic lives in TensorFlow 1.x +
    standalone Keras 2.1.6-2.2.4 (un-vendored, un-pinned third-party
    dependency, requirements.txt:9-10, not importable here: no Python-3.10
    wheels, no network).  The reference holds no tests or golden vectors for
    this path.  ==> "parity unpinned" for the NN graph: it is a literal
    restatement of net.py + the documented TF/Keras semantics (SURVEY.md
    Appendix A), realised twice (torch-CPU functional ops in ``graph_ref.py``,
    direct NumPy loops in ``numpy_direct.py``) which must agree with each
    other before either is used as the checker.
"""
