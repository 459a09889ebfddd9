"""CPU oracle for the flower-pose hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / reported CPU baseline.  The product path
(``flope_amd``) never imports this package and fails loudly when the HIP
library is missing.

Pinning status:
  * ``diff_quats`` and ``TransformerEncoder`` are pinned by fixtures generated
    by importing the reference's own files in the build container
    (tests/golden/make_reference_fixtures.py).
  * ``squarify_bb``/``bb_in_frame``/``filter_very_large_bb``/``get_points3d``
    are pinned by known-answer values derived by hand from the reference
    source.
  * The PoseResNet forward, ``roma.special_procrustes``, ``cv2.resize`` /
    ``cv2.erode`` live in third-party packages that are absent from the
    container and for which the reference holds no golden vectors:
    PARITY UNPINNED for those boundaries; the restatement follows the
    published algorithms (torchvision 0.20.1 resnet18 topology, roma 1.5.1
    special_procrustes formula, OpenCV 4.10 resize/erode semantics).
"""
