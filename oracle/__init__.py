"""CPU oracle for the semantic-depth hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / torch-CPU) of the algorithm on the path
BASELINE.json's north_star names.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; nothing under ``semantic_depth_amd/`` does,
and the product path raises when the HIP library is missing instead of falling back to this.

Pinning status:
  * oracle.pcl        — PINNED: checked function by function against the reference's own
                        ``semantic_depth_lib/pcl.py`` (imported in the build container by
                        tests/golden/make_golden.py; outputs committed under tests/golden/).
  * oracle.fusion     — ``post_processing``: PINNED — bit-exact (float64) against the reference's own
                        DepthFrame.post_processing (semantic_depth.py:656-664), lifted from the
                        reference's AST and executed by tests/golden/make_golden.py (the module
                        itself needs tf/cv2 and cannot be imported); vectors in
                        tests/golden/ref_pieces.npz, checked by tests/test_outputs.py.
                        mask gather: numpy boolean indexing, as the reference.
                        ``reproject`` restates cv2.reprojectImageTo3D [UPSTREAM OpenCV 4.0.0.21]
                        (Vec3f /= double is a multiply by 1./W, core/matx.hpp):
                        PARITY UNPINNED against OpenCV itself.
  * oracle.nets       — FCN-8s decoder follows fcn8s/fcn.py:159-224; the VGG16 body (Udacity
                        SavedModel) and the monodepth body (mrharicot/monodepth, unpinned copy)
                        are absent from the reference tree: PARITY UNPINNED, restated from the
                        published architectures.
  * oracle.o3d        — Open3D legacy statistical/radius outlier removal [UPSTREAM, version
                        unpinned, not in requirements.txt]: PARITY UNPINNED.
"""
