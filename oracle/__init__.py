"""CPU oracle for the DeMoN hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and
there only as the checker.  The product (demon_amd/, libdemon_hip.so) never does.

PARITY UNPINNED (see oracle/demon_oracle.c header): TensorFlow 1.4 and lmbspecialops, which hold the
reference arithmetic, are not in /root/reference; the reference has no tests or golden vectors for
this path.  depth_to_flow (output values), flow_to_depth (as its inverse), the angle-axis convention and the
evaluation metrics are pinned against the reference's in-tree code run in the build container
(tests/golden/, oracle/build_ref.py); warp2d, the stencil ops and the TensorFlow layers are not.
"""
